"""Register / scratch / occupancy table of one source file's kernels from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
usage: hipcc ... -Rpass-analysis=kernel-resource-usage -c file.hip 2> remarks.txt; python tools/kernel_resources.py remarks.txt [filter]"""
import re, subprocess, sys

text = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
names = re.findall(r"Function Name: (\S+)", text)
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
blocks = re.split(r"Function Name: \S+", text)[1:]
for d, b in zip(dem, blocks):
    if flt and flt not in d:
        continue
    g = lambda k: re.search(k + r": (\d+)", b).group(1)
    scr, occ = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")
    print("%-120s V %3s A %3s scratch %4s occ %s" % (d[:120], g("VGPRs"), g("AGPRs"), scr, occ))
