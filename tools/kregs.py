#!/usr/bin/env python3
"""VGPRs / scratch bytes / LDS of every kernel in a --save-temps .s file:  tools/kregs.py file.s [filter]"""
import re, subprocess, sys
s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', s, re.S):
    name, body = m.group(1), m.group(2)
    v = re.search(r'next_free_vgpr (\d+)', body).group(1)
    sc = re.search(r'private_segment_fixed_size (\d+)', body).group(1)
    lds = re.search(r'group_segment_fixed_size (\d+)', body).group(1)
    d = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    if flt in d:
        print("vgpr %3s scratch %4s lds %6s  %s" % (v, sc, lds, d[:110]))
