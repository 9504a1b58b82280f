"""profiles/traffic.json from the two PMC summaries of tools/gpu_profiles_r05.sh (FETCH_SIZE, WRITE_SIZE; KB per kernel name).

bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 / launches.  The factor 2 on reads is the gfx950 rocprofv3
correction of MI355X_MICROARCH.md (HBM section), re-checked here on kernels with known byte counts (prep_input,
mul_mask, tail_identity, final_euler: profiles/r01_pmc_fetch_v2.txt / r01_pmc_write_v2.txt); WRITE_SIZE is exact.
usage: python tools/make_traffic_json.py <fetch.txt> <write.txt> <workload> <B> <T> <out.json>
"""
import json
import sys


def demangle(name):
    """rocprofv3 leaves the bf16 template instances mangled (its demangler does not know DF16b): restore the printed form of
    the gtts:: kernels (int, float and __bf16 template arguments are all this library uses)."""
    import re
    m = re.match(r"_ZN4gtts(\d+)", name)
    if not m:
        return name
    n = int(m.group(1))
    base, rest = name[m.end():m.end() + n], name[m.end() + n:]
    if not rest.startswith("I"):
        return "gtts::" + base
    args, i = [], 1
    while i < len(rest) and rest[i] != "E":
        if rest.startswith("Li", i):
            j = rest.index("E", i)
            v = rest[i + 2:j]
            args.append("-" + v[1:] if v.startswith("n") else v)
            i = j + 1
        elif rest.startswith("DF16b", i):
            args.append("__bf16")
            i += 5
        elif rest[i] == "f":
            args.append("float")
            i += 1
        else:
            return name
    return "gtts::%s<%s>" % (base, ", ".join(args))


def parse(path):
    out = {}
    with open(path) as f:
        next(f)
        for line in f:
            parts = line.rsplit(None, 2)
            if len(parts) == 3:
                out[demangle(parts[0].strip())] = (int(parts[1]), float(parts[2]))
    return out


fetch, write = parse(sys.argv[1]), parse(sys.argv[2])
kern = {}
for k, (n, kb) in fetch.items():
    wn, wkb = write.get(k, (n, 0.0))
    kern[k] = {"launches": n, "read_bytes_per_launch": round(2 * kb * 1024 / n), "write_bytes_per_launch": round(wkb * 1024 / wn),
               "bytes_per_launch": round(2 * kb * 1024 / n + wkb * 1024 / wn)}
json.dump({"workload": sys.argv[3], "B": int(sys.argv[4]), "T": int(sys.argv[5]),
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, GTTS_STREAMS=1); read bytes = 2 x FETCH_SIZE KB (gfx950 correction), "
                     "written bytes = WRITE_SIZE KB; files " + sys.argv[1] + ", " + sys.argv[2],
           "kernels": kern}, open(sys.argv[6], "w"), indent=1)
print("wrote", sys.argv[6], len(kern), "kernels")
