#!/bin/bash
# A/B of library variants on one GPU box visit.  Usage: tools/ab2.sh "<tag> <lib.so> [bench flags...]" ...
# Each argument is one run; the variant library is selected with GTTS_LIB (see speech-backbones_amd/_lib.py).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for spec in "$@"; do
  set -- $spec
  tag=$1; lib=$2; shift 2
  GTTS_LIB=$PWD/speech-backbones_amd/$lib timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras "$@" \
      > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.txt
  echo "== $tag rc=$? $(python - <<PY
import json
try:
    d = json.load(open('gpurun_out/ab_$tag.json'))
    r = d.get('roofline') or {}
    print(d['value'], d['config'].get('ms_per_unet_call'), r.get('avg_us'), r.get('frac'))
except Exception as e:
    print('parse error', e)
PY
)"
done
