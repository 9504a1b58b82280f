#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; ROOT=$PWD
for rep in 1 2 3; do
for spec in "base libgradtts_gfx950.so" "rot libgtts_rot.so"; do
  set -- $spec
  GTTS_LIB=$PWD/speech-backbones_amd/$2 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-roofline > gpurun_out/abr_$1_$rep.json 2>/dev/null
  echo "rep $rep $1: $(python -c "import json;d=json.load(open('gpurun_out/abr_$1_$rep.json'));print(d['value'], d['config'].get('ms_per_unet_call'))")"
done
done
GTTS_LIB=$PWD/speech-backbones_amd/libgtts_rot.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "golden or batch_size or full_size" 2>&1 | tail -2
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tr -o prof -- python $ROOT/tools/train_prof.py > $ROOT/gpurun_out/train_steps_r03.txt 2>&1); echo "rocprof train rc=$?"
for f in $(find /tmp/prof_tr -name "*kernel_stats*.csv"); do cp $f gpurun_out/train_step_rocprof_kernel_stats_r03.csv; done
python tools/stats_summary.py gpurun_out/train_step_rocprof_kernel_stats_r03.csv 3; grep step gpurun_out/train_steps_r03.txt
timeout 600 python bench.py > gpurun_out/bench_r03.json 2> gpurun_out/bench_tables_r03.txt; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open('gpurun_out/bench_r03.json'))
print(d['value'], d['config']['ms_per_unet_call'], d['roofline']['kernel'], d['roofline']['avg_us'], d['roofline']['frac'], d['roofline'].get('frac_algorithmic'), d['roofline'].get('traffic'), d.get('cpu_baseline', {}).get('value'))
for k, v in d.get('extras', {}).items():
    print(k, {kk: vv for kk, vv in v.items() if kk not in ('workload', 'roofline')} if isinstance(v, dict) else v)
    if isinstance(v, dict) and 'roofline' in v: print('   roofline:', v['roofline'].get('kernel'), v['roofline'].get('avg_us'), v['roofline'].get('frac'), v['roofline'].get('bound'), v['roofline'].get('traffic'))
PY
