#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; ROOT=$PWD
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_full.py -m gpu -q -x -p no:cacheprovider -k "mas or MAS or compute_loss" 2>&1 | tail -2
cat > /tmp/masp.py <<PY
import importlib, sys, time, torch
sys.path.insert(0, "$ROOT")
S = importlib.import_module("speech-backbones_amd")
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(99)
for b, tx, ty in ((16, 200, 1024), (16, 400, 1024)):
    value = torch.randn(b, tx, ty, generator=g) * 4
    xl = torch.randint(tx // 2, tx + 1, (b,), generator=g); yl = torch.randint(ty // 2, ty + 1, (b,), generator=g)
    m = ((torch.arange(tx)[None, :] < xl[:, None]).unsqueeze(-1) * (torch.arange(ty)[None, :] < yl[:, None]).unsqueeze(1)).float()
    vd, md = value.to(dev), m.to(dev)
    for _ in range(10): S.mas_maximum_path(vd, md)
    torch.cuda.synchronize()
PY
cd /tmp; rm -rf /tmp/masp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/masp -o p -- python /tmp/masp.py > /tmp/masp.log 2>&1; f=$(find /tmp/masp -name "*kernel_stats*.csv" | head -1); grep mas_ $f | cut -c1-130
