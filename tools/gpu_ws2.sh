#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/ws_repro.py 16 1024 > gpurun_out/ws2_repro.txt 2>&1; echo "repro rc=$?"; tail -30 gpurun_out/ws2_repro.txt
timeout 300 python tools/ws_repro.py dump /tmp/new.pt 4 256 > /dev/null 2>&1; GTTS_LIB=$PWD/speech-backbones_amd/libgtts_nows.so timeout 300 python tools/ws_repro.py dump /tmp/old.pt 4 256 > /dev/null 2>&1
timeout 100 python tools/ws_repro.py cmp /tmp/new.pt /tmp/old.pt > gpurun_out/ws2_cmp.txt 2>&1; grep -c . gpurun_out/ws2_cmp.txt; grep "<--" gpurun_out/ws2_cmp.txt | head; sort -k9 -g gpurun_out/ws2_cmp.txt | tail -3
