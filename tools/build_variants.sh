#!/bin/bash
# build library variants in-tree: tools/build_variants.sh name "flags" [name "flags" ...]   ("product" = the default library)
cd /root/repo
while [ $# -gt 0 ]; do
  name=$1; flags=$2; shift 2
  if [ "$name" = product ]; then libname=libgradtts_gfx950.so; else libname=libgtts_$name.so; fi
  GTTS_LIB_NAME=$libname GTTS_EXTRA_FLAGS="$flags" python -c "
import importlib.util
spec = importlib.util.spec_from_file_location('b','speech-backbones_amd/build.py'); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m); print(m.build(force=True))" 2>&1 | tail -2 &
done
wait
