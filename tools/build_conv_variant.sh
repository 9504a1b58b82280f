#!/bin/bash
# A/B libraries that differ from the product in the convolution files only: recompile conv_ws.hip / conv_mfma.hip / conv_up.hip with extra flags and
# link against the product's other objects.  tools/build_conv_variant.sh name "flags" [name "flags" ...]  ->  speech-backbones_amd/libgtts_<name>.so
cd /root/repo/speech-backbones_amd
python build.py > /dev/null || exit 1
P=build/libgradtts_gfx950
while [ $# -gt 0 ]; do
  name=$1; flags=$2; shift 2
  (
  D=build/libgtts_$name; mkdir -p $D
  for f in conv_ws conv_mfma conv_up conv_up_ws; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-slp-vectorize $flags -c csrc/$f.hip -o $D/$f.o &
  done
  wait
  objs=""
  for o in $P/*.o; do b=$(basename $o); if [ -f $D/$b ]; then objs="$objs $D/$b"; else objs="$objs $o"; fi; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libgtts_$name.so $objs && echo "built libgtts_$name.so ($flags)"
  ) &
done
wait
