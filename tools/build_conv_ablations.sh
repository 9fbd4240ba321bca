#!/bin/bash
# Ablation builds of conv.hip for tools/bench_conv.py --lib: -DWM_CONV_ABLATE=1 (no tile fetches after the first),
# =2 (fetches, waits and barriers only: no fragment reads / MFMAs), =3 (neither: prologue + barriers + epilogue).
# The other objects are those of the production build (self-supervised-wafermaps_amd/build.py must have run).
set -e
cd "$(dirname "$0")/../self-supervised-wafermaps_amd/csrc"
mkdir -p _build/ablate
objs=$(ls _build/*.o | grep -v "/conv\.")
for a in 1 2 3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-function -Wno-pass-failed \
    -Wno-inline-asm -DWM_CONV_ABLATE=$a -c conv.hip -o _build/ablate/conv_a$a.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _build/ablate/libwafer_a$a.so $objs _build/ablate/conv_a$a.o
  echo "built _build/ablate/libwafer_a$a.so"
done
