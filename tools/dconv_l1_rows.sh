#!/bin/bash
# on the GPU box: layer 1's 3x3 kernels (56 x 56 x 64) with 4 / 7 / 8 / 14 output rows per tile (tools/micro/dconv_bench.cpp)
set -e
out=gpurun_out/tune; mkdir -p $out
LLVM=/opt/rocm/lib/llvm/bin
hipcc -O2 --offload-arch=gfx950 tools/micro/dconv_bench.cpp -o $out/dconv_bench
for base in dconv_l1_s1 dconv_l1_s2; do for rt in 4 7 8 14; do
  sfx="_r$rt"
  python3 sota_imagenet_amd/csrc/asm/dconv_gen.py --out $out --set ROWS_T=$rt --suffix $sfx $base > /dev/null
  $LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $out/$base$sfx.s -o $out/$base$sfx.o
  $LLVM/ld.lld -shared $out/$base$sfx.o -o $out/$base$sfx.hsaco
  $out/dconv_bench $out/$base$sfx.hsaco $base$sfx $out/$base$sfx.tbl 56 56 -$((56 / rt)) 64 64 256 1 40 64
done; done
