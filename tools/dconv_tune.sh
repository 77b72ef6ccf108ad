#!/bin/bash
# tools/dconv_tune.sh <base variant> "<set args>" ... — builds tuning variants of a direct-conv kernel and times each with
# tools/micro/dconv_bench.cpp on the GPU box (run under gpurun).  Example:
#   tools/dconv_tune.sh dconv_l3_s1 "" "skew=16" "skew=32"
set -e
base=$1; shift
out=gpurun_out/tune; mkdir -p $out
LLVM=/opt/rocm/lib/llvm/bin
hipcc -O2 --offload-arch=gfx950 tools/micro/dconv_bench.cpp -o $out/dconv_bench
i=0
for sets in "$@"; do
  args=""; for kv in $sets; do args="$args --set $kv"; done
  sfx="_t$i"
  gen=dconv_gen.py; case $base in pw_*) gen=pw_gen.py;; esac
  python3 sota_imagenet_amd/csrc/asm/$gen --out $out $args --suffix $sfx $base > /dev/null
  $LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $out/$base$sfx.s -o $out/$base$sfx.o
  $LLVM/ld.lld -shared $out/$base$sfx.o -o $out/$base$sfx.hsaco
  echo "== $base [$sets]"
  for nch in ${NCH:-4}; do
    $out/dconv_bench $out/$base$sfx.hsaco $base$sfx $out/$base$sfx.tbl ${GEOM:-14 14 1 256 256 256} $nch
  done
  i=$((i+1))
done
