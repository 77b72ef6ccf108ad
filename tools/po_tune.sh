#!/bin/bash
# tools/po_tune.sh <base variant> <M> <N> "<set args>" ... — builds tuning variants of an output-heavy pointwise kernel (asm/po_gen.py) and times
# each with tools/micro/po_bench.cpp on the GPU box (run under gpurun).  Example:
#   tools/po_tune.sh po_k256_b256_s2_a2 50176 1024 "" "probe=4" "probe=2"
set -e
base=$1; M=$2; N=$3; shift 3
out=gpurun_out/tune; mkdir -p $out
LLVM=/opt/rocm/lib/llvm/bin
hipcc -O2 --offload-arch=gfx950 tools/micro/po_bench.cpp -o $out/po_bench
K=$(echo $base | sed 's/po_k\([0-9]*\)_.*/\1/'); BN=$(echo $base | sed 's/.*_b\([0-9]*\)_.*/\1/')
ST=$(echo $base | sed 's/.*_s\([0-9]\)_.*/\1/'); AD=$(echo $base | sed 's/.*_a\([0-9]\)$/\1/')
i=0
for sets in "$@"; do
  args=""; TP=64; [ "$BN" = 64 ] && TP=128; WPC=1; for kv in $sets; do case $kv in WPC=*) WPC=${kv#WPC=}; continue;; MFR=*) TP=$((16 * ${kv#MFR=}));; esac; args="$args --set $kv"; done
  sfx="_t$i"
  python3 sota_imagenet_amd/csrc/asm/po_gen.py --out $out $args --suffix $sfx $base > /dev/null
  $LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $out/$base$sfx.s -o $out/$base$sfx.o
  $LLVM/ld.lld -shared $out/$base$sfx.o -o $out/$base$sfx.hsaco
  printf "%-22s " "[$sets]"
  $out/po_bench $out/$base$sfx.hsaco $base$sfx $K $BN $ST $AD $M $N 30 $TP $WPC
  i=$((i+1))
done
