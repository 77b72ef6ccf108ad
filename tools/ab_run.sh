#!/bin/bash
# on the GPU box: alternates the two builds (same device, same process layout) and prints ms/step of each round
DT=${1:-bf16}; ROUNDS=${2:-3}; MODEL=${3:-resnet50}
for r in $(seq $ROUNDS); do for v in old new; do
  MI355RN_LIB=$PWD/sota_imagenet_amd/lib/variant_$v.so timeout -k 10 300 python bench.py --model $MODEL --steps 30 --warmup 5 --dtype $DT \
    --no-cpu-baseline --no-roofline --no-secondary 2>&1 | grep "^{" | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v', '$DT', r['ms_per_step'])"
done; done
