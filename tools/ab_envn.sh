#!/bin/bash
# on the GPU box: one environment switch at several values, alternating on one box.  usage: ab_envn.sh NAME "v1 v2 v3" [model] [rounds]
NAME=$1; VALS=$2; MODEL=${3:-resnet50}; ROUNDS=${4:-3}
for r in $(seq $ROUNDS); do for v in $VALS; do
  env $NAME=$v timeout -k 10 300 python bench.py --model $MODEL --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary 2>&1 | grep "^{" | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$NAME=$v', '$MODEL', r['ms_per_step'])"
done; done
