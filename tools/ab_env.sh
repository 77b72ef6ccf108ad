#!/bin/bash
# on the GPU box: A/B of one environment switch on one box, alternating.  usage: ab_env.sh NAME VAL_A VAL_B [model] [rounds] [extra bench args]
NAME=$1; A=$2; B=$3; MODEL=${4:-resnet50}; ROUNDS=${5:-3}; EXTRA=${6:-}
for r in $(seq $ROUNDS); do for v in $A $B; do
  env $NAME=$v timeout -k 10 300 python bench.py --model $MODEL --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary $EXTRA 2>/dev/null | grep "^{" | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$NAME=$v', '$MODEL', r['ms_per_step'])"
done; done
