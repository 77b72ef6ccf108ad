#!/bin/bash
# on the GPU box: same build, alternates an environment switch (e.g. MI355_WGRAD_STREAM 0/1); prints ms/step
VAR=$1; DT=${2:-bf16}; ROUNDS=${3:-3}; LIB=${4:-$PWD/sota_imagenet_amd/lib/libmi355rn.so}
for r in $(seq $ROUNDS); do for v in 0 1; do
  export $VAR=$v
  MI355RN_LIB=$LIB timeout -k 10 300 python bench.py --steps 8 --warmup 3 --dtype $DT \
    --no-cpu-baseline --no-roofline --no-secondary 2>&1 | grep "^{" | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$VAR=$v', '$DT', r['ms_per_step'], r['config']['final_loss'])"
done; done
