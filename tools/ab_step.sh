#!/bin/bash
# Same-box A/B of one environment knob on the whole step: tools/ab_step.sh VAR valueA valueB [rounds]   (bf16, interleaved runs)
VAR=$1; A=$2; B=$3; R=${4:-3}
for r in $(seq $R); do
  for v in "$A" "$B"; do
    if [ "$v" = "unset" ]; then unset $VAR; else export $VAR="$v"; fi
    python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary --dtype bf16 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$VAR=$v', d['ms_per_step'])"
  done
done
