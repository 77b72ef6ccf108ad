#!/bin/bash
# On the GPU box: BASELINE configs[4] shapes — bs 512 at 160 / 224 / 320 px, bf16 and fp8 — step time (overlapped, default) and the
# per-conv-launch table of a serial step -> gpurun_out/<tag>/
TAG=${1:-config5}
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?not on a gpurun box}"
OUT=gpurun_out/$TAG
mkdir -p $OUT
for S in 160 224 320; do
  for dt in bf16 fp8; do
    python3 bench.py --batch 512 --size $S --dtype $dt --steps 10 --warmup 4 --no-cpu-baseline --no-secondary > $OUT/bench_${dt}_$S.json 2>> $OUT/err.log
    MI355_WGRAD_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_${dt}_$S -- python3 bench.py --batch 512 --size $S --dtype $dt --steps 5 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline > /dev/null 2>> $OUT/err.log
    python3 tools/trace_layers.py $OUT/trace_${dt}_$S/* 512 $S > $OUT/conv_per_layer_${dt}_$S.txt 2>> $OUT/err.log
    rm -rf $OUT/trace_${dt}_$S
  done
done
python3 - <<'PY'
import json, glob, os
out = os.environ.get("OUT", "")
PY
for f in $OUT/bench_*.json; do python3 -c "
import json,sys
r=json.loads(open('$f').read().strip().splitlines()[-1])
rf=r.get('roofline') or {}
print('%-22s %8.1f img/s %8.3f ms/step  step_tflops %7.1f  roofline frac %s serial %s  3x3 %s' % (os.path.basename('$f') if False else '$f'.split('/')[-1], r['value'], r['ms_per_step'], r['config']['step_tflops'], rf.get('frac'), rf.get('serial_frac'), (rf.get('conv3x3') or {}).get('serial_frac')))
"; done | tee $OUT/summary.txt
tail -n 1 $OUT/conv_per_layer_*.txt
