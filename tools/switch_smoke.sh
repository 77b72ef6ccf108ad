#!/bin/bash
# on the GPU box: every A/B switch at each of its values through a short bench run (does the path still run, what does it cost)
run() { env "$@" timeout -k 10 200 python bench.py --model ${MODEL:-resnet50} --dtype ${DT:-bf16} --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary 2>&1 | grep "^{" | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%-40s %s %s %.3f ms loss %.4f' % ('$*', '${MODEL:-resnet50}', '${DT:-bf16}', r['ms_per_step'], r['config']['final_loss']))" || echo "$* FAILED"; }
for kv in MI355_PO=0 MI355_PO=2 MI355_PO64=0 MI355_PO64=2 MI355_DCONV=0 MI355_WG3=0 MI355_PK=0 MI355_PW=0 MI355_DS_COMPACT=0 MI355_BN_FIN_WIDE=0 MI355_BN_FIN_WIDE=4 MI355_WGRAD_STREAM=0 MI355_FUSE_BN_BWD=0; do run $kv; done
DT=fp8 run MI355_FP8_PLAN=rule; DT=fp8 run MI355_FP8_WGRAD=0; DT=fp8 run MI355_DS_COMPACT=0; DT=fp32 run MI355_STREAM_K=0
for kv in MI355_BRESNET_ECA_SUMS=0 MI355_BRESNET_STEM_IM2COL=0 MI355_BRESNET_FUSED_ECA=0 MI355_BRESNET_LAZY_BN=0 MI355_BRESNET_BITS=0 MI355_POOL_SEG=1 MI355_DCONV=0; do MODEL=bresnet50 run $kv; done
