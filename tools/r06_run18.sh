#!/bin/bash
cd "$(dirname "$0")/.."
run() { env "$@" python bench.py --steps 20 --warmup 5 --no_cpu_baseline --roofline_steps 0 --extra_steps 0 --e2e_steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$*', d['ms_per_step'])"; }
for rep in 1 2; do
  run GM_NOP=0; run GM_GEMM_FUSED_ROUNDS=1; run GM_GEMM_FUSED_ROUNDS=3; run GM_GEMM_FUSED_ROUNDS=4; run GM_GEMM_PLAIN_ROUNDS=2; run GM_AGG_UNR=22; run GM_AGG_UNR=44
done | tee gpurun_out/r06_run18.txt
