#!/bin/bash
cd "$(dirname "$0")/.."
run() { env "$@" python bench.py --task_num $T --steps $N --warmup 5 --no_cpu_baseline --roofline_steps 0 --extra_steps 0 --e2e_steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('T=$T $*', d['ms_per_step'])"; }
for rep in 1 2 3; do
  T=32 N=20 run GM_GEMM_FUSED_ROUNDS=2; T=32 N=20 run GM_GEMM_FUSED_ROUNDS=3
  T=16 N=30 run GM_GEMM_FUSED_ROUNDS=2; T=16 N=30 run GM_GEMM_FUSED_ROUNDS=3
done | tee gpurun_out/r06_run19.txt
