#!/bin/bash
# 4-task arxiv shard (what one GPU runs at N = 8): CU-masked streams and grid caps, never tried at this size before round 6
cd "$(dirname "$0")/.."
O=gpurun_out/r06_t4; mkdir -p $O
run() { env "$@" python bench.py --task_num 4 --steps 40 --warmup 5 --no_cpu_baseline --roofline_steps 0 --extra_steps 0 --e2e_steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$*', d['ms_per_step'])"; }
for rep in 1 2; do
  run GM_NOP=0
  run GM_GEMM_SPLIT_GRID=224
  run GM_GEMM_SPLIT_GRID=240
  run GM_CU_MASK_SUPPORT=4
  run GM_CU_MASK_SUPPORT=8
  run GM_CU_MASK_SUPPORT=12
  run GM_CU_MASK_SUPPORT=16
  run GM_GEMM_FUSED_ROUNDS=2
  run GM_GEMM_FUSED_ROUNDS=6
  run GM_GEMM_FUSED_ROUNDS=8
  run GM_QUERY_STREAMS=2
  run GM_AGG_STREAM_MIN_ROWS=30000
done | tee $O/sweep.txt
