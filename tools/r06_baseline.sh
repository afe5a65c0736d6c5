#!/bin/bash
# round-6 baseline on one box: default bench line, host timing of a batch build, 4-task shard with grid caps
cd "$(dirname "$0")/.."
O=gpurun_out/r06_base; mkdir -p $O
python bench.py > $O/bench_arxiv.json 2> $O/bench_arxiv.err
tail -c 3000 $O/bench_arxiv.json
GM_TIMING=1 python tools/extract_prof.py > $O/extract_prof.txt 2>&1
for g in 0 224 192 160 128; do for rep in 1 2; do
  GM_GEMM_SPLIT_GRID=$g python bench.py --task_num 4 --steps 40 --warmup 5 --no_cpu_baseline --roofline_steps 0 --extra_steps 0 --e2e_steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('t4 grid=$g', d['ms_per_step'])"
done; done | tee $O/t4_grid.txt
