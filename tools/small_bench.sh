#!/bin/bash
# ms per meta-step of the small BASELINE configs and the 4-task arxiv shard (two runs each): what the latency work on k_head_loss / the host prologue is judged by
cd "$(dirname "$0")/.."
for c in firstmm tissue; do for i in 1 2; do
    python bench.py --config $c --steps 100 --warmup 10 --no_cpu_baseline --roofline_steps 0 --extra_steps 0 --e2e_steps 0 --no_eval 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$c', d['ms_per_step'])"
done; done
for i in 1 2; do
    python bench.py --task_num 4 --steps 40 --warmup 5 --no_cpu_baseline --roofline_steps 0 --extra_steps 0 --e2e_steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('arxiv-4-tasks', d['ms_per_step'])"
done
