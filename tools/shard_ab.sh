#!/bin/bash
# Same-box A/B of a knob on the task shards one GPU runs at N = 2 / 4 / 8 (16 / 8 / 4 tasks of the arxiv shape): ms per meta-step, alternating runs.
#   bash tools/shard_ab.sh GM_AGG_STREAM_MIN_ROWS 32768 100000
cd "$(dirname "$0")/.."
knob=$1; shift
for t in 16 8 4; do for rep in 1 2; do for v in "$@"; do
    env $knob=$v python bench.py --task_num $t --steps 30 --warmup 4 --no_cpu_baseline --roofline_steps 0 --extra_steps 0 --e2e_steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('tasks $t $knob=$v', d['ms_per_step'])"
done; done; done
