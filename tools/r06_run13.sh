#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r06_run13; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -k "round5 or fullsize or round6 or configs" > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
timeout 600 python bench.py --no_cpu_baseline --extra_steps 0 > $O/bench_arxiv.json 2> $O/bench_arxiv.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_run13/bench_arxiv.json').read().strip().splitlines()[-1])
print('arxiv', d['ms_per_step'], 'frac', d['roofline']['frac'], d['roofline']['strict_hbm_frac'], 'e2e', d['end_to_end']['ms_per_step'], d['two_queues'])
print({k:d['extraction'][k] for k in ('k_nodes_ms','k_fill_ms','finalize_span_ms','host_wall_ms_per_meta_batch','host_wall_ms_one_thread','frac')})
print(d['box'])
PY
