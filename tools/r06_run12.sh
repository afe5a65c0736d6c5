#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r06_run12; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -k "cone or configs or parity or round4 or fuzz" > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
timeout 600 python bench.py --no_cpu_baseline --roofline_steps 0 --steps 10 > $O/bench_arxiv.json 2> $O/bench_arxiv.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_run12/bench_arxiv.json').read().strip().splitlines()[-1])
c=d['extra']['cone+hoist_z1']; print('arxiv', d['ms_per_step'], 'cone+hoist', c['ms_per_step'], c.get('end_to_end'))
PY
