#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r06_run14; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -k "extraction or large_graph or replay or fuzz or parity or configs or dataloader or prefix or fullsize or joint or round6" > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
timeout 600 python bench.py --no_cpu_baseline --extra_steps 0 --roofline_steps 0 --steps 5 --e2e_steps 0 > $O/bench_arxiv.json 2> $O/bench_arxiv.err
for c in tissue firstmm; do timeout 600 python bench.py --config $c --no_cpu_baseline --extra_steps 0 --roofline_steps 0 --steps 5 --e2e_steps 0 --no_eval > $O/bench_$c.json 2> $O/bench_$c.err; done
python - <<'PY'
import json
for c in ['arxiv','tissue','firstmm']:
    d=json.loads(open('gpurun_out/r06_run14/bench_%s.json'%c).read().strip().splitlines()[-1])
    print(c, {k:d['extraction'][k] for k in ('k_nodes_ms','k_fill_ms','finalize_span_ms','host_wall_ms_per_meta_batch','host_wall_ms_two_calls','frac')})
PY
