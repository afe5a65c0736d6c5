#!/bin/bash
# full GPU suite + default bench + smoke after the build/teardown work (slab cache, chained scans, builder-paced host halves)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06_run22; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -x -q > $out/tests.txt 2>&1; tail -4 $out/tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > $out/bench.txt 2>&1; tail -1 $out/bench.txt > $out/bench.json; python - <<'PY'
import json
d = json.load(open('gpurun_out/r06_run22/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['box']['split_gemm']['fp32_equivalent_tflops'], d['box']['hbm_copy']['GBps'])
print('e2e', d['end_to_end']['ms_per_step'], d['end_to_end']['meta_tasks_per_s'])
x = d['extraction']; print('extraction', x['frac'], x['finalize_span_ms'], x['host_wall_ms_per_meta_batch'], x.get('host_wall_ms_two_calls'), x['k_nodes_ms'], x['k_fill_ms'])
c = d['extra']['cone+hoist_z1']; print('cone', c['ms_per_step'], c['end_to_end'])
PY
