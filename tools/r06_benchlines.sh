#!/bin/bash
# the bench lines of every config on one box (tag $1), no profiler
set -u
cd "$GRAFT_REPO_ROOT"
tag=${1:-r06_g}
timeout 900 python bench.py 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_arxiv.json
timeout 600 python bench.py --task_num 4 --no_cpu_baseline 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_t4_shard.json
for c in tissue firstmm syn0; do timeout 600 python bench.py --config $c --no_cpu_baseline 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_$c.json; done
python - <<PY
import json
for c in ['arxiv','t4_shard','tissue','firstmm','syn0']:
    d=json.load(open('gpurun_out/${tag}_bench_%s.json'%c))
    bx=d.get('box') or {}
    print(c, d['ms_per_step'], d['value'], 'frac', d['roofline']['frac'], 'box', (bx.get('hbm_copy') or {}).get('GBps'), (bx.get('split_gemm') or {}).get('fp32_equivalent_tflops'),
          'e2e', (d.get('end_to_end') or {}).get('ms_per_step'), 'cone e2e', ((d.get('extra') or {}).get('cone+hoist_z1') or {}).get('end_to_end',{}).get('ms_per_step') if isinstance((d.get('extra') or {}).get('cone+hoist_z1'), dict) else None)
PY
