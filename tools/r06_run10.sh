#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r06_run10; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
for c in firstmm tissue syn0; do
  timeout 600 python bench.py --config $c --no_cpu_baseline > $O/bench_$c.json 2> $O/bench_$c.err
done
timeout 600 python bench.py --task_num 4 --no_cpu_baseline > $O/bench_t4_shard.json 2> $O/bench_t4.err
timeout 900 python bench.py --no_cpu_baseline > $O/bench_arxiv.json 2> $O/bench_arxiv.err
python - <<'PY'
import json
for c in ['firstmm','tissue','syn0','t4_shard','arxiv']:
    try:
        d=json.loads(open('gpurun_out/r06_run10/bench_%s.json'%c).read().strip().splitlines()[-1])
    except Exception as e:
        print(c,'FAILED',e); continue
    ex=d.get('extra',{})
    print(c, d['ms_per_step'], 'frac', d['roofline']['frac'], 'e2e', d.get('end_to_end',{}).get('ms_per_step'), 'extr', {k:d['extraction'][k] for k in ('k_nodes_ms','k_fill_ms','finalize_span_ms','host_wall_ms_per_meta_batch','host_wall_ms_one_thread','frac')} if d.get('extraction') else None)
    print('    extra', {k:v.get('ms_per_step') for k,v in ex.items() if isinstance(v,dict)}, ex.get('cone+hoist_z1',{}).get('end_to_end'))
    print('    box', d.get('box'))
PY
