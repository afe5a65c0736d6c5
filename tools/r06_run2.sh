#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r06_run2; mkdir -p $O
GM_TIMING=1 python tools/extract_prof.py > $O/extract_prof.txt 2>&1
GM_AGG_STREAM=0 GM_TIMING=1 python tools/extract_prof.py > $O/extract_prof_nostream.txt 2>&1
for rep in 1 2; do for v in 1 0; do
  GM_AGG_STREAM=$v python bench.py --steps 20 --warmup 5 --no_cpu_baseline --extra_steps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('stream=$v', d['ms_per_step'], 'e2e', d['end_to_end']['ms_per_step'], 'roof', d['roofline']['frac'], 'host_wall', d['extraction']['host_wall_ms_per_meta_batch'], 'fin', d['extraction']['finalize_span_ms'], 'box', d.get('box'))"
done; done | tee $O/stream_ab.txt
