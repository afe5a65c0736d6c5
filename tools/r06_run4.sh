#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r06_run4; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
GM_TIMING=1 python tools/extract_prof.py > $O/extract_prof.txt 2>&1
grep -v "gm timing" $O/extract_prof.txt | sed -n 2,2p
python bench.py --no_cpu_baseline > $O/bench_arxiv.json 2> $O/bench_arxiv.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_run4/bench_arxiv.json').read().strip().splitlines()[-1])
print('ms', d['ms_per_step'], 'e2e', d['end_to_end'], 'extraction', {k:d['extraction'][k] for k in ('k_nodes_ms','k_fill_ms','finalize_span_ms','host_wall_ms_per_meta_batch','frac')})
c=d['extra']['cone+hoist_z1']; print('cone+hoist', c['ms_per_step'], c.get('end_to_end'), c.get('end_to_end_one_builder'))
PY
run() { env "$@" python bench.py --task_num $T --steps $N --warmup 5 --no_cpu_baseline --roofline_steps 0 --extra_steps 0 --e2e_steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('T=$T $*', d['ms_per_step'])"; }
for rep in 1 2; do
  T=32 N=20 run GM_NOP=0; T=32 N=20 run GM_GEMM_SPLIT_GRID=240
  T=16 N=30 run GM_NOP=0; T=16 N=30 run GM_GEMM_SPLIT_GRID=240
  T=8 N=30 run GM_NOP=0; T=8 N=30 run GM_GEMM_SPLIT_GRID=240; T=8 N=30 run GM_GEMM_SPLIT_GRID=248
done | tee $O/grid.txt
