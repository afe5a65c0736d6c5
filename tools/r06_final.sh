#!/bin/bash
# Round-6 evidence on ONE box: rocprofv3 summaries (tools/refresh_profiles.sh), the bench lines of every config, the two-stream kernel timelines of the small shapes
set -u
cd "$GRAFT_REPO_ROOT"
tag=${1:-r06_z}
bash tools/refresh_profiles.sh $tag > gpurun_out/${tag}_refresh.log 2>&1
tail -12 gpurun_out/${tag}_refresh.log | cut -c1-300
cd "$GRAFT_REPO_ROOT"
timeout 900 python bench.py > gpurun_out/${tag}_bench_arxiv.json 2> gpurun_out/${tag}_bench_arxiv.err
timeout 600 python bench.py --task_num 4 --no_cpu_baseline > gpurun_out/${tag}_bench_t4_shard.json 2>/dev/null
for c in tissue firstmm syn0; do timeout 600 python bench.py --config $c --no_cpu_baseline > gpurun_out/${tag}_bench_$c.json 2>/dev/null; done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
db() { find "$1" -name '*.db' | head -1; }
for c in "--task_num 4:t4" "--config firstmm --no_eval:firstmm" "--config tissue --no_eval:tissue"; do
  args=${c%%:*}; name=${c##*:}
  timeout 600 rocprofv3 --kernel-trace -d gpurun_out/p_$name -o x -- python bench.py $args --steps 3 --warmup 1 --no_cpu_baseline --extra_steps 0 --e2e_steps 0 --roofline_steps 0 > /dev/null 2>&1
  python tools/step_kernels.py "$(db gpurun_out/p_$name)" all > gpurun_out/${tag}_${name}_two_stream_kernels.txt 2>&1
  rm -rf gpurun_out/p_$name
done
python - <<PY
import json
for c in ['arxiv','t4_shard','tissue','firstmm','syn0']:
    try:
        d=json.loads(open('gpurun_out/${tag}_bench_%s.json'%c).read().strip().splitlines()[-1])
        print(c, d['ms_per_step'], d['value'], 'frac', d['roofline']['frac'], d['roofline'].get('strict_hbm_frac'))
    except Exception as e: print(c, 'FAILED', e)
PY
