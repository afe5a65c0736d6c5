#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
bash tools/refresh_profiles.sh r06_z > gpurun_out/r06_z_refresh.log 2>&1
tail -8 gpurun_out/r06_z_refresh.log | cut -c1-200
cd "$GRAFT_REPO_ROOT"
for i in 1 2 3; do python bench.py --config syn0 --no_cpu_baseline --extra_steps 0 --e2e_steps 0 --roofline_steps 0 --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('syn0', d['ms_per_step'])"; done
