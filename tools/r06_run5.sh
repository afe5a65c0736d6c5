#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r06_run5; mkdir -p $O
timeout 1200 python -m pytest tests/test_hip_round6.py tests/test_hip_round4.py tests/test_hip_fullsize.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
run() { env "$@" python bench.py --task_num $T --steps $N --warmup 5 --no_cpu_baseline --roofline_steps 0 --extra_steps 0 --e2e_steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('T=$T $*', d['ms_per_step'])"; }
for rep in 1 2; do
  T=32 N=20 run GM_FUSE_DIFF=0; T=32 N=20 run GM_FUSE_DIFF=1; T=32 N=20 run GM_FUSE_DIFF=1 GM_AGG_STREAM=0
  T=4 N=40 run GM_FUSE_DIFF=0; T=4 N=40 run GM_FUSE_DIFF=1; T=4 N=40 run GM_FUSE_DIFF=1 GM_AGG_STREAM=0
done | tee $O/fuse_diff_ab.txt
