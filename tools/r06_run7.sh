#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06_run7; mkdir -p $out
timeout 600 python -m pytest tests/test_hip_round6.py -m gpu -x -q -k "fused" 2>&1 | tail -3
B="python bench.py --warmup 1 --no_cpu_baseline --extra_steps 0 --e2e_steps 0 --roofline_steps 0"
db() { find "$1" -name '*.db' | head -1; }
for v in 1; do
  GM_FUSE_DIFF=$v GMETA_NO_BOX=1 timeout 600 rocprofv3 --kernel-trace --stats -d $out/p_ser$v -o x -- $B --serialize 1 --steps 3 > $out/fuse_diff${v}_serialized_bench.log 2>&1
  python tools/prof_summary.py "$(db $out/p_ser$v)" > $out/fuse_diff${v}_serialized_kernel_stats.txt
  head -8 $out/fuse_diff${v}_serialized_kernel_stats.txt | cut -c1-170
  rm -rf $out/p_ser$v
done
run() { env "$@" python bench.py --task_num $T --steps $N --warmup 5 --no_cpu_baseline --roofline_steps 0 --extra_steps 0 --e2e_steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('T=$T $*', d['ms_per_step'])"; }
for rep in 1 2; do
  T=32 N=20 run GM_FUSE_DIFF=0; T=32 N=20 run GM_FUSE_DIFF=1
  T=4 N=40 run GM_FUSE_DIFF=0; T=4 N=40 run GM_FUSE_DIFF=1
done | tee $out/fuse_diff_ab.txt
