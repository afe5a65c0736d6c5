#!/bin/bash
cd "$(dirname "$0")/.."
out=gpurun_out/r06_run8; mkdir -p $out
timeout 300 python -m pytest tests/test_hip_round6.py -m gpu -x -q -k "fused" 2>&1 | tail -3
GMETA_HIP_LIB=$PWD/g-meta_amd/libgmeta_hip_late1.so timeout 300 python -m pytest tests/test_hip_round6.py -m gpu -x -q -k "fused" 2>&1 | tail -3
run() { env "$@" timeout 300 python bench.py --task_num $T --steps $N --warmup 3 --no_cpu_baseline --roofline_steps 2 --extra_steps 0 --e2e_steps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); m=d['mfma']; r=d['roofline']
print('T=$T $*', d['ms_per_step'], 'one-stream', d['two_queues']['ms_per_step_one_stream'], 'wgrad', m['wgrad']['split_bf16']['ms_per_step'], 'gemm', m['gemm']['split_bf16']['ms_per_step'], 'agg', round(r['avg_launch_ms']*r['launches_measured']/2,3), 'frac', r['frac'], r['strict_hbm_frac'])"; }
L1=GMETA_HIP_LIB=$PWD/g-meta_amd/libgmeta_hip_late1.so
for rep in 1 2; do
  T=32 N=10 run GM_FUSE_DIFF=0; T=32 N=10 run GM_FUSE_DIFF=1; T=32 N=10 run GM_FUSE_DIFF=1 $L1
done | tee $out/fuse_diff_ab.txt
for rep in 1 2; do
  T=4 N=40 run GM_FUSE_DIFF=0; T=4 N=40 run GM_FUSE_DIFF=1; T=4 N=40 run GM_FUSE_DIFF=1 $L1
done | tee -a $out/fuse_diff_ab.txt
