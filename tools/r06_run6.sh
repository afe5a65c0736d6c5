#!/bin/bash
# serialised per-kernel stats with and without the fused differentiated passes (same box)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06_run6; mkdir -p $out
B="python bench.py --warmup 1 --no_cpu_baseline --extra_steps 0 --e2e_steps 0 --roofline_steps 0"
db() { find "$1" -name '*.db' | head -1; }
for v in 0 1; do
  GM_FUSE_DIFF=$v GMETA_NO_BOX=1 timeout 600 rocprofv3 --kernel-trace --stats -d $out/p_ser$v -o x -- $B --serialize 1 --steps 3 > $out/fuse_diff${v}_serialized_bench.log 2>&1
  python tools/prof_summary.py "$(db $out/p_ser$v)" > $out/fuse_diff${v}_serialized_kernel_stats.txt
  head -14 $out/fuse_diff${v}_serialized_kernel_stats.txt | cut -c1-170
  rm -rf $out/p_ser$v
done
