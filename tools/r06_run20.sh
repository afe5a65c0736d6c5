#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06_run20; mkdir -p $out
db() { find "$1" -name '*.db' | head -1; }
for L in 0 2; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $out/p$L -o x -- python tools/build_prof.py $L > $out/build_L$L.log 2>&1
  python tools/prof_summary.py "$(db $out/p$L)" > $out/build_L${L}_kernel_stats.txt
  grep "^build" $out/build_L$L.log; head -22 $out/build_L${L}_kernel_stats.txt | cut -c1-130; tail -1 $out/build_L${L}_kernel_stats.txt
  rm -rf $out/p$L
done
