#!/bin/bash
# Final round-6 evidence on ONE box (tag r06_f): everything tools/r06_final.sh takes + the kernel stats of a meta-batch build without / with receptive-field tables
set -u
cd "$GRAFT_REPO_ROOT"
tag=${1:-r06_f}
bash tools/r06_final.sh $tag
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
db() { find "$1" -name '*.db' | head -1; }
for L in 0 2; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $out/pb$L -o x -- python tools/build_prof.py $L > $out/${tag}_build_L$L.log 2>&1
  python tools/prof_summary.py "$(db $out/pb$L)" > $out/${tag}_build_L${L}_kernel_stats.txt
  rm -rf $out/pb$L
  PHASES=1 timeout 200 python tools/build_prof.py $L 2>&1 | tail -8 > $out/${tag}_build_L${L}_host_phases.txt
  tail -1 $out/${tag}_build_L${L}_kernel_stats.txt; cat $out/${tag}_build_L${L}_host_phases.txt
done
timeout 200 python tools/e2e_phases.py 1 2 > $out/${tag}_e2e_phases.txt 2>&1; grep -v "^  " $out/${tag}_e2e_phases.txt
timeout 200 python tools/pool_free_probe.py > $out/${tag}_pool_free_probe.txt 2>&1; tail -4 $out/${tag}_pool_free_probe.txt
