#!/bin/bash
# cone-table build after the eight-lanes-per-row k_out_edges: cone tests, build-only kernel stats, bench line
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06_run21; mkdir -p $out
db() { find "$1" -name '*.db' | head -1; }
timeout 900 python -m pytest tests -m gpu -x -q -k "cone" > $out/tests.txt 2>&1; tail -3 $out/tests.txt
L=2
timeout 600 rocprofv3 --kernel-trace --stats -d $out/p$L -o x -- python tools/build_prof.py $L > $out/build_L$L.log 2>&1
python tools/prof_summary.py "$(db $out/p$L)" > $out/build_L${L}_kernel_stats.txt
grep "^build" $out/build_L$L.log; head -22 $out/build_L${L}_kernel_stats.txt | cut -c1-130; tail -1 $out/build_L${L}_kernel_stats.txt
rm -rf $out/p$L
timeout 600 python bench.py > $out/bench.txt 2>&1; tail -1 $out/bench.txt | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
print(json.dumps(d['extra'].get('cone+hoist_z1'))[:1500])
print(json.dumps(d.get('extraction'))[:800])
"
