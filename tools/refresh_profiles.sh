#!/bin/bash
# Regenerates the rocprofv3 evidence for the current kernels on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 2400 -- 'bash tools/refresh_profiles.sh rNN_x'
# then copy gpurun_out/<tag>_* into profiles/.  Counter passes run on their own (no trace flags), as the pool requires.
set -u
tag=${1:-final}
out=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="env GMETA_NO_BOX=1 python bench.py --warmup 1 --no_cpu_baseline --extra_steps 0 --e2e_steps 0"
db() { find "$1" -name '*.db' | head -1; }
timeout 600 rocprofv3 --kernel-trace --stats -d $out/p_ser -o x -- $B --serialize 1 --steps 3 > $out/${tag}_serialized_bench.log 2>&1
python tools/prof_summary.py "$(db $out/p_ser)" > $out/${tag}_serialized_kernel_stats.txt
timeout 600 rocprofv3 --kernel-trace --stats -d $out/p_two -o x -- $B --steps 3 > $out/${tag}_two_stream_bench.log 2>&1
python tools/prof_summary.py "$(db $out/p_two)" > $out/${tag}_two_stream_kernel_stats.txt
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $out/p_f -o x -- $B --serialize 1 --steps 1 --roofline_steps 0 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $out/p_w -o x -- $B --serialize 1 --steps 1 --roofline_steps 0 > /dev/null 2>&1
python tools/pmc_summary.py "$(db $out/p_f)" > $out/${tag}_pmc_fetch_size.txt
python tools/pmc_summary.py "$(db $out/p_w)" > $out/${tag}_pmc_write_size.txt
python tools/pmc_summary.py --agg-traffic "$(db $out/p_f)" "$(db $out/p_w)" "${tag} $(date +%Y-%m-%d)" > $out/${tag}_agg_traffic.json
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \
    -d $out/p_sq -o x -- $B --serialize 1 --steps 1 --roofline_steps 0 > /dev/null 2>&1
python tools/pmc_summary.py "$(db $out/p_sq)" > $out/${tag}_pmc_sq.txt
grep -h '^{' $out/${tag}_serialized_bench.log $out/${tag}_two_stream_bench.log | cut -c1-400
head -6 $out/${tag}_serialized_kernel_stats.txt | cut -c1-150
cat $out/${tag}_agg_traffic.json
rm -rf $out/p_ser $out/p_two $out/p_f $out/p_w $out/p_sq      # only the summaries travel back (gpurun merges <= 64 MiB)
