#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r06_run16; mkdir -p $O
S=$(date +%s.%N); python bench.py > $O/bench_default.json 2> $O/bench_default.err; E=$(date +%s.%N); echo "bench.py default wall: $(echo "$E - $S" | bc) s"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_run16/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['box']['split_gemm']['fp32_equivalent_tflops'], d['box']['hbm_copy']['GBps'], d['cpu_baseline']['value'])
print(list(d.keys()))
PY
