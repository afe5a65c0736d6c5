#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r06_run11; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; tail -8 $O/pytest.txt
