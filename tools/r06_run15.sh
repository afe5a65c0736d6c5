#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r06_run15; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -rs > $O/pytest.txt 2>&1; tail -12 $O/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
