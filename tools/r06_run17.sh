#!/bin/bash
cd "$(dirname "$0")/.."
python -c "
import torch
print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')
for p in (-2,-1,0,1,2):
    try: print(p, torch.cuda.Stream(priority=p).priority)
    except Exception as e: print(p, 'err', e)
"
for rep in 1 2; do for pr in 0 -1 1; do
  GMETA_BUILD_PRIORITY=$pr python bench.py --steps 10 --warmup 3 --no_cpu_baseline --extra_steps 0 --roofline_steps 0 --e2e_steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('prio=$pr', d['ms_per_step'], 'e2e', d['end_to_end']['ms_per_step'])"
done; done
