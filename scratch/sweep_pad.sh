#!/bin/bash
for rep in 1 2; do
for pad in 0 64; do
  for st in 1 4 6; do
    echo -n "pad=$pad streams=$st: "
    GPEMU_GEMM_LDS_PAD=$pad python bench.py --no-cpu-baseline --no-predict --streams $st --steps 36 --warmup 4 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print(j['value'], j['ms_per_step'])"
  done
done
done
