#!/bin/bash
for rep in 1 2; do
for prio in 0 1; do
  for st in 4; do
    echo -n "prio=$prio streams=$st: "
    GPEMU_LEAF_PRIO=$prio python bench.py --no-cpu-baseline --no-predict --streams $st --steps 40 --warmup 4 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print(j['value'], j['ms_per_step'])"
  done
done
done
