#!/bin/bash
# sweep the outer panel width and the number of concurrent evaluation contexts
for nb in 256 512 1024; do
  for st in 1 4 6 8; do
    echo -n "nb=$nb streams=$st: "
    GPEMU_NB_TOP=$nb python bench.py --no-cpu-baseline --no-predict --streams $st --steps 24 --warmup 4 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print(j['value'], j['ms_per_step'])"
  done
done
