#!/bin/bash
# likelihood region against the number of contexts at batch 16 (round 3)
for v in "16 1" "16 2" "16 3" "16 4" "8 4"; do
  set -- $v
  python bench.py --steps 24 --batch $1 --streams $2 --no-predict --no-single --no-grad --no-train --no-pca8 --no-cpu-baseline 2>/dev/null \
    | python -c "import json,sys; j=json.load(sys.stdin); print('batch=$1 streams=$2', round(j['value'],2), 'evals/s', round(j['ms_per_evaluation'],3), 'ms/eval')"
done
