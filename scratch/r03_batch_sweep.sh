#!/bin/bash
# likelihood region of bench.py against batch size x contexts (round 3)
for v in "16 2" "32 1" "32 2" "24 2" "48 1" "64 1"; do
  set -- $v
  python bench.py --steps 12 --batch $1 --streams $2 --no-predict --no-single --no-grad --no-train --no-pca8 --no-cpu-baseline 2>/dev/null \
    | python -c "import json,sys; j=json.load(sys.stdin); print('batch=$1 streams=$2', round(j['value'],2), 'evals/s', round(j['ms_per_evaluation'],3), 'ms/eval', 'potrf_whole', round(j['roofline_other']['potrf_whole']['frac'],3), 'big', round(j['roofline']['frac'],3))"
done
