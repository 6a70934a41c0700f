#!/bin/bash
# value+gradient region of bench.py under outer-panel widths / batch sizes (round 3): one JSON line per variant
for v in "0 16" "2048 16" "512 16" "0 32" "2048 32"; do
  set -- $v
  GPEMU_NB_TOP=$1 python bench.py --steps 4 --grad-steps 8 --grad-batch $2 --no-predict --no-single --no-train --no-pca8 --no-cpu-baseline 2>/dev/null \
    | python -c "import json,sys; j=json.load(sys.stdin); g=j['value_grad']; print('NB_TOP=$1 batch=$2', round(g['value'],2), 'evals/s', round(g['one_context_blocking_ms_per_batch'],1), 'ms blocking per batch', round(g['roofline']['frac'],3))"
done
