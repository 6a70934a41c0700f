#!/bin/bash
# GPEMU_STAGGER_US A/B with the bench's own regions (two contexts, batches of 16; predictions; value+gradient), same box, A B A B
# usage: bash scratch/r04_stagger_bench_ab.sh [us]   -> gpurun_out/stagger_bench_ab.txt
US=${1:-20}
out=gpurun_out/stagger_bench_ab.txt
: > $out
for rep in 1 2; do
  for us in 0 $US; do
    GPEMU_STAGGER_US=$us timeout -k 10 300 python bench.py --no-cpu-baseline --no-interactive --no-train --no-pca8 --no-single > gpurun_out/_sb.json 2> gpurun_out/_sb.err || { echo "bench failed" >> $out; exit 1; }
    python - $us >> $out <<'PY'
import json, sys
j = json.loads(open("gpurun_out/_sb.json").read().strip().splitlines()[-1])
print("stagger %3s us: value %.1f evals/s  predictions %.0f /s  value+gradient %.1f /s  dominant kernel %.3f of peak" % (
    sys.argv[1], j["value"], j["predictions"]["value"], j["value_grad"]["value"], j["roofline"]["frac"]))
PY
  done
done
cat $out
