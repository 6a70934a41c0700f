#!/bin/bash
# which (seed, restarts) of the uni-2d-param example end in a snapshot interactive_mode accepts (the unbounded BFGS of the
# reference can walk the nugget to e^-33: a numerically singular model)
cd $GRAFT_REPO_ROOT
IN=tests/golden/ref_inputs/uni-2d-param.input_model_file.dat
Q=tests/golden/ref_inputs/uni-2d-param.sample_locations.dat
for restarts in 3 8; do for seed in 1 2 3 4 5 2718; do
  GPEMU_SEED=$seed GPEMU_RESTARTS=$restarts madaiemulator_amd/lib/interactive_emulator estimate_thetas $IN /tmp/M_$seed.dat --regression_order=1 > /tmp/train_$seed.log 2>&1
  best=$(grep "won with" /tmp/train_$seed.log | tail -1)
  if madaiemulator_amd/lib/interactive_emulator interactive_mode /tmp/M_$seed.dat -q < $Q > /tmp/out_$seed.txt 2> /tmp/err_$seed.txt; then s=ok; else s="FAILS: $(head -c 60 /tmp/err_$seed.txt)"; fi
  th=$(python3 -c "
import sys; sys.path.insert(0,'tests')
from test_host_api import parse_snapshot
print([round(float(v),3) for v in parse_snapshot(open('/tmp/M_$seed.dat').read().split())['models'][0]['thetas']])")
  echo "restarts $restarts seed $seed: $best thetas $th interactive_mode $s"
done; done
