#!/bin/bash
mkdir -p gpurun_out/r05
O=gpurun_out/r05
python -m pytest tests -m gpu -x -q -k "grad or golden_g6 or predict_setup_batch or multi" > $O/gpu_suite_4a.txt 2>&1
tail -4 $O/gpu_suite_4a.txt
grep -q failed $O/gpu_suite_4a.txt && exit 1
S=$O/grad_gram_ab.txt; : > $S
for cfg in "4096 16 16" "4096 8 16" "8192 8 16"; do
  set -- $cfg
  for g in 1 0; do echo "GPEMU_GRAD_GRAM=$g" >> $S; GPEMU_GRAD_GRAM=$g python scratch/r05_raw_vg_rate.py $1 $2 $3 2 12 exact >> $S 2>&1; done
done
cat $S
python -m pytest tests -m gpu -x -q > $O/gpu_suite_4.txt 2>&1
tail -4 $O/gpu_suite_4.txt
