#!/bin/bash
mkdir -p gpurun_out/r05
O=gpurun_out/r05
S=$O/leaf_tpw_ab.txt; : > $S
python scratch/r05_env_ab_n4096.py GPEMU_LEAF_TPW 1 2 4 0 >> $S 2>&1
python scratch/r04_env_ab.py GPEMU_LEAF_TPW 1 2 4 0 >> $S 2>&1
cat $S
python -m pytest tests -m gpu -x -q > $O/gpu_suite_3.txt 2>&1
tail -4 $O/gpu_suite_3.txt
