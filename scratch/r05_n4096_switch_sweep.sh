#!/bin/bash
mkdir -p gpurun_out/r05
O=gpurun_out/r05
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "kvectors or config4 or config5 or predict_setup_batch" > $O/gpu_suite_2.txt 2>&1
tail -5 $O/gpu_suite_2.txt
S=$O/n4096_knob_sweep.txt; : > $S
python scratch/r05_env_ab_n4096.py GPEMU_NB_TOP 0 512 1024 2048 4096 >> $S 2>&1
python scratch/r05_env_ab_n4096.py GPEMU_GEMM_BIG_TILES 256 512 1024 2048 4096 1000000 >> $S 2>&1
python scratch/r05_env_ab_n4096.py GPEMU_STAGGER_US 0 10 20 40 >> $S 2>&1
python scratch/r05_env_ab_n4096.py GPEMU_GEMM_TABLE 0 4 8 16 >> $S 2>&1
python scratch/r05_env_ab_n4096.py GPEMU_FACTOR_AHEAD 1 0 >> $S 2>&1
python scratch/r05_env_ab_n4096.py GPEMU_SPLIT_RHS_ROWS 1 0 >> $S 2>&1
python scratch/r05_env_ab_n4096.py GPEMU_IDLE_WAVES 1 0 >> $S 2>&1
cat $S
