#!/bin/bash
# round-5 evidence, call A: N=4096 by-class tables (likelihood batch of 64 at d=8 and d=16, value+gradient batch of 16/64 at d=16),
# start-up phases of an emulator (C-ABI and CLI)
set -e
mkdir -p gpurun_out/r05
O=gpurun_out/r05
python scratch/r05_setup_phases.py > $O/setup_phases.txt 2>&1
python scratch/r05_batch_by_class.py 4096 8 64 1 0 0 > $O/n4096_d8_b64_lik.txt 2>&1
python scratch/r05_batch_by_class.py 4096 16 64 1 0 0 > $O/n4096_d16_b64_lik.txt 2>&1
python scratch/r05_batch_by_class.py 4096 16 16 1 0 1 > $O/n4096_d16_b16_vg.txt 2>&1
python scratch/r05_batch_by_class.py 4096 16 64 1 0 1 > $O/n4096_d16_b64_vg.txt 2>&1
python scratch/r05_batch_by_class.py 4096 8 16 1 0 1 > $O/n4096_d8_b16_vg.txt 2>&1
python scratch/r05_batch_by_class.py 8192 8 16 3 1 0 > $O/n8192_d8_b16_lik.txt 2>&1
python scratch/r03_n4096_split.py > $O/n4096_split_eventtimes.txt 2>&1
python scratch/r05_cli_setup_trace.py > $O/cli_setup_trace.txt 2>&1
echo done A
