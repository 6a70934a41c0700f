#!/bin/bash
mkdir -p gpurun_out/r05
O=gpurun_out/r05/regionF_queues.txt
: > $O
for b in "16 2" "32 2" "64 1" "64 2" "8 4" "8 8" "4 8"; do python scratch/r05_raw_vg_rate.py 4096 16 $b 16 exact >> $O 2>&1; done
for q in 4 8 16; do
  for r in 16 50; do
    echo "== GPU_MAX_HW_QUEUES=$q restarts $r" >> $O
    GPU_MAX_HW_QUEUES=$q python scratch/r04_regionF_only.py $r >> $O 2>&1
  done
done
echo "== one at a time, restarts 50" >> $O
GPEMU_COMPONENTS_PER_SLOT=1 python scratch/r04_regionF_only.py 50 >> $O 2>&1
grep -E "^==|^N=|cli phases|cli_wall" $O | cut -c1-200
