#!/bin/bash
# round-5 call B: host-layer GPU tests, then region F of bench.py (configs[3] trained through the CLI) under the component
# pool: all components side by side against one at a time
mkdir -p gpurun_out/r05
O=gpurun_out/r05
python -m pytest tests/test_host_api.py tests/test_abi.py tests/test_interactive_io.py -m gpu -x -q > $O/hostapi_gpu.txt 2>&1
tail -5 $O/hostapi_gpu.txt
for cps in 0 1 2 4; do
  echo "== GPEMU_COMPONENTS_PER_SLOT=$cps (0 = by memory, up to 8)" >> $O/regionF_components_per_slot.txt
  GPEMU_COMPONENTS_PER_SLOT=$cps python scratch/r04_regionF_only.py 16 >> $O/regionF_components_per_slot.txt 2>&1
done
echo "== restarts 50, all side by side" >> $O/regionF_components_per_slot.txt
python scratch/r04_regionF_only.py 50 >> $O/regionF_components_per_slot.txt 2>&1
grep -E "^==|cli phases|wall_seconds|search stats" $O/regionF_components_per_slot.txt | cut -c1-400
