#!/bin/bash
# round-5 call E: the whole GPU suite, the start-up phases after the warm start / batched set-up, the default bench line
mkdir -p gpurun_out/r05
O=gpurun_out/r05
python -m pytest tests -m gpu -x -q --durations=15 > $O/gpu_suite_1.txt 2>&1
tail -25 $O/gpu_suite_1.txt
python scratch/r05_cli_setup_trace.py > $O/cli_setup_trace_after.txt 2>&1
grep -E "^==|setup trace|interactive stats" $O/cli_setup_trace_after.txt | cut -c1-260
timeout -k 10 600 python bench.py > $O/bench_1.json 2> $O/bench_1.err
tail -c 1500 $O/bench_1.json
