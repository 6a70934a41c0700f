# rocprofv3 --kernel-trace --stats of bench.py with ONE context (no concurrent kernels from a second stream, so the
# profiler's per-kernel durations can be compared with the HIP-event durations bench.py's roofline uses)
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/final1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kst1
timeout -k 10 400 rocprofv3 --kernel-trace --stats -f csv rocpd -d /tmp/kst1 -o r -- python3 $R/bench.py --no-cpu-baseline --streams 1 > $R/gpurun_out/final1/bench_rocprof.json 2> $R/gpurun_out/final1/rocprof.err
find /tmp/kst1 -name '*kernel_stats.csv' -exec cp {} $R/gpurun_out/final1/kernel_stats.csv \;
python3 $R/tools/rocpd_summary.py $(find /tmp/kst1 -name '*.db') --grid-y 16 > $R/gpurun_out/final1/kernel_stats_batch16.txt
cat $R/gpurun_out/final1/kernel_stats_batch16.txt
python3 -c "
import json;b=json.loads(open('$R/gpurun_out/final1/bench_rocprof.json').read().strip().splitlines()[-1]);print(b['value'],b['roofline'])"
