# the round's judged numbers: default bench line, then the same command (without the CPU legs) under rocprofv3 --kernel-trace --stats
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/final
timeout -k 10 500 python3 $R/bench.py > $R/gpurun_out/final/bench.json 2> $R/gpurun_out/final/bench.err
tail -c 600 $R/gpurun_out/final/bench.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kst
timeout -k 10 400 rocprofv3 --kernel-trace --stats -f csv rocpd -d /tmp/kst -o r -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/final/bench_rocprof.json 2> $R/gpurun_out/final/rocprof.err
find /tmp/kst -name '*kernel_stats.csv' -exec cp {} $R/gpurun_out/final/kernel_stats.csv \;
python3 $R/tools/rocpd_summary.py $(find /tmp/kst -name '*.db') > $R/gpurun_out/final/kernel_stats_all.txt || true
python3 $R/tools/rocpd_summary.py $(find /tmp/kst -name '*.db') --grid-y 16 > $R/gpurun_out/final/kernel_stats_batch16.txt || true
head -8 $R/gpurun_out/final/kernel_stats_batch16.txt
