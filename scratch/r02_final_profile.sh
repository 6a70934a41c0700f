# round 2: the judged bench line, then the same command (without the CPU legs) under rocprofv3 --kernel-trace --stats,
# with the default two contexts and with one (per-launch durations comparable with bench.py's HIP-event roofline)
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r02_final
timeout -k 10 600 python3 $R/bench.py > $R/gpurun_out/r02_final/bench.json 2> $R/gpurun_out/r02_final/bench.err
tail -c 300 $R/gpurun_out/r02_final/bench.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kst /tmp/kst1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -f csv rocpd -d /tmp/kst -o r -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/r02_final/bench_rocprof.json 2> $R/gpurun_out/r02_final/rocprof.err
find /tmp/kst -name '*kernel_stats.csv' -exec cp {} $R/gpurun_out/r02_final/kernel_stats.csv \;
python3 $R/tools/rocpd_summary.py $(find /tmp/kst -name '*.db') > $R/gpurun_out/r02_final/kernel_stats_all.txt || true
python3 $R/tools/rocpd_summary.py $(find /tmp/kst -name '*.db') --grid-y 16 > $R/gpurun_out/r02_final/kernel_stats_batch16.txt || true
timeout -k 10 300 rocprofv3 --kernel-trace --stats -f csv rocpd -d /tmp/kst1 -o r -- python3 $R/bench.py --no-cpu-baseline --streams 1 > $R/gpurun_out/r02_final/bench_rocprof_streams1.json 2> $R/gpurun_out/r02_final/rocprof1.err
find /tmp/kst1 -name '*kernel_stats.csv' -exec cp {} $R/gpurun_out/r02_final/kernel_stats_streams1.csv \;
python3 $R/tools/rocpd_summary.py $(find /tmp/kst1 -name '*.db') --grid-y 16 > $R/gpurun_out/r02_final/kernel_stats_streams1_batch16.txt
head -12 $R/gpurun_out/r02_final/kernel_stats_streams1_batch16.txt
