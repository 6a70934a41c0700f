# the judged bench line and the same command under rocprofv3 --kernel-trace --stats (two contexts / one / likelihood only)
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02c
mkdir -p $O
timeout -k 10 600 python3 $R/bench.py > $O/bench.json 2> $O/bench.err
tail -c 200 $O/bench.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kst /tmp/kst1 /tmp/kst2
timeout -k 10 300 rocprofv3 --kernel-trace --stats -f csv rocpd -d /tmp/kst -o r -- python3 $R/bench.py --no-cpu-baseline > $O/bench_rocprof.json 2> $O/rocprof.err
find /tmp/kst -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats.csv \;
python3 $R/tools/rocpd_summary.py $(find /tmp/kst -name '*.db') > $O/kernel_stats_all.txt || true
python3 $R/tools/rocpd_summary.py $(find /tmp/kst -name '*.db') --grid-y 16 > $O/kernel_stats_batch16.txt || true
echo "two-context profile done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -f csv rocpd -d /tmp/kst1 -o r -- python3 $R/bench.py --no-cpu-baseline --streams 1 > $O/bench_rocprof_streams1.json 2> $O/rocprof1.err
find /tmp/kst1 -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats_streams1.csv \;
python3 $R/tools/rocpd_summary.py $(find /tmp/kst1 -name '*.db') --grid-y 16 > $O/kernel_stats_streams1_batch16.txt
echo "one-context profile done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -f csv rocpd -d /tmp/kst2 -o r -- python3 $R/bench.py --no-cpu-baseline --streams 1 --no-grad --no-predict --no-single > $O/bench_rocprof_lik_only.json 2> $O/rocprof2.err
find /tmp/kst2 -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats_lik_only_streams1.csv \;
python3 $R/tools/rocpd_summary.py $(find /tmp/kst2 -name '*.db') --grid-y 16 > $O/kernel_stats_lik_only_streams1_batch16.txt
head -8 $O/kernel_stats_lik_only_streams1_batch16.txt
cd $R
timeout -k 10 300 python3 scratch/single_eval_cfg8.py > $O/single_eval_threshold_sweep.txt 2>&1
echo all done
