# rocprofv3 --kernel-trace --stats of the likelihood region alone, ONE context: the per-launch durations of the
# dominant kernel (gemm_nt_kernel<128,128,4,4,2>, grid_y = 16) are what bench.py's roofline.avg_launch_us must agree with
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r02_final
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kst2
timeout -k 10 300 rocprofv3 --kernel-trace --stats -f csv rocpd -d /tmp/kst2 -o r -- python3 $R/bench.py --no-cpu-baseline --streams 1 --no-grad --no-predict --no-single > $R/gpurun_out/r02_final/bench_rocprof_lik_only.json 2> $R/gpurun_out/r02_final/rocprof2.err
find /tmp/kst2 -name '*kernel_stats.csv' -exec cp {} $R/gpurun_out/r02_final/kernel_stats_lik_only_streams1.csv \;
python3 $R/tools/rocpd_summary.py $(find /tmp/kst2 -name '*.db') --grid-y 16 > $R/gpurun_out/r02_final/kernel_stats_lik_only_streams1_batch16.txt
cat $R/gpurun_out/r02_final/kernel_stats_lik_only_streams1_batch16.txt
python3 -c "
import json;b=json.loads(open('$R/gpurun_out/r02_final/bench_rocprof_lik_only.json').read().strip().splitlines()[-1]);print(b['value'],b['roofline'])"
