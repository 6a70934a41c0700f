# round 5, final tree: everything profiles/r05_* quotes beyond the experiment logs.  Two gpurun calls:
#   bash scratch/r05_profile.sh A     the default bench line, rocprofv3 kernel traces (likelihood region on one context = the
#                                     agreement check of `roofline`; the whole command minus the CPU / CLI legs)
#   bash scratch/r05_profile.sh B     FETCH_SIZE / WRITE_SIZE / SQ counter passes of two lock-step batches, the c2 and c5 workloads,
#                                     GEMM launches of a batch by K, fill time, value+gradient kernel stats (exact, N=4096 d=16)
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05r
mkdir -p $O
cd $R
if [ "$1" = "A" ]; then
  timeout -k 10 800 python3 bench.py > $O/bench.json 2> $O/bench.err
  tail -c 300 $O/bench.json; echo
  cd /tmp && export TMPDIR=/tmp
  rm -rf /tmp/kst /tmp/kst2
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -f csv rocpd -d /tmp/kst2 -o r -- python3 $R/bench.py --no-cpu-baseline --streams 1 --no-grad --no-predict --no-single --no-train --no-pca8 --no-interactive > $O/bench_rocprof_lik_only.json 2> $O/rocprof2.err
  find /tmp/kst2 -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats_lik_only_streams1.csv \;
  python3 $R/tools/rocpd_summary.py $(find /tmp/kst2 -name '*.db') --grid-y 16 > $O/kernel_stats_lik_only_streams1_batch16.txt
  head -8 $O/kernel_stats_lik_only_streams1_batch16.txt
  timeout -k 10 500 rocprofv3 --kernel-trace --stats -f csv rocpd -d /tmp/kst -o r -- python3 $R/bench.py --no-cpu-baseline --no-train --no-interactive > $O/bench_rocprof.json 2> $O/rocprof.err
  find /tmp/kst -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats.csv \;
  python3 $R/tools/rocpd_summary.py $(find /tmp/kst -name '*.db') > $O/kernel_stats_all.txt || true
  echo "A done"
else
  cd /tmp && export TMPDIR=/tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmf_$c
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmf_$c -o r -- python3 $R/scratch/one_batch.py > /tmp/logf_$c.txt 2>&1
  done
  python3 $R/tools/rocpd_pmc.py $(find /tmp/pmf_FETCH_SIZE -name '*.db') $(find /tmp/pmf_WRITE_SIZE -name '*.db') > $O/pmc_traffic.txt
  rm -rf /tmp/pmsq
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d /tmp/pmsq -o r -- python3 $R/scratch/one_batch.py > /tmp/logsq.txt 2>&1
  python3 $R/tools/rocpd_pmc.py $(find /tmp/pmsq -name '*.db') > $O/pmc_sq.txt
  echo "pmc done"
  rm -rf /tmp/kvg
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/kvg -o vg -- python3 $R/scratch/r05_raw_vg_rate.py 4096 16 16 1 8 exact > $O/vg_exact_n4096_d16.log 2>&1
  python3 $R/tools/rocpd_summary.py $(find /tmp/kvg -name '*.db') > $O/kernel_stats_value_grad_exact_n4096_d16.txt || true
  cd $R
  timeout -k 10 300 python3 bench.py --workload c2 --no-cpu-baseline --no-train > $O/bench_c2.json 2> $O/bench_c2.err
  timeout -k 10 400 python3 bench.py --workload c5 --no-cpu-baseline --no-train --no-pca8 > $O/bench_c5.json 2> $O/bench_c5.err
  timeout -k 10 200 python3 scratch/gemm_insitu_table.py > $O/gemm_insitu_by_K.txt 2>&1
  timeout -k 10 120 python3 scratch/r03_fill_time.py > $O/fill_time.txt 2>&1
  echo "B done"
fi
