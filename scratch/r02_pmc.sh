# round 2: fabric-side traffic (FETCH_SIZE / WRITE_SIZE, two separate --pmc passes) and SQ counters of scratch/one_batch.py
# (two lock-step batches of 16 evaluations, N=8192): per-kernel sums
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmf_$c
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmf_$c -o r -- python3 $R/scratch/one_batch.py > /tmp/logf_$c.txt 2>&1
done
python3 $R/tools/rocpd_pmc.py $(find /tmp/pmf_FETCH_SIZE -name '*.db') $(find /tmp/pmf_WRITE_SIZE -name '*.db') > $R/gpurun_out/r02_pmc_traffic.txt
rm -rf /tmp/pmsq
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d /tmp/pmsq -o r -- python3 $R/scratch/one_batch.py > /tmp/logsq.txt 2>&1
python3 $R/tools/rocpd_pmc.py $(find /tmp/pmsq -name '*.db') > $R/gpurun_out/r02_pmc_sq.txt
head -30 $R/gpurun_out/r02_pmc_traffic.txt
