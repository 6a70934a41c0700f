# HBM-side traffic of scratch/one_batch.py: two separate --pmc passes (FETCH_SIZE, WRITE_SIZE), per-kernel sums
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmf_$c
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmf_$c -o r -- python3 $R/scratch/one_batch.py > /tmp/logf_$c.txt 2>&1
done
python3 $R/tools/rocpd_pmc.py $(find /tmp/pmf_FETCH_SIZE -name '*.db') $(find /tmp/pmf_WRITE_SIZE -name '*.db') > $R/gpurun_out/pmc_full.txt
cat $R/gpurun_out/pmc_full.txt
