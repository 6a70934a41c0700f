set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for t in 0 8 16; do
 export GPEMU_GEMM_TABLE=$t
 for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm_$t_$c
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c -d /tmp/pm_${t}_$c -o r -- python3 $R/scratch/one_batch.py > /tmp/log_${t}_$c.txt 2>&1
 done
 echo "== TABLE=$t" >> $R/gpurun_out/pmc_table.txt
 python3 $R/tools/rocpd_pmc.py $(find /tmp/pm_${t}_FETCH_SIZE -name '*.db') $(find /tmp/pm_${t}_WRITE_SIZE -name '*.db') | grep -E "SIZE|gemm_nt" >> $R/gpurun_out/pmc_table.txt
done
cat $R/gpurun_out/pmc_table.txt
