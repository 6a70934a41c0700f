# LDS bank conflicts of the LDS-DMA GEMM loop (separate --pmc pass, kernel trace only)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
rm -rf /tmp/pmlds
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS -d /tmp/pmlds -o r -- python3 $R/scratch/one_batch.py > /tmp/loglds.txt 2>&1 || { tail -5 /tmp/loglds.txt; exit 1; }
python3 $R/tools/rocpd_pmc.py $(find /tmp/pmlds -name '*.db') > $R/gpurun_out/r02_pmc_lds.txt
grep -A3 "^SQ_LDS" $R/gpurun_out/r02_pmc_lds.txt | cut -c1-170
