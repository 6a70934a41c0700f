set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/ab7
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gemm or schedule or tile_order" > gpurun_out/ab7/tests.txt 2>&1 || true; tail -3 gpurun_out/ab7/tests.txt
timeout -k 10 300 python scratch/gemm_k.py 3 8 > gpurun_out/ab7/gemm_k.txt 2>&1; tail -12 gpurun_out/ab7/gemm_k.txt
timeout -k 10 200 python bench.py --no-cpu-baseline --no-predict --no-grad > gpurun_out/ab7/bench.json 2> gpurun_out/ab7/bench.err
python -c "
import json; j=json.loads(open('gpurun_out/ab7/bench.json').read().strip().splitlines()[-1]); print(j['value'], j['roofline']['frac'], j['roofline_other']['gemm_k512_and_longer']['frac'], j['roofline_other']['potrf_whole']['frac'], j['single_evaluation']['ms_per_evaluation'])"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmlds
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/pmlds -o r -- python3 $R/scratch/one_batch.py > /tmp/loglds.txt 2>&1
python3 $R/tools/rocpd_pmc.py $(find /tmp/pmlds -name '*.db') > $R/gpurun_out/ab7/pmc_lds.txt
grep -A2 "^SQ_LDS" $R/gpurun_out/ab7/pmc_lds.txt | cut -c1-170
