set -e
mkdir -p gpurun_out
for t in 0 4 8 16 0; do
 echo "== GPEMU_GEMM_TABLE=$t"
 GPEMU_GEMM_TABLE=$t timeout -k 10 200 python scratch/batch_sweep2.py 2
 GPEMU_GEMM_TABLE=$t timeout -k 10 100 python scratch/one_eval.py 2>&1 | tail -3
done
