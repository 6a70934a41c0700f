set -e
for bt in 256 512 1024 2048 4096; do
 echo "== GPEMU_GEMM_BIG_TILES=$bt"
 GPEMU_GEMM_BIG_TILES=$bt timeout -k 10 200 python scratch/nb_sweep_n.py 8192 16 3
 GPEMU_GEMM_BIG_TILES=$bt timeout -k 10 200 python scratch/nb_sweep_n.py 4096 64 1
done
