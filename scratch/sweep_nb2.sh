set -e
for nb in 1024 4096 8192 16384 1024 4096; do
 echo "== GPEMU_NB_TOP=$nb"
 GPEMU_NB_TOP=$nb timeout -k 10 200 python scratch/batch_sweep2.py 2
done
