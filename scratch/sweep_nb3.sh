set -e
for cfg in "2048 64 1" "4096 64 1" "8192 16 3" "12288 16 3" "16384 16 1"; do
 for nb in 1024 2048 4096 8192; do
  GPEMU_NB_TOP=$nb timeout -k 10 300 python scratch/nb_sweep_n.py $cfg
 done
done
