set -e
for nb in 256 512 1024 2048; do
 for st in 1 4; do
  echo -n "single-matrix launches NB_TOP=$nb streams=$st: "
  GPEMU_NB_TOP=$nb timeout -k 10 300 python bench.py --no-cpu-baseline --no-predict --batch 1 --streams $st --steps 48 --warmup 4 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print('%.1f evals/s'%j['value'])"
 done
done
