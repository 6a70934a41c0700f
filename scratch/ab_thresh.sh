#!/bin/bash
mkdir -p gpurun_out/ab2
timeout -k 10 200 python scratch/gemm_insitu_table.py > gpurun_out/ab2/insitu_cfg8.txt 2>&1
for r in 1 2; do
for t in 2048 1024 512 256; do
  GPEMU_GEMM_BIG_TILES=$t timeout -k 10 200 python bench.py --no-cpu-baseline --no-predict --no-grad --no-single > gpurun_out/ab2/bench_t${t}_r$r.json 2> gpurun_out/ab2/bench_t${t}_r$r.err
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/ab2/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "value %.1f ms/step %.2f dominant %.3f gemm_all %.3f potrf %.3f"%(j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline_other"]["gemm_all_launches"]["frac"], j["roofline_other"]["potrf_whole"]["frac"]))
    except Exception as e:
        print(f, "failed", e)
PY
tail -12 gpurun_out/ab2/insitu_cfg8.txt
