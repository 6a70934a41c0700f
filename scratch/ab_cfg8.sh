#!/bin/bash
# A/B of the big-tile configuration inside the bench (same box, alternating)
mkdir -p gpurun_out/ab
for r in 1 2; do
for cfg in 3 8; do
  GPEMU_GEMM_BIG_CFG=$cfg timeout -k 10 200 python bench.py --no-cpu-baseline --no-predict --no-grad --no-single > gpurun_out/ab/bench_cfg${cfg}_r$r.json 2> gpurun_out/ab/bench_cfg${cfg}_r$r.err
  GPEMU_GEMM_BIG_CFG=$cfg timeout -k 10 200 python bench.py --no-cpu-baseline --no-predict --no-grad --no-single --streams 1 > gpurun_out/ab/bench_cfg${cfg}_s1_r$r.json 2> gpurun_out/ab/bench_cfg${cfg}_s1_r$r.err
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/ab/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "value %.1f ms/step %.2f dominant %.3f gemm_all %.3f potrf %.3f"%(j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline_other"]["gemm_all_launches"]["frac"], j["roofline_other"]["potrf_whole"]["frac"]))
    except Exception as e:
        print(f, "failed", e)
PY
