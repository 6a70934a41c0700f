#!/bin/bash
mkdir -p gpurun_out/ab3
for r in 1 2; do
for s in 2 3 4; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-predict --no-grad --no-single --streams $s > gpurun_out/ab3/bench_s${s}_r$r.json 2> gpurun_out/ab3/bench_s${s}_r$r.err
done
done
timeout -k 10 300 python bench.py --no-cpu-baseline --no-predict --no-grad --no-single --streams 4 --batch 8 > gpurun_out/ab3/bench_s4b8_r1.json 2> gpurun_out/ab3/bench_s4b8.err
timeout -k 10 300 python bench.py --no-cpu-baseline --no-predict --no-grad --no-single --streams 3 --batch 12 > gpurun_out/ab3/bench_s3b12_r1.json 2> gpurun_out/ab3/bench_s3b12.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/ab3/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "value %.1f ms/step %.2f dominant %.3f k512 %.3f gemm_all %.3f potrf %.3f"%(j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline_other"]["gemm_k512_and_longer"]["frac"], j["roofline_other"]["gemm_all_launches"]["frac"], j["roofline_other"]["potrf_whole"]["frac"]))
    except Exception as e:
        print(f, "failed", e)
PY
