#!/bin/bash
mkdir -p gpurun_out/ab4
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gemm or schedule" > gpurun_out/ab4/tests.txt 2>&1; tail -3 gpurun_out/ab4/tests.txt
for r in 1 2; do
for v in 0 1; do
  GPEMU_GEMM_SMALL_DMA=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-predict --no-grad > gpurun_out/ab4/bench_v${v}_r$r.json 2> gpurun_out/ab4/bench_v${v}_r$r.err
  GPEMU_GEMM_SMALL_DMA=$v timeout -k 10 300 python bench.py --workload c2 --no-cpu-baseline --no-predict --no-grad > gpurun_out/ab4/benchc2_v${v}_r$r.json 2> gpurun_out/ab4/benchc2_v${v}_r$r.err
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/ab4/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "value %.1f ms/step %.2f dominant %.3f gemm_all %.3f potrf %.3f single %.3f"%(j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline_other"]["gemm_all_launches"]["frac"], j["roofline_other"]["potrf_whole"]["frac"], j["single_evaluation"]["ms_per_evaluation"]))
    except Exception as e:
        print(f, "failed", e)
PY
