#!/usr/bin/env python3
"""profiles/*_pmc_hbm_traffic.txt (tools/rocpd_pmc.py output of one FETCH_SIZE and one WRITE_SIZE pass) -> the JSON
bench.py reads `roofline.traffic` from.  rocprofv3 reports both counters in KB; on gfx950 FETCH_SIZE counts half the
bytes of the reads these kernels issue (MI355X_MICROARCH.md; profiles/r01_pmc_fetch_calibration.txt), WRITE_SIZE is exact.
usage: python tools/pmc_traffic_json.py profiles/r04_pmc_hbm_traffic.txt "<workload note>" [batches profiled, default 2] > profiles/r04_pmc_hbm_traffic.json
bench.py takes `roofline.traffic` from the newest such file only if the launch count per batch recorded here equals the
launch count of the dominant kernel in its own run (a stale file -- kernels or schedule changed since -- gives traffic: null)."""
import json
import re
import sys

KEYS = [("gemm_nt_kernel<128, 128", "gemm_nt_kernel_128x128_8waves_dma"), ("gemm_nt_kernel<64, 64, 4, 2, 2, 1", "gemm_nt_kernel_64x64_factor_ahead"),
        ("gemm_nt_kernel<64, 64, 4, 2, 2, 0", "gemm_nt_kernel_64x64"), ("cov_stage_gram_kernel", "cov_stage_kernel"), ("cov_stage_batch_kernel", "cov_stage_kernel"),
        ("leaf_solve_kernel", "leaf_solve_kernel"), ("leaf_factor_kernel", "leaf_factor_kernel"), ("leaf_pair_kernel", "leaf_pair_kernel")]
out = {"workload": sys.argv[2] if len(sys.argv) > 2 else "", "units": "rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KB; converted to bytes here",
       "correction": "FETCH_SIZE x 2 on gfx950 (MI355X_MICROARCH.md; the 8-B/lane C-tile reads calibrate to the same half, "
                     "profiles/r01_pmc_fetch_calibration.txt); WRITE_SIZE exact; Infinity-Cache hits are counted: fabric traffic, "
                     "an upper bound on HBM traffic", "source": sys.argv[1],
       "batches_profiled": int(sys.argv[3]) if len(sys.argv) > 3 else 2, "batch_size": 16, "N": 8192}
cur = None
for line in open(sys.argv[1]):
    if not line.startswith(" "):
        cur = line.strip()
        continue
    m = re.match(r"\s+(.*?)\s+calls\s+(\d+)\s+sum\s+(\d+)", line)
    if not m:
        continue
    name, calls, total = m.group(1), int(m.group(2)), float(m.group(3)) * 1024.0
    for pat, key in KEYS:
        if pat in name:
            # (several instantiations share a key -- e.g. the NEG = 0 / 1 forms of one tile shape: their counts add up)
            e = out.setdefault(key, {})
            seen = e.setdefault("_launches_by_counter", {})
            seen[cur] = seen.get(cur, 0) + calls
            e["launches"] = seen[cur]
            if cur == "FETCH_SIZE":
                e["fetch_bytes_raw"] = e.get("fetch_bytes_raw", 0.0) + total
                e["fetch_bytes_corrected"] = 2.0 * e["fetch_bytes_raw"]
            elif cur == "WRITE_SIZE":
                e["write_bytes"] = e.get("write_bytes", 0.0) + total
            break
for v in out.values():
    if isinstance(v, dict):
        v.pop("_launches_by_counter", None)
g = [v for k, v in out.items() if k.startswith("gemm_nt_kernel_")]
if g:
    out["gemm_nt_kernel"] = {"launches": sum(e["launches"] for e in g), "fetch_bytes_raw": sum(e.get("fetch_bytes_raw", 0) for e in g),
                             "fetch_bytes_corrected": sum(e.get("fetch_bytes_corrected", 0) for e in g),
                             "write_bytes": sum(e.get("write_bytes", 0) for e in g)}
print(json.dumps(out, indent=1))
