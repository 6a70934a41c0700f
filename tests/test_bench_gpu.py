"""bench.py itself in the GPU suite: the N > 1 path (the parent starts the ranks, one process group, the barrier / all_reduce
/ padded all_gather of every region, the component -> rank deal of the configs[3] region) on two gloo ranks that share the
one GPU of the box, with the plumbing workload -- what the driver's scaling run executes with RCCL on a real node."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_two_gloo_ranks_sharing_the_gpu():
    env = dict(os.environ, BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "tiny", "--steps", "3",
                          "--warmup", "1", "--no-cpu-baseline", "--no-train", "--pca8-steps", "1"],
                         env=env, capture_output=True, text=True, timeout=220, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                 # rank 0 prints ONE JSON line
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["dtype"] == "f64" and j["value"] > 0
    assert j["process_group"] == "gloo" and j["rccl_ranks"] == 0
    assert j["roofline"]["bound"] == "mfma" and "cpu_baseline" in j and j["cpu_baseline"] is None
    assert j["predictions"]["value"] > 0 and j["value_grad"]["value"] > 0
    p = j["pca8"]
    assert p["components"] == 8 and p["components_this_rank"] == 4 and len(p["best_neg_loglik_per_component"]) == 8
    assert all(v == v for v in p["best_neg_loglik_per_component"])          # every component's result arrived in the gather
