"""bench.py itself in the GPU suite: the N > 1 path (the parent starts the ranks, one process group, the barrier / all_reduce
/ padded all_gather of every region, the component -> rank deal of the configs[3] region) on two gloo ranks that share the
one GPU of the box, with the plumbing workload -- what the driver's scaling run executes with RCCL on a real node
(there the CLI ranks of regions D and F gather through gpemu_rccl_allgather; here, sharing a device, through files)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_two_gloo_ranks_sharing_the_gpu():
    env = dict(os.environ, BENCH_BACKEND="gloo", GPEMU_GATHER="file")     # (two RCCL ranks cannot share a device)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "tiny", "--steps", "3",
                          "--warmup", "1", "--no-cpu-baseline", "--train-runs", "6", "--pca8-steps", "1"],
                         env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
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
    # regions D and F: the C product as one CLI process per bench rank (csrc/host/ranks.c), the run list / the 8 PCA components
    # dealt to the two processes, one gather at the end, rank 0's snapshot hashed
    t = j["estimate_thetas_c_layer"]
    assert "error" not in t and t["processes"] == 2 and t["runs"] == 6 and len(t["snapshot_sha256"]) == 64
    f = j["pca8_trained_by_cli_ranks"]
    assert "error" not in f and f["processes"] == 2 and not f["ranks_failed"] and f["components_trained_here"] == 4
    assert len(f["snapshot_sha256"]) == 64
