#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on MI355X: estimate_thetas likelihood-evals/sec (+ predictions/sec)
at N=8192, d=8, fp64.

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank/GPU)

A step = one pass of the hot path over one batch of synthetic input: a lock-step batch of B independent
evalFnMulti-equivalent likelihood evaluations (covariance fill + Cholesky + solves + logL, sigma^2, beta), each at
its own FRESH theta (nothing cacheable), design resident in HBM -- what the reference runs as restart threads / a
theta list.  B = --batch (default 16 at N >= 8192); the K steps are dealt to --streams concurrent contexts.
value = evaluations per second = K * B * n_gpus / time ("evaluations_per_step" in the line).  After the K timed evaluation steps a second timed region pushes 1e6 query points
(resident in HBM, min(K, 20) batches) through the posterior mean+variance sweep.
Workload = BASELINE.json configs[2]: N=8192, d=8, Matern 5/2, regression order 1, 1e6 batched predictions.
The evaluation is at given (supplied) thetas: the reference cannot train a Matern model.

Multi-GPU: the evaluations / query blocks are independent units sharded one share per rank with no data-path
collective; a single all-gather of the per-rank results ends each region ("weak" scaling: per-rank work fixed).

Prints ONE JSON line on rank 0 (contract in the task statement) with "roofline" and "cpu_baseline" objects.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP64_MFMA_TFLOPS = 78.6     # MI355X dense fp64 matrix peak (256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz)
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec

WORKLOADS = {
    # name: (kind, N, d, regression_order, n_queries)
    "c3": (3, 8192, 8, 1, 1_000_000),   # BASELINE.json configs[2] -- the configuration the metric is quoted on
    "c2": (1, 4096, 8, 0, 1_000_000),   # configs[1]
    "c5": (1, 16384, 8, 0, 1_000_000),  # configs[4] (one rank's share)
    "tiny": (3, 512, 8, 1, 8192),       # plumbing check
}


def _cpu_eval_worker(args):
    """one oracle likelihood evaluation (reference operation sequence) -- cpu_baseline leg only"""
    kind, order, N, d, seed, i = args
    from oracle import oracle as O
    from madaiemulator_amd import synth
    X, y = synth.design(N, d, seed)
    th = synth.perturbed_thetas(kind, d, seed, i)
    t = time.perf_counter()
    if kind == 1:
        O.eval_fn_multi(kind, order, X, y, th[1:])
    else:
        e = O.Emulator(kind, order, X, y, th)       # fill + chol + explicit inverse + estimateBeta: same N^3 sequence
        r = y - e.H @ e.beta
        _ = r @ e.cinverse @ r
    return time.perf_counter() - t


def cpu_baseline(kind, order, N, d, seed):
    """Reference-faithful CPU restatement (oracle/, kind "port") on the host cores: one independent evaluation
    per core, the reference's own parallelisation (estimate_threaded.c:97,172).  Bounded sample at N_s < N,
    extrapolated by (N/N_s)^3 (the path is N^3: unblocked Cholesky + explicit inverse)."""
    from oracle import oracle as O
    O.build()
    cores = min(os.cpu_count() or 1, 16)
    Ns = 2048
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(cores) as pool:
        alone = pool.map(_cpu_eval_worker, [(kind, order, Ns, d, seed, 0)])[0]      # one core, the others idle
        times = pool.map(_cpu_eval_worker, [(kind, order, Ns, d, seed, i) for i in range(cores)])
    wall = time.perf_counter() - t0
    per_eval = float(np.mean(times))
    scale = (N / Ns) ** 3
    evals_per_s = cores / (per_eval * scale)
    # predictions: emulate_point on the oracle at Ns, scaled by N^2 (three N^2 dgemv + N^2*nreg dgemm per query)
    from madaiemulator_amd import synth
    X, y = synth.design(Ns, d, seed)
    e = O.Emulator(kind, order, X, y, synth.default_thetas(kind, d))
    tq = time.perf_counter()
    e.emulate(synth.queries(4, d, 3))
    per_q = (time.perf_counter() - tq) / 4
    preds_per_s = cores / (per_q * (N / Ns) ** 2)
    return {
        "value": evals_per_s, "unit": "likelihood-evals/s", "cores": cores, "kind": "port",
        "sample": (f"one oracle evaluation alone at N={Ns}, d={d}: {alone:.2f} s; {cores} concurrent (one per core): {per_eval:.2f} s each "
                   f"({wall:.1f} s wall); extrapolated to N={N} by (N/{Ns})^3; predictions: 4 oracle emulate_point "
                   f"calls at N={Ns} ({per_q*1e3:.1f} ms each) scaled by (N/{Ns})^2"),
        "value_1core": 1.0 / (alone * scale), "seconds_per_eval_at_sample_1core": alone,
        "predictions_per_s_1core": 1.0 / (per_q * (N / Ns) ** 2),
        "predictions_per_s": preds_per_s,
        "seconds_per_eval_at_sample": per_eval,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24, help="timed steps; one step = one lock-step batch of --batch evaluations")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--queries", type=int, default=None, help="total prediction points per rank (default 1e6)")
    ap.add_argument("--streams", type=int, default=2,
                    help="concurrent evaluation contexts (HIP streams) per GPU: the panel chain of one context's batch "
                         "overlaps the trailing updates of the other's")
    ap.add_argument("--batch", type=int, default=None,
                    help="independent evaluations factored in lock-step per context (gpemu_loglik_batch: the device "
                         "form of the reference's restart threads / callEvalLhoodList); 1 = one matrix per launch; "
                         "default 16 at N >= 8192, up to 64 for smaller models (the panel chain weighs more there)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-predict", action="store_true")
    args = ap.parse_args()

    import torch        # torch first: its bundled HIP runtime must be the one in the process (the reverse order
    torch.cuda.is_available()   # leaves torch without a device); the device library binds to the same soname
    from madaiemulator_amd import abi, shard, synth

    rank, world_size, local_rank = shard.world()
    distributed = world_size > 1
    backend = os.environ.get("BENCH_BACKEND", "nccl")     # "gloo": rehearse the N>1 path with ranks sharing one GPU
    ndev = max(1, torch.cuda.device_count())
    dev_index = local_rank % ndev
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    tdev = "cuda" if backend == "nccl" else "cpu"
    ngpus = world_size

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    kind, N, d, order, nq = WORKLOADS[args.workload]
    if args.queries:
        nq = args.queries
    seed = 20261003 + 2
    K, W = args.steps, args.warmup

    dev = dev_index if distributed else 0
    X, y = synth.design(N, d, seed)
    nstreams = max(1, args.streams)
    ctxs = [abi.Context(dev) for _ in range(nstreams)]      # one HIP stream + HBM workspace each
    for c in ctxs:
        c.set_model(kind, order, X, y)
    ctx = ctxs[0]
    # independent evaluations: global eval index g -> rank g mod world (each rank draws its own fresh thetas)
    def theta(i):
        return synth.perturbed_thetas(kind, d, seed, rank + world_size * i)

    # ---- correctness gate before timing (small N, same seeds): HIP vs oracle, 1e-8 relative
    gate = None
    if rank == 0:
        from oracle import oracle as O
        gate = {}
        # N=512 and N=2048 on the same seeds (the N=2048 oracle run is ~15 s of CPU: skipped together
        # with the CPU baseline by --no-cpu-baseline)
        for Ng in ([512] if args.no_cpu_baseline else [512, 2048]):
            Xg, yg = synth.design(Ng, d, seed)
            g = abi.Context(dev)
            g.set_model(kind, order, Xg, yg)
            thg = synth.default_thetas(kind, d)
            got = g.loglik(thg)
            gotb = g.loglik_batch(np.array([thg, synth.perturbed_thetas(kind, d, seed, 1), thg]))
            e = O.Emulator(kind, order, Xg, yg, thg)
            r = yg - e.H @ e.beta
            ref = -(-0.5 * e.logdet - Ng / 2.0 * 1.83788 - 0.5 * (r @ e.cinverse @ r))
            g.predict_setup(thg)
            qg = synth.queries(64, d, 5)
            m, v = g.predict(qg)
            mo, vo, _ = e.emulate(qg)
            gn = {"loglik_rel": abs(got["value"] - ref) / abs(ref),
                  "loglik_batch_rel": float(max(abs(gotb["value"][0] - ref), abs(gotb["value"][2] - ref)) / abs(ref)),
                  "sigma2_rel": abs(got["sigma2"] - yg @ e.cinverse @ r / Ng) / abs(got["sigma2"]),
                  "beta_rel": float(np.max(np.abs(got["beta"] - e.beta)) / np.max(np.abs(e.beta))),
                  "mean_abs": float(np.max(np.abs(m - mo))), "var_abs": float(np.max(np.abs(v - vo)))}
            assert max(gn.values()) < 1e-8, gn
            gate[f"N{Ng}"] = gn
            g.close()

    # ---- region A: likelihood evaluations
    # the K independent evaluations (each at its own fresh theta) are cut into lock-step batches of B and the
    # batches dealt round-robin to the contexts: every kernel of a factorisation handles its B matrices at once,
    # and the latency-bound panel chain of one context overlaps the big GEMMs of the other
    want = args.batch if args.batch else int(min(64, max(16, 16 * (8192 / N) ** 2)))
    B = max(1, want)
    chunks = [list(range(j * B, (j + 1) * B)) for j in range(K)]       # step j = evaluations j*B .. (j+1)*B-1
    for j in range(max(2, W)):                                          # W untimed warm-up steps per context, two at least
        for c in ctxs:                                                  # (plain launches first, then the graph is recorded)
            c.loglik_batch_enqueue(np.array([theta(100000 + 97 * j + i) for i in range(B)]))
    for c in ctxs:
        c.loglik_batch_collect()
    barrier()
    t0 = time.perf_counter()
    for j, ch in enumerate(chunks):
        ctxs[j % nstreams].loglik_batch_enqueue(np.array([theta(i) for i in ch]))
    used = min(len(chunks), nstreams)
    lasts = [c.loglik_batch_collect() for c in ctxs[:used]]
    barrier()
    tA = time.perf_counter() - t0
    lb = lasts[(len(chunks) - 1) % nstreams]
    last = {"value": float(lb["value"][-1])}
    for l in lasts:
        assert np.all(l["status"] == 0) and np.all(np.isfinite(l["value"])), l
    if distributed:
        tt = torch.tensor([tA], dtype=torch.float64, device=tdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        tA = float(tt.item())
        # the single collective of the path: gather (value, thetas...) per rank, arg-max on every rank
        rows = shard.all_gather_rows(np.concatenate([[last["value"]], theta(K * B - 1)])[None, :], 1 + len(theta(0)))
        assert len(rows) == world_size
    evals_per_s = ngpus * K * B / tA

    # ---- region B: batched predictions, queries resident in HBM
    pred = None
    if not args.no_predict:
        th0 = synth.default_thetas(kind, d)
        ctx.predict_setup(th0)
        nb = 20
        per = -(-nq // nb)
        Xq = synth.queries(per, d, seed + 11 + rank)
        dq, dm, dv = ctx.dev_alloc(Xq.nbytes), ctx.dev_alloc(per * 8), ctx.dev_alloc(per * 8)
        ctx.upload(dq, Xq)
        for _ in range(min(W, 2)):
            ctx.predict_dev(per, dq, dm, dv)
        ctx.sync()
        barrier()
        t0 = time.perf_counter()
        for _ in range(nb):
            ctx.predict_dev(per, dq, dm, dv)
        ctx.sync()
        barrier()
        tB = time.perf_counter() - t0
        if distributed:
            tt = torch.tensor([tB], dtype=torch.float64, device=tdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            tB = float(tt.item())
        mean = ctx.download(dm, (per,))
        assert np.all(np.isfinite(mean))
        pred = {"value": ngpus * nb * per / tB, "unit": "predictions/s", "points_per_rank": nb * per,
                "batches": nb, "ms_per_batch": tB / nb * 1e3}

    # ---- roofline of the dominant kernel (fp64 MFMA GEMM of the Cholesky trailing updates), HIP events on the
    #      ctx stream around every launch; rank 0 only
    roof, roof_other = None, {}
    if rank == 0:
        ctx.prof_begin(abi.PROF_GEMM)
        ctx.loglik_batch_enqueue(np.array([theta(2000 + i) for i in range(B)]))     # one lock-step batch, as timed above
        p = ctx.prof_end()
        ach = p["flops"] / (p["ms"] * 1e-3) / 1e12
        # memory-side bytes per launch from the PMC passes committed under profiles/ (separate rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE runs of the same evaluation).  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE counts half the bytes of
        # 16-B-per-lane reads; the 8-B-per-lane C-tile reads of this kernel calibrate to the same half
        # (profiles/r01_pmc_fetch_calibration.txt), so the whole raw fetch is doubled; WRITE_SIZE is exact.  Infinity-
        # Cache hits are included in both, i.e. this is fabric traffic, an upper bound on HBM traffic.
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic_v7_nb2048.json")
        if args.workload == "c3" and B == 16 and os.path.exists(tpath):      # measured for batches of 16
            tj = json.load(open(tpath))["gemm_nt_kernel"]
            traffic = (2.0 * tj["fetch_bytes_raw"] + tj["write_bytes"]) / tj["launches"]
        roof = {"bound": "mfma", "kernel": "gemm_nt_kernel (potrf trailing update, v_mfma_f64_16x16x4_f64)",
                "achieved": ach, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_FP64_MFMA_TFLOPS,
                "traffic": traffic, "launches": p["n"], "avg_launch_us": p["ms"] * 1e3 / max(p["n"], 1),
                "flops_per_eval": p["flops"] / B, "evaluations_per_launch": B}
        ctx.prof_begin(abi.PROF_POTRF)
        ctx.loglik_batch_enqueue(np.array([theta(3000 + i) for i in range(B)]))
        p = ctx.prof_end()
        roof_other["potrf_whole"] = {"bound": "mfma", "achieved": (N ** 3 / 3.0) * B / (p["ms"] * 1e-3) / 1e12,
                                     "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s", "ms_per_eval": p["ms"] / B,
                                     "evaluations_per_factorisation": B}
        roof_other["potrf_whole"]["frac"] = roof_other["potrf_whole"]["achieved"] / PEAK_FP64_MFMA_TFLOPS
        ctx.prof_begin(abi.PROF_FILL)
        for i in range(3):
            ctx.loglik_enqueue(theta(4000 + i))
        p = ctx.prof_end()
        gbs = p["bytes"] / (p["ms"] * 1e-3) / 1e9
        roof_other["cov_fill"] = {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                  "frac": gbs / PEAK_HBM_GBS, "avg_launch_us": p["ms"] * 1e3 / max(p["n"], 1)}
        if pred is not None:
            ctx.predict_setup(synth.default_thetas(kind, d))
            ctx.prof_begin(abi.PROF_GEMM)
            ctx.predict_dev(per, dq, dm, dv)
            p = ctx.prof_end()
            ach = p["flops"] / (p["ms"] * 1e-3) / 1e12
            roof_other["predict_gemm"] = {"bound": "mfma", "achieved": ach, "peak": PEAK_FP64_MFMA_TFLOPS,
                                          "unit": "TFLOP/s", "frac": ach / PEAK_FP64_MFMA_TFLOPS,
                                          "flops_per_prediction": p["flops"] / per}

    cpu = None
    if rank == 0 and ngpus == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(kind, order, N, d, seed)

    if rank == 0:
        out = {
            "metric": "estimate_thetas likelihood-evals/sec + predictions/sec at N=8192 d=8 fp64",
            "value": evals_per_s, "unit": "likelihood-evals/s",
            "n_gpus": ngpus, "steps": K, "warmup": W, "ms_per_step": tA / K * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.workload}: N={N}, d={d}, cov_fn={kind} "
                                   f"({'pow-exp' if kind == 1 else 'Matern32' if kind == 2 else 'Matern52'}), "
                                   f"regression_order={order}, {nq} prediction points per rank",
                       "parallelism": f"independent evaluations / query blocks x{ngpus} GPUs, one all-gather; per GPU "
                                      f"{nstreams} contexts x lock-step batches of {B} evaluations",
                       "streams_per_gpu": nstreams, "batch": B,
                       "step": f"one lock-step batch of {B} independent likelihood evaluations"},
            "evaluations_per_step": B, "ms_per_evaluation": tA / (K * B) * 1e3,
            "predictions": pred,
            "roofline": roof, "roofline_other": roof_other,
            "cpu_baseline": cpu,
            "speedup_vs_cpu_all_cores": (evals_per_s / cpu["value"]) if cpu else None,
            "parity_gate": gate,
        }
        print(json.dumps(out))
    for c in ctxs:
        c.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
