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

A third timed region measures what estimate_thetas really calls per BFGS step (libEmu/maxmultimin.c:615-618,
675-680): evalFnGradMulti, value + gradient from one factorisation, as lock-step batches (gpemu_loglik_grad_batch) on
a pow-exp model of the same size (the reference has no Matern gradient).

Multi-GPU: the evaluations / query blocks are independent units sharded one share per rank with no data-path
collective; a single all-gather of the per-rank results ends each region ("weak" scaling: per-rank work fixed).
`python bench.py --gpus N` with no RANK in the environment starts the N ranks itself (a child
`python -m torch.distributed.run --nproc-per-node N bench.py ...`, before this process touches the GPU) and forwards
rank 0's line; under the driver's own torchrun launch the ranks are already there.

Prints ONE JSON line on rank 0 (contract in the task statement) with "roofline" and "cpu_baseline" objects.
"""
import argparse
import ctypes as C
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


def usable_cores():
    """host cores this process may really use: the scheduler affinity, cut down to the cgroup CPU quota when there is one
    (the GPU box shows 256 CPUs and grants 16: `cpu.max` = 1600000 100000)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = int(q) / int(p)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    if quota:
        n = max(1, min(n, int(quota + 0.5)))
    return n, quota


def cpu_baseline(kind, order, N, d, seed, sizes=(1024, 1536, 2048)):
    """Reference-faithful CPU restatement (oracle/, kind "port") on the host cores: one independent evaluation
    per core, the reference's own parallelisation (estimate_threaded.c:97,172).  Bounded samples at N_s < N
    (default 1024, 1536, 2048: about a minute of wall time; `--cpu-sizes` adds larger ones for a one-off run), a power-law
    fit over them, and the N^3 extrapolation from the largest one (the path is N^3: unblocked Cholesky + explicit
    inverse) as the reported value."""
    from oracle import oracle as O
    O.build()
    cores, quota = usable_cores()
    sizes = sorted(int(v) for v in sizes)
    t0 = time.perf_counter()
    per = {}

    def wait(res, what):
        # (a heartbeat on stderr: a sample at N = 4096 runs for many minutes and a silent command is taken for hung)
        t_ = time.perf_counter()
        while not res.ready():
            res.wait(60.0)
            if not res.ready():
                print(f"[bench] cpu baseline: {what} still running after {time.perf_counter() - t_:.0f} s", file=sys.stderr, flush=True)
        return res.get()
    with mp.get_context("spawn").Pool(cores) as pool:
        mid = sizes[min(1, len(sizes) - 1)]
        alone = wait(pool.map_async(_cpu_eval_worker, [(kind, order, mid, d, seed, 0)]), f"one alone at N={mid}")[0]      # one core, the others idle
        print(f"[bench] cpu baseline: one evaluation alone at N={mid}: {alone:.1f} s", file=sys.stderr, flush=True)
        for Ns in sizes:
            per[Ns] = float(np.mean(wait(pool.map_async(_cpu_eval_worker, [(kind, order, Ns, d, seed, i) for i in range(cores)]),
                                         f"{cores} concurrent at N={Ns}")))
            print(f"[bench] cpu baseline: {cores} concurrent at N={Ns}: {per[Ns]:.1f} s each", file=sys.stderr, flush=True)
    wall = time.perf_counter() - t0
    Ns = sizes[-1]
    per_eval = per[Ns]
    scale = (N / Ns) ** 3
    evals_per_s = cores / (per_eval * scale)
    # t = c N^p over the three samples (least squares in log-log): p > 3 once the matrices leave the caches
    sizes = list(sizes)
    lx, ly = np.log(np.array(sizes, float)), np.log(np.array([per[n] for n in sizes]))
    pfit, cfit = np.polyfit(lx, ly, 1)
    fit_eval_s = float(np.exp(cfit) * N ** pfit)
    # predictions: emulate_point on the oracle at Ns, scaled by N^2 (three N^2 dgemv + N^2*nreg dgemm per query)
    # (at the largest sample unless that is a one-off big one: building the oracle's emulator is another full evaluation)
    from madaiemulator_amd import synth
    Nq = Ns if Ns <= 3072 else mid
    X, y = synth.design(Nq, d, seed)
    e = O.Emulator(kind, order, X, y, synth.default_thetas(kind, d))
    tq = time.perf_counter()
    e.emulate(synth.queries(4, d, 3))
    per_q = (time.perf_counter() - tq) / 4
    preds_per_s = cores / (per_q * (N / Nq) ** 2)
    return {
        "value": evals_per_s, "unit": "likelihood-evals/s", "cores": cores, "nproc": os.cpu_count(),
        "cgroup_cpu_quota": quota, "kind": "port",
        "sample": (f"{cores} concurrent oracle evaluations (one per core) at N={sizes}, d={d}: "
                   f"{', '.join('%.2f' % per[n] for n in sizes)} s each ({wall:.1f} s wall in all); one alone at N={mid}: "
                   f"{alone:.2f} s; value = N={Ns} sample extrapolated to N={N} by (N/{Ns})^3; power-law fit over the {len(sizes)} "
                   f"samples t ~ N^{pfit:.2f} gives {fit_eval_s:.0f} s per evaluation at N={N}; predictions: 4 oracle "
                   f"emulate_point calls at N={Nq} ({per_q*1e3:.1f} ms each) scaled by (N/{Nq})^2"),
        "seconds_per_eval_at_samples": {str(n): per[n] for n in sizes}, "fit_exponent": float(pfit),
        "value_from_fit": cores / fit_eval_s,
        "value_1core": 1.0 / (alone * (N / mid) ** 3), "seconds_per_eval_alone": {str(mid): alone},
        "predictions_per_s_1core": 1.0 / (per_q * (N / Nq) ** 2),
        "predictions_per_s": preds_per_s,
    }


def train_through_cli(N, d, seed, dev, runs, exact=True, rank=0, world=1, local_rank=0, rendezvous=None):
    """region D: `interactive_emulator estimate_thetas` (csrc/host, the reference's CLI contract) as a child process on a
    pow-exp N x d INPUT_MODEL_FILE (interactive_emulator.c:222-238 format), fixed seed, `runs` BFGS runs in lock-step
    groups (two per GPU).  Returns the search's own figures (GPEMU_SEARCH_STATS line of estimate_thetas_threaded).
    world > 1: the same search as ONE PROCESS PER GPU (csrc/host/ranks.c) -- every bench rank starts the CLI with its own
    GPEMU_RANK; run r of the list belongs to rank r mod world; one gpemu_rccl_allgather ends it; the figures are this
    rank's share (bench.py adds them up), best_loglik is the search's result on every rank."""
    import re
    import subprocess
    import tempfile
    from madaiemulator_amd import build, synth
    X, y = synth.design(N, d, seed)
    with tempfile.TemporaryDirectory(prefix="gpemu_bench_") as tmp:
        inp, snap = os.path.join(tmp, "model.dat"), os.path.join(tmp, "snapshot.txt")
        with open(inp, "w") as f:
            f.write(f"1\n{d}\n{N}\n")
            np.savetxt(f, X, fmt="%.17g")
            np.savetxt(f, y, fmt="%.17g")
        env = dict(os.environ, GPEMU_DEVICES=str(dev), GPEMU_SEED="20261003", GPEMU_JOBS="1", GPEMU_RESTARTS=str(runs),
                   GPEMU_SEARCH_STATS="1")
        if world > 1:
            os.makedirs(os.path.join(rendezvous, "region_d"), exist_ok=True)          # a fresh directory per search
            env.update(GPEMU_RANK=str(rank), GPEMU_WORLD_SIZE=str(world), GPEMU_LOCAL_RANK=str(local_rank),
                       GPEMU_RENDEZVOUS_DIR=os.path.join(rendezvous, "region_d"))
        cmd = [build.CLI_BIN, "estimate_thetas", inp, snap, "--covariance_fn=1", "--regression_order=0"]
        if exact:
            cmd.append("--exact_gradient")
        t0 = time.perf_counter()
        try:
            # (as ranks: a rank whose peer failed would wait for it; bounded so that the scaling run still ends in minutes)
            out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600 if world == 1 else 180)
        except (subprocess.TimeoutExpired, OSError) as ex:
            return {"error": repr(ex)[:300]}
        wall = time.perf_counter() - t0
        if out.returncode != 0:
            return {"error": out.stderr[-600:]}
        m = re.search(r"# search stats: runs (\d+) threads (\d+) groups (\d+) slots (\d+) value_grad_evals (\d+) value_evals (\d+) "
                      r"cached (\d+) rounds (\d+) round_elements (\d+) iterations (\d+) seconds ([0-9.]+) best (\S+)", out.stderr)
        if not m:
            return {"error": "no search stats line", "stderr": out.stderr[-600:]}
        g = m.groups()
        nruns, nthreads, ngroups, nslots, nvg, nv, ncached, nrounds, nelem, niter = (int(x) for x in g[:10])
        secs, best = float(g[10]), float(g[11])
        return {"value_grad_evals_per_s": nvg / secs, "unit": "evalFnGradMulti calls/s inside estimate_thetas_threaded",
                "runs": nruns, "host_threads": nthreads, "lockstep_groups": ngroups, "device_slots": nslots,
                "value_grad_evals": nvg, "value_only_evals": nv, "answered_from_cache": ncached, "device_rounds": nrounds,
                "mean_requests_per_round": nelem / max(nrounds, 1), "bfgs_iterations": niter,
                "ms_per_bfgs_iteration_per_run": secs * 1e3 * nthreads / max(niter, 1),
                "search_seconds": secs, "cli_wall_seconds": wall, "best_loglik": best,
                "gradient": "exact" if exact else "literal (reference formulas)",
                "snapshot_sha256": __import__("hashlib").sha256(open(snap, "rb").read()).hexdigest() if rank == 0 else None,
                "workload": f"pow-exp, N={N}, d={d}, regression_order=0; lib/interactive_emulator estimate_thetas, GPEMU_RESTARTS={runs}"}


def _io_stats(stderr_text):
    import re
    m = re.search(r"# interactive stats: points (\d+) batches (\d+) max_batch (\d+) parse_s ([0-9.]+) device_s ([0-9.]+) "
                  r"format_s ([0-9.]+) wall_s ([0-9.]+) load_snapshot_s ([0-9.]+) alloc_multi_emulator_s ([0-9.]+) components (\d+)",
                  stderr_text)
    if not m:
        return None
    g = m.groups()
    return {"points": int(g[0]), "batches": int(g[1]), "max_batch": int(g[2]), "parse_s": float(g[3]), "device_s": float(g[4]),
            "format_s": float(g[5]), "loop_wall_s": float(g[6]), "load_snapshot_s": float(g[7]),
            "alloc_multi_emulator_s": float(g[8]), "components": int(g[9])}


def interactive_cli_region(kind, N, d, order, seed, dev, nq, check):
    """region F: predictions/s a user of the drop-in sees -- `lib/interactive_emulator interactive_mode -q` as a child
    process on a MODEL_SNAPSHOT_FILE of the workload's model (supplied thetas), nq distinct points on stdin: the
    reference's text protocol (interactive_emulator.c:414-441: "%lf" in, "%.17f\\n" out) and its BINARY_INTERACTIVE_MODE
    framing (--binary).  The rate is points / the loop's own wall clock (start-up -- snapshot parse, alloc_multi_emulator --
    is reported beside it); the stage times name the bound.  `check(Xq)` -> (mean, var) of the same points through the
    C-ABI for a sanity comparison of what the CLI printed."""
    import subprocess
    import tempfile
    from madaiemulator_amd import build, synth
    X, y = synth.design(N, d, seed)
    th = synth.default_thetas(kind, d)
    out = {"workload": f"N={N}, d={d}, cov_fn={kind}, regression_order={order}; lib/interactive_emulator interactive_mode -q, "
                       f"{nq} distinct points on stdin (a file), results to a file"}
    with tempfile.TemporaryDirectory(prefix="gpemu_bench_") as tmp:
        snap, qtxt, qbin = os.path.join(tmp, "snap.txt"), os.path.join(tmp, "q.txt"), os.path.join(tmp, "q.bin")
        open(snap, "w").write(synth.single_output_snapshot(X, y, kind, order, th))
        Xq = synth.queries(nq, d, seed + 31)
        np.savetxt(qtxt, Xq, fmt="%.17g")
        Xq.tofile(qbin)
        env = dict(os.environ, GPEMU_DEVICE=str(dev), GPEMU_IO_STATS="1")
        lam = float(((y - y.mean()) ** 2).mean())
        for name, qf, extra in (("text", qtxt, []), ("binary", qbin, ["--binary"])):
            res = os.path.join(tmp, "out." + name)
            t0 = time.perf_counter()
            try:
                p = subprocess.run([build.CLI_BIN, "interactive_mode", snap, "-q"] + extra, stdin=open(qf, "rb"),
                                   stdout=open(res, "wb"), stderr=subprocess.PIPE, env=env, timeout=600)
            except (subprocess.TimeoutExpired, OSError) as ex:
                out[name] = {"error": repr(ex)[:300]}
                continue
            wall = time.perf_counter() - t0
            st = _io_stats(p.stderr.decode(errors="replace"))
            if p.returncode != 0 or not st or st["points"] != nq:
                out[name] = {"error": p.stderr.decode(errors="replace")[-600:]}
                continue
            if name == "text":
                vals = np.loadtxt(res, max_rows=2 * 4096).reshape(-1, 2)
                nnum = sum(1 for _ in open(res))
            else:
                raw = np.fromfile(res, dtype=np.float64)
                nnum, vals = raw.size, raw[:2 * 4096].reshape(-1, 2)
            assert nnum == 2 * nq, (name, nnum)
            m, v = check(Xq[:4096])                      # PCA-space mean/variance through the C-ABI -> observable space
            # (the CLI's GP is trained on the PCA column z = (y - ybar) / sqrt(lam) and back-projected, multivar_support.c:126-151:
            #  the mean is the direct model's, the variance is scaled by lam)
            assert np.max(np.abs(vals[:, 0] - m)) < 1e-7 and np.max(np.abs(vals[:, 1] - lam * v)) < 1e-7, name
            bound = max((("read+parse", st["parse_s"]), ("device", st["device_s"]), ("format+write", st["format_s"])), key=lambda t: t[1])
            out[name] = {"value": nq / st["loop_wall_s"], "unit": "predictions/s", "loop_wall_s": st["loop_wall_s"],
                         "process_wall_s": wall, "stage_busy_s": {"read_parse": st["parse_s"], "device": st["device_s"],
                                                                  "format_write": st["format_s"]},
                         "bound": bound[0], "batches": st["batches"], "max_batch": st["max_batch"],
                         "ns_per_number_read": st["parse_s"] / (nq * d) * 1e9, "ns_per_number_written": st["format_s"] / (2 * nq) * 1e9,
                         "startup": {"load_snapshot_s": st["load_snapshot_s"], "alloc_multi_emulator_s": st["alloc_multi_emulator_s"]}}
        # the MCMC pattern: one point, wait for its answer, next point (interactive_emulator.c:440 flushes per point)
        try:
            import select
            q = subprocess.Popen([build.CLI_BIN, "interactive_mode", snap, "-q"], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                 stderr=subprocess.DEVNULL, env=env)
            lat = []
            for i in range(60):
                line = (" ".join(repr(float(vv)) for vv in Xq[i]) + "\n").encode()
                t0 = time.perf_counter()
                q.stdin.write(line)
                q.stdin.flush()
                got = b""
                while got.count(b"\n") < 2:
                    if not select.select([q.stdout], [], [], 120.0)[0]:
                        raise TimeoutError("no answer to a lone point")
                    got += os.read(q.stdout.fileno(), 4096)
                lat.append(time.perf_counter() - t0)
            q.stdin.close()
            q.wait(timeout=60)
            out["lone_point_round_trip_ms"] = {"first": lat[0] * 1e3, "median_of_the_rest": float(np.median(lat[10:])) * 1e3}
        except (OSError, TimeoutError, subprocess.TimeoutExpired) as ex:
            out["lone_point_round_trip_ms"] = {"error": repr(ex)[:200]}
    return out


def pca8_emulator_cli(dev, npts=100000):
    """the QUERY side of BASELINE configs[3] through the drop-in: a MODEL_SNAPSHOT_FILE of the N=4096, d=16, t=9 -> 8 PCA
    component model at supplied thetas, `interactive_emulator interactive_mode -q` on `npts` points: what
    alloc_multi_emulator (multivar_support.c:30-52: eight alloc_emulator_struct) costs at start-up and how fast
    emulate_point_multi's batched form (eight component sweeps + the back-projection to 9 outputs, :103-157) answers"""
    import subprocess
    import tempfile
    from madaiemulator_amd import build, synth
    N, d, nt = 4096, 16, 9
    X, y = synth.design(N, d, 20261003 + 3)
    Y = synth.multi_outputs(X, y, nt)
    Z, evals, evecs, ybar = synth.pca_zmatrix(Y)
    nr = Z.shape[1]
    ths = [synth.perturbed_thetas(1, d, 77, c) for c in range(nr)]
    with tempfile.TemporaryDirectory(prefix="gpemu_bench_") as tmp:
        snap, qf, res = os.path.join(tmp, "snap.txt"), os.path.join(tmp, "q.txt"), os.path.join(tmp, "out.txt")
        open(snap, "w").write(synth.snapshot_text(X, Y, evals, evecs, Z, 1, 0, ths))
        Xq = synth.queries(npts, d, 5)
        np.savetxt(qf, Xq, fmt="%.17g")
        env = dict(os.environ, GPEMU_DEVICE=str(dev), GPEMU_IO_STATS="1")
        # parity gate in front of the clock: the first 64 points through the same CLI against an all-numpy / LAPACK chain
        # (tests/gradref.py: emulator.c:578-593, 672-785 and the back-projection of multivar_support.c:126-151 restated;
        # nothing of it comes from the device library): 8 components x one LAPACK factorisation at N = 4096
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests"))
        import gradref
        g64 = os.path.join(tmp, "q64.txt")
        np.savetxt(g64, Xq[:64], fmt="%.17g")
        try:
            g = subprocess.run([build.CLI_BIN, "interactive_mode", snap, "-q"], stdin=open(g64, "rb"), capture_output=True, env=env, timeout=600)
        except (subprocess.TimeoutExpired, OSError) as ex:
            return {"error": "parity gate: " + repr(ex)[:300]}
        if g.returncode != 0:
            return {"error": "parity gate: " + g.stderr.decode(errors="replace")[-600:]}
        got = np.array(g.stdout.split(), dtype=float).reshape(64, nt, 2)
        mo, vo = np.empty((64, nr)), np.empty((64, nr))
        for c in range(nr):
            mo[:, c], vo[:, c] = gradref.predict(X, Z[:, c], 0, ths[c], Xq[:64])
        ym, yv = gradref.backproject(mo, vo, evals, evecs, ybar)
        gate = {"points": 64, "reference": "numpy/LAPACK (tests/gradref.py predict + backproject), 8 factorisations at N=4096",
                "max_abs_mean_error": float(np.max(np.abs(got[:, :, 0] - ym))), "max_abs_variance_error": float(np.max(np.abs(got[:, :, 1] - yv))),
                "mean_scale": float(np.max(np.abs(ym))), "variance_scale": float(np.max(yv)), "tolerance": "1e-8 of the scale"}
        assert gate["max_abs_mean_error"] <= 1e-8 * gate["mean_scale"] and gate["max_abs_variance_error"] <= 1e-8 * gate["variance_scale"], gate
        try:
            p = subprocess.run([build.CLI_BIN, "interactive_mode", snap, "-q"], stdin=open(qf, "rb"), stdout=open(res, "wb"),
                               stderr=subprocess.PIPE, env=env, timeout=600)
        except (subprocess.TimeoutExpired, OSError) as ex:
            return {"error": repr(ex)[:300]}
        st = _io_stats(p.stderr.decode(errors="replace"))
        if p.returncode != 0 or not st or st["points"] != npts:
            return {"error": p.stderr.decode(errors="replace")[-600:]}
        vals = np.loadtxt(res, max_rows=2 * nt * 256).reshape(-1, nt, 2)
        assert np.all(np.isfinite(vals)) and np.all(vals[:, :, 1] > 0.0)
        return {"parity_gate": gate, "components": st["components"], "outputs": nt, "load_snapshot_s": st["load_snapshot_s"],
                "alloc_multi_emulator_s": st["alloc_multi_emulator_s"], "points": npts,
                "points_per_s": npts / st["loop_wall_s"], "component_predictions_per_s": nr * npts / st["loop_wall_s"],
                "stage_busy_s": {"read_parse": st["parse_s"], "device": st["device_s"], "format_write": st["format_s"]},
                "workload": f"N={N}, d={d}, {nt} outputs back-projected from {nr} PCA components; interactive_mode -q, text protocol"}


def pca8_train_ranks_cli(dev, rank, world, local_rank, rendezvous, restarts=4):
    """region F: BASELINE configs[3] TRAINED through the drop-in as ONE PROCESS PER GPU (csrc/host/ranks.c): every bench
    rank starts `interactive_emulator estimate_thetas` on the same N=4096, d=16, t=9 INPUT_MODEL_FILE with GPEMU_RANK /
    GPEMU_WORLD_SIZE of its own; the 8 PCA components are dealt to the ranks, the thetas meet in ONE gpemu_rccl_allgather
    (RCCL over xGMI when the ranks sit on different GPUs) and rank 0 writes the snapshot, whose sha256 is reported: it is
    the same string at every --gpus N, since a component's search depends on the seed alone."""
    import hashlib
    import re
    import subprocess
    from madaiemulator_amd import build, synth
    N, d, nt = 4096, 16, 9
    inp, snap = os.path.join(rendezvous, f"pca8_train_{rank}.dat"), os.path.join(rendezvous, f"pca8_snapshot_{rank}.txt")
    X, y = synth.design(N, d, 20261003 + 3)
    Y = synth.multi_outputs(X, y, nt) + 0.05 * synth.normal(12, N * nt).reshape(N, nt)
    with open(inp, "w") as f:
        f.write(f"{nt}\n{d}\n{N}\n")
        np.savetxt(f, X, fmt="%.17g")
        np.savetxt(f, Y, fmt="%.17g")
    env = dict(os.environ, GPEMU_DEVICES=str(dev), GPEMU_SEED="20261004", GPEMU_JOBS="1", GPEMU_RESTARTS=str(restarts),
               GPEMU_SEARCH_STATS="1", GPEMU_RANK=str(rank), GPEMU_WORLD_SIZE=str(world), GPEMU_LOCAL_RANK=str(local_rank),
               GPEMU_RENDEZVOUS_DIR=os.path.join(rendezvous, "region_f"))
    os.makedirs(os.path.join(rendezvous, "region_f"), exist_ok=True)                  # a fresh directory per search
    cmd = [build.CLI_BIN, "estimate_thetas", inp, snap, "--covariance_fn=1", "--regression_order=0", "--pca_variance=1.0",
           "--exact_gradient"]
    t0 = time.perf_counter()
    try:
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420 if world == 1 else 120)
    except (subprocess.TimeoutExpired, OSError) as ex:
        return {"error": repr(ex)[:300]}
    wall = time.perf_counter() - t0
    if out.returncode != 0:
        return {"error": out.stderr[-600:]}
    if os.environ.get("BENCH_KEEP_STDERR"):
        sys.stderr.write(out.stderr)
    secs = [float(v) for v in re.findall(r"# search stats: .* seconds ([0-9.]+) best", out.stderr)]
    nvg = [int(v) for v in re.findall(r"value_grad_evals (\d+)", out.stderr)]
    res = {"cli_wall_seconds": wall, "components_trained_here": len(secs), "search_seconds_here": sum(secs),
           "value_grad_evals_here": sum(nvg)}
    ph = re.search(r"# cli phases: rendezvous_read_input_s ([0-9.]+) pca_alloc_s ([0-9.]+) train_and_dump_s ([0-9.]+) snapshot_check_s ([0-9.]+)", out.stderr)
    if ph:
        res.update(rendezvous_read_input_s=float(ph.group(1)), pca_alloc_s=float(ph.group(2)), train_and_dump_s=float(ph.group(3)),
                   snapshot_check_s=float(ph.group(4)))
    if rank == 0:
        res["snapshot_sha256"] = hashlib.sha256(open(snap, "rb").read()).hexdigest()
        res["gather"] = ("none (one process)" if world == 1 else "files (GPEMU_GATHER=file: ranks sharing a device)"
                         if os.environ.get("GPEMU_GATHER") == "file" else "gpemu_rccl_allgather (RCCL), one per search")
        res["workload"] = (f"N={N}, d={d}, {nt} outputs -> 8 PCA components, pow-exp, regression_order=0, GPEMU_RESTARTS={restarts}; "
                           f"lib/interactive_emulator estimate_thetas as {world} process(es), components c = rank mod {world}")
    return res


def pca8_region(abi, shard, synth, dev, rank, world_size, steps, barrier, reduce_max):
    """region E (BASELINE configs[3]): N=4096, d=16, t=9 outputs -> 8 PCA components (multi_modelstruct.c:172-338), each an
    independent scalar GP on the shared design; component c -> rank c mod W.  Every component runs `steps` lock-step
    batches of 64 fresh-theta likelihood evaluations (pow-exp, regression order 0) and its best value travels in the one
    all-gather at the end.  Strong scaling: 8 * steps * 64 evaluations however many ranks share them."""
    N, d, nt, B = 4096, 16, 9, 64
    seed = 20261003 + 3
    X, y = synth.design(N, d, seed)
    Z = synth.pca_zmatrix(synth.multi_outputs(X, y, nt))[0]
    nr = Z.shape[1]
    mine = shard.cyclic_share(nr, rank, world_size)
    # a rank that holds a single component gives it TWO contexts (its batches alternate between them) so that one batch's
    # panel chain still runs beside another's trailing updates, as the components of a many-component rank do for each other
    nctx = 2 if len(mine) == 1 else 1
    nctx = int(os.environ.get("BENCH_PCA8_NCTX", nctx))          # (rehearsal of the one-component-per-rank form on one GPU)
    ctxs = {}
    for c in mine:
        ctxs[c] = [abi.Context(dev) for _ in range(nctx)]
        for cx in ctxs[c]:
            cx.set_model(1, 0, X, Z[:, c].copy())
            for j in range(2):
                cx.loglik_batch(np.array([synth.perturbed_thetas(1, d, seed + c, 9000 + 64 * j + i) for i in range(B)]))
    barrier()
    t0 = time.perf_counter()
    best = {c: np.inf for c in mine}
    busy = {c: [False] * nctx for c in mine}

    def collect(c, k):
        r = ctxs[c][k].loglik_batch_collect()
        assert np.all(r["status"] == 0) and np.all(np.isfinite(r["value"])), r
        best[c] = min(best[c], float(r["value"].min()))
        busy[c][k] = False
    for j in range(steps):                        # the components of a rank take turns: their batches overlap on the device
        for c in mine:
            k = j % nctx
            if busy[c][k]:
                collect(c, k)
            ctxs[c][k].loglik_batch_enqueue(np.array([synth.perturbed_thetas(1, d, seed + c, j * B + i) for i in range(B)]))
            busy[c][k] = True
    for c in mine:
        for k in range(nctx):
            if busy[c][k]:
                collect(c, k)
    barrier()
    t = reduce_max(time.perf_counter() - t0)
    rows = shard.farm_components(lambda c: [best[c]], nr, 1)           # the single collective: (component, best value)
    assert rows.shape == (nr, 1) and np.all(np.isfinite(rows))
    for c in mine:
        for cx in ctxs[c]:
            cx.close()
    # the raw value+gradient rate at this size (N=4096, d=16; exact gradient as region F's search uses): two contexts x
    # lock-step batches of 16 through gpemu_loglik_grad_batch_enqueue / collect_back, every batch collected -- what the CLI
    # search of region F (pca8_trained_by_cli_ranks) is measured against.  Rank 0, one GPU's worth.
    raw_vg = None
    if rank == 0:
        gcs = [abi.Context(dev) for _ in range(2)]
        for gc in gcs:
            gc.set_model(1, 0, X, Z[:, 0].copy())
            gc.set_mode(abi.MODE_EXACT_GRAD)
        Bg, nsteps = 16, 12
        thg = lambda j: np.array([synth.perturbed_thetas(1, d, seed + 5, j * Bg + i) for i in range(Bg)])
        for j in range(4):
            gcs[j % 2].loglik_grad_batch(thg(j))
        pend = [False, False]
        tg0 = time.perf_counter()
        for j in range(nsteps):
            k = j % 2
            if pend[k]:
                r = gcs[k].loglik_grad_batch_collect_back(0, Bg)
                assert np.all(r["status"] == 0) and np.all(np.isfinite(r["grad"]))
            gcs[k].loglik_grad_batch_enqueue(thg(10 + j))
            pend[k] = True
        for k in range(2):
            r = gcs[k].loglik_grad_batch_collect_back(0, Bg)
            assert np.all(r["status"] == 0) and np.all(np.isfinite(r["grad"]))
        tg = time.perf_counter() - tg0
        for gc in gcs:
            gc.close()
        raw_vg = {"value": nsteps * Bg / tg, "unit": "value+gradient evaluations/s (exact gradient), 2 contexts x batches of 16, every batch collected",
                  "roofline_frac_of_N3_flops": nsteps * Bg / tg * float(N) ** 3 / 1e12 / PEAK_FP64_MFMA_TFLOPS}
    return {"value": nr * steps * B / t, "raw_value_grad": raw_vg, "contexts_per_component": nctx, "unit": "likelihood-evals/s over the 8 components (total work fixed: strong scaling)",
            "components": nr, "components_this_rank": len(mine), "batches_per_component": steps, "evaluations_per_batch": B,
            "seconds": t, "workload": f"N={N}, d={d}, t={nt} outputs -> {nr} PCA components, pow-exp, regression_order=0",
            "best_neg_loglik_per_component": rows[:, 0].tolist()}


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
    ap.add_argument("--cpu-sizes", default="1024,1536,2048",
                    help="sample sizes of the CPU baseline (one oracle evaluation per core at each); the default takes about a "
                         "minute, adding 4096 about a quarter of an hour")
    ap.add_argument("--no-predict", action="store_true")
    ap.add_argument("--no-grad", action="store_true", help="skip the value+gradient region")
    ap.add_argument("--no-single", action="store_true", help="skip the one-evaluation-at-a-time latency figure")
    ap.add_argument("--grad-steps", type=int, default=None, help="timed value+gradient batches (default min(steps, 8))")
    ap.add_argument("--no-train", action="store_true", help="skip the estimate_thetas-through-the-C-layer region")
    ap.add_argument("--grad-batch", type=int, default=None, help="evaluations per value+gradient batch (default min(batch, 16))")
    ap.add_argument("--train-runs", type=int, default=64, help="BFGS runs (restarts) of that region (GPEMU_RESTARTS)")
    ap.add_argument("--train-literal", action="store_true",
                    help="that region with the reference's literal gradient formulas instead of the exact gradient")
    ap.add_argument("--no-pca8", action="store_true", help="skip the 8-PCA-component region (BASELINE configs[3])")
    ap.add_argument("--no-interactive", action="store_true", help="skip the interactive_mode-through-the-CLI region")
    ap.add_argument("--interactive-queries", type=int, default=None, help="points piped through the CLI (default: --queries / 1e6)")
    ap.add_argument("--pca8-steps", type=int, default=8, help="lock-step batches of evaluations per PCA component")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # no launcher: start the N ranks ourselves, as fresh children, BEFORE anything in this process touches the GPU
        # (no torch import yet), one rank per GPU over RCCL; rank 0 prints the line, which is forwarded
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.exit(subprocess.call(cmd, env=env))

    import torch        # torch first: its bundled HIP runtime must be the one in the process (the reverse order
    torch.cuda.is_available()   # leaves torch without a device); the device library binds to the same soname
    from madaiemulator_amd import abi, shard, synth

    rank, world_size, local_rank = shard.world()
    # BENCH_FORCE_DIST=1: initialise the process group (RCCL when BENCH_BACKEND is "nccl") even for ONE rank, so that the
    # collective code path of an N > 1 run -- init, barrier, all_reduce, the padded all_gather -- can be rehearsed on a
    # one-GPU box (two RCCL ranks cannot share a device)
    distributed = world_size > 1 or bool(os.environ.get("BENCH_FORCE_DIST"))
    if distributed and "RANK" not in os.environ:
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(so.getsockname()[1]))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("LOCAL_RANK", "0")
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
    assert world_size == max(1, args.gpus), f"--gpus {args.gpus} but the launcher started {world_size} rank(s)"
    if distributed:
        assert dist.get_world_size() == args.gpus

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def allreduce_max(t):
        tt = torch.tensor([t], dtype=torch.float64, device=tdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

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
        # the value+gradient region (C) is gated like the others: evalFnGradMulti on region C's own pow-exp design at
        # N=512 -- a lock-step batch through the asynchronous entry against the oracle's gradFnMulti / evalFnMulti
        # (maxmultimin.c:416-550, 288-394 restated), 1e-8 of the largest gradient component
        if not args.no_grad:
            Xg, yg = synth.design(512, d, seed + 1)
            g = abi.Context(dev)
            g.set_model(1, 0, Xg, yg)
            thb = np.array([synth.perturbed_thetas(1, d, seed + 1, i) for i in range(3)])
            thb[:, 0] = 0.0
            g.loglik_grad_batch_enqueue(thb)
            gotg = g.loglik_grad_batch_collect_back(0, 3)
            gerr, verr = 0.0, 0.0
            for b in (0, 2):
                gref, st = O.grad_fn_multi(1, 0, Xg, yg, thb[b][1:])
                vref = O.eval_fn_multi(1, 0, Xg, yg, thb[b][1:])["value"]
                assert st == 0 and gotg["status"][b] == 0
                gerr = max(gerr, float(np.max(np.abs(gotg["grad"][b] - gref)) / np.max(np.abs(gref))))
                verr = max(verr, abs(gotg["value"][b] - vref) / abs(vref))
            gate["value_grad_N512"] = {"grad_rel_to_largest_component": gerr, "value_rel": verr}
            assert gerr < 1e-8 and verr < 1e-8, gate["value_grad_N512"]
            g.close()

    def note(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)
    note("parity gate done" if gate is not None else "start")

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
    # every step's results are collected: a context keeps up to RESULT_RING - 1 batches in flight and hands back the
    # oldest one (pinned result ring, gpemu_loglik_batch_collect_back) before the ring wraps
    RING = abi.RESULT_RING
    pending = [[] for _ in ctxs]            # per context: step indices enqueued and not yet collected
    values = np.full((K, B), np.nan)
    t0 = time.perf_counter()
    for j, ch in enumerate(chunks):
        s_ = j % nstreams
        c = ctxs[s_]
        if len(pending[s_]) == RING - 1:
            r = c.loglik_batch_collect_back(RING - 2, B)
            assert np.all(r["status"] == 0), r
            values[pending[s_].pop(0)] = r["value"]
        c.loglik_batch_enqueue(np.array([theta(i) for i in ch]))
        pending[s_].append(j)
    for s_, c in enumerate(ctxs):
        while pending[s_]:
            r = c.loglik_batch_collect_back(len(pending[s_]) - 1, B)
            assert np.all(r["status"] == 0), r
            values[pending[s_].pop(0)] = r["value"]
    barrier()
    tA = time.perf_counter() - t0
    assert np.all(np.isfinite(values)), "a likelihood value of the timed region is not finite"
    last = {"value": float(values[-1, -1])}
    if distributed:
        tt = torch.tensor([tA], dtype=torch.float64, device=tdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        tA = float(tt.item())
        # the single collective of the path: gather (value, thetas...) per rank, arg-max on every rank
        rows = shard.all_gather_rows(np.concatenate([[last["value"]], theta(K * B - 1)])[None, :], 1 + len(theta(0)))
        assert len(rows) == world_size
    evals_per_s = ngpus * K * B / tA
    note(f"region A: {evals_per_s:.1f} evaluations/s")

    # ---- roofline of the dominant kernel (fp64 MFMA GEMM of the Cholesky trailing updates), HIP events on the
    #      ctx stream around every launch; rank 0 only.  Taken right behind region A: the same kernels in the state of the
    #      card region A was timed in (at the end of the whole line, a minute of load later, they read 3 % lower)
    roof, roof_other = None, {}
    if rank == 0:
        # the dominant kernel = gemm_nt_kernel<128,128,4,4,2,0,1> (128x128 tiles, 8 waves, operands by LDS-DMA, A negated by the NEG bit): every
        # update of a lock-step batch with >= 1024 such tiles and n >= 256 (contraction lengths 256 .. 2048), 95 % of its flops and
        # 79 % of the GPU time of the likelihood region (profiles/r03_kernel_stats_*)
        # (a first profiled batch is discarded: the plain-launch path with an event pair around every launch is taken for the
        # first time there; then THREE lock-step batches, as timed above, every launch of the dominant kernel between its own
        # pair of events -- round 5: a single batch read 0.83 on a box whose other classes, and rocprofv3, said 0.865)
        ctx.prof_begin(abi.PROF_GEMM_BIG)
        ctx.loglik_batch_enqueue(np.array([theta(1900 + i) for i in range(B)]))
        ctx.prof_end()
        ctx.loglik_batch_collect()
        pp = []
        for rep in range(3):
            ctx.prof_begin(abi.PROF_GEMM_BIG)
            ctx.loglik_batch_enqueue(np.array([theta(2000 + B * rep + i) for i in range(B)]))
            pp.append(ctx.prof_end())
            ctx.loglik_batch_collect()
        assert len({q_["n"] for q_ in pp}) == 1
        p = {"n": pp[0]["n"], "ms": sum(q_["ms"] for q_ in pp) / len(pp), "flops": sum(q_["flops"] for q_ in pp) / len(pp)}   # per batch
        ach = p["flops"] / (p["ms"] * 1e-3) / 1e12 if p["ms"] > 0 else 0.0
        # memory-side bytes per launch from the PMC passes committed under profiles/ (separate rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE runs of the same evaluation).  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE counts half the bytes of
        # 16-B-per-lane reads; the 8-B-per-lane C-tile reads of this kernel calibrate to the same half
        # (profiles/r01_pmc_fetch_calibration.txt), so the whole raw fetch is doubled; WRITE_SIZE is exact.  Infinity-
        # Cache hits are included in both, i.e. this is fabric traffic, an upper bound on HBM traffic.
        traffic, traffic_src, traffic_why = None, None, "no profiles/r*_pmc_hbm_traffic.json"
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_hbm_traffic.json")))
        if cands:
            tpath = cands[-1]                                   # the newest round's PMC passes
            tj_all = json.load(open(tpath))
            tj = tj_all.get("gemm_nt_kernel_128x128_8waves_dma")
            nbat = tj_all.get("batches_profiled", 2)
            # the file is used only if it describes THIS run: same workload and batch size, and the dominant kernel was
            # launched as often per batch there as here (a changed kernel split or schedule makes the file stale -> null)
            if args.workload != "c3" or B != tj_all.get("batch_size", 16):
                traffic_why = "PMC file is for workload c3, batches of 16"
            elif not tj or "fetch_bytes_raw" not in tj or "write_bytes" not in tj:
                traffic_why = f"{os.path.basename(tpath)} has no FETCH_SIZE/WRITE_SIZE entry for the dominant kernel"
            elif tj["launches"] != nbat * p["n"]:
                traffic_why = (f"{os.path.basename(tpath)} is stale: {tj['launches']} launches of the dominant kernel in {nbat} "
                               f"batches there, {p['n']} per batch in this run")
            else:
                traffic = (2.0 * tj["fetch_bytes_raw"] + tj["write_bytes"]) / tj["launches"]
                traffic_src, traffic_why = os.path.basename(tpath), None
        roof = {"bound": "mfma", "kernel": "gemm_nt_kernel<128,128,4,4,2,0,1> (potrf trailing updates on 128x128 tiles, LDS-DMA staging, v_mfma_f64_16x16x4_f64)",
                "achieved": ach, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_FP64_MFMA_TFLOPS,
                "traffic": traffic, "traffic_source": traffic_src, "traffic_null_because": traffic_why,
                "launches": p["n"], "avg_launch_us": p["ms"] * 1e3 / max(p["n"], 1), "batches_averaged": len(pp),
                "flops_per_launch": p["flops"] / max(p["n"], 1), "evaluations_per_launch": B}
        # every GEMM launch of the batch (the narrow K <= 256 updates on 64x64 tiles included; with factor-ahead their
        # tile (0,0) also factors the next diagonal block, so their durations contain ~10 us of pivots each)
        ctx.prof_begin(abi.PROF_GEMM)
        ctx.loglik_batch_enqueue(np.array([theta(2500 + i) for i in range(B)]))
        p = ctx.prof_end()
        ach = p["flops"] / (p["ms"] * 1e-3) / 1e12
        roof_other["gemm_all_launches"] = {"bound": "mfma", "achieved": ach, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                                           "frac": ach / PEAK_FP64_MFMA_TFLOPS, "launches": p["n"],
                                           "avg_launch_us": p["ms"] * 1e3 / max(p["n"], 1), "flops_per_eval": p["flops"] / B}
        # the compute-bound part: launches with a contraction length of 512 or more
        ctx.prof_begin(abi.PROF_GEMM_K512)
        ctx.loglik_batch_enqueue(np.array([theta(2700 + i) for i in range(B)]))
        p = ctx.prof_end()
        if p["n"] and p["ms"] > 0:
            ach = p["flops"] / (p["ms"] * 1e-3) / 1e12
            roof_other["gemm_k512_and_longer"] = {"bound": "mfma", "achieved": ach, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                                                  "frac": ach / PEAK_FP64_MFMA_TFLOPS, "launches": p["n"],
                                                  "avg_launch_us": p["ms"] * 1e3 / max(p["n"], 1), "flops_per_eval": p["flops"] / B}
        ctx.prof_begin(abi.PROF_POTRF)
        ctx.loglik_batch_enqueue(np.array([theta(3000 + i) for i in range(B)]))
        p = ctx.prof_end()
        roof_other["potrf_whole"] = {"bound": "mfma", "achieved": (N ** 3 / 3.0) * B / (p["ms"] * 1e-3) / 1e12,
                                     "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s", "ms_per_eval": p["ms"] / B,
                                     "evaluations_per_factorisation": B}
        roof_other["potrf_whole"]["frac"] = roof_other["potrf_whole"]["achieved"] / PEAK_FP64_MFMA_TFLOPS
        # covariance fill: the staging launch of a lock-step batch (what the timed region runs: ONE launch fills the B lower
        # triangles), and beside it the one-matrix launch of a sequential caller; algorithmic bytes = the lower tiles written
        ctx.prof_begin(abi.PROF_FILL)
        for i in range(2):
            ctx.loglik_batch_enqueue(np.array([theta(4000 + B * i + k) for k in range(B)]))
        p = ctx.prof_end()
        gbs = p["bytes"] / (p["ms"] * 1e-3) / 1e9
        ctx.prof_begin(abi.PROF_FILL)
        for i in range(3):
            ctx.loglik_enqueue(theta(4100 + i))
        p1 = ctx.prof_end()
        roof_other["cov_fill"] = {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                  "frac": gbs / PEAK_HBM_GBS, "avg_launch_us": p["ms"] * 1e3 / max(p["n"], 1),
                                  "us_per_matrix": p["ms"] * 1e3 / max(p["n"], 1) / B, "matrices_per_launch": B,
                                  "one_matrix_per_launch_us": p1["ms"] * 1e3 / max(p1["n"], 1),
                                  "one_matrix_per_launch_frac": p1["bytes"] / (p1["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                  "note": "plain-fill write bandwidth of the box 6.0-6.8 TB/s (profiles/r03_write_bandwidth.txt)"}
    if rank == 0:
        ctx.sync()
    note("likelihood rooflines done")

    # one evaluation at a time on one context (the shape of a sequential caller: one BFGS run, alloc_emulator_struct;
    # emulator_struct.c:28-32): host call to host result, i.e. `--batch 1 --streams 1`
    single = None
    if not args.no_single:
        for i in range(3):
            ctx.loglik(theta(700000 + i))
        t0 = time.perf_counter()
        ns = 10
        for i in range(ns):
            r1 = ctx.loglik(theta(710000 + i))
            assert r1["status"] == 0 and np.isfinite(r1["value"])
        ts = (time.perf_counter() - t0) / ns
        single = {"ms_per_evaluation": ts * 1e3, "evals_per_s": 1.0 / ts,
                  "frac_of_mfma_peak": (N ** 3 / 3.0) / ts / 1e12 / PEAK_FP64_MFMA_TFLOPS}

    # ---- region B: batched predictions, queries resident in HBM
    pred = None
    if not args.no_predict:
        th0 = synth.default_thetas(kind, d)
        ctx.predict_setup(th0)
        # what a sequential caller pays ONCE per emulator (alloc_emulator_struct, emulator_struct.c:13-37: fill, factorisation
        # with the inverse rows, L^-1 by transposition, C^-1 [y|H], host finishing): host call to host return; and what ONE
        # emulate_point (emulator_struct.c:124-143) costs it afterwards (the skinny split-K product)
        t0 = time.perf_counter()
        nsu = 5
        for i in range(nsu):
            ctx.predict_setup(synth.perturbed_thetas(kind, d, seed, 800000 + i))
        t_setup = (time.perf_counter() - t0) / nsu
        ctx.predict_setup(th0)
        one_q = synth.queries(64, d, seed + 13)
        ctx.predict(one_q[:1])
        t0 = time.perf_counter()
        for i in range(64):
            ctx.predict(one_q[i:i + 1])
        t_one = (time.perf_counter() - t0) / 64
        setup_info = {"predict_setup_ms": t_setup * 1e3, "flops": 2.0 * N ** 3 / 3.0,
                      "frac_of_mfma_peak": (2.0 * N ** 3 / 3.0) / t_setup / 1e12 / PEAK_FP64_MFMA_TFLOPS,
                      "emulate_point_latency_ms": t_one * 1e3,
                      "note": "gpemu_predict_setup = alloc_emulator_struct (one-off per emulator); emulate_point = one query "
                              "through gpemu_predict_batch, host buffers, host call to host result"}
        nb = 20
        per = -(-nq // nb)
        # nb * per DISTINCT query points, resident in HBM before the clock starts; every batch reads its own block
        Xq = synth.queries(nb * per, d, seed + 11 + rank)
        dq, dm, dv = ctx.dev_alloc(Xq.nbytes), ctx.dev_alloc(nb * per * 8), ctx.dev_alloc(nb * per * 8)
        ctx.upload(dq, Xq)
        for _ in range(min(W, 2)):
            ctx.predict_dev(per, dq, dm, dv)
        ctx.sync()
        barrier()
        t0 = time.perf_counter()
        for b in range(nb):
            ctx.predict_dev(per, C.c_void_p(dq.value + b * per * d * 8), C.c_void_p(dm.value + b * per * 8),
                            C.c_void_p(dv.value + b * per * 8))
        ctx.sync()
        barrier()
        tB = time.perf_counter() - t0
        if distributed:
            tt = torch.tensor([tB], dtype=torch.float64, device=tdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            tB = float(tt.item())
        mean = ctx.download(dm, (nb * per,))
        var = ctx.download(dv, (nb * per,))
        assert np.all(np.isfinite(mean)) and np.all(np.isfinite(var)) and len(np.unique(mean[:4096])) > 4000
        pred = {"value": ngpus * nb * per / tB, "unit": "predictions/s", "points_per_rank": nb * per,
                "distinct_points": True, "batches": nb, "ms_per_batch": tB / nb * 1e3, "sequential_caller": setup_info}
        # the same sweep through the host-buffer entry (queries uploaded, results downloaded per call: PCIe inclusive)
        t0 = time.perf_counter()
        hb = min(4, nb)
        for b in range(hb):
            ctx.predict(Xq[b * per:(b + 1) * per])
        pred["host_buffer_entry_predictions_per_s"] = hb * per / (time.perf_counter() - t0)

    note("region B (predictions) done")
    # ---- region F: interactive_mode THROUGH THE DROP-IN (the metric's "predictions/sec ... inside interactive_mode")
    inter = None
    if rank == 0 and ngpus == 1 and pred is not None and not args.no_interactive:
        inter = interactive_cli_region(kind, N, d, order, seed, dev, args.interactive_queries or nq, lambda Q: ctx.predict(Q))
        for k in ("text", "binary"):
            if isinstance(inter.get(k), dict) and "value" in inter[k]:
                inter[k]["fraction_of_resident_rate"] = inter[k]["value"] / pred["value"]
                inter[k]["fraction_of_host_buffer_rate"] = inter[k]["value"] / pred["host_buffer_entry_predictions_per_s"]
        note("region F (interactive_mode through the CLI) done")
    # ---- region C: value + gradient (evalFnGradMulti, maxmultimin.c:615-618 -- what estimate_thetas calls per BFGS
    #      step), lock-step batches of Bg on a pow-exp model of the same N and d; as in region A the batches are dealt to
    #      `--streams` contexts through the asynchronous entry (gpemu_loglik_grad_batch_enqueue / _collect_back): the panel
    #      chain and the host finishing of one context's batch overlap the GEMMs of the other's; every batch is collected
    vg = None
    if not args.no_grad:
        Bg = args.grad_batch if args.grad_batch else min(B, 16)
        Kg = args.grad_steps if args.grad_steps else max(2, min(K, 8))
        gns = min(nstreams, 2)
        gctxs = [abi.Context(dev) for _ in range(gns)]
        Xg_, yg_ = synth.design(N, d, seed + 1)
        for gc in gctxs:
            gc.set_model(1, 0, Xg_, yg_)

        def gtheta(i):
            return synth.perturbed_thetas(1, d, seed + 1, rank + world_size * i)
        for j in range(2):
            for gc in gctxs:
                gc.loglik_grad_batch(np.array([gtheta(50000 + 31 * j + i) for i in range(Bg)]))
        barrier()
        gpend = [[] for _ in gctxs]
        gvals = np.full((Kg, Bg), np.nan)
        ggrad_ok = True
        t0 = time.perf_counter()
        for j in range(Kg):
            s_ = j % gns
            gc = gctxs[s_]
            if len(gpend[s_]) == RING - 1:
                r = gc.loglik_grad_batch_collect_back(RING - 2, Bg)
                assert np.all(r["status"] == 0), r
                gvals[gpend[s_].pop(0)] = r["value"]
                ggrad_ok = ggrad_ok and bool(np.all(np.isfinite(r["grad"])))
            gc.loglik_grad_batch_enqueue(np.array([gtheta(j * Bg + i) for i in range(Bg)]))
            gpend[s_].append(j)
        for s_, gc in enumerate(gctxs):
            while gpend[s_]:
                r = gc.loglik_grad_batch_collect_back(len(gpend[s_]) - 1, Bg)
                assert np.all(r["status"] == 0), r
                gvals[gpend[s_].pop(0)] = r["value"]
                ggrad_ok = ggrad_ok and bool(np.all(np.isfinite(r["grad"])))
        barrier()
        tC = time.perf_counter() - t0
        assert np.all(np.isfinite(gvals)) and ggrad_ok, "a value or gradient of the timed region is not finite"
        if distributed:
            tt = torch.tensor([tC], dtype=torch.float64, device=tdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            tC = float(tt.item())
        gflops = float(N) ** 3 * Bg * Kg                # 2N^3/3 (factorisation with the inverse rows) + N^3/3 (C^-1 = U U^T)
        ach = gflops / tC / 1e12
        # the same batch through the blocking entry on ONE context (what a single lock-step group sees per round)
        t0 = time.perf_counter()
        for j in range(2):
            gctxs[0].loglik_grad_batch(np.array([gtheta(90000 + j * Bg + i) for i in range(Bg)]))
        t_block = (time.perf_counter() - t0) / 2
        vg = {"value": ngpus * Kg * Bg / tC, "unit": "value+gradient evals/s", "steps": Kg, "evaluations_per_step": Bg,
              "contexts": gns, "ms_per_evaluation": tC / (Kg * Bg) * 1e3, "workload": f"pow-exp, N={N}, d={d}, regression_order=0",
              "one_context_blocking_ms_per_batch": t_block * 1e3, "one_context_blocking_evals_per_s": Bg / t_block,
              "gradient": "literal (reference formulas)" if not (gctxs[0].get_mode() & abi.MODE_EXACT_GRAD) else "exact",
              "roofline": {"bound": "mfma", "achieved": ach, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                           "frac": ach / PEAK_FP64_MFMA_TFLOPS, "flops_per_eval": float(N) ** 3,
                           "note": "whole host-visible region (staging, factorisation with inverse rows, U U^T, gradient "
                                   "reductions, host finishing, every batch collected) against N^3 algorithmic flops per evaluation"}}
        for gc in gctxs:
            gc.close()

    note("region C (value+gradient) done")

    # ---- region D: estimate_thetas THROUGH THE DROP-IN (libEmuMI + the interactive_emulator CLI, a child process): the
    #      product's own restart pool -- lock-step groups of BFGS threads over gpemu_loglik_grad_batch_* -- trains the
    #      pow-exp model of region C from an INPUT_MODEL_FILE with a fixed seed and a bounded run list; what is reported is
    #      what that search achieved (its own GPEMU_SEARCH_STATS line), beside the raw C-ABI figure of region C
    #      With --gpus N > 1 the SAME search (same run list, same seed) runs as one CLI process per GPU (ranks.c): the runs are
    #      dealt to the ranks and one gpemu_rccl_allgather picks the winner; value_grad_evals_per_s then counts all ranks'
    #      evaluations over the slowest rank's search time, and best_loglik / snapshot_sha256 are those of the N=1 line.
    import shutil
    import tempfile
    rdv = [tempfile.mkdtemp(prefix="gpemu_bench_ranks_") if (rank == 0 and distributed) else None]
    if distributed:
        dist.broadcast_object_list(rdv, src=0)
    train = None
    if not args.no_train and not args.no_grad and (rank == 0 or distributed):
        barrier()
        mine = train_through_cli(N, d, seed + 1, dev, args.train_runs, exact=not args.train_literal,
                                 rank=rank, world=world_size, local_rank=local_rank, rendezvous=rdv[0])
        if distributed:
            ok = "value_grad_evals" in mine
            worst = allreduce_max(mine["search_seconds"] if ok else 0.0)
            failed = allreduce_max(0.0 if ok else 1.0)
            tot = torch.tensor([float(mine.get("value_grad_evals", 0)), float(mine.get("runs", 0))], dtype=torch.float64, device=tdev)
            dist.all_reduce(tot)
            if rank == 0:
                train = mine
                if ok and not failed:
                    train.update(value_grad_evals=int(tot[0].item()), runs=int(tot[1].item()), search_seconds=worst,
                                 value_grad_evals_per_s=float(tot[0].item()) / worst, processes=world_size,
                                 gather="files (GPEMU_GATHER=file: ranks sharing a device)" if os.environ.get("GPEMU_GATHER") == "file"
                                 else "gpemu_rccl_allgather (RCCL), one per search")
                    for k in ("value_only_evals", "answered_from_cache", "device_rounds", "mean_requests_per_round", "bfgs_iterations",
                              "ms_per_bfgs_iteration_per_run", "host_threads", "lockstep_groups"):
                        train[k + "_rank0"] = train.pop(k)
                elif failed:
                    train = {"error": mine.get("error", "another rank's CLI failed")}
        else:
            train = mine
        if train and vg and "value_grad_evals_per_s" in train:
            train["fraction_of_raw_value_grad"] = train["value_grad_evals_per_s"] / vg["value"]
        if not distributed and not args.train_literal and train and "error" not in train:
            # the same search with the reference's LITERAL gradient formulas (the CLI's default mode, emulator.c:173-209 /
            # maxmultimin.c:416-550): the reference's own behaviour, driver-timed beside the exact one
            lit = train_through_cli(N, d, seed + 1, dev, args.train_runs, exact=False)
            if vg and "value_grad_evals_per_s" in lit:
                lit["fraction_of_raw_value_grad"] = lit["value_grad_evals_per_s"] / vg["value"]
            train["literal_gradient_search"] = lit
        note("region D (estimate_thetas through the C layer) done")

    # ---- region E: BASELINE configs[3] -- the 8 PCA components of an N=4096, d=16, t=9 multi-output model
    #      (multivar_support.c:20-28: independent scalar GPs on one design), component c -> rank c mod W
    #      (shard.farm_components); every component gets the same number of likelihood evaluations, so the total work is
    #      FIXED as ranks are added: the strong-scaling figure the ">= 6x at 8 GPUs" target is quoted on
    pca8 = None
    if not args.no_pca8:
        pca8 = pca8_region(abi, shard, synth, dev, rank, world_size, args.pca8_steps, barrier,
                           (lambda t: allreduce_max(t)) if distributed else (lambda t: t))
        if rank == 0 and ngpus == 1 and not args.no_interactive:
            pca8["emulator_through_cli"] = pca8_emulator_cli(dev)
        note("region E (8 PCA components) done")
    # ---- region F: configs[3] TRAINED by the C product as one process per GPU (ranks.c): the bench ranks each start the CLI
    #      with their rank; the CLI processes gather the thetas among themselves through RCCL (gpemu_rccl_allgather)
    ranks_cli = None
    if distributed and train is not None and "error" in train:
        ranks_cli = {"skipped": "the CLI ranks of region D failed; not tried again"}
    d_failed = allreduce_max(1.0 if ranks_cli is not None else 0.0) > 0 if distributed else False
    if not args.no_pca8 and not args.no_train and not d_failed:
        barrier()
        t0 = time.perf_counter()
        with tempfile.TemporaryDirectory(prefix="gpemu_bench_") as own:
            # 50 runs per component: the reference's own restart count per job (estimate_threaded.c:113)
            mine = pca8_train_ranks_cli(dev, rank, world_size, local_rank, rdv[0] if distributed else own, restarts=50)
        f_wall, f_fail, f_evals = time.perf_counter() - t0, 1.0 if "error" in mine else 0.0, float(mine.get("value_grad_evals_here", 0))
        if distributed:
            f_wall, f_fail = allreduce_max(f_wall), allreduce_max(f_fail)
            tot = torch.tensor([f_evals], dtype=torch.float64, device=tdev)
            dist.all_reduce(tot)
            f_evals = float(tot.item())
        if rank == 0:
            ranks_cli = dict(mine, wall_seconds_max_over_ranks=f_wall, ranks_failed=bool(f_fail > 0), processes=world_size,
                             value_grad_evals_all_ranks=int(f_evals),
                             value_grad_evals_per_s_wall=f_evals / f_wall if f_wall > 0 else None)
            raw = (pca8 or {}).get("raw_value_grad")
            if raw and f_wall > 0:
                # against the raw C-ABI rate of ONE GPU at this size (pca8.raw_value_grad): at --gpus 1 the fraction of the device
                # rate the whole CLI process -- start-up, parsing, PCA, the searches, the snapshot -- delivers; at N GPUs N x raw
                ranks_cli["raw_value_grad_evals_per_s_one_gpu"] = raw["value"]
                ranks_cli["fraction_of_raw_wall"] = ranks_cli["value_grad_evals_per_s_wall"] / (raw["value"] * world_size)
                if mine.get("train_and_dump_s"):
                    ranks_cli["fraction_of_raw_in_training_phase_rank0"] = float(mine.get("value_grad_evals_here", 0)) / mine["train_and_dump_s"] / raw["value"]
        note("region F (configs[3] trained as one CLI process per GPU) done")
    # ---- roofline of the prediction GEMM (the likelihood rooflines were taken right behind region A)
    if rank == 0:
        if pred is not None:
            ctx.predict_setup(synth.default_thetas(kind, d))
            ctx.prof_begin(abi.PROF_GEMM)
            ctx.predict_dev(per, dq, dm, dv)
            p = ctx.prof_end()
            ach = p["flops"] / (p["ms"] * 1e-3) / 1e12
            roof_other["predict_gemm"] = {"bound": "mfma", "achieved": ach, "peak": PEAK_FP64_MFMA_TFLOPS,
                                          "unit": "TFLOP/s", "frac": ach / PEAK_FP64_MFMA_TFLOPS,
                                          "flops_per_prediction": p["flops"] / per}

    note("roofline sections done")
    cpu = None
    if rank == 0 and ngpus == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(kind, order, N, d, seed, [int(v) for v in args.cpu_sizes.split(",")])

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
            "interactive_mode_cli": inter,
            "value_grad": vg,
            "estimate_thetas_c_layer": train,
            "pca8": pca8,
            "pca8_trained_by_cli_ranks": ranks_cli,
            "single_evaluation": single,
            "rccl_ranks": (world_size if (distributed and backend == "nccl") else (1 if not distributed else 0)),
            "process_group": (backend if distributed else None),
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
        if rank == 0 and rdv[0]:
            shutil.rmtree(rdv[0], ignore_errors=True)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
