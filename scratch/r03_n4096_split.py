"""where a lock-step batch of 64 evaluations at N=4096 (BASELINE configs[1] / the components of configs[3]) spends its
time: GEMM by K, leaves, fill (HIP events per class), for outer panel widths 1024 / 2048 / 4096"""
import sys, os, re, subprocess, collections, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    from madaiemulator_amd import abi, synth
    B, N, d = int(sys.argv[2]), 4096, int(sys.argv[3])
    ctx = abi.Context(0)
    X, y = synth.design(N, d, 5)
    ctx.set_model(1, 0, X, y)
    ths = np.array([synth.perturbed_thetas(1, d, 7, i) for i in range(B)])
    for i in range(3):
        ctx.loglik_batch(ths)
    for cls, name in ((abi.PROF_GEMM, "GEMM"), (abi.PROF_LEAF, "LEAF"), (abi.PROF_FILL, "FILL"), (abi.PROF_POTRF, "POTRF")):
        ctx.prof_begin(cls); ctx.loglik_batch_enqueue(ths); p = ctx.prof_end(); ctx.loglik_batch_collect()
        print("TOTAL", name, "n %d ms %.3f" % (p["n"], p["ms"]))
    t0 = time.perf_counter()
    for i in range(5): ctx.loglik_batch(ths)
    print("WALL ms per batch %.3f -> %.1f evals/s (one context)" % ((time.perf_counter() - t0) / 5 * 1e3, B * 5 / (time.perf_counter() - t0)))
    sys.exit(0)
for nbt in ("0", "1024", "4096"):
    for B in ("64", "32"):
        env = dict(os.environ, GPEMU_PROF_DUMP="1", GPEMU_NB_TOP=nbt)
        out = subprocess.run([sys.executable, __file__, "child", B, "8"], env=env, capture_output=True, text=True)
        acc = collections.OrderedDict()
        for line in out.stderr.splitlines():
            m = re.search(r"gemm m=(\d+) n=(\d+) k=(\d+) tri=(\d) flops=([\d.e+]+) ms=([\d.]+)", line)
            if m:
                k = int(m.group(3)); a = acc.setdefault(k, [0, 0.0, 0.0]); a[0] += 1; a[1] += float(m.group(5)); a[2] += float(m.group(6))
        print("NB_TOP", nbt, "batch", B, " ".join("K=%d:%d launches %.2f ms %.0f TF" % (k, n, ms, fl / ms / 1e9) for k, (n, fl, ms) in sorted(acc.items())))
        print("   ", " | ".join(l for l in out.stdout.splitlines() if l.startswith(("TOTAL", "WALL"))))
        if out.returncode: print('child failed', out.returncode, out.stderr[-1500:])
