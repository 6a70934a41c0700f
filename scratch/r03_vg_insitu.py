"""per-launch HIP-event times of the GEMM launches of one value+gradient batch of 16 (pow-exp, N=8192), grouped by shape
class: factorisation updates by K, the right-hand-side-row launches, the identity-row launches, the C^-1 = U U^T product"""
import sys, os, re, subprocess, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    from madaiemulator_amd import abi, synth
    B, N = 16, 8192
    ctx = abi.Context(0)
    X, y = synth.design(N, 8, 5)
    ctx.set_model(1, 0, X, y)
    ths = np.array([synth.perturbed_thetas(1, 8, 7, i) for i in range(B)])
    for i in range(2):
        ctx.loglik_grad_batch(ths)
    for cls, name in ((abi.PROF_GEMM, "GEMM"), (abi.PROF_LEAF, "LEAF"), (abi.PROF_FILL, "FILL"), (abi.PROF_POTRF, "POTRF")):
        ctx.prof_begin(cls); ctx.loglik_grad_batch_enqueue(ths); p = ctx.prof_end(); ctx.loglik_grad_batch_collect()
        print("TOTAL", name, p)
    import time
    t0 = time.perf_counter(); ctx.loglik_grad_batch(ths); print("WALL ms", (time.perf_counter() - t0) * 1e3)
    sys.exit(0)
env = dict(os.environ, GPEMU_PROF_DUMP="1")
out = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
acc = collections.OrderedDict()
first = True
for line in out.stderr.splitlines():
    m = re.search(r"gemm m=(\d+) n=(\d+) k=(\d+) tri=(\d) flops=([\d.e+]+) ms=([\d.]+)", line)
    if m:
        mm, n, k, tri = int(m.group(1)), int(m.group(2)), int(m.group(3)), int(m.group(4)); fl = float(m.group(5)); ms = float(m.group(6))
        key = ("UUT corner" if (k >= 8192 and mm == n) else "rhs rows" if mm == 64 else "identity rows" if (tri == 0 and k < 8192) else "update") + " K=%d" % k
        a = acc.setdefault(key, [0, 0.0, 0.0]); a[0] += 1; a[1] += fl; a[2] += ms
tot_fl = tot_ms = 0
for k, (n, fl, ms) in acc.items():
    print("%-28s launches %3d  flops %.3e  ms %8.3f  TF/s %.1f" % (k, n, fl, ms, fl / ms / 1e9 if ms else 0)); tot_fl += fl; tot_ms += ms
print("all GEMM launches: %.3e flops in %.2f ms = %.1f TF/s" % (tot_fl, tot_ms, tot_fl / tot_ms / 1e9))
print(out.stdout[-900:])
if out.returncode: print('child failed', out.returncode, out.stderr[-2000:])
