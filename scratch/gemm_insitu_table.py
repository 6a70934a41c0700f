"""per-launch HIP-event times of the 127 GEMM launches of one lock-step batch of 16 (N=8192), grouped by K"""
import sys, os, re, subprocess, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    from madaiemulator_amd import abi, synth
    B, N = 16, 8192
    ctx = abi.Context(0)
    X, y = synth.design(N, 8, 5)
    ctx.set_model(3, 1, X, y)
    ths = np.array([synth.perturbed_thetas(3, 8, 7, i) for i in range(B)])
    for i in range(3):
        ctx.loglik_batch(ths)
    ctx.prof_begin(abi.PROF_GEMM); ctx.loglik_batch_enqueue(ths); p = ctx.prof_end(); ctx.loglik_batch_collect()
    print("TOTAL", p)
    sys.exit(0)
env = dict(os.environ, GPEMU_PROF_DUMP="1")
out = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
acc = collections.OrderedDict()
for line in out.stderr.splitlines():
    m = re.search(r"gemm m=(\d+) n=(\d+) k=(\d+) tri=(\d) flops=([\d.e+]+) ms=([\d.]+)", line)
    if m:
        k = int(m.group(3)); fl = float(m.group(5)); ms = float(m.group(6))
        a = acc.setdefault(k, [0, 0.0, 0.0]); a[0] += 1; a[1] += fl; a[2] += ms
        if k >= 512: print("  m=%s n=%s k=%d  %.3f ms  %.1f TF/s" % (m.group(1), m.group(2), k, ms, fl / ms / 1e9))
for k, (n, fl, ms) in sorted(acc.items()):
    print("K=%5d launches %3d  flops %.3e  ms %.3f  TF/s %.1f" % (k, n, fl, ms, fl / ms / 1e9))
print(out.stdout[-300:])
if out.returncode: print('child failed', out.returncode, out.stderr[-2000:])
