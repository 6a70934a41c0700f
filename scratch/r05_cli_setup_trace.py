"""start-up of `interactive_emulator interactive_mode` on the c3 and the pca8 snapshots with GPEMU_SETUP_TRACE=1 / GPEMU_IO_STATS=1:
the phases of every alloc_emulator_struct and the process wall time to the first answer"""
import os, sys, time, subprocess, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from madaiemulator_amd import build, synth
env = dict(os.environ, GPEMU_DEVICE="0", GPEMU_IO_STATS="1", GPEMU_SETUP_TRACE="1")
with tempfile.TemporaryDirectory() as tmp:
    snaps = {}
    X, y = synth.design(8192, 8, 20261003 + 2)
    snaps["c3"] = (os.path.join(tmp, "c3.txt"), 8)
    open(snaps["c3"][0], "w").write(synth.single_output_snapshot(X, y, 3, 1, synth.default_thetas(3, 8)))
    N, d, nt = 4096, 16, 9
    X, y = synth.design(N, d, 20261003 + 3)
    Y = synth.multi_outputs(X, y, nt)
    Z, evals, evecs, ybar = synth.pca_zmatrix(Y)
    ths = [synth.perturbed_thetas(1, d, 77, c) for c in range(Z.shape[1])]
    snaps["pca8"] = (os.path.join(tmp, "pca8.txt"), d)
    open(snaps["pca8"][0], "w").write(synth.snapshot_text(X, Y, evals, evecs, Z, 1, 0, ths))
    for name, (snap, dd) in snaps.items():
        for extra_env in ({},):
            for rep in range(2):
                q = " ".join("0.5" for _ in range(dd)) + "\n"
                t0 = time.perf_counter()
                p = subprocess.run([build.CLI_BIN, "interactive_mode", snap, "-q"], input=q.encode(), capture_output=True, env=dict(env, **extra_env))
                w = time.perf_counter() - t0
                print("== %s %s rep %d: process wall %.3f s, rc %d" % (name, extra_env, rep, w, p.returncode))
                print(p.stderr.decode()[-3000:])
