"""region F's CLI run (configs[3] trained through `interactive_emulator estimate_thetas`) under rocprofv3 --kernel-trace:
writes the INPUT_MODEL_FILE, runs the CLI itself under the profiler (the program after `--` is the C binary), then prints the
device busy time of the run (tools/rocpd_timeline.py).  usage: python scratch/r05_regionF_trace.py restarts [components_per_slot]"""
import os, sys, subprocess, tempfile, glob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from madaiemulator_amd import build, synth
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
restarts = sys.argv[1]
N, d, nt = 4096, 16, 9
X, y = synth.design(N, d, 20261003 + 3)
Y = synth.multi_outputs(X, y, nt) + 0.05 * synth.normal(12, N * nt).reshape(N, nt)
tmp = tempfile.mkdtemp(prefix="r05F_")
inp, snap = os.path.join(tmp, "in.dat"), os.path.join(tmp, "snap.txt")
with open(inp, "w") as f:
    f.write(f"{nt}\n{d}\n{N}\n"); np.savetxt(f, X, fmt="%.17g"); np.savetxt(f, Y, fmt="%.17g")
env = dict(os.environ, GPEMU_DEVICES="0", GPEMU_SEED="20261004", GPEMU_JOBS="1", GPEMU_RESTARTS=restarts, GPEMU_SEARCH_STATS="1", TMPDIR="/tmp")
if len(sys.argv) > 2: env["GPEMU_COMPONENTS_PER_SLOT"] = sys.argv[2]
out = os.path.join(tmp, "prof")
cmd = ["rocprofv3", "--kernel-trace", "-d", out, "-o", "r", "--", build.CLI_BIN, "estimate_thetas", inp, snap, "--covariance_fn=1", "--regression_order=0",
       "--pca_variance=1.0", "--exact_gradient"]
p = subprocess.run(cmd, env=env, cwd="/tmp", capture_output=True, text=True, timeout=600)
print("\n".join(l for l in p.stderr.splitlines() if l.startswith("# search stats") or l.startswith("# cli phases")))
dbs = glob.glob(os.path.join(out, "**", "*.db"), recursive=True)
print(subprocess.run([sys.executable, os.path.join(R, "tools", "rocpd_timeline.py"), dbs[0]], capture_output=True, text=True).stdout)
print(subprocess.run([sys.executable, os.path.join(R, "tools", "rocpd_summary.py"), dbs[0]], capture_output=True, text=True).stdout[:3000])
