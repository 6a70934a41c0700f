"""what the CLI does as TWO ranks with the RCCL transport on ONE device (the box has one GPU; two RCCL ranks cannot share a
device): the start-up handshake carries the ncclUniqueId, both ranks call ncclCommInitRank -- expected: an error from RCCL in
at least one rank, the other ended by the watchdog, both promptly with status 1 (no hang)."""
import os, sys, subprocess, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madaiemulator_amd import build
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
inp = os.path.join(R, "tests", "golden", "ref_inputs", "multi-simple.input_model_file.dat")
with tempfile.TemporaryDirectory() as tmp:
    procs = []
    t0 = time.time()
    for r in range(2):
        env = dict(os.environ, GPEMU_RANK=str(r), GPEMU_WORLD_SIZE="2", GPEMU_RENDEZVOUS_DIR=tmp, GPEMU_DEVICES="0", GPEMU_SEED="3", GPEMU_RESTARTS="2",
                   NCCL_DEBUG="WARN")
        procs.append(subprocess.Popen([build.CLI_BIN, "estimate_thetas", inp, os.path.join(tmp, f"snap{r}"), "--regression_order=1"], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for r, p in enumerate(procs):
        try:
            so, se = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            p.kill(); so, se = p.communicate(); print("rank", r, "TIMED OUT")
        print("rank", r, "returncode", p.returncode, "after %.1f s" % (time.time() - t0))
        print(se[-1500:])
    print("files left:", sorted(os.listdir(tmp)))
