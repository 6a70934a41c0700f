"""The C host layer (csrc/host, libEmuMI.so) that mirrors the reference's libEmu interface and the
interactive_emulator CLI, driven the way the reference's callers drive them; results checked against the oracle."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import scipy.linalg as sl

from madaiemulator_amd import build, synth
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INP = os.path.join(ROOT, "tests", "golden", "ref_inputs")
UNI = os.path.join(INP, "uni-simple.input_model_file.dat")
TWOD = os.path.join(INP, "uni-2d-param.input_model_file.dat")
MULTI = os.path.join(INP, "multi-simple.input_model_file.dat")
RTOL = 1e-8


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    build.build_all()
    exe = str(tmp_path_factory.mktemp("drv") / "host_api_driver")
    subprocess.check_call(["gcc", "-std=gnu99", "-O1", "-I", os.path.join(ROOT, "include"), "-I", build.HOST_SRC,
                           "-o", exe, os.path.join(ROOT, "tests", "c", "host_api_driver.c"),
                           "-L", build.LIBDIR, "-lEmuMI", "-lgpemu_hip", f"-Wl,-rpath,{build.LIBDIR}", "-lm"])
    return exe


def run(cmd, **kw):
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, **kw)
    assert out.returncode == 0, out.stderr[-3000:]
    return out.stdout


def parse(stdout):
    res = {}
    for line in stdout.splitlines():
        if line.startswith("#") or not line.strip():
            continue
        k, *vals = line.split()
        try:
            res.setdefault(k, []).append([float(v) for v in vals])
        except ValueError:
            pass
    return res


def test_host_library_exports_reference_symbols():
    import ctypes
    build.build_all()
    lib = ctypes.CDLL(build.HIP_LIB, mode=ctypes.RTLD_GLOBAL)
    lib = ctypes.CDLL(build.HOST_LIB)
    for name in ("evalFnMulti", "gradFnMulti", "evalFnGradMulti", "estimateSigmaFull", "maxWithMultiMin",
                 "doOptimizeMultiMin", "set_random_init_value", "estimate_thetas_threaded", "alloc_modelstruct_2",
                 "free_modelstruct_2", "dump_modelstruct_2", "load_modelstruct_2", "set_global_ptrs",
                 "fill_sample_scales_vec", "setup_optimization_ranges", "alloc_emulator_struct", "free_emulator_struct",
                 "emulate_point", "makeCovMatrix_fnptr", "makeKVector_fnptr", "makeHMatrix_fnptr", "makeHVector_linear",
                 "covariance_fn_gaussian", "covariance_fn_matern_three", "covariance_fn_matern_five",
                 "alloc_multimodelstruct", "gen_pca_decomp", "dump_multi_modelstruct", "load_multi_modelstruct",
                 "alloc_multi_emulator", "emulate_point_multi", "emulate_point_multi_pca", "estimate_multi",
                 "chol_inverse_cov_matrix", "estimateBeta", "estimateSigma", "getLogLikelyhood", "makeEmulatedMean",
                 "makeEmulatedVariance", "callEvalLhoodList", "evalFnMultiList", "emulate_points", "makeCovMatrix",
                 "makeKVector", "makeHMatrix", "covariance_fn", "makeHVector", "makeGradMatLength", "getGradientCn",
                 "makeHMatrix_es", "makeCovMatrix_es", "makeKVector_es", "estimateBeta_es", "callEstimate",
                 "callEmulateAtList", "callEmulateAtPt", "derivative_l_gauss", "derivative_l_matern_three",
                 "derivative_l_matern_five", "setupEmulateMC", "callEmulateMC",
                 "freeEmulateMC", "setupEmulateMCMulti", "callEmulateMCMulti", "freeEmulateMCMulti"):
        assert hasattr(lib, name), name


def test_documented_consumer_build_lines_build_and_export_the_installed_headers(tmp_path):
    """INTEGRATION.md's "Build lines for a consumer" block, run VERBATIM from the repository root (it went stale once: two
    source files were missing from the list), must give a libEmuMI.so that exports every function the reference's installed
    headers declare (tests/golden/installed_header_symbols.txt, made by make_installed_header_symbols.py from
    CMakeLists.txt:99 / src/CMakeLists.txt:50-51) -- a consumer that links it in place of libEmu links -- and a CLI that runs."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = text[text.index("Build lines for a consumer"):]
    block = block[block.index("```\n") + 4:]
    block = block[:block.index("```")]
    lines = [ln for ln in block.splitlines() if ln.strip()]
    assert len(lines) == 3 and lines[0].startswith("hipcc ") and lines[1].startswith("gcc ") and lines[2].startswith("gcc "), lines
    out = tmp_path / "out"
    out.mkdir()
    env = dict(os.environ, OUT=str(out), PATH=os.environ.get("PATH", "") + ":/opt/rocm/bin")
    for ln in lines:
        r = subprocess.run(["bash", "-c", "set -e -o pipefail; " + ln], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (ln, r.stderr[-3000:])
    nm = subprocess.run(["nm", "-D", "--defined-only", str(out / "libEmuMI.so")], capture_output=True, text=True, check=True).stdout
    have = {ln.split()[-1] for ln in nm.splitlines() if ln.strip()}
    want = [ln.split() for ln in open(os.path.join(ROOT, "tests", "golden", "installed_header_symbols.txt")) if ln.strip()]
    missing = [(h, n) for h, n in want if n not in have]
    # declared in libEmu/emulator.h:23, defined by no source file of the reference: nothing to provide
    assert missing == [["libEmu/emulator.h", "covariance_fn_gaussian_exact"]] or missing == [("libEmu/emulator.h", "covariance_fn_gaussian_exact")], missing
    for name in ("copy_modelstruct", "copy_optstruct", "free_optstruct", "setup_cov_fn", "setup_regression"):   # round-4 verdict
        assert name in have
    r = subprocess.run([str(out / "interactive_emulator")], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and r.stderr.startswith("useage:")
    # the package's own build (build.py) exports the same set
    build.build_all()
    nm2 = subprocess.run(["nm", "-D", "--defined-only", build.HOST_LIB], capture_output=True, text=True, check=True).stdout
    assert {ln.split()[-1] for ln in nm2.splitlines() if ln.strip()} == have


def test_installed_header_helper_functions(tmp_path):
    """copy_modelstruct / copy_optstruct / free_optstruct / setup_cov_fn / setup_regression (src/modelstruct.c:35-49,
    src/optstruct.c:8-119) and the rest of csrc/host/legacy_api.c with the reference's semantics: deep copies, the
    process-wide function pointers, nthetas / nregression_fns forced, dump -> load round trips; called as
    libEmu/estimate_threaded.c:57-68 (setup_params) and libRbind/rbind.c:61-62, 689-691 call them.  No device."""
    build.build_all()
    exe = str(tmp_path / "legacy_api_driver")
    subprocess.check_call(["gcc", "-std=gnu99", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", build.HOST_SRC,
                           "-o", exe, os.path.join(ROOT, "tests", "c", "legacy_api_driver.c"),
                           "-L", build.LIBDIR, "-lEmuMI", "-lgpemu_hip", f"-Wl,-rpath,{build.LIBDIR}", "-lm", "-lpthread"])
    out = subprocess.run([exe, str(tmp_path / "dump.txt")], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "legacy api ok" in out.stdout, out.stderr[-2000:]
    assert "setup_cov_fn has changed nthetas from 99" in out.stderr              # optstruct.c:107-108


def test_host_logic_without_gpu(tmp_path):
    """mt19937 known answers, regression basis, PCA decomposition (Jacobi eigen-solver) against numpy, snapshot
    dump -> load -> dump byte identity: the parts of the host mirror that never touch the device"""
    build.build_all()
    exe = str(tmp_path / "host_cpu_driver")
    subprocess.check_call(["gcc", "-std=gnu99", "-O1", "-I", os.path.join(ROOT, "include"), "-I", build.HOST_SRC,
                           "-o", exe, os.path.join(ROOT, "tests", "c", "host_cpu_driver.c"),
                           "-L", build.LIBDIR, "-lEmuMI", "-lgpemu_hip", f"-Wl,-rpath,{build.LIBDIR}", "-lm"])
    s1, s2, dfile = tmp_path / "snap1", tmp_path / "snap2", tmp_path / "deriv.bin"
    out = subprocess.run([exe, MULTI, str(s1), str(s2), str(dfile)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    res = parse(out.stdout)
    # MT19937, init_genrand(5489): Matsumoto & Nishimura's reference output
    assert [int(v) for v in res["mt"][0]] == [3499211612, 581869302, 3890346734, 3586334585]
    assert res["uniform"][0][0] == 3499211612 / 4294967296.0
    X, Y = synth.read_input_model_file(MULTI)
    d = X.shape[1]
    x0 = X[0]
    assert res["h0"][0] == [1.0]
    assert np.allclose(res["h1"][0], np.concatenate([[1.0], x0]), rtol=0, atol=0)
    assert np.allclose(res["h2"][0], np.concatenate([[1.0], x0, x0 ** 2]), rtol=1e-15)
    assert np.allclose(res["h3"][0], np.concatenate([[1.0], x0, x0 ** 2, x0 ** 3]), rtol=1e-15)
    # PCA: eigen-decomposition of the (1/N) covariance of the centred outputs (multi_modelstruct.c:172-338)
    Yc = Y - Y.mean(axis=0)
    w, V = np.linalg.eigh(Yc.T @ Yc / len(Y))
    w, V = w[::-1], V[:, ::-1]
    nr = int(res["nr"][0][0])
    assert 1 <= nr <= Y.shape[1]
    assert np.allclose(res["evals"][0], w[:nr], rtol=1e-10)
    E = np.array(res["evec"])
    for r in range(nr):                                   # eigenvectors up to sign
        assert min(np.max(np.abs(E[:, r] - V[:, r])), np.max(np.abs(E[:, r] + V[:, r]))) < 1e-8
    z0 = (Yc[0] @ E) / np.sqrt(np.array(res["evals"][0]))
    assert np.allclose(res["z0"][0], z0, rtol=1e-9, atol=1e-12)
    assert s1.read_bytes() == s2.read_bytes() and len(s1.read_bytes()) > 100
    # the harness's own PCA (synth.pca_zmatrix: what bench.py's configs[3] region decomposes with) is the product's:
    # same number of kept components at the same variance fraction, same eigenvalues, same z up to the eigenvector signs
    Zs, ls, Us, ybar = synth.pca_zmatrix(Y, 0.99)
    assert Zs.shape[1] == nr and np.allclose(ls, res["evals"][0], rtol=1e-10) and np.allclose(ybar, Y.mean(axis=0))
    assert np.allclose(np.abs(Zs[0]), np.abs(res["z0"][0]), rtol=1e-8, atol=1e-11)
    # a6: derivative_l_matern_three / _five (emulator.c:401-433, 497-532), the literal recurrence with the carried
    # rtemp -- host code, bit for bit the oracle's restatement
    n = len(X)
    dd = np.fromfile(dfile, dtype=np.float64).reshape(2, n, n)
    assert np.array_equal(dd[0], O.derivative_l(2, X, 0.7, 2))
    assert np.array_equal(dd[1], O.derivative_l(3, X, 0.7, 2))
    assert not np.allclose(dd[0], dd[0].T)               # path dependent: not even symmetric


def _noisy_model_file(path, N=150, d=2, seed=31337, nt=1):
    X, y = synth.design(N, d, seed)
    Y = np.stack([y + 0.15 * synth.normal(99 + t, N) + 0.3 * t * X[:, t % d] for t in range(nt)], axis=1)
    with open(path, "w") as f:
        f.write(f"{nt}\n{d}\n{N}\n")
        np.savetxt(f, X, fmt="%.17g")
        np.savetxt(f, Y, fmt="%.17g")


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="needs a machine WITHOUT a HIP device (the GPU form is test_device_error_in_a_threaded_search...)")
def test_threaded_search_without_a_device_ends_with_status_1_not_a_signal(tmp_path):
    """the host error path (csrc/host/fatal.c): a threaded search whose device context cannot be made -- every lock-step group
    of every component thread fails at its first round, under the group mutex, with the sibling threads waiting -- ends
    like the reference does where it cannot go on (maxmultimin.c:495: message, exit status 1): ONE message, status 1, no
    signal, promptly.  Three components over three device slots, two groups each: six failing threads at once."""
    build.build_all()
    inp, snap = tmp_path / "in.dat", tmp_path / "snap.txt"
    _noisy_model_file(inp, nt=4)
    env = dict(os.environ, GPEMU_DEVICES="0,0,0", GPEMU_SEED="7", GPEMU_RESTARTS="24")
    out = subprocess.run([build.CLI_BIN, "estimate_thetas", str(inp), str(snap), "--pca_variance=0.999"], env=env, capture_output=True,
                         text=True, timeout=60)
    assert out.returncode == 1, (out.returncode, out.stderr[-2000:])
    assert out.stderr.count("gpemu error") == 1 and "no usable HIP device" in out.stderr, out.stderr[-2000:]


def test_run_list_layout_over_threads_groups_and_slots():
    """how estimate_thetas_threaded deals a run list to host threads, lock-step groups and device slots
    (optimizer.c gpemu_host_plan_groups: pure arithmetic, no GPU): full groups of 16 first, two groups per slot, every slot
    gets work, thread ranges tile [0, nthreads) without gaps"""
    import ctypes as C
    build.build_all()
    C.CDLL(build.HIP_LIB, mode=C.RTLD_GLOBAL)
    lib = C.CDLL(build.HOST_LIB)
    ia = C.c_int * 512

    def plan(total, lockstep, per_slot, nslots):
        lo, hi, slot, nt = ia(), ia(), ia(), C.c_int(0)
        g = lib.gpemu_host_plan_groups(total, lockstep, per_slot, nslots, C.byref(nt), lo, hi, slot, 512)
        groups = [(lo[i], hi[i], slot[i]) for i in range(g)]
        assert groups[0][0] == 0 and groups[-1][1] == nt.value and all(a[1] == b[0] for a, b in zip(groups, groups[1:]))
        assert all(0 < h - l <= lockstep for l, h, _ in groups)
        return nt.value, groups
    assert plan(50, 16, 2, 1) == (32, [(0, 16, 0), (16, 32, 0)])                 # the default search on one GPU
    assert plan(20, 16, 2, 1) == (20, [(0, 16, 0), (16, 20, 0)])                 # full group first, then the remainder
    assert plan(3, 16, 2, 1) == (3, [(0, 3, 0)])
    nt, g = plan(50, 16, 2, 8)                                                   # 8 GPUs: every slot gets a group
    assert nt == 50 and len(g) == 8 and sorted(s for _, _, s in g) == list(range(8)) and {h - l for l, h, _ in g} <= {6, 7}
    nt, g = plan(400, 16, 2, 8)                                                  # enough runs: two full groups per GPU
    assert nt == 256 and len(g) == 16 and [s for _, _, s in g] == list(range(8)) * 2 and all(h - l == 16 for l, h, _ in g)
    nt, g = plan(300, 16, 2, 2)
    assert nt == 64 and [s for _, _, s in g] == [0, 1, 0, 1]
    assert plan(24, 5, 2, 1) == (10, [(0, 5, 0), (5, 10, 0)])
    assert lib.gpemu_host_plan_groups(0, 16, 2, 1, None, ia(), ia(), ia(), 512) == 0


def test_device_slots_from_the_environment():
    """GPEMU_DEVICES parsing and the slot -> device map of the host layer (device_bridge.c "devices"): host logic, no GPU"""
    import sys
    code = ("import ctypes as C, sys\n"
            f"C.CDLL({build.HIP_LIB!r}, mode=C.RTLD_GLOBAL); L = C.CDLL({build.HOST_LIB!r})\n"
            "n = L.gpemu_host_device_slots()\n"
            "print(n, [L.gpemu_host_slot_device(i) for i in range(n + 2)], L.gpemu_host_device())\n"
            "L.gpemu_host_thread_device(5); print(L.gpemu_host_device(), L.gpemu_host_thread_device_get())\n"
            "L.gpemu_host_thread_device(-1); print(L.gpemu_host_device())\n")
    build.build_all()
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, GPEMU_DEVICES="3,1,1"), capture_output=True,
                         text=True, timeout=60)
    assert out.returncode == 0, out.stderr[-1000:]
    lines = out.stdout.strip().splitlines()
    assert lines[0] == "3 [3, 1, 1, 3, 1] 3"          # three slots, wrapping, slot 0 is the calling thread's default
    assert lines[1] == "5 5" and lines[2] == "3"      # a thread working for a slot declares its device; -1: back to slot 0
    env = dict(os.environ)
    env.pop("GPEMU_DEVICES", None)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and int(out.stdout.split()[0]) >= 1      # every visible device, at least one slot


G6SNAP = os.path.join(ROOT, "tests", "golden", "g6_multi_snapshot.txt")


def test_snapshot_golden_g7_roundtrips_byte_for_byte(driver, tmp_path):
    """G7: the hand-written MODEL_SNAPSHOT_FILE fixture (reference grammar and printf formats, multi_modelstruct.c:346-401,
    modelstruct.c:375-409) goes through load_multi_modelstruct -> dump_multi_modelstruct unchanged.  Host code only."""
    out = tmp_path / "g6_again.txt"
    assert "nt 6 nr 3 N 100 d 3" in run([driver, "roundtrip", G6SNAP, str(out)])
    assert out.read_bytes() == open(G6SNAP, "rb").read()


def test_print_thetas_mode(tmp_path):
    """`interactive_emulator print_thetas MODEL_SNAPSHOT_FILE` (interactive_emulator.c:455-510 of the reference): the table of
    exp(theta) per PCA component with the component's share of the kept variance ("%lf" fields, tab separated), and one
    pca_emu_summary_<i>.dat per component (design columns and the component's training values) in the working directory.
    Host logic only: no device is touched."""
    out = subprocess.run([build.CLI_BIN, "print_thetas", G6SNAP], capture_output=True, text=True, timeout=60, cwd=tmp_path)
    assert out.returncode == 0, out.stderr
    sd = parse_snapshot(open(G6SNAP).read().split())
    lines = out.stdout.splitlines()
    assert lines[0] == "#-- EMULATOR LENGTH SCALES (thetas) IN PCA SPACE -- #"
    assert lines[1] == "#-- id\tpca-var\tScale\tNugget" + "".join(f"\tlength_{k}" for k in range(sd["d"])) + " -- #"
    assert len(lines) == 2 + sd["nr"]
    for i, comp in enumerate(sd["models"]):
        want = "%d\t" % i + "%f\t" % (sd["evals"][i] / sd["evals"].sum()) + "".join("%f\t" % np.exp(t) for t in comp["thetas"])
        assert lines[2 + i] == want
        rows = np.loadtxt(tmp_path / f"pca_emu_summary_{i}.dat")
        assert rows.shape == (sd["N"], sd["d"] + 1)
        assert np.allclose(rows[:, :-1], comp["X"], atol=5e-7) and np.allclose(rows[:, -1], comp["z"], atol=5e-7)


def test_cli_usage_and_argument_errors(tmp_path):
    """the argument handling of interactive_emulator.c:255-345,520-545 of the reference: the usage text for too few
    arguments, -h, an unknown mode or a missing file name; the reference's messages for files that cannot be opened; a
    --pca_variance outside [0, 1] reported and replaced by 0.95 (and, the kept quirk, -v switching on --pca_output and
    --quiet).  Host logic only."""
    cli = build.CLI_BIN
    def go(*args):
        return subprocess.run([cli, *args], capture_output=True, text=True, timeout=60, cwd=tmp_path)
    for args in ((), ("interactive_mode",), ("-h", "x", "y"), ("no_such_mode", "a", "b"), ("estimate_thetas", "only_one_file")):
        r = go(*args)
        assert r.returncode != 0 and r.stderr.startswith("useage:") and "print_thetas MODEL_SNAPSHOT_FILE" in r.stderr, args
    r = go("print_thetas", "no_such_snapshot")
    assert r.returncode != 0 and "Error opening file" in r.stderr
    r = go("estimate_thetas", "no_such_input", "snap")
    assert r.returncode != 0 and "Input File read failed." in r.stderr
    r = go("print_thetas", G6SNAP, "--pca_variance=2.5")
    assert r.returncode == 0 and "# err pca_variance argument given incorrect value: 2.500000" in r.stderr
    assert "# using default value: 0.950000" in r.stderr and "# var-frac: 0.950000" in r.stderr


def test_harness_snapshot_writer_is_the_products_dump_byte_for_byte(driver, tmp_path):
    """synth.snapshot_text / single_output_snapshot (what bench.py's interactive_mode regions feed the CLI: a
    MODEL_SNAPSHOT_FILE at SUPPLIED thetas) written in the reference's grammar and printf formats
    (multi_modelstruct.c:346-401, modelstruct.c:375-409): load_multi_modelstruct -> dump_multi_modelstruct reproduces the
    file byte for byte, for a 4-component multi-output model and for a scalar Matern model.  Host code only."""
    X, y = synth.design(256, 4, 3)
    Y = synth.multi_outputs(X, y, 5)
    Z, evals, evecs, _ = synth.pca_zmatrix(Y)
    ths = [synth.perturbed_thetas(1, 4, 77, c) for c in range(Z.shape[1])]
    a, b = tmp_path / "multi.txt", tmp_path / "multi_again.txt"
    a.write_text(synth.snapshot_text(X, Y, evals, evecs, Z, 1, 0, ths))
    assert "nt 5 nr 4 N 256 d 4" in run([driver, "roundtrip", str(a), str(b)])
    assert a.read_bytes() == b.read_bytes()
    c, d_ = tmp_path / "single.txt", tmp_path / "single_again.txt"
    c.write_text(synth.single_output_snapshot(X, y, 3, 1, synth.default_thetas(3, 4)))
    assert "nt 1 nr 1 N 256 d 4" in run([driver, "roundtrip", str(c), str(d_)])
    assert c.read_bytes() == d_.read_bytes()


@pytest.mark.gpu
def test_multi_output_golden_g6_through_the_c_layer(driver, tmp_path):
    """G6: load the fixture snapshot, alloc_multi_emulator, emulate_point_multi at the fixture's 16 queries (13 random +
    3 training points): observable-space means and variances against the golden vectors"""
    g2 = np.load(os.path.join(ROOT, "tests", "golden", "golden_v2.npz"))
    qf = tmp_path / "q.dat"
    np.savetxt(qf, g2["g6_q"], fmt="%.17g")
    res = parse(run([driver, "multi", G6SNAP, str(qf)]))
    pred = np.array(res["pred"]).reshape(len(g2["g6_q"]), -1, 2)
    assert np.max(np.abs(pred[:, :, 0] - g2["g6_mean"])) < 1e-8 * max(1.0, np.abs(g2["g6_mean"]).max())
    assert np.max(np.abs(pred[:, :, 1] - g2["g6_var"])) < 1e-8 * max(1e-3, np.abs(g2["g6_var"]).max())


@pytest.mark.gpu
def test_interactive_mode_stream_lone_points_and_binary_framing(tmp_path):
    """interactive_mode through the CLI on the G6 snapshot (nt = 6 outputs, 3 PCA components, d = 3): (a) 40 000 points
    piped at once (several pipelined device batches) give, for the fixture's 16 queries, the golden observable-space
    values, and every point's answer equals -- byte for byte -- the answer it gets when it is sent alone and waited for
    (interactive_emulator.c:414-441: one point per loop turn); (b) --binary (the reference's BINARY_INTERACTIVE_MODE
    framing, :392-396,418-438) returns the doubles whose "%.17f" rendering is the text output; (c) -z keeps nt pairs."""
    import select
    import time
    g2 = np.load(os.path.join(ROOT, "tests", "golden", "golden_v2.npz"))
    cli = build.CLI_BIN
    Q = np.vstack([g2["g6_q"], synth.queries(40000 - len(g2["g6_q"]), 3, 77)])
    text = "\n".join(" ".join(repr(float(v)) for v in row) for row in Q) + "\n"
    p = subprocess.run([cli, "interactive_mode", G6SNAP, "-q"], input=text.encode(), capture_output=True, timeout=300,
                       env=dict(os.environ, GPEMU_IO_STATS="1"))
    assert p.returncode == 0, p.stderr[-2000:]
    lines = p.stdout.decode().split("\n")[:-1]
    assert len(lines) == 40000 * 6 * 2
    vals = np.array(lines, float).reshape(40000, 6, 2)
    n16 = len(g2["g6_q"])
    assert np.max(np.abs(vals[:n16, :, 0] - g2["g6_mean"])) < 1e-8 * max(1.0, np.abs(g2["g6_mean"]).max())
    assert np.max(np.abs(vals[:n16, :, 1] - g2["g6_var"])) < 1e-8 * max(1e-3, np.abs(g2["g6_var"]).max())
    stats = [l for l in p.stderr.decode().splitlines() if l.startswith("# interactive stats")]
    assert stats and int(stats[0].split()[4]) == 40000 and int(stats[0].split()[8]) >= 256       # points; waiting points ride in batches
    # one point at a time, each answer waited for
    q = subprocess.Popen([cli, "interactive_mode", G6SNAP, "-q"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    for i in list(range(5)) + [20000, 39999]:
        q.stdin.write((" ".join(repr(float(v)) for v in Q[i]) + "\n").encode())
        q.stdin.flush()
        got, t0 = b"", time.time()
        while got.count(b"\n") < 12:
            assert time.time() - t0 < 120, "no answer to a lone point"
            if select.select([q.stdout], [], [], 1.0)[0]:
                got += os.read(q.stdout.fileno(), 65536)
        alone = np.array(got.decode().split(), float).reshape(6, 2)
        # (a lone point takes the skinny split-K product, a point inside a batch the unsplit GEMM: equal to rounding)
        assert np.max(np.abs(alone[:, 0] - vals[i, :, 0])) < 1e-11 * max(1.0, np.abs(vals[i, :, 0]).max())
        assert np.max(np.abs(alone[:, 1] - vals[i, :, 1])) < 1e-11 * max(1e-3, np.abs(vals[i, :, 1]).max())
    q.stdin.close()
    assert q.wait(timeout=60) == 0
    # binary framing: the same batches, raw doubles in and out
    b = subprocess.run([cli, "interactive_mode", G6SNAP, "-q", "--binary"], input=Q.tobytes(), capture_output=True, timeout=300)
    assert b.returncode == 0, b.stderr[-2000:]
    raw = np.frombuffer(b.stdout, dtype=np.float64)
    assert raw.size == 40000 * 6 * 2
    assert ["%.17f" % v for v in raw] == lines
    # header (not quiet) and pca-space output: nt pairs per point, those beyond nr are zeros
    z = subprocess.run([cli, "interactive_mode", G6SNAP, "-z"], input=text[:200].rsplit("\n", 1)[0].encode() + b"\n",
                       capture_output=True, timeout=120)
    zl = z.stdout.decode().split()
    npts = len(text[:200].rsplit("\n", 1)[0].split()) // 3
    zz = np.array(zl, float).reshape(npts, 6, 2)                   # -z implies -q (the reference's fall-through)
    assert np.all(zz[:, 3:, :] == 0.0) and np.all(zz[:, :3, 1] > 0.0)


@pytest.mark.gpu
@pytest.mark.parametrize("path,d,order", [(UNI, 1, 0), (UNI, 1, 1), (TWOD, 2, 1)])
def test_evalfn_gradfn_like_gsl_multimin_would_call_them(driver, path, d, order):
    X, Y = synth.read_input_model_file(path)
    y = Y[:, 0]
    th = np.array([-3.0] + [-0.4 - 0.3 * k for k in range(d)])
    res = parse(run([driver, "eval", path, "1", str(order)] + [repr(float(t)) for t in th]))
    o = O.eval_fn_multi(1, order, X, y, th)
    g, _ = O.grad_fn_multi(1, order, X, y, th)
    assert res["evalFnMulti"][0][0] == pytest.approx(o["value"], rel=RTOL)
    assert res["estimateSigmaFull"][0][0] == pytest.approx(o["sigma2"], rel=RTOL)
    assert np.allclose(res["gradFnMulti"][0], g, rtol=1e-7, atol=1e-7 * np.abs(g).max())
    assert res["evalFnGradMulti"][0][0] == pytest.approx(o["value"], rel=RTOL)
    assert np.allclose(res["evalFnGradMulti"][0][1:], g, rtol=1e-7, atol=1e-7 * np.abs(g).max())


@pytest.mark.gpu
def test_evalfn_matern_returns_nan_like_the_reference(driver):
    # amp is zeroed by evalFnMulti and read raw by the Matern kernels -> C = theta_1 I with theta_1 < 0 -> GSL_NAN
    out = subprocess.run([driver, "eval", UNI, "2", "0", "-3.0", "0.0"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0
    assert "evalfnmulti nan" in out.stdout.lower()
    assert "non postive def" in out.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("cov,order", [(1, 0), (1, 1), (3, 1), (2, 0)])
def test_emulator_struct_and_emulate_point(driver, tmp_path, cov, order):
    X, Y = synth.read_input_model_file(TWOD)
    y = Y[:, 0]
    th = np.array([-0.2, -3.5, -1.0, -0.7]) if cov == 1 else np.array([1.3, 0.02, np.log(0.8)])
    Q = np.vstack([synth.queries(40, 2, 3), X[:2]])
    qf = tmp_path / "q.dat"
    np.savetxt(qf, Q, fmt="%.17g")
    res = parse(run([driver, "emu", TWOD, str(cov), str(order), str(qf)] + [repr(float(t)) for t in th]))
    e = O.Emulator(cov, order, X, y, th)
    m, v, _ = e.emulate(Q)
    kappa = O.cov(cov, Q[0], Q[0], th)
    pred = np.array(res["pred"])
    batch = np.array(res["batch"])
    assert np.allclose(res["beta"][0], e.beta, rtol=RTOL)
    assert np.max(np.abs(pred[:, 0] - m)) <= RTOL * max(1.0, np.abs(m).max())
    assert np.max(np.abs(pred[:, 1] - v)) <= RTOL * kappa
    assert np.array_equal(pred, batch)                      # one point at a time == one batch
    assert res["cinverse00"][0][0] == pytest.approx(e.cinverse[0, 0], rel=RTOL)
    assert np.allclose(res["beta_es"][0], e.beta, rtol=RTOL)                # emulator_struct.c:63-118 wrappers
    k0 = O.kvector(cov, X, Q[0], th)
    assert res["kvec_es"][0] == pytest.approx([k0[0], k0[-1]], rel=1e-12, abs=1e-300)
    assert res["hmat_es"][0][0] == pytest.approx(e.H[-1, -1], rel=1e-15)
    # libEmu/emulate-fns.h:13-27 (csrc/host/legacy_api.c): the list, one-point and resultstruct forms are the batched sweep
    # of a freshly made emulator -- the same numbers; the caller's-matrices forms go through the host copy of C^-1
    assert np.array_equal(np.array(res["atlist"]), batch) and np.array_equal(np.array(res["results"]), batch)
    assert res["atpoint"][0] == list(batch[0])
    for key, row in (("quick", 0), ("ith", 1)):
        assert res[key][0][0] == pytest.approx(m[row], rel=RTOL, abs=RTOL) and abs(res[key][0][1] - v[row]) <= RTOL * kappa


@pytest.mark.gpu
@pytest.mark.parametrize("cov,order", [(1, 1), (3, 0)])
def test_lowlevel_host_matrix_interface(driver, tmp_path, cov, order):
    """chol_inverse_cov_matrix, estimateBeta, estimateSigma, getLogLikelyhood, makeEmulatedMean / Variance with the
    N x N matrices in host memory, called in the order the reference's R bindings call them (rbind.c:121-210)"""
    X, Y = synth.read_input_model_file(TWOD)
    y = Y[:, 0]
    th = np.array([-0.2, -3.5, -1.0, -0.7]) if cov == 1 else np.array([1.3, 0.02, np.log(0.8)])
    Q = np.vstack([synth.queries(12, 2, 3), X[:1]])
    qf = tmp_path / "q.dat"
    np.savetxt(qf, Q, fmt="%.17g")
    res = parse(run([driver, "lowlevel", TWOD, str(cov), str(order), str(qf)] + [repr(float(t)) for t in th]))
    e = O.Emulator(cov, order, X, y, th)
    r = y - e.H @ e.beta
    quad = r @ e.cinverse @ r
    assert res["logdet"][0][0] == pytest.approx(e.logdet, rel=RTOL)
    assert np.allclose(res["beta"][0], e.beta, rtol=RTOL)
    assert res["loglik"][0][0] == pytest.approx(-0.5 * e.logdet - len(y) / 2.0 * 1.83788 - 0.5 * quad, rel=RTOL)
    assert res["sigma2"][0][0] == pytest.approx(y @ e.cinverse @ r / len(y), rel=RTOL)
    assert res["gradcn"][0][0] == pytest.approx(-len(y) / 2.0 + 0.5 * (y @ e.cinverse @ y), rel=RTOL)
    m, v, _ = e.emulate(Q)
    pred = np.array(res["pred"])
    kappa = O.cov(cov, Q[0], Q[0], th)
    assert np.max(np.abs(pred[:, 0] - m)) <= RTOL * max(1.0, np.abs(m).max())
    assert np.max(np.abs(pred[:, 1] - v)) <= RTOL * kappa


@pytest.mark.gpu
def test_cli_train_snapshot_predict_end_to_end(driver, tmp_path):
    """test/uni-simple of the reference: estimate_thetas --regression_order=1, then interactive_mode on its
    sample locations (BASELINE.json configs[0]); the snapshot must round-trip byte for byte and the predictions
    must equal the oracle's emulate_point at the thetas the snapshot holds."""
    cli = build.CLI_BIN
    snap = tmp_path / "univariate_snapshot_file"
    env = dict(os.environ, GPEMU_SEED="12345", GPEMU_RESTARTS="4")
    run([cli, "estimate_thetas", UNI, str(snap), "--regression_order=1"], env=env)
    toks = snap.read_text().split()
    nt, nr, d, N, cov, order = (int(t) for t in toks[:6])
    assert (nt, nr, d, N, cov, order) == (1, 1, 1, 34, 1, 1)
    # byte-exact dump -> load -> dump
    snap2 = tmp_path / "snap2"
    run([driver, "roundtrip", str(snap), str(snap2)])
    assert snap.read_bytes() == snap2.read_bytes()
    # thetas sit after: 6 header ints, N*d, N*nt, nr evals, nt*nr evecs, N*nr z, then the model block
    pos = 6 + N * d + N * nt + nr + nt * nr + N * nr
    nthetas = int(toks[pos])
    blk = pos + 10 + 2 * nthetas + N * d
    z = np.array(toks[blk:blk + N], float)
    thetas = np.array(toks[blk + N:blk + N + nthetas], float)
    assert nthetas == 3 and np.all(np.isfinite(thetas))       # BFGS is unbounded (as in the reference): only the start points respect grad_ranges
    X, Y = synth.read_input_model_file(UNI)
    qpath = os.path.join(INP, "uni-simple.sample_locations.dat")
    out = run([cli, "interactive_mode", str(snap)], stdin=open(qpath))
    lines = out.split()
    assert lines[:6] == ["1", "param_0", "2", "mean_0", "variance_0"] + lines[5:6]
    vals = np.array(lines[5:], float).reshape(-1, 2)
    Q = np.array(open(qpath).read().split(), float).reshape(-1, 1)
    assert len(vals) == len(Q) == 100
    # oracle prediction in PCA space, then the reference's back-projection (nt = nr = 1)
    e = O.Emulator(1, 1, X, z, thetas)
    m, v, _ = e.emulate(Q)
    lam = float(toks[6 + N * d + N * nt])
    u = float(toks[6 + N * d + N * nt + nr])
    mean = Y[:, 0].mean() + u * np.sqrt(lam) * m
    var = u * u * lam * v
    assert np.max(np.abs(vals[:, 0] - mean)) < 1e-7 * max(1.0, np.abs(mean).max())
    assert np.max(np.abs(vals[:, 1] - var)) < 1e-7 * max(1e-3, np.abs(var).max())
    # the emulator actually learnt the curve: training points are reproduced
    out2 = run([cli, "interactive_mode", str(snap), "-q"], input="\n".join(repr(float(x)) for x in X[:5, 0]) + "\n")
    got = np.array(out2.split(), float).reshape(-1, 2)
    assert np.max(np.abs(got[:, 0] - Y[:5, 0])) < 1e-6


@pytest.mark.gpu
def test_cli_trains_a_matern_model_with_the_corrected_forms(driver, tmp_path):
    """The reference cannot train a Matern model (amplitude zeroed and read raw -> C = theta_1 I -> exit).
    With --matern_fixed --exact_gradient the CLI trains test/uni-simple with the Matern 5/2 kernel; the snapshot holds
    log-scale amplitude and nugget, and interactive_mode --matern_fixed predicts from it what the oracle's literal
    kernel predicts at (e^theta0, e^theta1, theta2)."""
    cli = build.CLI_BIN
    snap = tmp_path / "matern_snapshot"
    env = dict(os.environ, GPEMU_SEED="31", GPEMU_RESTARTS="4")
    run([cli, "estimate_thetas", UNI, str(snap), "--regression_order=1", "--covariance_fn=3", "--matern_fixed",
         "--exact_gradient"], env=env)
    toks = snap.read_text().split()
    nt, nr, d, N, cov, order = (int(t) for t in toks[:6])
    assert (nt, nr, d, N, cov, order) == (1, 1, 1, 34, 3, 1)
    pos = 6 + N * d + N * nt + nr + nt * nr + N * nr
    nthetas = int(toks[pos])
    blk = pos + 10 + 2 * nthetas + N * d
    z = np.array(toks[blk:blk + N], float)
    thetas = np.array(toks[blk + N:blk + N + nthetas], float)
    assert nthetas == 3 and np.all(np.isfinite(thetas)) and np.any(thetas != 0.0)
    X, Y = synth.read_input_model_file(UNI)
    qpath = os.path.join(INP, "uni-simple.sample_locations.dat")
    out = run([cli, "interactive_mode", str(snap), "-q", "--matern_fixed"], stdin=open(qpath))
    vals = np.array(out.split(), float).reshape(-1, 2)
    Q = np.array(open(qpath).read().split(), float).reshape(-1, 1)
    th_raw = np.array([np.exp(thetas[0]), np.exp(thetas[1]), thetas[2]])
    e = O.Emulator(3, 1, X, z, th_raw)
    m, v, _ = e.emulate(Q)
    lam = float(toks[6 + N * d + N * nt])
    u = float(toks[6 + N * d + N * nt + nr])
    mean = Y[:, 0].mean() + u * np.sqrt(lam) * m
    var = u * u * lam * v
    assert np.max(np.abs(vals[:, 0] - mean)) < 1e-7 * max(1.0, np.abs(mean).max())
    assert np.max(np.abs(vals[:, 1] - var)) < 1e-7 * max(1e-3, np.abs(var).max())
    # the trained emulator reproduces the training outputs (nugget-level error)
    out2 = run([cli, "interactive_mode", str(snap), "-q", "--matern_fixed"],
               input="\n".join(repr(float(x)) for x in X[:5, 0]) + "\n")
    got = np.array(out2.split(), float).reshape(-1, 2)
    assert np.max(np.abs(got[:, 0] - Y[:5, 0])) < 0.05 * np.ptp(Y[:, 0])
    # the snapshot records that its Matern thetas are on the log scale (a trailing line the reference's loader never
    # reads): queried WITHOUT --matern_fixed it is still read in that mode, with a note -- not silently as amp = theta0
    assert snap.read_text().rstrip().endswith("#gpemu matern_log_scale 1")
    env2 = {k: v for k, v in os.environ.items() if k not in ("GPEMU_MATERN_FIXED", "GPEMU_EXACT_GRAD")}
    plain = subprocess.run([cli, "interactive_mode", str(snap), "-q"], stdin=open(qpath), capture_output=True, text=True,
                           timeout=300, env=env2)
    assert plain.returncode == 0 and plain.stdout == out and "log scale" in plain.stderr
    # ... and the round trip keeps the record: load -> dump gives the same bytes
    rt = tmp_path / "matern_rt"
    run([driver, "roundtrip", str(snap), str(rt)], env=env2)
    assert rt.read_bytes() == snap.read_bytes()


@pytest.mark.gpu
def test_training_in_lockstep_group_matches_per_thread_scheme(driver, tmp_path):
    """estimate_thetas_threaded: the restarts as a lock-step group (their likelihood/gradient requests batched on the
    device) find the same optimum as one sequential BFGS run after the other, and its value agrees with the oracle.
    Noisy synthetic outputs, so that the optimum is interior (the reference's toy models are noise-free: their
    likelihood grows without bound as the nugget vanishes)."""
    N, d = 150, 2
    X, y = synth.design(N, d, 31337)
    y = y + 0.15 * synth.normal(99, N)
    f = tmp_path / "noisy.dat"
    f.write_text(f"1\n{d}\n{N}\n" + "\n".join(" ".join(repr(float(v)) for v in row) for row in X) + "\n" +
                 "\n".join(repr(float(v)) for v in y) + "\n")
    best = {}
    for mode in ("8", "1", "threads"):
        env = dict(os.environ, GPEMU_SEED="4711", GPEMU_RESTARTS="16", GPEMU_LOCKSTEP=mode)
        if mode == "threads":        # three worker threads, each with its own device context, 6 restarts per job
            env.update(GPEMU_LOCKSTEP="1", GPEMU_NTHREADS="3", GPEMU_JOBS="3", GPEMU_RESTARTS="6")
        res = parse(run([driver, "train", str(f), "1", "1"], env=env))
        th = np.array(res["thetas"][0])
        val = res["neglogl"][0][0]
        assert np.all(np.isfinite(th)) and np.isfinite(val)
        assert val == pytest.approx(O.eval_fn_multi(1, 1, X, y, th[1:])["value"], rel=1e-6)
        best[mode] = (val, th)
    # 16 random restarts each: both searches end at the same optimum, to the minimiser's own stopping tolerance
    # (|gradient| < 0.1 as in the reference, maxmultimin.c:725)
    for other in ("1", "threads"):
        assert abs(best["8"][0] - best[other][0]) < 1.0, best
        assert np.max(np.abs(best["8"][1] - best[other][1])) < 0.3, best


@pytest.mark.gpu
def test_device_error_in_a_threaded_search_ends_with_status_1_and_one_message(tmp_path):
    """round-4 record: a device error inside a threaded search (a failed graph capture) ended the CLI with signal 11 --
    exit() from a group thread under the group mutex, beside ~30 sibling threads inside HIP calls, two threads entering it
    together.  GPEMU_FAULT_ENQUEUE=3 makes the third value+gradient round of the process report GPEMU_ERR_HIP in the middle
    of a search over three component threads x two lock-step groups: the process must end as the reference does where it
    cannot go on (maxmultimin.c:495, emulate-fns.c:282-285): message on stderr, exit status 1 -- not a signal, not a hang,
    and the message ONCE (csrc/host/fatal.c: single entry, no atexit teardown of the HIP runtime)."""
    build.build_all()
    inp, snap = tmp_path / "in.dat", tmp_path / "snap.txt"
    _noisy_model_file(inp, nt=4)
    env = dict(os.environ, GPEMU_DEVICES="0,0,0", GPEMU_SEED="7", GPEMU_RESTARTS="24", GPEMU_FAULT_ENQUEUE="3")
    out = subprocess.run([build.CLI_BIN, "estimate_thetas", str(inp), str(snap), "--pca_variance=0.999"], env=env, capture_output=True,
                         text=True, timeout=120)
    assert out.returncode == 1, (out.returncode, out.stderr[-2000:])
    assert out.stderr.count("gpemu error 3") == 1 and "injected device error" in out.stderr, out.stderr[-2000:]
    # the same search without the fault trains and writes its snapshot
    del env["GPEMU_FAULT_ENQUEUE"]
    ok = subprocess.run([build.CLI_BIN, "estimate_thetas", str(inp), str(snap), "--pca_variance=0.999"], env=env, capture_output=True,
                        text=True, timeout=300)
    assert ok.returncode == 0 and snap.stat().st_size > 1000, ok.stderr[-2000:]


@pytest.mark.gpu
def test_multi_output_pca_snapshot_and_backprojection(driver, tmp_path):
    """test/multi-simple (N=100, d=3, t=6): PCA keeps nr <= t-1 components; predictions at the training points
    come back in the observable space close to the training outputs."""
    cli = build.CLI_BIN
    snap = tmp_path / "multi_snapshot_file"
    env = dict(os.environ, GPEMU_SEED="777", GPEMU_RESTARTS="2")
    run([cli, "estimate_thetas", MULTI, str(snap), "--regression_order=0"], env=env)
    toks = snap.read_text().split()
    nt, nr, d, N = (int(t) for t in toks[:4])
    assert (nt, d, N) == (6, 3, 100) and 1 <= nr <= nt - 1
    X, Y = synth.read_input_model_file(MULTI)
    qf = tmp_path / "q.dat"
    np.savetxt(qf, X[:10], fmt="%.17g")
    res = parse(run([driver, "multi", str(snap), str(qf)]))
    pred = np.array(res["pred"]).reshape(10, nt, 2)
    # oracle: emulate_point per PCA component at the snapshot's thetas, then the reference's back-projection
    snapd = parse_snapshot(toks)
    m_pca, v_pca = np.empty((10, nr)), np.empty((10, nr))
    for c, comp in enumerate(snapd["models"]):
        e = O.Emulator(comp["cov"], comp["order"], comp["X"], comp["z"], comp["thetas"])
        m_pca[:, c], v_pca[:, c], _ = e.emulate(X[:10])
    ybar = snapd["Y"].mean(axis=0)
    for q in range(10):
        mo, vo = O.pca_backproject(ybar, snapd["evals"], snapd["evecs"], m_pca[q], v_pca[q])
        assert np.max(np.abs(pred[q, :, 0] - mo)) < 1e-7 * max(1.0, np.abs(mo).max())
        assert np.max(np.abs(pred[q, :, 1] - vo)) < 1e-7 * max(1e-3, np.abs(vo).max())
    # and the emulator is a sensible fit of the training outputs (PCA truncation + nugget only)
    assert np.max(np.abs(pred[:, :, 0] - Y[:10])) < 0.2


@pytest.mark.gpu
def test_multi_output_training_over_device_slots_writes_the_same_snapshot(driver, tmp_path):
    """SURVEY 8(e) axis 1 in the C product: estimate_multi (multivar_support.c:20-28) deals the PCA components to the
    device slots of GPEMU_DEVICES, one host thread and one set of device contexts per slot -- here two slots on the
    one GPU of the box.  A component's search does not depend on the slot that ran it, so under a fixed seed the
    snapshot is byte for byte the one the serial loop writes; the predictions of the multi-slot emulator
    (alloc_multi_emulator: component c on slot c mod S) are the serial ones too."""
    cli = build.CLI_BIN
    snaps = {}
    # ("serial": one slot, one component at a time, the reference's loop; "side_by_side": one slot, the slot's components
    # trained by a pool of component threads at the same time -- round 5, the default; two components at a time; slots)
    for name, devs, cps in (("serial", "0", "1"), ("side_by_side", "0", "0"), ("two_at_a_time", "0", "2"), ("two_slots", "0,0", "0"),
                            ("three_slots", "0,0,0", "1")):
        snap = tmp_path / f"snap_{name}"
        env = dict(os.environ, GPEMU_SEED="2024", GPEMU_RESTARTS="2", GPEMU_DEVICES=devs, GPEMU_COMPONENTS_PER_SLOT=cps, GPEMU_SEARCH_STATS="1")
        out = subprocess.run([cli, "estimate_thetas", MULTI, str(snap), "--regression_order=1"], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        snaps[name] = snap.read_bytes()
        assert len(snaps[name]) > 1000
        # every search reports its OWN evaluations (they were process-wide counters before round 5): the same figures
        # whatever ran beside it
        stats = sorted(re.findall(r"# search stats: runs (\d+) threads \d+ groups \d+ slots \d+ value_grad_evals (\d+) value_evals (\d+) cached (\d+)", out.stderr))
        assert len(stats) >= 2
        snaps[name + "_stats"] = stats
    assert snaps["serial"] == snaps["side_by_side"] == snaps["two_at_a_time"] == snaps["two_slots"] == snaps["three_slots"]
    assert snaps["serial_stats"] == snaps["side_by_side_stats"] == snaps["two_slots_stats"]
    X, _ = synth.read_input_model_file(MULTI)
    qf = tmp_path / "q.dat"
    np.savetxt(qf, X[:7] + 0.01, fmt="%.17g")
    preds = [run([driver, "multi", str(tmp_path / "snap_serial"), str(qf)], env=dict(os.environ, GPEMU_DEVICES=devs))
             for devs in ("0", "0,0")]
    assert preds[0] == preds[1] and "pred" in preds[0]


@pytest.mark.gpu
def test_eight_pca_components_over_eight_device_slots_write_the_serial_snapshot(tmp_path):
    """BASELINE configs[3]'s shape in small (t = 9 outputs -> 8 PCA components at variance fraction 1) through the CLI:
    GPEMU_DEVICES=0,0,0,0,0,0,0,0 -- eight device slots, here all on the one GPU of the box, component c trained by the
    host thread of slot c (multi.c, multivar_support.c:20-28) -- writes byte for byte the snapshot of the serial loop
    (GPEMU_DEVICES=0); so does a two-slot run.  On an 8-GPU node the same command line puts one component on each GPU."""
    cli = build.CLI_BIN
    N, d, nt = 160, 4, 9
    X, y = synth.design(N, d, 8088)
    Y = synth.multi_outputs(X, y, nt) + 0.05 * synth.normal(12, N * nt).reshape(N, nt)
    f = tmp_path / "multi9.dat"
    with open(f, "w") as fh:
        fh.write(f"{nt}\n{d}\n{N}\n")
        np.savetxt(fh, X, fmt="%.17g")
        np.savetxt(fh, Y, fmt="%.17g")
    snaps = {}
    for name, devs in (("serial", "0"), ("eight_slots", "0,0,0,0,0,0,0,0"), ("two_slots", "0,0")):
        snap = tmp_path / f"snap_{name}"
        env = dict(os.environ, GPEMU_SEED="77", GPEMU_RESTARTS="3", GPEMU_DEVICES=devs)
        run([cli, "estimate_thetas", str(f), str(snap), "--regression_order=0", "--pca_variance=1.0"], env=env)
        snaps[name] = snap.read_bytes()
    toks = snaps["serial"].split()
    assert int(toks[0]) == nt and int(toks[1]) == 8                     # nr = nt - 1 (multi_modelstruct.c:267-272)
    assert snaps["serial"] == snaps["eight_slots"] == snaps["two_slots"]


def test_rank_gather_file_transport_three_processes(tmp_path):
    """ranks.c: gpemu_host_allgather between three processes named by GPEMU_RANK / GPEMU_WORLD_SIZE, through the file
    transport (GPEMU_GATHER=file; the RCCL transport needs one GPU per rank): two gathers in a row (sequence numbers keep
    them apart), every rank ends with every share in rank order.  Host logic, no GPU."""
    code = (
        "import ctypes, os, sys\n"
        "from madaiemulator_amd import build\n"
        "L = ctypes.CDLL(build.HOST_LIB)\n"
        "L.gpemu_host_allgather.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.POINTER(ctypes.c_double)]\n"
        "r, w = L.gpemu_host_rank(), L.gpemu_host_world_size()\n"
        "out = []\n"
        "for n in (3, 5):\n"
        "    send = (ctypes.c_double * n)(*[100.0 * r + i + 0.25 * n for i in range(n)])\n"
        "    recv = (ctypes.c_double * (n * w))()\n"
        "    L.gpemu_host_allgather(send, n, recv)\n"
        "    out.append(list(recv))\n"
        "print(r, w, out)\n")
    procs = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              cwd=ROOT,
                              env=dict(os.environ, GPEMU_RANK=str(r), GPEMU_WORLD_SIZE="3", GPEMU_GATHER="file",
                                       GPEMU_RENDEZVOUS_DIR=str(tmp_path)))
             for r in range(3)]
    want = [[100.0 * r + i + 0.25 * n for r in range(3) for i in range(n)] for n in (3, 5)]
    for r, p in enumerate(procs):
        so, se = p.communicate(timeout=120)
        assert p.returncode == 0, se
        assert so.strip() == f"{r} 3 {want}"
    # a rank outside the world is refused, and so is a world without a rendezvous directory
    bad = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True,
                         cwd=ROOT,
                         env=dict(os.environ, GPEMU_RANK="3", GPEMU_WORLD_SIZE="3", GPEMU_RENDEZVOUS_DIR=str(tmp_path)))
    assert bad.returncode != 0 and "outside" in bad.stderr
    env = dict(os.environ, GPEMU_RANK="0", GPEMU_WORLD_SIZE="2")
    env.pop("GPEMU_RENDEZVOUS_DIR", None)
    bad = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env,
                         cwd=ROOT)
    assert bad.returncode != 0 and "GPEMU_RENDEZVOUS_DIR" in bad.stderr


_RANK_CODE = (
    "import ctypes, os, signal, sys, time\n"
    "from madaiemulator_amd import build\n"
    "L = ctypes.CDLL(build.HOST_LIB)\n"
    "L.gpemu_host_allgather.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.POINTER(ctypes.c_double)]\n"
    "r, w = L.gpemu_host_rank(), L.gpemu_host_world_size()\n"
    "time.sleep(float(os.environ.get('TEST_SLEEP_BEFORE_JOIN', '0')))\n"
    "L.gpemu_host_rank_device()\n"                         # the start-up rendezvous (ranks.c join_run)
    "what = os.environ.get('TEST_AFTER_JOIN', '')\n"
    "if what == 'fatal': L.gpemu_host_fatal(b'rank %d: boom\\n', r)\n"
    "if what == 'kill': os.kill(os.getpid(), signal.SIGKILL)\n"
    "time.sleep(float(os.environ.get('TEST_SLEEP_BEFORE_GATHER', '0')))\n"
    "send = (ctypes.c_double * 2)(10.0 * r, 10.0 * r + 1)\n"
    "recv = (ctypes.c_double * (2 * w))()\n"
    "L.gpemu_host_allgather(send, 2, recv)\n"
    "L.gpemu_host_ranks_finish()\n"
    "print(r, list(recv))\n")


def _start_ranks(world, rdv, per_rank_env=None, ranks=None, **common):
    procs = {}
    for r in (range(world) if ranks is None else ranks):
        env = dict(os.environ, GPEMU_RANK=str(r), GPEMU_WORLD_SIZE=str(world), GPEMU_GATHER="file", GPEMU_RENDEZVOUS_DIR=str(rdv))
        env.update({k: str(v) for k, v in common.items()})
        env.update((per_rank_env or {}).get(r, {}))
        procs[r] = subprocess.Popen([sys.executable, "-c", _RANK_CODE], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, env=env)
    return procs


def test_ranks_second_run_in_a_directory_that_holds_an_earlier_runs_files(tmp_path):
    """ranks.c run nonce (round-4 advisor: file names carried only a per-process sequence number, so a reused
    GPEMU_RENDEZVOUS_DIR handed a run the previous run's shares -- wrong thetas, no error -- or a dead communicator's id):
    a directory littered with what a crashed run leaves (its run_id, a go file, acks, gather files under the old AND the new
    naming, an rccl id) does not disturb a new run; a delayed rank 0 is waited for; a clean run leaves the directory as it
    found it."""
    import time
    stale = {"run_id": b"4242-deadbeef00000".ljust(48, b"\0"), "go_4242-deadbeef00000": b"4242-deadbeef00000".ljust(48, b"\0") * 3 + b"\0" * 128,
             "ack_4242-deadbeef00000_1": b"77-1".ljust(48, b"\0"), "gather_0_0.bin": np.array([666.0, 666.0]).tobytes(),
             "gather_0_1.bin": np.array([666.0, 666.0]).tobytes(), "gather_4242-deadbeef00000_0_1.bin": np.array([666.0, 666.0]).tobytes(),
             "rccl_id_0": b"x" * 128}
    for name, data in stale.items():
        (tmp_path / name).write_bytes(data)
    want = [10.0 * r + i for r in range(3) for i in range(2)]
    t0 = time.time()
    procs = _start_ranks(3, tmp_path, per_rank_env={0: {"TEST_SLEEP_BEFORE_JOIN": "1.5"}})      # ranks 1, 2 meet the stale run_id first
    for r, p in procs.items():
        so, se = p.communicate(timeout=60)
        assert p.returncode == 0, (r, se[-2000:])
        assert so.strip() == f"{r} {want}"
    assert time.time() - t0 < 30
    left = sorted(f.name for f in tmp_path.iterdir())
    assert left == sorted(n for n in stale if n != "run_id" and n != "ack_4242-deadbeef00000_1") or left == sorted(n for n in stale if n != "run_id"), left
    # a second run right behind it, same directory
    procs = _start_ranks(3, tmp_path)
    for r, p in procs.items():
        so, se = p.communicate(timeout=60)
        assert p.returncode == 0 and so.strip() == f"{r} {want}", (r, se[-2000:])


@pytest.mark.parametrize("how", ["fatal", "kill"])
def test_ranks_do_not_wait_for_a_rank_that_is_gone(tmp_path, how):
    """a rank that ends on the layer's error path after the rendezvous (fatal.c drops failed_<run>_<rank>) or is killed
    outright (its pid disappears): the others sit in the gather, which has no deadline of its own (training may take
    hours), and must end promptly with a message and status 1 instead of waiting (round 4: 600 s / 3600 s) -- the watchdog
    thread of ranks.c."""
    import time
    t0 = time.time()
    procs = _start_ranks(3, tmp_path, per_rank_env={1: {"TEST_AFTER_JOIN": how}})
    # (the survivors first: the killed rank stays an unreaped zombie of this test process meanwhile, as under a launcher that
    # waits for its ranks one after the other -- its pid still answers kill(pid, 0), /proc tells)
    for r in (0, 2):
        so, se = procs[r].communicate(timeout=60)
        assert procs[r].returncode == 1, (r, so, se[-2000:])
        # (rank 1's end, or the end it caused in the other survivor: whichever marker the watchdog met first)
        assert f"rank {r}: rank " in se and "of this run" in se and "not waiting for it" in se, se
    so1, se1 = procs[1].communicate(timeout=60)
    assert procs[1].returncode == (1 if how == "fatal" else -9), se1
    if how == "fatal":
        assert "rank 1: boom" in se1
    assert time.time() - t0 < 30


def test_ranks_rendezvous_gives_up_on_a_rank_that_never_starts(tmp_path):
    """the start-up rendezvous has a deadline (launch skew, GPEMU_RENDEZVOUS_WAIT_S), the only one there is"""
    for ranks, msg in (((1,), "rank 1: rank 0 has not opened a run"), ((0,), "rank 0: rank 1 has not joined")):
        procs = _start_ranks(2, tmp_path, ranks=ranks, GPEMU_RENDEZVOUS_WAIT_S="1")
        for r, p in procs.items():
            so, se = p.communicate(timeout=60)
            assert p.returncode == 1 and msg in se, (r, se[-2000:])


def _run_ranks(cmd, world, rendezvous, env, snap_of_rank):
    """start `world` processes of the CLI, rank r writing to snap_of_rank(r); returns their stdouts"""
    procs = []
    for r in range(world):
        e = dict(env, GPEMU_RANK=str(r), GPEMU_WORLD_SIZE=str(world), GPEMU_GATHER="file", GPEMU_RENDEZVOUS_DIR=str(rendezvous))
        procs.append(subprocess.Popen([c if c != "@SNAP@" else str(snap_of_rank(r)) for c in cmd], env=e,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for r, p in enumerate(procs):
        so, se = p.communicate(timeout=900)
        assert p.returncode == 0, (r, se[-2000:])
        outs.append(so)
    return outs


@pytest.mark.gpu
def test_one_process_per_gpu_ranks_write_the_serial_snapshot(tmp_path):
    """SURVEY 8(e) as PROCESSES (ranks.c): `interactive_emulator estimate_thetas` started W times with GPEMU_RANK /
    GPEMU_WORLD_SIZE.  A multi-output model deals its PCA components to the ranks (multivar_support.c:20-28), a
    single-output model the runs of its run list (estimate_threaded.c:101-113); ONE all-gather of a few doubles per rank
    ends the search and rank 0 writes the snapshot -- byte for byte the one a single process writes, for W = 2 and for a
    W larger than the number of components / not dividing the run list.  The ranks here share the one GPU of the box, so the gather goes through files
    (GPEMU_GATHER=file); on a node each rank has its own GPU and the gather is gpemu_rccl_allgather."""
    cli = build.CLI_BIN
    base = dict(os.environ, GPEMU_SEED="2024", GPEMU_RESTARTS="2", GPEMU_DEVICES="0")
    serial = tmp_path / "serial"
    run([cli, "estimate_thetas", MULTI, str(serial), "--regression_order=1"], env=base)
    for world in (2, 4):                                            # (4 ranks for 3 components: one rank trains nothing)
        rdv = tmp_path / f"rdv_multi_{world}"
        rdv.mkdir()
        _run_ranks([cli, "estimate_thetas", MULTI, "@SNAP@", "--regression_order=1"], world, rdv, base,
                   lambda r: tmp_path / f"multi_w{world}_r{r}")
        assert (tmp_path / f"multi_w{world}_r0").read_bytes() == serial.read_bytes()
        for r in range(1, world):
            assert not (tmp_path / f"multi_w{world}_r{r}").exists()          # rank 0 alone writes
    # single-output model: the run list (GPEMU_JOBS x GPEMU_RESTARTS = 10 runs) over 2 and over 4 ranks
    N, d = 220, 3
    X, y = synth.design(N, d, 1357)
    y = y + 0.2 * synth.normal(9, N)
    f = tmp_path / "one.dat"
    _write_model_file(f, X, y)
    base = dict(os.environ, GPEMU_SEED="31", GPEMU_JOBS="2", GPEMU_RESTARTS="5", GPEMU_EXACT_GRAD="1", GPEMU_DEVICES="0")
    serial = tmp_path / "serial_one"
    run([cli, "estimate_thetas", str(f), str(serial), "--regression_order=0"], env=base)
    for world in (2, 4):
        rdv = tmp_path / f"rdv_one_{world}"
        rdv.mkdir()
        _run_ranks([cli, "estimate_thetas", str(f), "@SNAP@", "--regression_order=0"], world, rdv, base,
                   lambda r: tmp_path / f"one_w{world}_r{r}")
        assert (tmp_path / f"one_w{world}_r0").read_bytes() == serial.read_bytes()


@pytest.mark.gpu
def test_two_rccl_ranks_on_one_device_end_promptly_with_status_1(tmp_path):
    """the RCCL transport of ranks.c as far as a one-GPU box can take it: two CLI ranks (GPEMU_GATHER unset) meet in the
    start-up handshake, rank 0's ncclUniqueId travels in its `go` file, both call ncclCommInitRank -- which RCCL refuses for
    two ranks on ONE device.  What must hold: no rank hangs in the rendezvous or in the communicator; both end within seconds
    with status 1 and a message that names the call (fatal.c; a rank whose peer failed first is ended by the watchdog).
    Between two physical GPUs the same path has never run."""
    import time
    procs = []
    t0 = time.time()
    for r in range(2):
        env = dict(os.environ, GPEMU_RANK=str(r), GPEMU_WORLD_SIZE="2", GPEMU_RENDEZVOUS_DIR=str(tmp_path), GPEMU_DEVICES="0",
                   GPEMU_SEED="3", GPEMU_RESTARTS="2")
        env.pop("GPEMU_GATHER", None)
        procs.append(subprocess.Popen([build.CLI_BIN, "estimate_thetas", MULTI, str(tmp_path / f"snap{r}"), "--regression_order=1"], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    errs = []
    for p in procs:
        so, se = p.communicate(timeout=180)
        assert p.returncode == 1, (p.returncode, se[-1500:])
        errs.append(se)
    assert time.time() - t0 < 120
    assert any("ncclCommInitRank" in e for e in errs), errs
    for r, e in enumerate(errs):
        assert "ncclCommInitRank" in e or "not waiting for it" in e, (r, e[-1500:])


@pytest.mark.gpu
def test_rccl_allgather_entry(tmp_path):
    """gpemu_rccl_allgather (include/gpemu.h; csrc/hip/rccl_gather.hip): librccl opened at run time, communicator from
    an id file, ncclAllGather of doubles on a stream of its own.  One GPU on the box = a world of one rank (two RCCL ranks
    cannot share a device); the id file is gone afterwards; bad arguments come back as GPEMU_ERR_ARG with a message."""
    import ctypes
    import torch                           # the hard case on purpose: torch brings ITS OWN librccl and HIP runtime into the
    import torch.distributed               # process; the entry must still open the librccl beside the libamdhip64 it links
    assert torch.zeros(1).item() == 0.0
    from madaiemulator_amd import abi
    L = abi.load()
    L.gpemu_rccl_allgather.restype = ctypes.c_int
    L.gpemu_rccl_allgather.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.POINTER(ctypes.c_double),
                                       ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.c_char_p, ctypes.c_size_t]
    n = 11
    send = (ctypes.c_double * n)(*[1.5 * i - 3.0 for i in range(n)])
    recv = (ctypes.c_double * n)()
    err = ctypes.create_string_buffer(256)
    idf = tmp_path / "id0"
    rc = L.gpemu_rccl_allgather(0, 0, 1, str(idf).encode(), send, n, recv, err, 256)
    assert rc == 0, err.value
    assert list(recv) == list(send) and not idf.exists()
    rc = L.gpemu_rccl_allgather(0, 2, 2, str(idf).encode(), send, n, recv, err, 256)
    assert rc != 0 and err.value


def _write_model_file(path, X, y):
    N, d = X.shape
    path.write_text(f"1\n{d}\n{N}\n" + "\n".join(" ".join(repr(float(v)) for v in row) for row in X) + "\n" +
                    "\n".join(repr(float(v)) for v in y) + "\n")


@pytest.mark.gpu
@pytest.mark.parametrize("cov", [1, 3])
def test_training_with_the_corrected_gradient_converges(driver, tmp_path, cov):
    """SURVEY App. C2-C4 flags: with GPEMU_EXACT_GRAD=1 (and GPEMU_MATERN_FIXED=1 for the Matern 5/2 model, which the
    reference cannot train at all) estimate_thetas_threaded trains N=1024, d=8: the winning BFGS run ends at
    |gradient| < 0.1 (maxmultimin.c:725), no line search needs the 'lowest trial value' fallback the literal
    gradient forces, and the gradient at the optimum -- the exact one -- is indeed small."""
    N, d = 1024, 8
    X, y = synth.design(N, d, 4242)
    y = y + 0.2 * synth.normal(17, N)
    f = tmp_path / "train.dat"
    _write_model_file(f, X, y)
    env = dict(os.environ, GPEMU_SEED="99", GPEMU_JOBS="16", GPEMU_RESTARTS="1", GPEMU_EXACT_GRAD="1", GPEMU_DEVICES="0")
    if cov != 1:
        env["GPEMU_MATERN_FIXED"] = "1"
    res = parse(run([driver, "train", str(f), str(cov), "0"], env=env))
    th = np.array(res["thetas"][0])
    runs, conv, noprog, fallbacks, best_gnorm = res["search"][0]
    assert runs == 16 and np.all(np.isfinite(th)) and np.isfinite(res["neglogl"][0][0])
    assert fallbacks == 0
    assert 0.0 <= best_gnorm < 0.1
    assert np.sqrt(np.sum(np.square(res["grad_at_best"][0]))) < 0.1
    if cov == 1:
        # the value at the optimum against the oracle (pow-exp: the literal kernel is unchanged by the flags)
        assert res["neglogl"][0][0] == pytest.approx(O.eval_fn_multi(1, 0, X, y, th[1:])["value"], rel=1e-6)


@pytest.mark.gpu
def test_training_at_n4096_where_the_reference_argmax_floor_would_fail(driver, tmp_path):
    """The reference starts its arg-max over restarts at -2000 (maxmultimin.c:38,60,110): at N = 4096 every
    log-likelihood of a noisy model is below that and the reference ends with 'maximisation didn't work at all' and
    thetas = 0.  Here the arg-max starts at -infinity: the search returns its best run, which -- with
    the exact gradient -- has converged, and the value at the returned thetas is what evalFnMulti says it is."""
    N, d = 4096, 8
    X, y = synth.design(N, d, 777)
    y = y + 0.8 * synth.normal(3, N)
    f = tmp_path / "train4096.dat"
    _write_model_file(f, X, y)
    env = dict(os.environ, GPEMU_SEED="5", GPEMU_JOBS="4", GPEMU_RESTARTS="1", GPEMU_EXACT_GRAD="1", GPEMU_DEVICES="0")
    res = parse(run([driver, "train", str(f), "1", "0"], env=env))
    th = np.array(res["thetas"][0])
    val = res["neglogl"][0][0]
    runs, conv, noprog, fallbacks, best_gnorm = res["search"][0]
    assert runs == 4 and np.all(np.isfinite(th)) and np.any(th != 0.0)
    assert val > 2000.0                      # log-likelihood below the reference's floor of -2000
    assert conv >= 1 and fallbacks == 0 and 0.0 <= best_gnorm < 0.1


@pytest.mark.gpu
def test_model_buffers_rewritten_in_place_are_seen(driver, tmp_path):
    """The reference re-reads the model on every call (maxmultimin.c:317: the matrix is filled from the_model->xmodel
    each time), so a caller may rewrite training_vector->data or xmodel->data IN PLACE -- same pointers -- between calls
    (libRbind-style loops, an MCMC re-fit).  The device copy is keyed by a checksum of every value, not by pointer
    identity: each step must return the likelihood of the data as they are at that call, and the per-caller value cache
    must not survive a change of the data (tests/c/host_api_driver.c `rewrite` does the rewriting)."""
    N, d = 200, 3
    X, y = synth.design(N, d, 606)
    f = tmp_path / "rw.dat"
    _write_model_file(f, X, y)
    th = synth.default_thetas(1, d)[1:]
    res = parse(run([driver, "rewrite", str(f), "1", "1"] + [repr(float(v)) for v in th], env=dict(os.environ, GPEMU_DEVICES="0")))
    y1 = 0.5 * y + 0.01 * (np.arange(N) % 7)
    X2 = 0.9 * X + 0.003 * ((np.arange(N)[:, None] + 3 * np.arange(d)[None, :]) % 11)
    for key, (Xs, ys) in (("step0", (X, y)), ("step1", (X, y1)), ("step2", (X2, y1))):
        ref = O.eval_fn_multi(1, 1, Xs, ys, th)
        assert res[key][0][0] == pytest.approx(ref["value"], rel=RTOL), key
        assert res[key][0][1] == pytest.approx(ref["sigma2"], rel=RTOL), key
    assert res["step2b"][0][0] == res["step2"][0][0]
    # three data states -> three device evaluations (two value-only, one value+gradient); the other four requests -- the
    # sigma^2 of each state and the repeated value of the last -- were answered from the caller's cache of its last results
    assert res["evalstats"][0] == [2.0, 1.0, 4.0]


@pytest.mark.gpu
def test_search_result_does_not_depend_on_the_device_slots(driver, tmp_path):
    """the run list of estimate_thetas_threaded (how many BFGS runs, where each starts) is derived from GPEMU_JOBS x
    GPEMU_RESTARTS and the seed alone; the runs are then dealt to however many threads, lock-step groups and device slots
    there are, and the arg-max breaks ties by run index: one slot, three slots, one or two groups per slot, groups of 16
    or of 5 and the one-context-per-thread scheme all return the same thetas, bit for bit.  (An element of a lock-step
    batch equals the same evaluation done alone, so a run's trajectory does not depend on its batch mates.)  The end-of-run
    evalFnMulti / estimateSigmaFull calls never reach the device: they are answered from the member's cache."""
    N, d = 300, 3
    X, y = synth.design(N, d, 2718)
    y = y + 0.2 * synth.normal(5, N)
    f = tmp_path / "det.dat"
    _write_model_file(f, X, y)
    outs = {}
    for name, extra in (("one_slot", dict(GPEMU_DEVICES="0")), ("three_slots", dict(GPEMU_DEVICES="0,0,0")),
                        ("one_group", dict(GPEMU_DEVICES="0", GPEMU_GROUPS_PER_SLOT="1")),
                        ("groups_of_5", dict(GPEMU_DEVICES="0", GPEMU_LOCKSTEP="5")),
                        ("threads", dict(GPEMU_DEVICES="0", GPEMU_LOCKSTEP="1", GPEMU_NTHREADS="3"))):
        env = dict(os.environ, GPEMU_SEED="31", GPEMU_JOBS="3", GPEMU_RESTARTS="8", GPEMU_EXACT_GRAD="1", **extra)
        res = parse(run([driver, "train", str(f), "1", "0"], env=env))
        outs[name] = (res["thetas"][0], res["neglogl"][0][0])
        runs = res["search"][0][0]
        nv, nvg, ncached = res["evalstats"][0][:3]
        assert runs == 24
        # per run: estimateSigmaFull + evalFnMulti at the final point, both from the cache; the driver's own evalFnMulti of
        # the winning thetas afterwards is the one value-only evaluation on the device
        assert ncached + (nv - 1) == 48 and nv <= 3 and nvg > 24, res["evalstats"]
    for name in outs:
        assert outs[name] == outs["one_slot"], (name, outs)
    assert outs["one_slot"][1] == pytest.approx(O.eval_fn_multi(1, 0, X, y, np.array(outs["one_slot"][0][1:]))["value"], rel=1e-6)


@pytest.mark.gpu
def test_training_at_n8192_through_the_lockstep_groups(driver, tmp_path):
    """estimate_thetas at the size the headline metric is quoted on (N=8192, d=8, pow-exp): 32 BFGS runs as two lock-step
    groups of 16 on the one GPU (value+gradient batches of 16 through gpemu_loglik_grad_batch_enqueue / _collect_back), a
    bounded number of iterations' worth of device time (a minute).  The value at the returned thetas is what gpemu_loglik
    says there (the driver's evalFnMulti), the winning run converged, and the search's own counters show full batches."""
    N, d = 8192, 8
    X, y = synth.design(N, d, 20261003 + 3)
    y = y + 0.3 * synth.normal(11, N)
    f = tmp_path / "train8192.dat"
    with open(f, "w") as fh:
        fh.write(f"1\n{d}\n{N}\n")
        np.savetxt(fh, X, fmt="%.17g")
        np.savetxt(fh, y, fmt="%.17g")
    env = dict(os.environ, GPEMU_SEED="7", GPEMU_JOBS="1", GPEMU_RESTARTS="32", GPEMU_EXACT_GRAD="1", GPEMU_DEVICES="0",
               GPEMU_SEARCH_STATS="1")
    out = subprocess.run([driver, "train", str(f), "1", "0"], env=env, capture_output=True, text=True, timeout=220)
    assert out.returncode == 0, out.stderr[-2000:]
    res = parse(out.stdout)
    th = np.array(res["thetas"][0])
    runs, conv, noprog, fallbacks, best_gnorm = res["search"][0]
    nv, nvg, ncached, rounds, elems = res["evalstats"][0]
    assert runs == 32 and np.all(np.isfinite(th)) and np.isfinite(res["neglogl"][0][0])
    assert "# search stats: runs 32 threads 32 groups 2 slots 1" in out.stderr
    assert conv >= 1 and 0.0 <= best_gnorm < 0.1
    assert ncached + (nv - 1) == 64 and nv <= 3
    assert elems / rounds > 8.0                      # mean requests per device round (16 while every run is alive)
    # the returned thetas against an independent evaluation of the same likelihood on a fresh context
    from madaiemulator_amd import abi
    c = abi.Context(0)
    c.set_model(1, 0, X, y)
    full = th.copy()
    full[0] = 0.0
    assert c.loglik(full)["value"] == res["neglogl"][0][0]
    g = c.loglik_grad(full)
    c.close()
    assert g["value"] == res["neglogl"][0][0]


@pytest.mark.gpu
def test_call_eval_lhood_list_without_r():
    """libRbind's batched likelihood entry (rbind.c:626-724): flat .C() signature, column-major arrays"""
    import ctypes as C
    build.build_all()
    C.CDLL(build.HIP_LIB, mode=C.RTLD_GLOBAL)
    lib = C.CDLL(build.HOST_LIB)
    X, Y = synth.read_input_model_file(TWOD)
    y = Y[:, 0]
    N, d = X.shape
    nthetas = d + 2
    rows = np.array([[-3.0, -1.0, -0.7, 0.0], [-4.0, -0.5, -0.9, 0.0], [-2.5, -1.5, -0.2, 0.0],
                     [-3.5, -0.8, -0.4, 0.0], [-2.0, -1.2, -1.1, 0.0]])   # {nug, lengths..., unused}
    ans = np.zeros(len(rows))
    os.environ["GPEMU_HOST_BATCH"] = "2"       # 5 rows -> lock-step batches of 2, 2, 1 (evalFnMultiList)
    dp = C.POINTER(C.c_double)
    ip = lambda v: C.byref(C.c_int(v))
    xin = np.asfortranarray(X).ravel(order="F").copy()
    pin = np.asfortranarray(rows).ravel(order="F").copy()
    lib.callEvalLhoodList(xin.ctypes.data_as(dp), ip(d), pin.ctypes.data_as(dp), ip(len(rows)), y.ctypes.data_as(dp),
                          ip(N), ip(nthetas), ans.ctypes.data_as(dp), ip(1), ip(1))
    os.environ.pop("GPEMU_HOST_BATCH")
    for r, a in zip(rows, ans):
        assert a == pytest.approx(O.eval_fn_multi(1, 1, X, y, r[:nthetas - 1])["value"], rel=RTOL)
    # callEmulateAtList / callEmulateAtPt (rbind.c:121,214): mean and variance at given thetas, points column-major
    th = np.array([-0.2, -3.5, -1.0, -0.7])
    Q = synth.queries(9, d, 4)
    qin = np.asfortranarray(Q).ravel(order="F").copy()
    m, v = np.zeros(len(Q)), np.zeros(len(Q))
    lib.callEmulateAtList(xin.ctypes.data_as(dp), ip(d), qin.ctypes.data_as(dp), ip(len(Q)), y.ctypes.data_as(dp), ip(N),
                          th.ctypes.data_as(dp), ip(nthetas), m.ctypes.data_as(dp), v.ctypes.data_as(dp), ip(1), ip(1))
    e = O.Emulator(1, 1, X, y, th)
    mo, vo, _ = e.emulate(Q)
    kappa = O.cov(1, Q[0], Q[0], th)
    assert np.max(np.abs(m - mo)) <= RTOL * max(1.0, np.abs(mo).max()) and np.max(np.abs(v - vo)) <= RTOL * kappa
    m1, v1 = np.zeros(1), np.zeros(1)
    q0 = Q[0].copy()
    lib.callEmulateAtPt(xin.ctypes.data_as(dp), ip(d), q0.ctypes.data_as(dp), y.ctypes.data_as(dp), ip(N),
                        th.ctypes.data_as(dp), ip(nthetas), m1.ctypes.data_as(dp), v1.ctypes.data_as(dp), ip(1), ip(1))
    assert m1[0] == pytest.approx(mo[0], rel=1e-8, abs=1e-10) and v1[0] == pytest.approx(vo[0], abs=RTOL * kappa)
    # callEstimate (rbind.c:35): trained thetas come back finite and beat the start of the search box
    os.environ.update(GPEMU_SEED="7", GPEMU_RESTARTS="4")
    final = np.zeros(nthetas)
    lib.callEstimate(xin.ctypes.data_as(dp), ip(d), y.ctypes.data_as(dp), ip(N), ip(nthetas), final.ctypes.data_as(dp),
                     ip(0), C.byref(C.c_double(0.0)), ip(1), ip(1))
    assert np.all(np.isfinite(final))
    best = O.eval_fn_multi(1, 1, X, y, final[1:])["value"]
    assert np.isfinite(best) and best < O.eval_fn_multi(1, 1, X, y, np.array([-3.0, -1.0, -0.7]))["value"]
    # setupEmulateMC / callEmulateMC / freeEmulateMC (rbind.c:299-460): one emulator kept between calls
    lib.setupEmulateMC(xin.ctypes.data_as(dp), ip(d), y.ctypes.data_as(dp), ip(N), th.ctypes.data_as(dp), ip(nthetas), ip(1), ip(1))
    for q in range(3):
        qq = Q[q].copy()
        lib.callEmulateMC(qq.ctypes.data_as(dp), m1.ctypes.data_as(dp), v1.ctypes.data_as(dp))
        assert m1[0] == pytest.approx(mo[q], rel=1e-8, abs=1e-10) and v1[0] == pytest.approx(vo[q], abs=RTOL * kappa)
    lib.freeEmulateMC()
    lib.freeEmulateMC()                                   # idempotent
    # ...Multi (rbind.c:483-600): nydims outputs of one design, one theta row each; arrays flattened column by column
    Ym = np.column_stack([y, -2.0 * y + X[:, 0], np.cos(3.0 * y)])
    ths = np.array([th, th + [0.1, 0.2, -0.1, 0.05], th + [-0.3, 0.5, 0.2, -0.2]])
    nyd = Ym.shape[1]
    lib.setupEmulateMCMulti(xin.ctypes.data_as(dp), ip(d), np.asfortranarray(Ym).ravel(order="F").copy().ctypes.data_as(dp),
                            ip(nyd), ip(N), np.asfortranarray(ths).ravel(order="F").copy().ctypes.data_as(dp), ip(nthetas),
                            ip(1), ip(1))
    mm, vv = np.zeros(nyd), np.zeros(nyd)
    for q in (0, 5):
        qq = Q[q].copy()
        lib.callEmulateMCMulti(qq.ctypes.data_as(dp), ip(nyd), mm.ctypes.data_as(dp), vv.ctypes.data_as(dp))
        for c in range(nyd):
            ec = O.Emulator(1, 1, X, Ym[:, c].copy(), ths[c])
            mc, vc, _ = ec.emulate(Q[q:q + 1])
            assert mm[c] == pytest.approx(mc[0], rel=1e-8, abs=1e-10)
            assert vv[c] == pytest.approx(vc[0], abs=RTOL * O.cov(1, Q[q], Q[q], ths[c]))
    lib.freeEmulateMCMulti(ip(nyd))


@pytest.mark.gpu
def test_emuplusplus_class(tmp_path):
    """the C++ query class: QueryEmulator returns means and sqrt(variance) (EmuPlusPlus.cpp:137-178)"""
    build.build_all()
    exe = str(tmp_path / "emupp_driver")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-I", os.path.join(ROOT, "include"), "-I", build.HOST_SRC, "-o", exe,
                           os.path.join(ROOT, "tests", "c", "emupp_driver.cpp"), "-L", build.LIBDIR, "-lEmuPlusPlusMI",
                           "-lEmuMI", "-lgpemu_hip", f"-Wl,-rpath,{build.LIBDIR}", "-lm"])
    snap = tmp_path / "snap"
    env = dict(os.environ, GPEMU_SEED="99", GPEMU_RESTARTS="2")
    run([build.CLI_BIN, "estimate_thetas", MULTI, str(snap), "--regression_order=0"], env=env)
    X, Y = synth.read_input_model_file(MULTI)
    qf = tmp_path / "q.dat"
    np.savetxt(qf, np.vstack([X[:4], synth.queries(6, 3, 1) * 0.1]), fmt="%.17g")
    res = parse(run([exe, str(snap), str(qf)]))
    cli = np.array(run([build.CLI_BIN, "interactive_mode", str(snap), "-q"], stdin=open(qf)).split(), float).reshape(10, -1, 2)
    single = np.array(res["single"]).reshape(10, -1, 2)
    batch = np.array(res["batch"]).reshape(10, -1, 2)
    assert res["info"][0][:2] == [3.0, 6.0]
    assert np.array_equal(single, batch, equal_nan=True)
    assert np.allclose(single[:, :, 0], cli[:, :, 0], rtol=1e-12, atol=1e-12)
    # Errors = sqrt(variance): at a training point the variance is 0 up to rounding, sqrt of a tiny negative is NaN
    # exactly as in the reference (EmuPlusPlus.cpp:176)
    ok = cli[:, :, 1] > 0
    assert ok.sum() > 30
    assert np.allclose(single[:, :, 1][ok] ** 2, cli[:, :, 1][ok], rtol=1e-9, atol=1e-14)
    rest = single[:, :, 1][~ok]                      # the CLI prints %.17f: variances below 5e-18 read as 0
    assert np.all(np.isnan(rest) | (np.nan_to_num(rest) < 1e-6))
    pca = parse(run([exe, str(snap), str(qf), "pca"]))
    assert int(pca["info"][0][1]) == int(open(snap).read().split()[1])          # number_outputs == nr in PCA mode


def _oracle_through_snapshot(snap_text, Xq):
    """oracle emulate_point per PCA component at the thetas of a MODEL_SNAPSHOT_FILE, back-projected as the reference does"""
    sd = parse_snapshot(snap_text.split())
    nr = sd["nr"]
    m_pca, v_pca = np.empty((len(Xq), nr)), np.empty((len(Xq), nr))
    for c, comp in enumerate(sd["models"]):
        m_pca[:, c], v_pca[:, c], _ = O.Emulator(comp["cov"], comp["order"], comp["X"], comp["z"], comp["thetas"]).emulate(Xq)
    ybar = sd["Y"].mean(axis=0)
    out = [O.pca_backproject(ybar, sd["evals"], sd["evecs"], m_pca[q], v_pca[q]) for q in range(len(Xq))]
    return sd, np.array([o[0] for o in out]), np.array([o[1] for o in out])


@pytest.mark.gpu
def test_reference_example_scripts_uni_2d_param_and_multi_simple(tmp_path):
    """the two other example directories of the reference, with the command lines of their scripts:
    test/uni-2d-param (train-emulator.sh: `estimate_thetas Latin_square_sampling_2d_samp_fn_200.dat M.dat
    --regression_order=1`; sample-emulator.sh: `interactive_mode M.dat < sample_locations.dat`, nskip = 4 + nparams header
    lines, then mean / variance alternating) and test/multi-simple (train-emulator.sh: `--regression_order=0`;
    sample-emu.sh: the two points "0.03 0.04 0.01" and "0.05 0.02 0.01").  Every printed value against the oracle at the
    thetas the snapshot holds."""
    cli = build.CLI_BIN
    env = dict(os.environ, GPEMU_SEED="2718", GPEMU_RESTARTS="3")
    # uni-2d-param.  Its training values are a smooth function WITHOUT noise: the likelihood grows as the nugget -> 0 and the
    # search (unbounded, as the reference's gsl bfgs2) ends at a nugget of e^-33 whatever the seed -- a covariance matrix of
    # condition 1e16 that LAPACK refuses too.  The product says so when it writes the snapshot, and interactive_mode then ends
    # with the reference's own message (emulate-fns.c:282-285); what is checked is that the two agree with an independent
    # judgement of the matrix.
    snap = tmp_path / "M.dat"
    tr = subprocess.run([cli, "estimate_thetas", TWOD, str(snap), "--regression_order=1"], env=env, capture_output=True, text=True, timeout=600)
    assert tr.returncode == 0
    qpath = os.path.join(INP, "uni-2d-param.sample_locations.dat")
    Q = np.array(open(qpath).read().split(), float).reshape(-1, 2)
    sd = parse_snapshot(snap.read_text().split())
    assert (sd["nt"], sd["nr"], sd["d"], sd["N"]) == (1, 1, 2, 200) and sd["models"][0]["order"] == 1
    import gradref
    Cm, _ = gradref.powexp_matrix(sd["models"][0]["X"], sd["models"][0]["thetas"])
    try:
        sl.cho_factor(Cm, lower=True)
        lapack_ok = True
    except np.linalg.LinAlgError:
        lapack_ok = False
    im = subprocess.run([cli, "interactive_mode", str(snap)], stdin=open(qpath), capture_output=True, text=True, timeout=600)
    # ONE outcome, asserted (round-4 verdict: the test accepted either): with this seed and run list the search is deterministic
    # -- run r starts from (seed, r), the device arithmetic is bit-reproducible -- and ends, as for every seed tried
    # (scratch/r04_uni2d_seeds.sh), at a nugget below e^-25 where the amplitude-scaled matrix no longer factors: the CLI warns at
    # training time, interactive_mode refuses the snapshot with the reference's message and status 1, LAPACK agrees
    assert "numerically singular" in tr.stderr and "GPEMU_NUGGET_FLOOR" in tr.stderr
    assert im.returncode == 1, (im.returncode, im.stderr[-500:])
    assert "trying to cholesky a non postive def matrix, in emulate-fns.c sorry..." in im.stderr
    assert not lapack_ok and sd["models"][0]["thetas"][1] < -25.0
    # the same example with the lower wall the warning names (GPEMU_NUGGET_FLOOR, not in the reference): the search stops at
    # the wall, the snapshot is usable, and sample-emulator.sh's output agrees with the oracle at the snapshot's thetas
    snap2 = tmp_path / "M_floor.dat"
    tr2 = subprocess.run([cli, "estimate_thetas", TWOD, str(snap2), "--regression_order=1"], env=dict(env, GPEMU_NUGGET_FLOOR="-12"),
                         capture_output=True, text=True, timeout=600)
    assert tr2.returncode == 0 and "numerically singular" not in tr2.stderr
    th2 = parse_snapshot(snap2.read_text().split())["models"][0]["thetas"]
    assert -12.0 <= th2[1] < -5.0                                   # walked down from the start range [-5, -2] to the wall
    lines = run([cli, "interactive_mode", str(snap2)], stdin=open(qpath)).split()
    assert lines[:6] == ["2", "param_0", "param_1", "2", "mean_0", "variance_0"]              # nskip = 6 = 4 + nparams
    vals = np.array(lines[6:], float).reshape(-1, 2)
    assert len(vals) == len(Q) == 1024
    _, mean, var = _oracle_through_snapshot(snap2.read_text(), Q)
    assert np.max(np.abs(vals[:, 0] - mean[:, 0])) < 1e-7 * max(1.0, np.abs(mean).max())
    assert np.max(np.abs(vals[:, 1] - var[:, 0])) < 1e-7 * max(1e-3, np.abs(var).max())
    # multi-simple
    snap = tmp_path / "multi_snapshot_file"
    run([cli, "estimate_thetas", MULTI, str(snap), "--regression_order=0"], env=env)
    lines = run([cli, "interactive_mode", str(snap)], input="0.03 0.04 0.01\n0.05 0.02 0.01\n").split()
    sd, mean, var = _oracle_through_snapshot(snap.read_text(), np.array([[0.03, 0.04, 0.01], [0.05, 0.02, 0.01]]))
    nt = sd["nt"]
    header = 1 + 3 + 1 + 2 * nt
    assert lines[0] == "3" and lines[4] == str(2 * nt) and lines[5] == "mean_0" and lines[header - 1] == f"variance_{nt - 1}"
    vals = np.array(lines[header:], float).reshape(2, nt, 2)
    assert np.max(np.abs(vals[:, :, 0] - mean)) < 1e-7 * max(1.0, np.abs(mean).max())
    assert np.max(np.abs(vals[:, :, 1] - var)) < 1e-7 * max(1e-3, np.abs(var).max())


@pytest.mark.gpu
def test_cli_reads_the_input_model_file_from_stdin(tmp_path):
    """INPUT_MODEL_FILE "-" (interactive_emulator.c:212-251 of the reference, "can be - to read from standard input"): the
    same snapshot, byte for byte, as with the file name"""
    cli = build.CLI_BIN
    env = dict(os.environ, GPEMU_SEED="8", GPEMU_RESTARTS="2")
    a, b = tmp_path / "from_file", tmp_path / "from_stdin"
    run([cli, "estimate_thetas", UNI, str(a), "--regression_order=1"], env=env)
    run([cli, "estimate_thetas", "-", str(b), "--regression_order=1"], env=env, stdin=open(UNI))
    assert a.read_bytes() == b.read_bytes() and len(a.read_bytes()) > 500


@pytest.mark.gpu
def test_emuplusplus_reference_example_flow(tmp_path):
    """the reference's own test/emuplusplus-test, step for step: train-emu.sh (`interactive_emulator estimate_thetas
    input_model_file.dat univariate_snapshot_file --regression_order=1` on the uni-simple data, emu-dir/train-emulator.sh)
    then sample-emu.sh (src/example.cpp: `emulator my_emu(filename)`, QueryEmulator for every location of
    sample_locations.dat).  Means and errors = sqrt(variance) against the oracle's emulate_point at the thetas the search
    wrote into the snapshot; the class's answers equal interactive_mode's on the same snapshot."""
    build.build_all()
    exe = str(tmp_path / "emupp_driver")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-I", os.path.join(ROOT, "include"), "-I", build.HOST_SRC, "-o", exe,
                           os.path.join(ROOT, "tests", "c", "emupp_driver.cpp"), "-L", build.LIBDIR, "-lEmuPlusPlusMI",
                           "-lEmuMI", "-lgpemu_hip", f"-Wl,-rpath,{build.LIBDIR}", "-lm"])
    snap = tmp_path / "univariate_snapshot_file"
    run([build.CLI_BIN, "estimate_thetas", UNI, str(snap), "--regression_order=1"], env=dict(os.environ, GPEMU_SEED="4", GPEMU_RESTARTS="3"))
    qpath = os.path.join(INP, "emuplusplus-test.sample_locations.dat")
    Xq = np.loadtxt(qpath).reshape(-1, 1)
    assert len(Xq) == 16
    res = parse(run([exe, str(snap), qpath]))
    assert res["info"][0] == [1.0, 1.0, 1.0, 1.0]                 # one parameter, one output, regression order 1, pow-exp
    got = np.array(res["single"]).reshape(16, 2)
    assert np.array_equal(got, np.array(res["batch"]).reshape(16, 2), equal_nan=True)
    snapshot = parse_snapshot(open(snap).read().split())
    mdl = snapshot["models"][0]
    X, Y = synth.read_input_model_file(UNI)
    # oracle prediction in PCA space, then the reference's back-projection (nt = nr = 1, multivar_support.c:103-157)
    m, v, _ = O.Emulator(1, 1, X, mdl["z"], mdl["thetas"]).emulate(Xq)
    lam, u = float(snapshot["evals"][0]), float(snapshot["evecs"][0, 0])
    mean = Y[:, 0].mean() + u * np.sqrt(lam) * m
    var = u * u * lam * v
    assert np.max(np.abs(got[:, 0] - mean)) < 1e-7 * max(1.0, np.abs(mean).max())
    ok = var > 1e-9
    assert ok.sum() >= 10 and np.allclose(got[ok, 1] ** 2, var[ok], rtol=1e-6)
    cli = np.array(run([build.CLI_BIN, "interactive_mode", str(snap), "-q"], stdin=open(qpath)).split(), float).reshape(16, 2)
    assert np.allclose(cli[:, 0], got[:, 0], rtol=1e-12, atol=1e-12)


def parse_snapshot(toks):
    """MODEL_SNAPSHOT_FILE grammar"""
    it = iter(toks)
    nxt = lambda n, typ=float: np.array([typ(next(it)) for _ in range(n)])
    nt, nr, d, N, cov, order = (int(next(it)) for _ in range(6))
    out = dict(nt=nt, nr=nr, d=d, N=N, X=nxt(N * d).reshape(N, d), Y=nxt(N * nt).reshape(N, nt), evals=nxt(nr),
               evecs=nxt(nt * nr).reshape(nt, nr), Z=nxt(N * nr).reshape(N, nr), models=[])
    for _ in range(nr):
        nthetas, dd, NN, _nemu, order_c, nreg, _fnm = (int(next(it)) for _ in range(7))
        _fn = float(next(it))
        cov_c, _uds = int(next(it)), int(next(it))
        ranges = nxt(2 * nthetas).reshape(nthetas, 2)
        Xc = nxt(NN * dd).reshape(NN, dd)
        z = nxt(NN)
        th = nxt(nthetas)
        scales = nxt(dd)
        out["models"].append(dict(cov=cov_c, order=order_c, X=Xc, z=z, thetas=th, ranges=ranges, scales=scales))
    return out
