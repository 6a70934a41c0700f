"""interactive_mode's I/O pipeline (csrc/host/interactive_io.c) without a GPU: the reference's text framing
(src/interactive_emulator.c:420 fscanf("%lf%*c"), :434-435 "%.17f\\n") and its BINARY_INTERACTIVE_MODE framing (:418,
:431-432 raw doubles), batching of waiting points, immediate answers to a lone point, input order kept.  The device stage
is a stand-in (tests/c/io_loop_driver.c); the GPU tests drive the real CLI (test_host_api.py)."""
import os
import select
import subprocess
import time

import numpy as np
import pytest

from madaiemulator_amd import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def drv(tmp_path_factory):
    build.build_all()
    exe = str(tmp_path_factory.mktemp("io") / "io_loop_driver")
    subprocess.check_call(["gcc", "-std=gnu99", "-O1", "-I", os.path.join(ROOT, "include"), "-I", build.HOST_SRC,
                           "-o", exe, os.path.join(ROOT, "tests", "c", "io_loop_driver.c"),
                           "-L", build.LIBDIR, "-lEmuMI", "-lgpemu_hip", f"-Wl,-rpath,{build.LIBDIR}", "-lm", "-lpthread"])
    return exe


def expected(X, nout, nprint):
    s = X.sum(axis=1)
    mean = np.zeros((len(X), nprint))
    var = np.zeros((len(X), nprint))
    with np.errstate(over="ignore"):
        for i in range(nout):
            mean[:, i] = (i + 1) * s
            var[:, i] = X[:, 0] * X[:, 0] + i
    return mean, var


def as_text(mean, var):
    out = []
    for q in range(len(mean)):
        for i in range(mean.shape[1]):
            out.append("%.17f\n%.17f\n" % (mean[q, i], var[q, i]))
    return "".join(out)


@pytest.mark.parametrize("threads", ["0", "2"])
def test_text_stream_is_answered_in_order_byte_for_byte(drv, threads):
    """60 000 points of d = 3 on stdin at once (several 16 384-point batches, the helper threads awake): every answer in
    input order, formatted as glibc's "%.17f" -- with the conversions on one thread and dealt to helpers alike"""
    rng = np.random.default_rng(5)
    X = rng.normal(size=(60000, 3)) * 10.0 ** rng.integers(-3, 4, size=(60000, 1))
    X[17] = [1e300, -2.5e-300, 0.0]                             # 300-digit "%.17f" fields
    text = "\n".join(" ".join(repr(float(v)) for v in row) for row in X) + "\n"
    p = subprocess.run([drv, "3", "2", "2", "0"], input=text.encode(), capture_output=True, timeout=120,
                       env=dict(os.environ, GPEMU_IO_THREADS=threads))
    assert p.returncode == 0, p.stderr[-500:]
    mean, var = expected(X, 2, 2)
    assert p.stdout.decode() == as_text(mean, var)
    pts, batches, maxb = (int(v) for v in p.stderr.decode().split()[1:4])
    # (at least the four full batches; a few more when the pipe ran dry for a moment while the stand-in device was idle)
    assert pts == 60000 and maxb == 16384 and 4 <= batches <= 200


def test_separators_partial_point_and_a_non_number_end_the_input_like_fscanf(drv):
    """separators the loop accepts (blank, tab, CR, LF, comma, semicolon); a trailing partial point is dropped
    (r < expected_r, interactive_emulator.c:423); a token that is no number ends the input there"""
    p = subprocess.run([drv, "2", "1", "1", "0"], input=b"1,2;3\t4\r\n5 6 7", capture_output=True, timeout=30)
    X = np.array([[1, 2], [3, 4], [5, 6]], float)
    assert p.stdout.decode() == as_text(*expected(X, 1, 1))
    p = subprocess.run([drv, "2", "1", "1", "0"], input=b"1 2 3 4 x 5 6 7 8\n", capture_output=True, timeout=30)
    assert p.stdout.decode() == as_text(*expected(X[:2], 1, 1))
    p = subprocess.run([drv, "2", "1", "1", "0"], input=b"", capture_output=True, timeout=30)
    assert p.returncode == 0 and p.stdout == b""


def test_pca_space_output_keeps_nt_pairs(drv):
    """nout < nprint (--pca_output with nr < nt): the pairs beyond nr are zeros, nt pairs per point as the reference prints"""
    X = np.array([[0.5, 1.5], [2.0, -1.0]])
    p = subprocess.run([drv, "2", "2", "3", "0"], input=b"0.5 1.5\n2.0 -1.0\n", capture_output=True, timeout=30)
    assert p.stdout.decode() == as_text(*expected(X, 2, 3))


def _ask(p, payload, nbytes=None, nlines=None, timeout=10.0):
    p.stdin.write(payload)
    p.stdin.flush()
    got = b""
    t0 = time.time()
    while (nbytes is not None and len(got) < nbytes) or (nlines is not None and got.count(b"\n") < nlines):
        assert time.time() - t0 < timeout, "no answer to a lone point: the loop is waiting for more input"
        r, _, _ = select.select([p.stdout], [], [], 0.5)
        if r:
            chunk = os.read(p.stdout.fileno(), 65536)
            assert chunk
            got += chunk
    return got


@pytest.mark.parametrize("delay_us", ["0", "20000"])
def test_a_lone_point_is_answered_at_once(drv, delay_us):
    """the MCMC pattern (one point, wait for its answer, next point): every point comes back before the next is sent,
    with an idle and with a slow device stage"""
    p = subprocess.Popen([drv, "2", "1", "1", "0", delay_us], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    rng = np.random.default_rng(1)
    for i in range(12):
        x = rng.normal(size=(1, 2))
        got = _ask(p, ("%r %r\n" % (float(x[0, 0]), float(x[0, 1]))).encode(), nlines=2)
        assert got.decode() == as_text(*expected(x, 1, 1))
    p.stdin.close()
    assert p.wait(timeout=10) == 0
    assert p.stderr.read().decode().split()[1:3] == ["12", "12"]       # twelve points, twelve batches


def test_points_arriving_while_the_device_is_busy_are_batched(drv):
    """a slow device stage (50 ms per batch) and a producer that writes 400 points one line at a time: the waiting
    points ride in a few batches instead of 400 device calls, in order"""
    p = subprocess.Popen([drv, "1", "1", "1", "0", "50000"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    X = np.arange(400, dtype=float).reshape(-1, 1) / 7.0
    for v in X[:, 0]:
        p.stdin.write(("%r\n" % float(v)).encode())
        p.stdin.flush()
    out, err = p.communicate(timeout=60)
    assert out.decode() == as_text(*expected(X, 1, 1))
    pts, batches = (int(v) for v in err.decode().split()[1:3])
    assert pts == 400 and batches < 100


def test_binary_framing_raw_doubles_in_and_out(drv):
    """the reference's BINARY_INTERACTIVE_MODE (interactive_emulator.c:392-396,418-438): d raw doubles per point in,
    nt x (mean, variance) raw doubles out, bit for bit; a trailing partial point is dropped; a lone point is answered"""
    rng = np.random.default_rng(9)
    X = rng.normal(size=(40000, 4))
    p = subprocess.run([drv, "4", "3", "3", "1"], input=X.tobytes() + b"\x00" * 11, capture_output=True, timeout=120)
    assert p.returncode == 0
    got = np.frombuffer(p.stdout, dtype=np.float64).reshape(40000, 3, 2)
    mean, var = expected(X, 3, 3)
    assert np.array_equal(got[:, :, 0], mean) and np.array_equal(got[:, :, 1], var)
    q = subprocess.Popen([drv, "4", "3", "3", "1"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    for i in range(5):
        got = _ask(q, X[i].tobytes(), nbytes=48)
        assert np.array_equal(np.frombuffer(got, dtype=np.float64).reshape(3, 2)[:, 0], mean[i])
    q.stdin.close()
    assert q.wait(timeout=10) == 0
