"""The C-ABI shared library loads and exports exactly what include/gpemu.h declares (no GPU needed)."""
import ctypes
import os
import re

from madaiemulator_amd import abi, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "gpemu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gpemu_[a-z_0-9]+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    path = build.build_hip()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in gpemu.h but not exported"
    assert sorted(abi.SYMBOLS) == names, "abi.py binding table and gpemu.h disagree"


def test_binding_loads_and_reports_version():
    L = abi.load()
    assert b"gfx950" in L.gpemu_version()


def test_no_cpu_fallback():
    # without a HIP device the context cannot be created: the product path fails loudly, it never computes on the CPU
    L = abi.load()
    if L.gpemu_device_count() == 0:
        h = ctypes.c_void_p()
        assert L.gpemu_ctx_create(ctypes.byref(h), 0) == abi.ERR_NO_DEVICE
        assert not h.value


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "madaiemulator_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hip", ".hpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower(), \
                    f"{f} mentions the oracle: the product path must not depend on test infrastructure"


def test_header_is_plain_c_and_links(tmp_path):
    """include/gpemu.h is what a C host (the reference is C99) includes: it must compile as C99 and as C++, with no
    HIP or torch headers, and a program that only uses it must link against the library"""
    import subprocess
    build.build_hip()
    src = tmp_path / "use_header.c"
    src.write_text(
        '#include "gpemu.h"\n#include <stdio.h>\n'
        'int main(void) { gpemu_ctx *c = 0; int rc = gpemu_ctx_create(&c, 0);\n'
        '  printf("%s devices=%d rc=%d max_batch=%d\\n", gpemu_version(), gpemu_device_count(), rc, GPEMU_MAX_BATCH);\n'
        '  if (c) { gpemu_ctx_destroy(c); }\n  return 0; }\n')
    exe = tmp_path / "use_header"
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", inc, "-o", str(exe), str(src),
                           "-L", build.LIBDIR, "-lgpemu_hip", f"-Wl,-rpath,{build.LIBDIR}"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "gfx950" in out.stdout, out.stderr
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Werror", "-I", inc, "-x", "c++", "-fsyntax-only", str(src)])


def test_gemm_tile_table_is_a_balanced_permutation():
    """host logic of the XCD-blocked GEMM tile order (kernels_linalg.hip: build_tile_table): every valid tile exactly
    once, the 8 XCD shares equal to within one tile, unused slots only at the tail of a share, and the tiles of a
    share taken super-block by super-block"""
    import numpy as np
    L = abi.load()
    for tiles_m, tiles_n, tri, sb in [(60, 60, 1, 8), (57, 8, 0, 8), (1, 1, 1, 8), (3, 700, 0, 8), (33, 33, 1, 5),
                                      (128, 128, 1, 16), (9, 9, 1, 1), (121, 120, 1, 8)]:
        n = L.gpemu_test_tile_table(tiles_m, tiles_n, tri, sb, None, 0)
        assert n > 0 and n % 8 == 0
        buf = (ctypes.c_int * n)()
        assert L.gpemu_test_tile_table(tiles_m, tiles_n, tri, sb, buf, n) == n
        t = np.array(buf[:], dtype=np.int64).reshape(-1, 8)          # row q, column x = q-th tile of XCD x
        want = {(tm, tn) for tm in range(tiles_m) for tn in range(tiles_n) if not (tri and tn > tm)}
        got = [(int(e) >> 16, int(e) & 0xffff) for e in t.ravel() if e >= 0]
        assert len(got) == len(want) and set(got) == want
        counts = (t >= 0).sum(axis=0)
        assert counts.max() - counts.min() <= 1
        for x in range(8):                                           # padding only at the tail of a share
            col = t[:, x]
            assert np.all(col[:counts[x]] >= 0) and np.all(col[counts[x]:] < 0)
        # a share is a contiguous piece of the super-block sequence: the tiles of 64 consecutive slots span few blocks
        if len(want) >= 8 * 4 * sb * sb:
            share = [(int(e) >> 16, int(e) & 0xffff) for e in t[:, 3] if e >= 0][:sb * sb]
            blocks = {(tm // sb, tn // sb) for tm, tn in share}
            assert len(blocks) <= 4          # (triangular diagonal blocks hold half the tiles)
    assert L.gpemu_test_tile_table(0, 4, 0, 8, None, 0) < 0 and L.gpemu_test_tile_table(4, 40000, 0, 8, None, 0) < 0


def test_row_table_gives_whole_tile_rows_to_an_xcd():
    """the workgroup -> tile table of C^-1 = U U^T (host logic, no device): every lower tile exactly once, a tile row on
    ONE XCD, an XCD's rows in order of decreasing k-range, the XCDs' shares of work within a few per cent"""
    import numpy as np
    L = abi.load()
    for tiles_m, bm, off, k1 in [(33, 128, 64, 4096), (65, 128, 64, 8192), (17, 128, 64, 2048), (129, 128, 64, 16384), (40, 64, 64, 2496)]:
        n = L.gpemu_test_row_table(tiles_m, bm, off, 0, k1, None, 0)
        assert n > 0 and n % 8 == 0
        buf = (ctypes.c_int * n)()
        assert L.gpemu_test_row_table(tiles_m, bm, off, 0, k1, buf, n) == n
        t = np.array(buf[:], dtype=np.int64).reshape(-1, 8)
        got = [(int(e) >> 16, int(e) & 0xffff) for e in t.ravel() if e >= 0]
        want = {(r, c) for r in range(tiles_m) for c in range(r + 1)}
        assert len(got) == len(want) and set(got) == want
        owner, work = {}, np.zeros(8)
        for x in range(8):
            col = [int(e) for e in t[:, x]]
            cnt = sum(e >= 0 for e in col)
            assert all(e >= 0 for e in col[:cnt]) and all(e < 0 for e in col[cnt:])
            rows = [e >> 16 for e in col[:cnt]]
            assert rows == sorted(rows)                              # decreasing k-range = increasing row, a row's tiles together
            for r in set(rows):
                assert owner.setdefault(r, x) == x
                ks = max(0, ((r * bm - off) // 16) * 16)
                work[x] += (r + 1) * (k1 - ks)
        assert len(owner) == tiles_m
        assert work.max() / work.mean() < (1.08 if tiles_m >= 33 else 1.15)
    assert L.gpemu_test_row_table(0, 128, 64, 0, 4096, None, 0) < 0
