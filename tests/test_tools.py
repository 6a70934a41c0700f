"""The profiling helpers under tools/ parse what the device library / rocprofv3 write (no GPU needed)."""
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_trace_timeline_merges_two_contexts(tmp_path):
    # the line format of gpemu_trace_dump: tag | start_ns end_ns sum_wg_ns workgroups sum_wg_clocks stamp1 stamp2 stamp3
    a = tmp_path / "ctx0.txt"
    b = tmp_path / "ctx1.txt"
    a.write_text("gemm m=7232 n=7168 k=1024 | 1000 501000 16000000 64 36800000 100 200 0\n"
                 "leaf_factor c0=0 | 501000 511000 10000 1 24000 1700 6400 8100\n"
                 "gemm m=8192 n=64 k=64 | 512000 520000 500000 50 1150000 10 20 0\n")
    b.write_text("gemm m=6208 n=6144 k=512 | 200000 450000 8000000 64 18400000 100 200 0\n"
                 "leaf_solve c0=0 m=8000 | 450000 470000 320000 16 736000 0 0 0\n")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "trace_timeline.py"), str(a), str(b)],
                         capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    assert "ctx 0" in out.stdout and "ctx 1" in out.stdout
    assert "gemm_k1024 0.500 ms" in out.stdout and "gemm_k512 0.250 ms" in out.stdout
    assert "gemm_narrow" in out.stdout and "leaf_factor" in out.stdout and "leaf_solve" in out.stdout
    assert "big GEMMs in flight" in out.stdout


def test_rocpd_summary_groups_kernels(tmp_path):
    db = tmp_path / "r.db"
    con = sqlite3.connect(db)
    con.execute("create table kernels (name text, start integer, end integer, grid_y integer)")
    rows = [("void gpemu::gemm_nt_kernel<128, 128, 4, 4, 2>(gpemu::GemmArgs)", 0, 4000000, 16),
            ("void gpemu::gemm_nt_kernel<64, 64, 4, 2, 2>(gpemu::GemmArgs)", 4000000, 4100000, 16),
            ("void gpemu::gemm_nt_kernel<64, 64, 4, 2, 2>(gpemu::GemmArgs)", 4100000, 4120000, 1),
            ("gpemu::leaf_factor_kernel(double*, long, int, int*)", 4120000, 4130000, 16)]
    con.executemany("insert into kernels values (?,?,?,?)", rows)
    con.commit()
    con.close()
    tool = os.path.join(ROOT, "tools", "rocpd_summary.py")
    full = subprocess.run([sys.executable, tool, str(db)], capture_output=True, text=True, timeout=60)
    assert full.returncode == 0, full.stderr
    assert "gemm_nt_kernel (all tile shapes)" in full.stdout and "      3 " in full.stdout
    only = subprocess.run([sys.executable, tool, str(db), "--grid-y", "16"], capture_output=True, text=True, timeout=60)
    assert only.returncode == 0 and "      2 " in only.stdout.split("gemm_nt_kernel (all tile shapes)")[1]


def test_every_environment_variable_the_library_reads_is_documented():
    """INTEGRATION.md lists the switches: a GPEMU_* name read with getenv anywhere in csrc/ must appear there"""
    import glob, os, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    names = set()
    for f in glob.glob(os.path.join(root, "madaiemulator_amd", "csrc", "*", "*")):
        if not f.endswith((".c", ".cpp", ".hip", ".h", ".hpp")):
            continue
        for line in open(f, errors="replace"):
            if "getenv" in line or "geti(" in line or "env_int(" in line:
                names.update(re.findall(r'"(GPEMU_[A-Z0-9_]+)"', line))
    assert len(names) > 30
    missing = sorted(n for n in names if n not in doc)
    assert not missing, missing
