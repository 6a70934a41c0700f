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
