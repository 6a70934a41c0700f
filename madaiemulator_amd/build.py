"""Builds the in-tree native libraries.

  lib/libgpemu_hip.so : HIP kernels + C-ABI (include/gpemu.h), hipcc --offload-arch=gfx950
  lib/libEmuMI.so     : C99 host layer mirroring the reference's libEmu interface on top of the C-ABI

hipcc cross-compiles without a GPU, so this also runs in the CPU-only build container.
"""
import glob
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
LIBDIR = os.path.join(PKG, "lib")
HIP_SRC = os.path.join(PKG, "csrc", "hip")
HOST_SRC = os.path.join(PKG, "csrc", "host")
INCLUDE = os.path.join(ROOT, "include")

HIP_LIB = os.path.join(LIBDIR, "libgpemu_hip.so")
HOST_LIB = os.path.join(LIBDIR, "libEmuMI.so")
CLI_BIN = os.path.join(LIBDIR, "interactive_emulator")
EPP_LIB = os.path.join(LIBDIR, "libEmuPlusPlusMI.so")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def build_hip(force=False, verbose=False):
    srcs = sorted(glob.glob(os.path.join(HIP_SRC, "*.hip")))
    deps = srcs + glob.glob(os.path.join(HIP_SRC, "*.hpp")) + glob.glob(os.path.join(INCLUDE, "*.h"))
    os.makedirs(LIBDIR, exist_ok=True)
    if not force and not _newer(HIP_LIB, deps):
        return HIP_LIB
    objs = []
    for s in srcs:
        o = os.path.join(LIBDIR, os.path.basename(s) + ".o")
        if force or _newer(o, [s] + deps[len(srcs):]):
            cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-Wno-unused-value", "-I", INCLUDE, "-I", HIP_SRC,
                   "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        objs.append(o)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", HIP_LIB] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return HIP_LIB


def build_host(force=False, verbose=False):
    srcs = sorted(glob.glob(os.path.join(HOST_SRC, "*.c")))
    lib_srcs = [s for s in srcs if os.path.basename(s) != "interactive_emulator.c"]
    deps = srcs + glob.glob(os.path.join(HOST_SRC, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h"))
    if not lib_srcs:
        return None
    if force or _newer(HOST_LIB, deps + [HIP_LIB]):
        cmd = ["gcc", "-std=gnu99", "-O2", "-fPIC", "-shared", "-Wall", "-I", INCLUDE, "-I", HOST_SRC,
               "-o", HOST_LIB] + lib_srcs + ["-L", LIBDIR, "-lgpemu_hip", "-Wl,-rpath,$ORIGIN", "-lm", "-lpthread"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    cpp = os.path.join(HOST_SRC, "EmuPlusPlus.cpp")
    if os.path.exists(cpp) and (force or _newer(EPP_LIB, deps + [cpp, HOST_LIB])):
        cmd = ["g++", "-std=c++11", "-O2", "-fPIC", "-shared", "-Wall", "-I", INCLUDE, "-I", HOST_SRC, "-o", EPP_LIB, cpp,
               "-L", LIBDIR, "-lEmuMI", "-lgpemu_hip", "-Wl,-rpath,$ORIGIN", "-lm", "-lpthread"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    cli = os.path.join(HOST_SRC, "interactive_emulator.c")
    if os.path.exists(cli) and (force or _newer(CLI_BIN, deps + [HOST_LIB])):
        cmd = ["gcc", "-std=gnu99", "-O2", "-Wall", "-I", INCLUDE, "-I", HOST_SRC, "-o", CLI_BIN, cli,
               "-L", LIBDIR, "-lEmuMI", "-lgpemu_hip", "-Wl,-rpath,$ORIGIN", "-lm", "-lpthread"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return HOST_LIB


def build_all(force=False, verbose=False):
    build_hip(force, verbose)
    build_host(force, verbose)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
