"""madaiemulator_amd -- MI355X-native GP likelihood / prediction hot path of MADAIEmulator.

The compute path is the HIP library behind include/gpemu.h (lib/libgpemu_hip.so);
this package only carries the build driver and the ctypes binding used by the
tests and bench.py.
"""
from . import build  # noqa: F401

__all__ = ["build", "abi"]
