"""Loads the emulator build of the kernels (tests/emu) and binds the product ctypes prototypes to it."""
import ctypes
import functools
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'emu'))


@functools.lru_cache(maxsize=1)
def emu_cdll():
    import build_emu
    from mvector import _hip
    return _hip.bind(ctypes.CDLL(build_emu.build()))
