"""Test helper: builds and loads the SIMT-emulated build of libtkz (tests/hostemu/) so that the real
kernel sources can be exercised on a CPU by the `not gpu` tests.  Test infrastructure only."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
EMU_DIR = os.path.join(_HERE, "hostemu")
EMU_LIB = os.path.join(EMU_DIR, "_build", "libtkz_hostemu.so")

_lib = None


def library():
    global _lib
    if _lib is None:
        subprocess.check_call(["make", "-C", EMU_DIR, "-s"])
        from tokenizer_amd import _native
        _lib = _native.Library(EMU_LIB)
    return _lib
