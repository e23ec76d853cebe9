"""Test aids (not product): ctypes loader of tools/diag/liblenv_diag.so."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblenv_diag.so")
_lib = None


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.lenv_diag_occupy_cus.restype = C.c_int
        _lib.lenv_diag_occupy_cus.argtypes = [C.c_int32, C.c_int32, C.c_int64, C.c_void_p]
    return _lib


def occupy_cus(blocks, lds_bytes, ticks, stream):
    """Hold `blocks` CUs for `ticks` of the 100 MHz clock on the HIP stream handle `stream` (an int)."""
    rc = lib().lenv_diag_occupy_cus(int(blocks), int(lds_bytes), int(ticks), C.c_void_p(stream))
    if rc != 0:
        raise RuntimeError("lenv_diag_occupy_cus: %d" % rc)
