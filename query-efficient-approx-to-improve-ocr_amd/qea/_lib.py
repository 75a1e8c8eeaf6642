"""ctypes binding of libqea_hip.so (the C ABI declared in include/qea_hip.h).

There is deliberately no fallback: if the shared library is missing or a call fails the
caller gets an exception, never a silently different code path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "libqea_hip.so")

_lib = None


class QeaError(RuntimeError):
    pass


def lib():
    """Load (once) and return the CDLL; raises QeaError if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise QeaError(
                f"{LIB_PATH} not found: build it with `python __graft_entry__.py build` "
                "(hipcc --offload-arch=gfx950); there is no CPU fallback"
            )
        _lib = C.CDLL(LIB_PATH)
        _declare(_lib)
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().qea_last_error().decode("utf-8", "replace")
        raise QeaError(f"{what} failed ({rc}): {msg}")


_fp = C.c_void_p
_i32 = C.c_int32


class ConvDesc(C.Structure):
    _fields_ = [
        ("x", _fp), ("w", _fp), ("y", _fp), ("scale", _fp), ("bias", _fp), ("mask", _fp),
        ("B", _i32), ("H", _i32), ("W", _i32), ("Cin", _i32),
        ("OH", _i32), ("OW", _i32), ("N", _i32),
        ("KH", _i32), ("KW", _i32), ("pad_h", _i32), ("pad_w", _i32),
        ("stride_h", _i32), ("stride_w", _i32),
        ("ldx", _i32), ("ldy", _i32), ("ldmask", _i32),
        ("relu", _i32), ("accumulate", _i32), ("out_mode", _i32), ("tile", _i32),
    ]


class WgradDesc(C.Structure):
    _fields_ = [
        ("p", _fp), ("q", _fp), ("dw", _fp), ("workspace", _fp), ("workspace_bytes", C.c_size_t),
        ("B", _i32), ("PH", _i32), ("PW", _i32), ("QH", _i32), ("QW", _i32), ("R", _i32), ("C", _i32),
        ("KH", _i32), ("KW", _i32), ("pad_h", _i32), ("pad_w", _i32), ("stride_h", _i32), ("stride_w", _i32),
        ("ldp", _i32), ("ldq", _i32), ("accumulate", _i32), ("splits", _i32), ("tile", _i32),
    ]


def _declare(L):
    L.qea_version.restype = C.c_int
    L.qea_last_error.restype = C.c_char_p
    L.qea_prof_enable.argtypes = [C.c_int, C.c_int]
    L.qea_prof_read.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.qea_conv_igemm.argtypes = [C.POINTER(ConvDesc), _fp]
    L.qea_conv_wgrad.argtypes = [C.POINTER(WgradDesc), _fp]
    L.qea_conv_wgrad_workspace_bytes.argtypes = [C.POINTER(WgradDesc)]
    L.qea_conv_wgrad_workspace_bytes.restype = C.c_size_t
