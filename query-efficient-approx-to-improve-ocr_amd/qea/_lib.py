"""ctypes binding of libqea_hip.so (the C ABI declared in include/qea_hip.h).

Prototypes are read from the header itself, so the Python side cannot drift from the ABI.
There is deliberately no fallback: if the shared library is missing or a call fails the
caller gets an exception, never a silently different code path.
"""
import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.dirname(_HERE)
LIB_PATH = os.environ.get("QEA_HIP_LIB") or os.path.join(PKG_DIR, "libqea_hip.so")   # QEA_HIP_LIB: A/B a second build
HEADER_PATH = os.path.join(os.path.dirname(PKG_DIR), "include", "qea_hip.h")

_lib = None


class QeaError(RuntimeError):
    pass


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
        ("x_planes", _fp), ("w_planes", _fp), ("stats", _fp), ("w_frag_planes", _fp), ("x_absmax", _fp), ("y_absmax", _fp),
        ("pool_y", _fp), ("ldpool", _i32), ("pool_kw", _i32), ("pool_absmax", _fp),
        ("bst_y", _fp), ("ldbst", _i32), ("bst_stat64", _fp), ("bst_scale", _fp), ("bst_shift", _fp),
    ]


class WgradDesc(C.Structure):
    _fields_ = [
        ("p", _fp), ("q", _fp), ("dw", _fp), ("workspace", _fp), ("workspace_bytes", C.c_size_t),
        ("B", _i32), ("PH", _i32), ("PW", _i32), ("QH", _i32), ("QW", _i32), ("R", _i32), ("C", _i32),
        ("KH", _i32), ("KW", _i32), ("pad_h", _i32), ("pad_w", _i32), ("stride_h", _i32), ("stride_w", _i32),
        ("ldp", _i32), ("ldq", _i32), ("accumulate", _i32), ("splits", _i32), ("tile", _i32),
        ("p_absmax", _fp), ("q_absmax", _fp), ("dbias", _fp),
    ]


_SCALARS = {
    "int": C.c_int, "int32_t": C.c_int32, "int64_t": C.c_int64, "uint64_t": C.c_uint64,
    "size_t": C.c_size_t, "float": C.c_float, "double": C.c_double,
}
_PROTO = re.compile(r"^(int|size_t|const char\*)\s+(qea_\w+)\s*\(([^;{]*?)\)\s*;", re.M | re.S)


class WformJob(C.Structure):
    """qea_wform_job of include/qea_hip.h (qea_weight_forms_multi)"""
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("amax", C.c_void_p), ("kind", C.c_int32), ("a", C.c_int32), ("b", C.c_int32),
                ("c", C.c_int32), ("d", C.c_int32)]


def header_prototypes(path=HEADER_PATH):
    """[(name, restype, [argtypes])] for every function the header declares."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = []
    for ret, name, args in _PROTO.findall(text):
        restype = {"int": C.c_int, "size_t": C.c_size_t, "const char*": C.c_char_p}[ret]
        argtypes = []
        args = " ".join(args.split())
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(C.c_void_p)
                else:
                    ty = a.replace("const ", "").split()[0]
                    argtypes.append(_SCALARS[ty])
        out.append((name, restype, argtypes))
    return out


def _try_build():
    """Build the library in-tree with hipcc when it is missing (same recipe as __graft_entry__.build()).
    This builds the ONLY implementation there is; it is not a fallback to a different code path."""
    import shutil
    import subprocess
    script = os.path.join(PKG_DIR, "csrc", "build.sh")
    if shutil.which("hipcc") and os.path.exists(script):
        subprocess.run(["bash", script], check=False, stdout=subprocess.DEVNULL)


def lib():
    """Load (once) and return the CDLL; raises QeaError if it is not built and cannot be built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH) and not os.environ.get("QEA_NO_AUTOBUILD"):
            _try_build()
        if not os.path.exists(LIB_PATH):
            raise QeaError(
                f"{LIB_PATH} not found: build it with `python __graft_entry__.py build` "
                "(hipcc --offload-arch=gfx950); there is no CPU fallback"
            )
        L = C.CDLL(LIB_PATH)
        for name, restype, argtypes in header_prototypes():
            fn = getattr(L, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().qea_last_error().decode("utf-8", "replace")
        raise QeaError(f"{what} failed ({rc}): {msg}")
