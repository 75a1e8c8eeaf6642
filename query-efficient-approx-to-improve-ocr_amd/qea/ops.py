"""Thin tensor-level wrappers over the C ABI (include/qea_hip.h).

Every function takes CUDA (ROCm) fp32 tensors, passes raw device pointers + the current
stream to libqea_hip.so and raises on any error.  No CPU implementation exists here.
"""
import ctypes as C

import torch

from . import _lib

OUT_NHWC, OUT_TBC, OUT_CONVT = 0, 1, 2
PROF_CONV_IGEMM, PROF_CONV_WGRAD, PROF_LSTM_STEP = 0, 1, 2

_ws = {}


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.QeaError("qea ops need CUDA tensors (there is no CPU path)")
    if t.dtype not in (torch.float32, torch.int32, torch.int64, torch.float64, torch.uint8):
        raise _lib.QeaError(f"unsupported dtype {t.dtype}")
    return t.data_ptr()


def workspace(nbytes, device):
    """Grow-only scratch buffer per device (caller-owned workspace of the C ABI)."""
    key = (device.type, device.index)
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws[key] = buf
    return buf


def conv_igemm(x, w, y, *, B, H, W, Cin, OH, OW, N, KH, KW, pad=(0, 0), stride=(1, 1), ldx, ldy,
               scale=None, bias=None, mask=None, ldmask=0, relu=False, accumulate=False,
               out_mode=OUT_NHWC, tile=0):
    d = _lib.ConvDesc(x=_ptr(x), w=_ptr(w), y=_ptr(y), scale=_ptr(scale), bias=_ptr(bias), mask=_ptr(mask),
                      B=B, H=H, W=W, Cin=Cin, OH=OH, OW=OW, N=N, KH=KH, KW=KW, pad_h=pad[0], pad_w=pad[1],
                      stride_h=stride[0], stride_w=stride[1], ldx=ldx, ldy=ldy, ldmask=ldmask,
                      relu=int(relu), accumulate=int(accumulate), out_mode=out_mode, tile=tile)
    _lib.check(_lib.lib().qea_conv_igemm(C.byref(d), _stream()), "qea_conv_igemm")


def conv_wgrad(p, q, dw, *, B, PH, PW, QH, QW, R, Cc, KH, KW, pad=(0, 0), stride=(1, 1), ldp, ldq,
               accumulate=False, splits=0, tile=0):
    L = _lib.lib()
    d = _lib.WgradDesc(p=_ptr(p), q=_ptr(q), dw=_ptr(dw), workspace=None, workspace_bytes=0,
                       B=B, PH=PH, PW=PW, QH=QH, QW=QW, R=R, C=Cc, KH=KH, KW=KW, pad_h=pad[0], pad_w=pad[1],
                       stride_h=stride[0], stride_w=stride[1], ldp=ldp, ldq=ldq, accumulate=int(accumulate),
                       splits=splits, tile=tile)
    need = L.qea_conv_wgrad_workspace_bytes(C.byref(d))
    if need:
        ws = workspace(need, p.device)
        d.workspace = ws.data_ptr()
        d.workspace_bytes = ws.numel()
    _lib.check(L.qea_conv_wgrad(C.byref(d), _stream()), "qea_conv_wgrad")


def prof_enable(klass, on=True):
    _lib.check(_lib.lib().qea_prof_enable(klass, int(on)), "qea_prof_enable")


def prof_reset():
    _lib.check(_lib.lib().qea_prof_reset(), "qea_prof_reset")


def prof_read(klass):
    ms, fl, by, n = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
    _lib.check(_lib.lib().qea_prof_read(klass, C.byref(ms), C.byref(fl), C.byref(by), C.byref(n)), "qea_prof_read")
    return {"ms": ms.value, "flops": fl.value, "bytes": by.value, "launches": n.value}
