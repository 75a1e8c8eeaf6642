"""Thin tensor-level wrappers over the C ABI (include/qea_hip.h).

Every function takes CUDA (ROCm) fp32 tensors, passes raw device pointers + the current
stream to libqea_hip.so and raises on any error.  No CPU implementation exists here.
"""
import ctypes as C
import os
import weakref

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
    """Grow-only scratch buffer per (device, stream): kernels on different streams never share one."""
    if device.type != "cuda":
        raise _lib.QeaError("qea ops need CUDA tensors (there is no CPU path)")
    key = (device.type, device.index, _stream())
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws[key] = buf
    return buf


def split_planes(x, ld, M, Cc):
    """fp32 rows [M][ld] (Cc channels used) -> the P3 pre-split format (include/qea_hip.h: qea_split_planes) as a byte tensor."""
    L = _lib.lib()
    out = torch.empty(L.qea_split_planes_bytes(M, Cc), dtype=torch.uint8, device=x.device)
    _lib.check(L.qea_split_planes(_ptr(x), ld, M, Cc, out.data_ptr(), _stream()), "qea_split_planes")
    return out


FUSE_BN_STATS = {"on": True}     # tests switch the fused statistics epilogue off to compare with the separate pass
PRESPLIT = {"on": True, "x": False}     # tests / tools: "on" = pre-split filters (default), "x" = pre-split activations too

# ---- derived forms of WEIGHTS (dgrad filter layouts, transposes, P3 planes), kept until the weights change.
# A weight tensor changes either through torch (load_state_dict, copy_: its _version moves) or through the fused Adam kernel
# (raw pointer writes: qea.optim.FusedAdam.step calls bump_weight_epoch()).  With the CRNN frozen in Phase B its derived forms
# are built once, not once per step (round 1 re-derived 390 filter layouts per step).  Inside a hipGraph capture the cache is
# scoped to that capture (see weight_cached): a replay re-derives every form from the weights it finds, once per weight epoch.
# RULE for any other writer: a write that goes through `p.data` (torch.distributed.broadcast(p.data), p.data.copy_, an EMA) or
# through a raw pointer moves neither _version nor the epoch — call bump_weight_epoch() after it (qea.graph.GraphedStep does
# after every replay, TrainerCore after the start-up broadcast, qea.params.FlatState when it re-homes the parameters).
_wcache = {}
_wepoch = [0]
WEIGHT_CACHE = {"on": True}
CAPTURE = {"token": None}


MODEL_EPOCHS = {"on": os.environ.get("QEA_MODEL_EPOCHS", "1") != "0"}   # 0: every raw-pointer writer makes ALL derived forms stale (round-3 behaviour)


def bump_weight_epoch(params=None):
    """Derived weight forms are stale from here on.  params: the parameters a raw-pointer writer has just touched — when every one of
    them lives in a flat parameter buffer, only the forms of those models go stale (the fused Adam of one model must not make
    the other model's filters be re-packed: round 4 found every form built twice per step); otherwise, or without
    params, everything."""
    if params is not None and MODEL_EPOCHS["on"]:
        from .params import flat_state_of
        seen = {}
        for p_ in params:
            fs = flat_state_of(p_, check=False)
            if fs is None:
                seen = None
                break
            seen[id(fs)] = fs
        if seen:
            for fs in seen.values():
                fs.epoch += 1
            return
    _wepoch[0] += 1


def _model_epoch(w):
    from .params import flat_state_of
    fs = flat_state_of(w, check=False)
    return fs.epoch if fs is not None else 0


def weight_cached(kind, w, build, also=(), extra=None, batch=None):
    """build() -> tensor(s) derived from the weight tensor `w` (and the tensors in `also`) only; cached per (kind, the
    tensor OBJECT) until one of them changes.  Entries die with the tensor object (weakref), so a new tensor that happens
    to reuse the address of a freed one can never hit.  `extra`: any further hashable the value depends on."""
    token = None
    if torch.cuda.is_current_stream_capturing():
        # inside a capture made by qea.graph.GraphedStep a derived form is built once per weight epoch OF THAT CAPTURE (the build
        # launches are part of the graph and run before their uses on every replay); an entry carries the capture's token, so
        # neither eager code nor another capture can ever hit it.  Any other capture caches nothing.
        token = CAPTURE["token"]
        if token is None:
            return build()
    if not WEIGHT_CACHE["on"]:
        return build()
    key = (kind, id(w))
    ver = _cache_ver(w, also, token, extra)
    hit = _wcache.get(key)
    if hit is not None and hit[0] == ver and hit[2]() is w:
        return hit[1]
    if batch is not None and BATCH_FORMS["on"] and extra is None and w.is_cuda and (w.data_ptr() & 15) == 0:
        return _batched_forms(key, w, also, batch, token)
    val = build()
    _wcache[key] = (ver, val, weakref.ref(w, lambda _r, key=key: _wcache.pop(key, None)))
    return val


# ---- derived forms of one model in ONE launch (round 4).  A form that qea_weight_forms_multi can make registers a FormJob under its
# cache key; the first miss after a weight update then builds every stale registered form of the SAME group whose weight lives in the
# same flat parameter buffer (= the same model) with one launch, and the later requests of the step hit the cache.  Groups: "flip"
# (flip-transposed filters) and "pack" (fragment planes; a pack of a flipped filter asks for the flipped filter first, which runs
# the flip group).  Forms outside a flat buffer (tests, ad-hoc weights) keep their single launches.
BATCH_FORMS = {"on": os.environ.get("QEA_BATCH_FORMS", "1") != "0"}
_form_jobs = {}


class FormJob:
    """plain data (no reference to the weight: a registered job must not keep its tensor alive).  srck: "self" = the weight itself,
    "flipT" = its flip-transposed form (dims of the 3x3 pack: N, Cin of the input-gradient conv); out: shape or byte count;
    amax: (ld, M, Cc) of the filter_absmax call, or None."""
    __slots__ = ("group", "kind", "dims", "srck", "out", "amax", "wref", "also")

    def __init__(self, group, kind, dims, srck, out, amax):
        self.group, self.kind, self.dims, self.srck, self.out, self.amax = group, kind, dims, srck, out, amax

    def source(self, w):
        return w if self.srck == "self" else flip_transposed(w, self.dims[1], self.dims[0], 3, 3)

    def alloc(self, w):
        return torch.empty(self.out, device=w.device) if isinstance(self.out, tuple) else torch.empty(self.out, dtype=torch.uint8, device=w.device)

    def scale(self, w):
        return None if self.amax is None else filter_absmax((self.srck, w), w, *self.amax)


def _cache_ver(w, also, token, extra=None):
    return (w.data_ptr(), w._version, _wepoch[0], _model_epoch(w), _stream(), token, extra) + tuple((id(t), t.data_ptr(), t._version, _model_epoch(t)) for t in also)


def _batched_forms(key, w, also, job, token):
    """builds `key` and every other stale registered form of its group and model; returns key's tensor"""
    from .params import flat_state_of
    job.wref, job.also = weakref.ref(w), tuple(weakref.ref(t) for t in also)
    _form_jobs[key] = job
    fs = flat_state_of(w, check=False)
    todo = [(key, w, also, job)]
    if fs is not None:
        dead = []                                            # jobs of weights that are gone: dropped here, never from a finalizer
        for k2, j2 in _form_jobs.items():                    # (a callback that edits the dict could run inside this very loop)
            w2 = j2.wref()
            also2 = tuple(r() for r in j2.also)
            if w2 is None or any(t is None for t in also2):
                dead.append(k2)
                continue
            if k2 == key or j2.group != job.group or len(todo) >= 64 or flat_state_of(w2, check=False) is not fs:
                continue
            hit = _wcache.get(k2)
            if hit is not None and hit[0] == _cache_ver(w2, also2, token) and hit[2]() is w2:
                continue
            todo.append((k2, w2, also2, j2))
        for k2 in dead:
            del _form_jobs[k2]
    jobs = (_lib.WformJob * len(todo))()
    outs, keep = [], []
    for i, (_k, wi, _a, ji) in enumerate(todo):
        src, out, am = ji.source(wi), ji.alloc(wi), ji.scale(wi)
        keep.append((src, am))
        outs.append(out)
        jobs[i].src, jobs[i].dst, jobs[i].amax, jobs[i].kind = src.data_ptr(), out.data_ptr(), (am.data_ptr() if am is not None else None), ji.kind
        jobs[i].a, jobs[i].b, jobs[i].c, jobs[i].d = ji.dims
    _lib.check(_lib.lib().qea_weight_forms_multi(jobs, len(todo), _stream()), "qea_weight_forms_multi")
    for (ki, wi, ai, _j), out in zip(todo, outs):
        _wcache[ki] = (_cache_ver(wi, ai, token), out, weakref.ref(wi, lambda _r, key=ki: _wcache.pop(key, None)))
    return outs[0]


def flip_transposed(w, Co, Ci, KH, KW):
    """wt[ci][KH-1-kh][KW-1-kw][co] = w[co][kh][kw][ci]: the filter of the input-gradient convolution (cached)."""
    def build():
        wt = torch.empty(Ci, KH, KW, Co, device=w.device)
        filter_flip_transpose(w, wt, Co, Ci, KH, KW)
        return wt
    return weight_cached(("flipT", Co, Ci, KH, KW), w, build, batch=FormJob("flip", 0, (Co, Ci, KH, KW), "self", (Ci, KH, KW, Co), None))


def transposed(w, R, Cc):
    """out[c][r] = w[r][c] (cached)."""
    def build():
        out = torch.empty(Cc, R, device=w.device)
        transpose2d(w, out, R, Cc)
        return out
    return weight_cached(("T", R, Cc), w, build)


def conv_igemm(x, w, y, *, B, H, W, Cin, OH, OW, N, KH, KW, pad=(0, 0), stride=(1, 1), ldx, ldy,
               scale=None, bias=None, mask=None, ldmask=0, relu=False, accumulate=False,
               out_mode=OUT_NHWC, tile=0, x_planes=None, w_planes=None, w_src=None, want_stats=False, x_amax=None, y_amax=None, pool=None,
               bwd_stats=None):
    """x_planes / w_planes: operands already in the P3 format (a caller that uses a tensor in several launches splits it
    once); when the launch runs on a split-bf16 tile and they are not given, they are made here (one HBM pass each).
    w_src = (kind, weight tensor[, (further weight tensors)]): `w` is a function of that weight (and of the further ones) only (itself:
    kind "fwd"; its cached flip_transposed / transposed form: "flipT" / "T"; the BiLSTM's concatenated projections: "catT" of both
    directions), so its planes are cached with it until one of them changes.
    want_stats: ask for the fused BatchNorm-statistics epilogue; returns (partials [blocks][N][2] fp64, blocks) when the chosen
    kernel has it, else None (the caller then runs bn_train_stats over y).
    pool = (pooled, ldpool, kw, amax slot or None): the 2 x kw max-pool of y leaves with the epilogue (same bits as maxpool_fwd on y) —
    only for a launch conv_can_pool(...) accepted.
    bwd_stats = (y_ref, ld, stat64 [2][N], scale, shift): y (this launch's output) is the gradient entering a train-mode BatchNorm+ReLU
    whose input was y_ref: the epilogue also leaves the two reductions of that BatchNorm's backward; returns (partials, blocks) for
    bn_bwd(partials=...), or None when this launch has no such epilogue (the caller's bn_bwd then makes its own pass)."""
    L = _lib.lib()
    d = _lib.ConvDesc(x=_ptr(x), w=_ptr(w), y=_ptr(y), scale=_ptr(scale), bias=_ptr(bias), mask=_ptr(mask),
                      B=B, H=H, W=W, Cin=Cin, OH=OH, OW=OW, N=N, KH=KH, KW=KW, pad_h=pad[0], pad_w=pad[1],
                      stride_h=stride[0], stride_w=stride[1], ldx=ldx, ldy=ldy, ldmask=ldmask,
                      relu=int(relu), accumulate=int(accumulate), out_mode=out_mode, tile=tile, x_planes=None, w_planes=None, stats=None,
                      w_frag_planes=None)
    frag = xmax = None
    wants = L.qea_conv_igemm_wants_frag_planes(C.byref(d)) if PRESPLIT["on"] else 0
    if wants == 2 and SPLIT_F16["on"] and x_planes is None and w_planes is None:
        # 1x1 GEMM on the 128-row LDS tile (tile 26, fp16 split only): filter [N][K] in fragment-order fp16 planes (cached)
        def build():
            out = torch.empty(L.qea_pack_frag_planes_f16_1x1_bytes(N, Cin), dtype=torch.uint8, device=x.device)
            wmax = filter_absmax(w_src, w, Cin, N, Cin)
            _lib.check(L.qea_pack_frag_planes_f16_1x1(_ptr(w), N, Cin, _ptr(wmax), out.data_ptr(), _stream()), "qea_pack_frag_planes_f16_1x1")
            return out
        job = None
        if w_src is not None and w_src[0] == "fwd" and len(w_src) == 2 and w_src[1].data_ptr() == w.data_ptr():
            job = FormJob("pack", 2, (N, Cin, 0, 0), "self", L.qea_pack_frag_planes_f16_1x1_bytes(N, Cin), (Cin, N, Cin))
        frag = weight_cached(("frag1x1", w_src[0], N, Cin), w_src[1], build, also=tuple(w_src[2]) if len(w_src) > 2 else (), batch=job) if w_src is not None else build()
        d.w_frag_planes = frag.data_ptr()
        xmax = x_amax if x_amax is not None else absmax(x, ldx, B * H * W, Cin)
        d.x_absmax = xmax.data_ptr()
    elif wants == 1:
        # 3x3 layer on the split LDS-halo kernel: its filter in fragment-order planes (a few hundred KB, cached)
        f16 = SPLIT_F16["on"]

        def build():
            if f16:
                out = torch.empty(L.qea_pack_frag_planes_f16_bytes(N, Cin), dtype=torch.uint8, device=x.device)
                wmax = filter_absmax(w_src, w, 9 * Cin, N, 9 * Cin)
                _lib.check(L.qea_pack_frag_planes_f16(_ptr(w), N, Cin, _ptr(wmax), out.data_ptr(), _stream()), "qea_pack_frag_planes_f16")
                return out
            out = torch.empty(L.qea_pack_frag_planes_bytes(N, Cin), dtype=torch.uint8, device=x.device)
            _lib.check(L.qea_pack_frag_planes(_ptr(w), N, Cin, out.data_ptr(), _stream()), "qea_pack_frag_planes")
            return out
        job = None
        if f16 and w_src is not None and len(w_src) == 2 and KH == 3 and KW == 3:
            nb = L.qea_pack_frag_planes_f16_bytes(N, Cin)
            if w_src[0] == "fwd" and w_src[1].data_ptr() == w.data_ptr():
                job = FormJob("pack", 1, (N, Cin, 0, 0), "self", nb, (9 * Cin, N, 9 * Cin))
            elif w_src[0] == "flipT":
                # the input-gradient filter: planes of the flip-transposed weight (itself a cached form: asked for first, which
                # builds all stale flipped filters of the model in one launch)
                job = FormJob("pack", 1, (N, Cin, 0, 0), "flipT", nb, (9 * N, Cin, 9 * N))
        frag = weight_cached(("fragf16" if f16 else "frag", w_src[0], N, Cin), w_src[1], build, also=tuple(w_src[2]) if len(w_src) > 2 else (), batch=job) if w_src is not None else build()
        d.w_frag_planes = frag.data_ptr()
        if f16:
            # the input's abs-max: carried by the tensor's producer (x_amax), else one pass over the input here
            xmax = x_amax if x_amax is not None else absmax(x, ldx, B * H * W, Cin)
            d.x_absmax = xmax.data_ptr()
    elif PRESPLIT["on"] and Cin % 16 == 0 and L.qea_conv_igemm_uses_split_bf16(C.byref(d)):
        K = KH * KW * Cin
        if N * K * 6 < (1 << 31) - 256:
            # the FILTER's planes are made once per weight update and shared by every M-tile; an activation is consumed by one
            # launch, so it is only taken pre-split when the caller has the planes anyway (PRESPLIT["x"]: tools / tests)
            if x_planes is None and PRESPLIT.get("x") and B * H * W * Cin * 6 < (1 << 31) - 256:
                x_planes = split_planes(x, ldx, B * H * W, Cin)
            f16 = SPLIT_F16["on"] and x_planes is None and w_planes is None and N * K * 4 < (1 << 31) - 256
            if w_planes is None:
                build = (lambda: split_planes_f16(w, K, N, K, w_src)) if f16 else (lambda: split_planes(w, K, N, K))
                w_planes = weight_cached(("planesf16" if f16 else "planes", w_src[0], N, K), w_src[1], build, also=tuple(w_src[2]) if len(w_src) > 2 else ()) if w_src is not None else build()
            d.x_planes = x_planes.data_ptr() if x_planes is not None else None
            d.w_planes = w_planes.data_ptr()
            if f16:                                       # hybrid tile on the two-way fp16 split: the activations' abs-max
                xmax = x_amax if x_amax is not None else absmax(x, ldx, B * H * W, Cin)
                d.x_absmax = xmax.data_ptr()
    if y_amax is not None:
        d.y_absmax = y_amax.data_ptr()
    if pool is not None:
        d.pool_y, d.ldpool, d.pool_kw = pool[0].data_ptr(), pool[1], pool[2]
        d.pool_absmax = pool[3].data_ptr() if pool[3] is not None else None
    partials = None
    if bwd_stats is not None and wants == 1 and d.x_absmax and FUSE_BN_STATS["on"]:
        blocks = L.qea_conv_igemm_stats_blocks(C.byref(d))
        if blocks > 0:
            partials = torch.empty(blocks + 256, N, 2, dtype=torch.float64, device=x.device)   # + QEA_BN_PARTIAL_SCRATCH_ROWS
            d.stats = partials.data_ptr()
            d.bst_y, d.ldbst, d.bst_stat64 = bwd_stats[0].data_ptr(), bwd_stats[1], bwd_stats[2].data_ptr()
            d.bst_scale, d.bst_shift = bwd_stats[3].data_ptr(), bwd_stats[4].data_ptr()
    elif want_stats and FUSE_BN_STATS["on"]:
        blocks = L.qea_conv_igemm_stats_blocks(C.byref(d))
        if blocks > 0:
            partials = torch.empty(blocks + 256, N, 2, dtype=torch.float64, device=x.device)   # + QEA_BN_PARTIAL_SCRATCH_ROWS
            d.stats = partials.data_ptr()
    _lib.check(L.qea_conv_igemm(C.byref(d), _stream()), "qea_conv_igemm")
    return (partials, partials.shape[0] - 256) if partials is not None else None


def conv_can_pool(*, B, H, W, Cin, N, kw, ldx, ldy, mask=None):
    """True when the 3x3 pad-1 layer of this shape can carry the 2 x kw max-pool that follows it in its epilogue (conv_igemm's pool=):
    the two-way fp16 split is on and the LDS-halo kernel has the instance."""
    if not (PRESPLIT["on"] and SPLIT_F16["on"] and mfma_mode() == "split_f16") or mask is not None:
        return False
    d = _lib.ConvDesc(x=None, w=None, y=None, scale=None, bias=None, mask=None, B=B, H=H, W=W, Cin=Cin, OH=H, OW=W, N=N, KH=3, KW=3, pad_h=1, pad_w=1,
                      stride_h=1, stride_w=1, ldx=ldx, ldy=ldy, ldmask=0, relu=0, accumulate=0, out_mode=OUT_NHWC, tile=0, x_planes=None,
                      w_planes=None, stats=None, w_frag_planes=None)
    return bool(_lib.lib().qea_conv_igemm_can_pool(C.byref(d), kw))


def conv_wgrad(p, q, dw, *, B, PH, PW, QH, QW, R, Cc, KH, KW, pad=(0, 0), stride=(1, 1), ldp, ldq,
               accumulate=False, splits=0, tile=0, p_amax=None, q_amax=None, dbias=None):
    """p_amax / q_amax: device scalars holding the abs-max of p / q when the caller has them (the two-way fp16 split of the
    nine-tap kernel needs both); in "split_f16" mode a missing one is computed here (one pass over the tensor).
    dbias [R]: the bias gradient (column sums of p), same `accumulate`: inside the launch where the kernel has it (the producer /
    consumer nine-tap form), else one colsum pass here."""
    L = _lib.lib()
    d = _lib.WgradDesc(p=_ptr(p), q=_ptr(q), dw=_ptr(dw), workspace=None, workspace_bytes=0,
                       B=B, PH=PH, PW=PW, QH=QH, QW=QW, R=R, C=Cc, KH=KH, KW=KW, pad_h=pad[0], pad_w=pad[1],
                       stride_h=stride[0], stride_w=stride[1], ldp=ldp, ldq=ldq, accumulate=int(accumulate),
                       splits=splits, tile=tile)
    nine_tap = (KH == 3 and KW == 3 and pad == (1, 1) and stride == (1, 1) and PH == QH and PW == QW and R % 32 == 0 and Cc % 32 == 0
                and (PW % 32 == 0 or PW == 16) and tile in (0, 23, 29))
    generic_split = tile in (20, 21, 22) or (tile == 0 and ((R >= 128 and Cc >= 128) or (R >= 128 and Cc == 64) or (R == 64 and Cc >= 128)))
    if SPLIT_F16["on"] and (nine_tap or generic_split) and mfma_mode() != "f32":
        if p_amax is None:
            p_amax = absmax(p, ldp, B * PH * PW, R)
        if q_amax is None:
            q_amax = absmax(q, ldq, B * QH * QW, Cc)
        d.p_absmax, d.q_absmax = p_amax.data_ptr(), q_amax.data_ptr()
    if dbias is not None:
        if L.qea_conv_wgrad_fuses_bias(C.byref(d)):
            d.dbias = dbias.data_ptr()
        else:
            colsum(p, ldp, B * PH * PW, R, dbias, accumulate=accumulate)
    need = L.qea_conv_wgrad_workspace_bytes(C.byref(d))
    if need:
        ws = workspace(need, p.device)
        d.workspace = ws.data_ptr()
        d.workspace_bytes = ws.numel()
    _lib.check(L.qea_conv_wgrad(C.byref(d), _stream()), "qea_conv_wgrad")


# "split_f16" (default since round 3): the LDS-halo 3x3 kernels take the TWO-way fp16 split (ABI v6: three MFMAs per product,
# operands scaled by powers of two from their abs-max); every other GEMM-class launch as in "split_bf16".  QEA_SPLIT=bf16 in the
# environment keeps the three-way bf16 split everywhere.
SPLIT_F16 = {"on": os.environ.get("QEA_SPLIT", "f16").lower() != "bf16"}


def set_mfma_mode(mode):
    """"split_bf16" (default dispatch), "split_f16" (as split_bf16 with the two-way fp16 split in the LDS-halo convs) or "f32"
    (every product on the fp32 MFMA); returns the previous mode's name."""
    lib_mode = {"split_bf16": 0, "split_f16": 0, "f32": 1}[mode]
    prev_name = mfma_mode()
    prev = _lib.lib().qea_set_mfma_mode(lib_mode)
    if prev < 0:
        _lib.check(prev, "qea_set_mfma_mode")
    SPLIT_F16["on"] = mode == "split_f16"
    return prev_name


def mfma_mode():
    m = ("split_bf16", "f32")[_lib.lib().qea_set_mfma_mode(-1)]
    return "split_f16" if (m == "split_bf16" and SPLIT_F16["on"]) else m


class AmaxPool:
    """Zero-filled 4-byte slots for producer-carried abs-max values: one allocation + one memset per forward / backward pass."""

    def __init__(self, device, n=192):
        self.buf = torch.zeros(n, device=device)
        self.used = 0

    def slot(self):
        if self.used == self.buf.numel():
            self.buf = torch.zeros(self.buf.numel(), device=self.buf.device)
            self.used = 0
        self.used += 1
        return self.buf[self.used - 1:self.used]


def amax_pool(device):
    """-> AmaxPool when the fp16 split is active (its launches want abs-max values), else None (producers then skip the atomics)"""
    return AmaxPool(device) if (SPLIT_F16["on"] and mfma_mode() == "split_f16") else None


def filter_absmax(w_src, w, ld, M, Cc):
    """The abs-max a filter's fp16 planes are scaled from.  A filter that lives in a model's flat parameter buffer (qea.params)
    takes the abs-max of the WHOLE buffer, computed once per weight update for all its filters (one launch instead of one per
    filter and derived form): any bound >= the filter's own max is a valid scale source, and the fp16 pair keeps all 22 bits of
    every element down to 2^-17 of the bound, far below the spread between a model's layers."""
    from .params import flat_state_of
    fs = flat_state_of(w_src[1], check=False) if w_src is not None else None
    if fs is None or fs.total % 4 or fs.data.data_ptr() > w_src[1].data_ptr() or w_src[1].data_ptr() >= fs.data.data_ptr() + 4 * fs.total:
        return absmax(w, ld, M, Cc)
    # keyed on the version counters of ALL the model's parameters (fs.vsum, refreshed by the engines' ensure_flat() walk before
    # every forward): a torch-side write to one re-homed parameter moves that parameter's _version, not the flat buffer's, and a
    # bound taken before e.g. load_state_dict would scale the new, larger filter to inf in its h plane
    # ... and on the model's own epoch (the fused Adam writes through raw pointers and bumps only that)
    return weight_cached("flat_absmax", fs.data, lambda: absmax(fs.data, fs.total, 1, fs.total), extra=(fs.vsum, fs.epoch))


def split_planes_f16(x, ld, M, Cc, w_src=None):
    """the two-plane fp16 row format of a FILTER (qea_split_planes_f16), scaled by the power of two of its abs-max"""
    L = _lib.lib()
    out = torch.empty(L.qea_split_planes_f16_bytes(M, Cc), dtype=torch.uint8, device=x.device)
    xmax = filter_absmax(w_src, x, ld, M, Cc)
    _lib.check(L.qea_split_planes_f16(_ptr(x), ld, M, Cc, _ptr(xmax), out.data_ptr(), _stream()), "qea_split_planes_f16")
    return out


def absmax(x, ld, M, Cc):
    """device scalar: the largest finite |x| of a strided [M][Cc] tensor (one pass; the scale source of the fp16 split)"""
    out = torch.empty(1, device=x.device)
    _lib.check(_lib.lib().qea_absmax(_ptr(x), ld, M, Cc, _ptr(out), _stream()), "qea_absmax")
    return out


_overlap = {"on": None}


def set_overlap(on):
    """Switch the wgrad side stream on/off at run time (bench.py times with it on and measures per-kernel
    durations with it off, because a launch's event-to-event time otherwise includes a co-running kernel)."""
    _overlap["on"] = bool(on)


def overlap_enabled():
    if _overlap["on"] is None:
        import os
        _overlap["on"] = os.environ.get("QEA_OVERLAP", "1") != "0"
    return _overlap["on"]


class SideStream:
    """Runs weight-gradient work beside the main backward chain.  The dgrad -> BN-backward -> dgrad chain is a
    strict dependency chain with HBM-bound links, while every wgrad only needs its layer's dy: launching the
    wgrads on a second HIP stream lets the MFMA-bound wgrad kernels fill the matrix pipes while the chain's
    HBM-bound kernels run.  Set QEA_OVERLAP=0 to keep everything on one stream."""

    def __init__(self, device):
        self.side = torch.cuda.Stream(device)

    @property
    def enabled(self):
        return overlap_enabled()

    def run(self, fn, *reads):
        """fn() is enqueued on the side stream after everything already enqueued on the current stream; `reads`
        are tensors it reads that were allocated on the current stream (kept from reuse until it has run)."""
        if not self.enabled:
            fn()
            return
        main = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(main)
        with torch.cuda.stream(self.side):
            self.side.wait_event(ev)
            fn()
        for t in reads:
            if t is not None:
                t.record_stream(self.side)

    def join(self):
        if self.enabled:
            ev = torch.cuda.Event()
            ev.record(self.side)
            torch.cuda.current_stream().wait_event(ev)


def prof_enable(klass, on=True):
    _lib.check(_lib.lib().qea_prof_enable(klass, int(on)), "qea_prof_enable")


def prof_reset():
    _lib.check(_lib.lib().qea_prof_reset(), "qea_prof_reset")


def prof_read(klass):
    ms, fl, by, n = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
    _lib.check(_lib.lib().qea_prof_read(klass, C.byref(ms), C.byref(fl), C.byref(by), C.byref(n)), "qea_prof_read")
    sp, sf = C.c_double(), C.c_double()
    _lib.check(_lib.lib().qea_prof_read_split_bf16(klass, C.byref(sp)), "qea_prof_read_split_bf16")
    _lib.check(_lib.lib().qea_prof_read_split_f16(klass, C.byref(sf)), "qea_prof_read_split_f16")
    return {"ms": ms.value, "flops": fl.value, "bytes": by.value, "launches": n.value, "flops_split_bf16": sp.value, "flops_split_f16": sf.value}


def prof_tag_halo_bf3(cin_chunk, cout_group, stats, f16=False):
    """QEA_PROF_TAG_HALO_BF3 of include/qea_hip.h (+ 5 for the two-way fp16 instantiation of the same kernel)"""
    return 24000 + (1000 if cin_chunk == 64 else 0) + cout_group + (500 if stats else 0) + (5 if f16 else 0)


def prof_read_tagged(klass, tag):
    ms, fl, by, n = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
    _lib.check(_lib.lib().qea_prof_read_tagged(klass, tag, C.byref(ms), C.byref(fl), C.byref(by), C.byref(n)), "qea_prof_read_tagged")
    return {"ms": ms.value, "flops": fl.value, "bytes": by.value, "launches": n.value}


# ----------------------------------------------------------------------------- BN / pool / misc
def _colws(M, C_, device):
    need = _lib.lib().qea_colreduce_workspace_bytes(M, C_)
    ws = workspace(need, device)
    return ws.data_ptr(), ws.numel()


def bn_train_stats(y, ldy, M, C_, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift, stat64=None):
    wp, wn = _colws(M, C_, y.device)
    _lib.check(_lib.lib().qea_bn_train_stats(_ptr(y), ldy, M, C_, _ptr(gamma), _ptr(beta), eps, momentum,
                                             _ptr(running_mean), _ptr(running_var), _ptr(mean), _ptr(invstd),
                                             _ptr(scale), _ptr(shift), _ptr(stat64), wp, wn, _stream()), "qea_bn_train_stats")


def bn_train_stats_from_partials(partials, blocks, M, C_, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift,
                                 stat64=None):
    _lib.check(_lib.lib().qea_bn_train_stats_from_partials(_ptr(partials), blocks, M, C_, _ptr(gamma), _ptr(beta), eps, momentum,
                                                           _ptr(running_mean), _ptr(running_var), _ptr(mean), _ptr(invstd),
                                                           _ptr(scale), _ptr(shift), _ptr(stat64), _stream()),
               "qea_bn_train_stats_from_partials")


def bn_eval_coeff(C_, gamma, beta, running_mean, running_var, eps, conv_bias, mean, invstd, scale, shift):
    _lib.check(_lib.lib().qea_bn_eval_coeff(C_, _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var), eps,
                                            _ptr(conv_bias), _ptr(mean), _ptr(invstd), _ptr(scale), _ptr(shift),
                                            _stream()), "qea_bn_eval_coeff")


def bn_apply(y, ldy, a, lda, M, C_, scale, shift, relu=True, amax=None):
    """amax (here and in bn_bwd / maxpool_fwd / maxpool_bwd / conv_igemm's y_amax): a zero-filled 1-element slot (AmaxPool) that receives
    the abs-max of what the launch stores — the scale source of the fp16-split launch that consumes the tensor."""
    _lib.check(_lib.lib().qea_bn_apply(_ptr(y), ldy, _ptr(a), lda, M, C_, _ptr(scale), _ptr(shift), int(relu), _ptr(amax), _stream()),
               "qea_bn_apply")


def bn_bwd(da, ldda, a, lda, y, ldy, M, C_, gamma, mean, invstd, training, dgamma, dbeta, dy, lddy, accumulate=False, stat64=None,
           relu_scale=None, relu_shift=None, amax=None, partials=None):
    """ReLU mask: pass the activation `a`, or a=None with the forward's scale/shift (mask recomputed from y, one tensor read less).
    partials = (tensor, blocks) from conv_igemm(bwd_stats=...): the reductions came with da's producer (no pass over da and y here)."""
    wp, wn = _colws(M, C_, da.device)
    if partials is not None:
        _lib.check(_lib.lib().qea_bn_bwd_from_partials(partials[0].data_ptr(), partials[1], _ptr(da), ldda, _ptr(relu_scale), _ptr(relu_shift), _ptr(y),
                                                       ldy, M, C_, _ptr(gamma), _ptr(mean), _ptr(invstd), _ptr(stat64), int(training), _ptr(dgamma),
                                                       _ptr(dbeta), int(accumulate), _ptr(dy), lddy, wp, wn, _ptr(amax), _stream()),
                   "qea_bn_bwd_from_partials")
        return
    _lib.check(_lib.lib().qea_bn_bwd(_ptr(da), ldda, _ptr(a), lda, _ptr(relu_scale), _ptr(relu_shift), _ptr(y), ldy, M, C_, _ptr(gamma), _ptr(mean),
                                     _ptr(invstd), _ptr(stat64), int(training), _ptr(dgamma), _ptr(dbeta), int(accumulate),
                                     _ptr(dy), lddy, wp, wn, _ptr(amax), _stream()), "qea_bn_bwd")


def bn_bwd_pool(da, ldda, dpool, lddp, kw, y, ldy, B, H, W, C_, gamma, mean, invstd, training, dgamma, dbeta, dy, lddy, accumulate=False,
                stat64=None, relu_scale=None, relu_shift=None, amax=None):
    """maxpool_bwd(accumulate into da) + bn_bwd in one (ABI v8): da = the skip path's gradient or None, dpool = the pooled tensor's gradient;
    winners and ReLU mask are recomputed from y (relu_scale / relu_shift required)."""
    wp, wn = _colws(B * H * W, C_, y.device)
    _lib.check(_lib.lib().qea_bn_bwd_pool(_ptr(da), ldda, _ptr(dpool), lddp, kw, _ptr(relu_scale), _ptr(relu_shift), _ptr(y), ldy, B, H, W, C_,
                                          _ptr(gamma), _ptr(mean), _ptr(invstd), _ptr(stat64), int(training), _ptr(dgamma), _ptr(dbeta),
                                          int(accumulate), _ptr(dy), lddy, wp, wn, _ptr(amax), _stream()), "qea_bn_bwd_pool")


def colsum(x, ldx, M, C_, out, accumulate=False):
    wp, wn = _colws(M, C_, x.device)
    _lib.check(_lib.lib().qea_colsum(_ptr(x), ldx, M, C_, _ptr(out), int(accumulate), wp, wn, _stream()), "qea_colsum")


def bn_apply_pool(y, ldy, a, lda, pooled, ldp, B, H, W, C_, scale, shift, kh, kw, relu=True, amax=None, pooled_amax=None):
    """bn_apply + maxpool_fwd in one pass over y (bit-identical to the two calls)"""
    _lib.check(_lib.lib().qea_bn_apply_pool(_ptr(y), ldy, _ptr(a), lda, _ptr(pooled), ldp, B, H, W, C_, _ptr(scale), _ptr(shift), int(relu), kh, kw,
                                            _ptr(amax), _ptr(pooled_amax), _stream()), "qea_bn_apply_pool")


def maxpool_fwd(x, ldx, y, ldy, B, H, W, C_, kh, kw, amax=None):
    _lib.check(_lib.lib().qea_maxpool_fwd(_ptr(x), ldx, _ptr(y), ldy, B, H, W, C_, kh, kw, _ptr(amax), _stream()), "qea_maxpool_fwd")


def conv_c1_pool_bwd(x, w, bias, dpool, lddp, dw, db, dx, B, H, W, Co, accumulate=False):
    """backward of conv1 -> ReLU -> max_pool2d(2, 2) from the pooled tensor's gradient and the 1-channel input (ABI v8): dpool is overwritten
    with its ReLU-masked values; dw / db (None: no parameter gradients), dx (None: not wanted)."""
    L = _lib.lib()
    need = L.qea_conv_c1_pool_bwd_workspace_bytes(B, H, W, Co, int(dx is not None))
    ws = torch.empty(max(need, 256), dtype=torch.uint8, device=x.device)
    _lib.check(L.qea_conv_c1_pool_bwd(_ptr(x), _ptr(w), _ptr(bias), _ptr(dpool), lddp, _ptr(dw), _ptr(db), _ptr(dx), B, H, W, Co, int(accumulate),
                                      ws.data_ptr(), ws.numel(), _stream()), "qea_conv_c1_pool_bwd")


def maxpool_bwd(x, ldx, dy, lddy, dx, lddx, B, H, W, C_, kh, kw, relu_mask=False, accumulate=False, amax=None):
    _lib.check(_lib.lib().qea_maxpool_bwd(_ptr(x), ldx, _ptr(dy), lddy, _ptr(dx), lddx, B, H, W, C_, kh, kw,
                                          int(relu_mask), int(accumulate), _ptr(amax), _stream()), "qea_maxpool_bwd")


def transpose2d(src, dst, R, Cc):
    _lib.check(_lib.lib().qea_transpose2d(_ptr(src), _ptr(dst), R, Cc, _stream()), "qea_transpose2d")


def filter_flip_transpose(w, wt, Co, Ci, KH, KW):
    _lib.check(_lib.lib().qea_filter_flip_transpose(_ptr(w), _ptr(wt), Co, Ci, KH, KW, _stream()), "qea_filter_flip_transpose")


# ----------------------------------------------------------------------------- single-channel convs / head
def conv_c1_fwd(x, w, bias, y, ldy, B, H, W, Co, relu=False):
    _lib.check(_lib.lib().qea_conv_c1_fwd(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), ldy, B, H, W, Co, int(relu), _stream()),
               "qea_conv_c1_fwd")


def conv_c1_fwd_pool(x, w, bias, y, ldy, pooled, ldp, B, H, W, Co, relu=False, pooled_amax=None):
    """conv_c1_fwd + maxpool_fwd(2, 2) in one pass (bit-identical); False when the shape is not taken (the caller then runs the two)"""
    if H % 2 or W % 4 or Co not in (32, 64, 128):
        return False
    _lib.check(_lib.lib().qea_conv_c1_fwd_pool(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), ldy, _ptr(pooled), ldp, B, H, W, Co, int(relu),
                                               _ptr(pooled_amax), _stream()), "qea_conv_c1_fwd_pool")
    return True


def conv_c1_wgrad(x, dy, lddy, dw, db, B, H, W, Co, accumulate=False):
    L = _lib.lib()
    ws = workspace(L.qea_conv_c1_wgrad_workspace_bytes(B, H, W, Co), x.device)
    _lib.check(L.qea_conv_c1_wgrad(_ptr(x), _ptr(dy), lddy, _ptr(dw), _ptr(db), B, H, W, Co, int(accumulate),
                                   ws.data_ptr(), ws.numel(), _stream()), "qea_conv_c1_wgrad")


def conv_c1_dgrad(dy, lddy, w, dx, B, H, W, Co, accumulate=False):
    _lib.check(_lib.lib().qea_conv_c1_dgrad(_ptr(dy), lddy, _ptr(w), _ptr(dx), B, H, W, Co, int(accumulate), _stream()),
               "qea_conv_c1_dgrad")


def head_fwd(x, ldx, w, b, y, M, C_):
    _lib.check(_lib.lib().qea_head_fwd(_ptr(x), ldx, _ptr(w), _ptr(b), _ptr(y), M, C_, _stream()), "qea_head_fwd")


def head_bwd(x, ldx, y, dyy, w, dx, lddx, dw, db, M, C_, accumulate=False):
    L = _lib.lib()
    ws = workspace(L.qea_head_bwd_workspace_bytes(M, C_), x.device)
    _lib.check(L.qea_head_bwd(_ptr(x), ldx, _ptr(y), _ptr(dyy), _ptr(w), _ptr(dx), lddx, _ptr(dw), _ptr(db),
                              int(accumulate), M, C_, ws.data_ptr(), ws.numel(), _stream()), "qea_head_bwd")


# ----------------------------------------------------------------------------- sequence ops
def lstm_pack_whh(w_hh, packed_fwd, packed_bwd):
    _lib.check(_lib.lib().qea_lstm_pack_whh(_ptr(w_hh), _ptr(packed_fwd), _ptr(packed_bwd), _stream()), "qea_lstm_pack_whh")


def lstm_layer_fwd(gates, c, y, packed_fwd, T, B):
    _lib.check(_lib.lib().qea_lstm_layer_fwd(_ptr(gates), _ptr(c), _ptr(y), _ptr(packed_fwd), T, B, _stream()), "qea_lstm_layer_fwd")


def lstm_layer_bwd(gates, c, dy, packed_bwd, dc_scratch, T, B):
    _lib.check(_lib.lib().qea_lstm_layer_bwd(_ptr(gates), _ptr(c), _ptr(dy), _ptr(packed_bwd), _ptr(dc_scratch), T, B,
                                             _stream()), "qea_lstm_layer_bwd")


def lstm_pack_whh_split(w_hh, planes_fwd, planes_bwd):
    _lib.check(_lib.lib().qea_lstm_pack_whh_split(_ptr(w_hh), _ptr(planes_fwd), _ptr(planes_bwd), _stream()), "qea_lstm_pack_whh_split")


LSTM_SEQ = {"on": os.environ.get("QEA_LSTM", "seq") != "step"}


def lstm_mode():
    """"seq": one launch per layer pass, W_hh resident in LDS as fp16 planes (csrc/lstm_seq.hip; the default fp16-split mode);
    "bf3": one launch per step, split-bf16 recurrent GEMMs (QEA_LSTM=step, or the split-bf16 MFMA mode); "f32": QEA_MFMA=f32."""
    m = mfma_mode()
    if m == "f32":
        return "f32"
    return "seq" if (m == "split_f16" and LSTM_SEQ["on"]) else "bf3"


def lstm_packs(whf, whr):
    """(fwd pack, bwd pack, mode) of one layer's two W_hh for the CURRENT mode (lstm_mode), cached with the weights: fp16 planes +
    the two abs-max values for the one-launch layer kernels, three-plane bf16 fragments for the split-bf16 step kernels, fp32
    fragments for QEA_MFMA=f32.  A "seq" pack is the tuple (planes [2][bytes], w_absmax [2])."""
    mode = lstm_mode()
    dev = whf.device

    def pack():
        if mode == "seq":
            nb = _lib.lib().qea_lstm_seq_pack_bytes()
            pf_ = torch.empty(2, nb, dtype=torch.uint8, device=dev)
            pb_ = torch.empty(2, nb, dtype=torch.uint8, device=dev)
            # the scale source of the planes: any bound >= max |W_hh| (the model's flat abs-max when the weight lives in one: no launch)
            bounds = [filter_absmax(("fwd", wh), wh, 256, 1024, 256) for wh in (whf, whr)]
            am = torch.cat([b.reshape(1) for b in bounds])
            for d_, wh in enumerate((whf, whr)):
                _lib.check(_lib.lib().qea_lstm_seq_pack(_ptr(wh), _ptr(pf_[d_]), _ptr(pb_[d_]), _ptr(am[d_:]), _stream()), "qea_lstm_seq_pack")
            return (pf_, am), (pb_, am)
        if mode == "bf3":
            nb = _lib.lib().qea_lstm_pack_whh_split_bytes()
            pf_ = torch.empty(2, nb, dtype=torch.uint8, device=dev)
            pb_ = torch.empty(2, nb, dtype=torch.uint8, device=dev)
            for d_, wh in enumerate((whf, whr)):
                lstm_pack_whh_split(wh, pf_[d_], pb_[d_])
        else:
            pf_ = torch.empty(2, whf.numel(), device=dev)
            pb_ = torch.empty(2, whf.numel(), device=dev)
            for d_, wh in enumerate((whf, whr)):
                lstm_pack_whh(wh, pf_[d_], pb_[d_])
        return pf_, pb_
    pf, pb = weight_cached(("whh_pack", mode), whf, pack, also=(whr,))
    return pf, pb, mode


def _lstm_seq_ws(B, dev):
    return torch.empty(_lib.lib().qea_lstm_seq_workspace_bytes(B), dtype=torch.uint8, device=dev)


def lstm_layer_fwd_any(gates, c, y, pack, mode, T, B, y_amax=None):
    """mode: "seq" / "bf3" / "f32" (True / False of the round-2 callers = "bf3" / "f32").  y_amax: a ZEROED abs-max slot; returns True
    when the pass has left the layer output's abs-max in it (the one-launch kernels), else the caller makes its own pass."""
    if os.environ.get("QEA_LSTM_TRACE"):
        print("lstm fwd", mode, T, B, flush=True)
    if mode == "seq":
        planes, am = pack
        ws = _lstm_seq_ws(B, gates.device)
        _lib.check(_lib.lib().qea_lstm_seq_fwd(_ptr(gates), _ptr(c), _ptr(y), _ptr(planes), _ptr(am), T, B, _ptr(ws), _ptr(y_amax), _stream()), "qea_lstm_seq_fwd")
        return y_amax is not None
    elif mode is True or mode == "bf3":
        _lib.check(_lib.lib().qea_lstm_layer_fwd_split(_ptr(gates), _ptr(c), _ptr(y), _ptr(pack), T, B, _stream()), "qea_lstm_layer_fwd_split")
    else:
        lstm_layer_fwd(gates, c, y, pack, T, B)


def lstm_layer_bwd_any(gates, c, dy, pack, mode, dc_scratch, T, B, g_amax=None):
    if mode == "seq":
        planes, am = pack
        ws = _lstm_seq_ws(B, gates.device)
        _lib.check(_lib.lib().qea_lstm_seq_bwd(_ptr(gates), _ptr(c), _ptr(dy), _ptr(planes), _ptr(am), T, B, _ptr(ws), _ptr(g_amax), _stream()), "qea_lstm_seq_bwd")
        return g_amax is not None
    elif mode is True or mode == "bf3":
        _lib.check(_lib.lib().qea_lstm_layer_bwd_split(_ptr(gates), _ptr(c), _ptr(dy), _ptr(pack), _ptr(dc_scratch), T, B, _stream()),
                   "qea_lstm_layer_bwd_split")
    else:
        lstm_layer_bwd(gates, c, dy, pack, dc_scratch, T, B)


def log_softmax_fwd(x, ldx, y, ldy, M, C_):
    _lib.check(_lib.lib().qea_log_softmax_fwd(_ptr(x), ldx, _ptr(y), ldy, M, C_, _stream()), "qea_log_softmax_fwd")


def log_softmax_bwd(g, ldg, lp, ldlp, dx, lddx, M, C_, Cpad, nan_scrub):
    _lib.check(_lib.lib().qea_log_softmax_bwd(_ptr(g), ldg, _ptr(lp), ldlp, _ptr(dx), lddx, M, C_, Cpad, int(nan_scrub),
                                              _stream()), "qea_log_softmax_bwd")


def ctc_loss(lp, ld_t, ld_n, targets, target_offsets, input_lengths, target_lengths, T, N, C_, blank, S_max, reduction,
             grad_scale, nll, loss, grad, gld_t, gld_n):
    L = _lib.lib()
    ws = workspace(L.qea_ctc_workspace_bytes(T, N, S_max), lp.device)
    _lib.check(L.qea_ctc_loss(_ptr(lp), ld_t, ld_n, _ptr(targets), _ptr(target_offsets), _ptr(input_lengths),
                              _ptr(target_lengths), T, N, C_, blank, S_max, reduction, grad_scale, _ptr(nll), _ptr(loss),
                              _ptr(grad), gld_t, gld_n, ws.data_ptr(), ws.numel(), _stream()), "qea_ctc_loss")


# ----------------------------------------------------------------------------- optimiser / jitter / selection / crops
def adam_step(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    _lib.check(_lib.lib().qea_adam_step(_ptr(p), _ptr(g), _ptr(m), _ptr(v), n, lr, beta1, beta2, eps, weight_decay, step,
                                        grad_scale, _stream()), "qea_adam_step")


def adam_step_capturable(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, coef, grad_scale=1.0):
    """step: 1-element CUDA float tensor (incremented on the device), coef: 2-element CUDA float scratch."""
    _lib.check(_lib.lib().qea_adam_step_capturable(_ptr(p), _ptr(g), _ptr(m), _ptr(v), n, lr, beta1, beta2, eps, weight_decay,
                                                   _ptr(step), _ptr(coef), grad_scale, _stream()), "qea_adam_step_capturable")


def jitter(img, sigma, out, noise_out, K, R, HW, coef, seed, offset):
    _lib.check(_lib.lib().qea_jitter(_ptr(img), _ptr(sigma), _ptr(out), _ptr(noise_out), K, R, HW, coef, seed, offset,
                                     _stream()), "qea_jitter")


def jitter_apply(img, noise, out, K, R, HW, coef):
    _lib.check(_lib.lib().qea_jitter_apply(_ptr(img), _ptr(noise), _ptr(out), K, R, HW, coef, _stream()), "qea_jitter_apply")


def topk_desc_stable(keys, n, k, idx_out):
    _lib.check(_lib.lib().qea_topk_desc_stable(_ptr(keys), n, k, _ptr(idx_out), _stream()), "qea_topk_desc_stable")


def crop_pad_gather(img, H, W, boxes, N, OH, OW, out):
    _lib.check(_lib.lib().qea_crop_pad_gather(_ptr(img), H, W, _ptr(boxes), N, OH, OW, _ptr(out), _stream()), "qea_crop_pad_gather")


def crop_pad_scatter(dout, boxes, N, OH, OW, dimg, H, W):
    _lib.check(_lib.lib().qea_crop_pad_scatter(_ptr(dout), _ptr(boxes), N, OH, OW, _ptr(dimg), H, W, _stream()), "qea_crop_pad_scatter")


def greedy_decode(scores, ld_t, ld_n, T, N, C_, blank, tokens, lengths):
    _lib.check(_lib.lib().qea_greedy_decode(_ptr(scores), ld_t, ld_n, T, N, C_, blank, _ptr(tokens), _ptr(lengths), _stream()),
               "qea_greedy_decode")


def prof_read_launches(klass, capacity=4096):
    import numpy as np
    ms = np.zeros(capacity)
    fl = np.zeros(capacity)
    n = C.c_int64()
    _lib.check(_lib.lib().qea_prof_read_launches(klass, ms.ctypes.data, fl.ctypes.data, capacity, C.byref(n)), "qea_prof_read_launches")
    k = min(n.value, capacity)
    return ms[:k], fl[:k]


def edit_distance(pred_tokens, ldp, pred_len, gt_tokens, gt_offsets, gt_len, N, out):
    _lib.check(_lib.lib().qea_edit_distance(_ptr(pred_tokens), ldp, _ptr(pred_len), _ptr(gt_tokens), _ptr(gt_offsets), _ptr(gt_len), N,
                                            _ptr(out), _stream()), "qea_edit_distance")
