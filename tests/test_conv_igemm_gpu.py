"""HIP implicit-GEMM conv (qea_conv_igemm) vs torch-CPU conv2d on every distinct conv shape
of the hot path, through the C ABI."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _run(B, H, W, Cin, Cout, KH, KW, pad, stride=1, relu=False, bias=False, tile=0, seed=0):
    from qea import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, KH, KW, generator=g) / (Cin * KH * KW) ** 0.5
    b = torch.randn(Cout, generator=g) if bias else None
    ref = F.conv2d(x.double(), w.double(), None if b is None else b.double(), stride=stride, padding=pad)
    if relu:
        ref = ref.relu()
    OH, OW = ref.shape[2], ref.shape[3]
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    wd = w.permute(0, 2, 3, 1).contiguous().cuda()
    bd = b.cuda() if bias else None
    y = torch.full((B, OH, OW, Cout), float("nan"), device="cuda")
    d = _lib.ConvDesc(
        x=xd.data_ptr(), w=wd.data_ptr(), y=y.data_ptr(), scale=None,
        bias=bd.data_ptr() if bias else None, mask=None,
        B=B, H=H, W=W, Cin=Cin, OH=OH, OW=OW, N=Cout, KH=KH, KW=KW, pad_h=pad, pad_w=pad,
        stride_h=stride, stride_w=stride, ldx=Cin, ldy=Cout, ldmask=0, relu=int(relu),
        accumulate=0, out_mode=0, tile=tile)
    _lib.check(L.qea_conv_igemm(C.byref(d), torch.cuda.current_stream().cuda_stream), "qea_conv_igemm")
    torch.cuda.synchronize()
    got = y.cpu().permute(0, 3, 1, 2).double()
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= 2e-5 * max(scale, 1.0), (err, scale)


SHAPES = [
    # (H, W, Cin, Cout)  UNet levels and CRNN backbone
    (32, 128, 32, 32), (32, 128, 64, 32), (16, 64, 32, 64), (16, 64, 64, 64), (16, 64, 128, 64),
    (8, 32, 64, 128), (8, 32, 128, 128), (8, 32, 256, 128), (4, 16, 128, 256), (4, 16, 256, 256),
    (4, 16, 512, 256), (2, 8, 256, 512), (2, 8, 512, 512),
    (16, 64, 64, 128), (8, 32, 128, 256), (8, 32, 256, 256), (4, 32, 256, 512), (4, 32, 512, 512),
]


@pytest.mark.parametrize("H,W,Cin,Cout", SHAPES)
def test_conv3x3_shapes(H, W, Cin, Cout):
    _run(3, H, W, Cin, Cout, 3, 3, 1)


@pytest.mark.parametrize("tile", [1, 2, 3, 5, 6, 7, 8, 9, 20, 21, 22, 23, 25])
def test_conv3x3_all_tiles_ragged(tile):
    # M = 5*7*9 = 315 and N = 40 are not multiples of any tile edge
    _run(5, 7, 9, 64, 40, 3, 3, 1, relu=True, bias=True, tile=tile)


def test_conv7_2x2_pad0():
    _run(4, 2, 32, 512, 512, 2, 2, 0, bias=True)


def test_conv_2x2_stride2():
    _run(2, 8, 16, 64, 128, 2, 2, 0, stride=2)


def test_gemm_1x1():
    _run(1, 1, 300, 512, 95, 1, 1, 0, bias=True)


@pytest.mark.parametrize("Cin,Cout,H,W", [(32, 32, 8, 32), (32, 64, 16, 64), (64, 32, 4, 32), (64, 64, 12, 96)])
def test_conv3x3_halo_kernel_strided_buffers(Cin, Cout, H, W):
    """tile 4 = LDS-halo kernel of the narrow layers, reading from / writing into wider (concat) buffers."""
    from qea import ops
    B = 3
    g = torch.Generator().manual_seed(Cin + Cout)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5
    ref = F.conv2d(x.double(), w.double(), padding=1)
    ldx, ldy = Cin + 32, Cout + 64
    xb = torch.full((B, H, W, ldx), 7.0, device="cuda")
    xb[..., 32:] = x.permute(0, 2, 3, 1).cuda()
    yb = torch.full((B, H, W, ldy), -3.0, device="cuda")
    ops.conv_igemm(xb[..., 32:], w.permute(0, 2, 3, 1).contiguous().cuda(), yb[..., 64:], B=B, H=H, W=W, Cin=Cin, OH=H, OW=W, N=Cout,
                   KH=3, KW=3, pad=(1, 1), ldx=ldx, ldy=ldy, tile=4)
    torch.cuda.synchronize()
    got = yb[..., 64:].cpu().permute(0, 3, 1, 2).double()
    assert (got - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    assert (yb[..., :64] == -3.0).all()


def test_split_bf16_is_not_reduced_precision():
    """The split-bf16 kernels (tiles 20-22, x = h+m+l, six bf16 MFMAs per product, fp32 accumulate) against the native
    fp32-MFMA kernels, both measured against fp64: on a long reduction (forward, K = 9*512) the split form must be at least as
    close (it rounds the accumulator 16 products at a time instead of 2); on a short one (wgrad over 512 pixels) its floor is
    the 3-plane representation error of about one fp32 rounding unit (2^-23).  Wide-dynamic-range operands (1e-3 ... 1e3)
    included, since the split is relative to each element."""
    ULP = 2.0 ** -23
    from qea import ops
    g = torch.Generator().manual_seed(7)
    B, H, W, Ci, Co = 4, 4, 32, 512, 512
    for spread in (1.0, 3.0):
        x = torch.randn(B, H, W, Ci, generator=g) * torch.exp(spread * torch.randn(B, H, W, Ci, generator=g))
        w = torch.randn(Co, 3, 3, Ci, generator=g) * 0.02 * torch.exp(spread * torch.randn(Co, 3, 3, Ci, generator=g))
        ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), padding=1).permute(0, 2, 3, 1)
        errs = {}
        for tile in (8, 22):
            y = torch.empty(B, H, W, Co, device="cuda")
            ops.conv_igemm(x.cuda(), w.cuda(), y, B=B, H=H, W=W, Cin=Ci, OH=H, OW=W, N=Co, KH=3, KW=3, pad=(1, 1), ldx=Ci, ldy=Co, tile=tile)
            errs[tile] = ((y.cpu().double() - ref).norm() / ref.norm()).item()
        assert errs[22] <= max(1.25 * errs[8], 2 * ULP), errs
        dy = torch.randn(B, H, W, Co, generator=g) * torch.exp(spread * torch.randn(B, H, W, Co, generator=g))
        refw = torch.einsum("bhwo,bhwkc->okc", dy.double(),
                            F.unfold(x.permute(0, 3, 1, 2).double(), 3, padding=1).view(B, Ci, 9, H, W).permute(0, 3, 4, 2, 1))
        errs = {}
        for tile in (9, 20):
            dw = torch.empty(Co, 3, 3, Ci, device="cuda")
            ops.conv_wgrad(dy.cuda(), x.cuda(), dw, B=B, PH=H, PW=W, QH=H, QW=W, R=Co, Cc=Ci, KH=3, KW=3, pad=(1, 1), ldp=Co, ldq=Ci, tile=tile)
            errs[tile] = ((dw.cpu().double().view(Co, 9, Ci) - refw).norm() / refw.norm()).item()
        assert errs[20] <= max(1.25 * errs[9], 2 * ULP), errs


@pytest.mark.parametrize("tile", [20, 21, 22, 23, 25])
@pytest.mark.parametrize("shape", [(5, 7, 9, 64, 40, 3, 1), (3, 8, 32, 128, 256, 3, 1), (2, 4, 32, 512, 512, 3, 1), (4, 2, 32, 512, 512, 2, 0),
                                   (1, 1, 300, 512, 95, 1, 0), (2, 8, 16, 64, 128, 2, 0)])
def test_presplit_planes_bit_identical(tile, shape):
    """conv_igemm_p3_kernel (operands pre-split into bf16 planes once, staged by LDS-DMA) runs the very MFMA sequence of
    conv_igemm_bf3_kernel (operands split on the fly) on the very same bf16 values: the outputs must agree BIT FOR BIT on
    every tile, ragged M / N, zero padding, 2x2 stride-2 and 1x1 GEMM shapes included."""
    from qea import ops
    B, H, W, Cin, Cout, k, pad = shape
    stride = 2 if (k == 2 and pad == 0 and H == 8) else 1
    g = torch.Generator().manual_seed(tile * 100 + Cin)
    x = (torch.randn(B, H, W, Cin, generator=g) * torch.exp(2 * torch.randn(B, H, W, Cin, generator=g))).cuda()
    w = (torch.randn(Cout, k, k, Cin, generator=g) * 0.05).cuda()
    bias = torch.randn(Cout, generator=g).cuda()
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    outs = []
    prev_mode = ops.set_mfma_mode("split_bf16")                          # (the statement is about the three-way bf16 forms)
    for pre, prex in ((False, False), (True, False), (True, True)):       # split on the fly / filter planes (hybrid) / both operands
        ops.PRESPLIT["on"], ops.PRESPLIT["x"] = pre, prex
        try:
            y = torch.full((B, OH, OW, Cout), float("nan"), device="cuda")
            ops.conv_igemm(x, w, y, B=B, H=H, W=W, Cin=Cin, OH=OH, OW=OW, N=Cout, KH=k, KW=k, pad=(pad, pad), stride=(stride, stride),
                           ldx=Cin, ldy=Cout, bias=bias, relu=True, tile=tile)
            outs.append(y)
        finally:
            ops.PRESPLIT["on"], ops.PRESPLIT["x"] = True, False
    ops.set_mfma_mode(prev_mode)
    torch.cuda.synchronize()
    assert torch.isfinite(outs[1]).all() and torch.isfinite(outs[2]).all()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    # the two-way fp16 form of the hybrid tile (default mode): same values to fp32 rounding, non-finite free
    y = torch.full((B, OH, OW, Cout), float("nan"), device="cuda")
    ops.conv_igemm(x, w, y, B=B, H=H, W=W, Cin=Cin, OH=OH, OW=OW, N=Cout, KH=k, KW=k, pad=(pad, pad), stride=(stride, stride),
                   ldx=Cin, ldy=Cout, bias=bias, relu=True, tile=tile)
    assert torch.isfinite(y).all()
    assert (y - outs[0]).abs().max().item() <= 2e-5 * max(1.0, outs[0].abs().max().item())


def test_presplit_from_a_strided_concat_slice():
    """x is the right half of a wider (UNet concat) buffer: the split pass reads with the pixel stride, the planes are compact."""
    from qea import ops
    B, H, W, Cin, Cout = 3, 8, 32, 128, 128
    g = torch.Generator().manual_seed(5)
    cat = torch.randn(B, H, W, 2 * Cin, generator=g).cuda()
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) * 0.05).cuda()
    x = cat[..., Cin:]
    outs = []
    for pre in (False, True):
        ops.PRESPLIT["on"], ops.PRESPLIT["x"] = pre, pre
        try:
            y = torch.empty(B, H, W, Cout, device="cuda")
            ops.conv_igemm(x, w, y, B=B, H=H, W=W, Cin=Cin, OH=H, OW=W, N=Cout, KH=3, KW=3, pad=(1, 1), ldx=2 * Cin, ldy=Cout, tile=21)
            outs.append(y)
        finally:
            ops.PRESPLIT["on"], ops.PRESPLIT["x"] = True, False
    assert torch.equal(outs[0], outs[1])
    ref = F.conv2d(x.permute(0, 3, 1, 2).double().cpu(), w.permute(0, 3, 1, 2).double().cpu(), padding=1).permute(0, 2, 3, 1)
    assert (outs[1].cpu().double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()


def test_non_finite_operand_stays_confined_and_visible():
    """qea_split3 turns an infinite operand into (h = inf, m = l = NaN), so a split-bf16 tile yields NaN
    where v_mfma_f32_32x32x2_f32 yields +-inf.  Documented difference: both modes mark exactly the output
    pixels whose receptive field holds the non-finite input as non-finite, and every other output is bit-identical to the
    clean run — a diverged activation can neither hide nor spread."""
    from qea import ops
    B, H, W, Cin, Cout = 2, 8, 32, 128, 128
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, H, W, Cin, generator=g).cuda()
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) * 0.05).cuda()
    xb = x.clone()
    xb[1, 4, 10, 7] = float("inf")
    hit = torch.zeros(B, H, W, dtype=torch.bool)
    hit[1, 3:6, 9:12] = True
    for mode in ("split_f16", "split_bf16", "f32"):
        prev = ops.set_mfma_mode(mode)
        try:
            ys = []
            for inp in (x, xb):
                y = torch.empty(B, H, W, Cout, device="cuda")
                ops.conv_igemm(inp, w, y, B=B, H=H, W=W, Cin=Cin, OH=H, OW=W, N=Cout, KH=3, KW=3, pad=(1, 1), ldx=Cin, ldy=Cout)
                ys.append(y.cpu())
        finally:
            ops.set_mfma_mode(prev)
        clean, bad = ys
        assert torch.isfinite(clean).all()
        assert (~torch.isfinite(bad[hit])).all(), mode
        assert torch.equal(bad[~hit], clean[~hit]), mode


@pytest.mark.parametrize("shape", [(3, 8, 32, 128, 128), (2, 4, 32, 256, 512), (5, 16, 64, 64, 64), (3, 32, 128, 32, 32), (2, 16, 64, 32, 64),
                                   (7, 4, 16, 256, 256), (3, 2, 8, 512, 512), (2, 16, 64, 128, 64), (40, 32, 128, 32, 32), (24, 16, 64, 64, 320)])
def test_fused_bn_statistics_epilogue(shape):
    """batch statistics of the BatchNorm that follows a conv (models/model_unet.py:78-109) as per-block fp64
    column sums from the conv's own epilogue.  The conv output is bit-identical with and without the epilogue, and mean /
    invstd / scale / shift / running statistics equal those of the separate pass over y (same fp64 sums up to their order)."""
    from qea import ops
    B, H, W, Cin, Cout = shape
    g = torch.Generator().manual_seed(H + Cin)
    x = torch.randn(B, H, W, Cin, generator=g).cuda()
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) * 0.05).cuda()
    gamma, beta = (1 + 0.1 * torch.randn(Cout, generator=g)).cuda(), (0.1 * torch.randn(Cout, generator=g)).cuda()
    M = B * H * W
    res = {}
    for fuse in (False, True):
        ops.FUSE_BN_STATS["on"] = fuse
        try:
            y = torch.empty(M, Cout, device="cuda")
            got = ops.conv_igemm(x, w, y, B=B, H=H, W=W, Cin=Cin, OH=H, OW=W, N=Cout, KH=3, KW=3, pad=(1, 1), ldx=Cin, ldy=Cout, want_stats=True)
        finally:
            ops.FUSE_BN_STATS["on"] = True
        coef = torch.empty(4, Cout, device="cuda")
        rm, rv = torch.zeros(Cout, device="cuda"), torch.ones(Cout, device="cuda")
        st = torch.empty(2, Cout, device="cuda", dtype=torch.float64)
        if fuse:
            assert got is not None, "every shape here runs on a kernel with the statistics epilogue"
            ops.bn_train_stats_from_partials(got[0], got[1], M, Cout, gamma, beta, 1e-5, 0.1, rm, rv, coef[0], coef[1], coef[2], coef[3], st)
        else:
            assert got is None
            ops.bn_train_stats(y, Cout, M, Cout, gamma, beta, 1e-5, 0.1, rm, rv, coef[0], coef[1], coef[2], coef[3], st)
        res[fuse] = (y, coef, rm, rv, st)
    torch.cuda.synchronize()
    assert torch.equal(res[False][0], res[True][0])
    yd = res[True][0].double()
    assert (res[True][4][0] - yd.mean(0)).abs().max().item() <= 1e-12 * max(1.0, yd.abs().max().item())
    for a, b in zip(res[False][1:], res[True][1:]):
        assert (a.double() - b.double()).abs().max().item() <= 2e-7 * max(1.0, a.double().abs().max().item())


@pytest.mark.parametrize("Cin,Cout,H,W,B", [(32, 32, 8, 32, 3), (32, 64, 16, 64, 2), (64, 32, 4, 32, 5), (64, 64, 12, 96, 2), (32, 32, 32, 128, 2),
                                            (64, 64, 16, 64, 3), (128, 64, 16, 64, 2), (64, 128, 8, 32, 3), (128, 128, 8, 32, 2), (256, 128, 4, 32, 3),
                                            (32, 128, 8, 64, 2), (192, 32, 4, 32, 2), (256, 256, 8, 32, 2), (512, 512, 4, 32, 2), (128, 384, 4, 32, 3),
                                            # small-image tiles: 4x16 images two per tile, 2x8 images eight per tile; image counts that
                                            # leave the last tile partly empty (UNet levels 4 and 5, models/model_unet.py:19-29)
                                            (128, 256, 4, 16, 3), (256, 256, 4, 16, 4), (512, 256, 4, 16, 5), (64, 128, 4, 16, 2),
                                            (256, 512, 2, 8, 8), (512, 512, 2, 8, 11), (64, 128, 2, 8, 3), (128, 128, 2, 8, 17)])
def test_narrow_split_bf16_halo_kernel(Cin, Cout, H, W, B):
    """tile 24 = conv3x3_halo_bf3_kernel (input halo split ONCE into bf16 planes in LDS, filter in fragment-order planes): against
    fp64, against the fp32 LDS-halo kernel (tile 4), reading from / writing into wider (concat) buffers, with the fused
    BatchNorm-statistics epilogue and with the scale / bias / ReLU epilogue."""
    from qea import ops
    g = torch.Generator().manual_seed(Cin * 7 + Cout + H)
    x = torch.randn(B, Cin, H, W, generator=g) * torch.exp(torch.randn(B, Cin, H, W, generator=g))
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5
    ref = F.conv2d(x.double(), w.double(), padding=1)
    ldx, ldy = Cin + 32, Cout + 64
    xb = torch.full((B, H, W, ldx), 7.0, device="cuda")
    xb[..., 32:] = x.permute(0, 2, 3, 1).cuda()
    wd = w.permute(0, 2, 3, 1).contiguous().cuda()
    outs = {}
    narrow = Cin <= 64 and Cout <= 64
    for tile in ((4, 24) if narrow else (21, 24)):          # reference kernel: fp32 LDS-halo (narrow) / hybrid split tile
        yb = torch.full((B, H, W, ldy), -3.0, device="cuda")
        got = ops.conv_igemm(xb[..., 32:], wd, yb[..., 64:], B=B, H=H, W=W, Cin=Cin, OH=H, OW=W, N=Cout, KH=3, KW=3, pad=(1, 1), ldx=ldx, ldy=ldy,
                             tile=tile, want_stats=True)
        torch.cuda.synchronize()
        assert (yb[..., :64] == -3.0).all()
        y = yb[..., 64:].cpu().permute(0, 3, 1, 2).double()
        assert (y - ref).abs().max().item() <= 2e-5 * ref.abs().max().item(), tile
        if got is None:
            assert tile == 21 and B * H * W < 256                   # (a hybrid tile too large for this M has no statistics blocks)
            outs[tile] = y
            continue
        part = got[0][:got[1]].sum(0).cpu()                          # [Cout][2] (the last 256 rows are scratch)
        flat = yb[..., 64:].reshape(-1, Cout).double().cpu()
        assert (part[:, 0] - flat.sum(0)).abs().max().item() <= 1e-9 * flat.abs().sum(0).max().item()
        assert (part[:, 1] - (flat * flat).sum(0)).abs().max().item() <= 1e-9 * (flat * flat).sum(0).max().item()
        outs[tile] = y
    assert (outs[24] - outs[4 if narrow else 21]).abs().max().item() <= 2e-5 * ref.abs().max().item()
    # auto dispatch takes the split kernel; scale + bias + ReLU epilogue
    sc, bi = torch.rand(Cout, generator=g).cuda() + 0.5, torch.randn(Cout, generator=g).cuda()
    y2 = torch.empty(B, H, W, Cout, device="cuda")
    ops.conv_igemm(xb[..., 32:], wd, y2, B=B, H=H, W=W, Cin=Cin, OH=H, OW=W, N=Cout, KH=3, KW=3, pad=(1, 1), ldx=ldx, ldy=Cout, scale=sc, bias=bi,
                   relu=True)
    want = (ref.permute(0, 2, 3, 1) * sc.cpu().double() + bi.cpu().double()).clamp_min(0)
    assert (y2.cpu().double() - want).abs().max().item() <= 3e-5 * want.abs().max().item()
    # ReLU-mask epilogue (the input gradient through a bare ReLU: CRNN conv4's dgrad, models/model_crnn.py:50-51)
    mk = torch.randn(B, H, W, Cout + 32, generator=g).cuda()
    y3 = torch.empty(B, H, W, Cout, device="cuda")
    ops.conv_igemm(xb[..., 32:], wd, y3, B=B, H=H, W=W, Cin=Cin, OH=H, OW=W, N=Cout, KH=3, KW=3, pad=(1, 1), ldx=ldx, ldy=Cout, mask=mk[..., 32:],
                   ldmask=Cout + 32)
    want3 = ref.permute(0, 2, 3, 1) * (mk[..., 32:].cpu() > 0)
    assert (y3.cpu().double() - want3).abs().max().item() <= 2e-5 * ref.abs().max().item()


@pytest.mark.parametrize("M,K,N,mode", [(1000, 64, 128, "nhwc"), (128 * 5, 512, 1024, "nhwc"), (777, 2048, 512, "tbc"), (4 * 6 * 10, 128, 256, "convt"),
                                       (3 * 2 * 8, 512, 1024, "convt"), (16 * 16 * 64, 64, 128, "convt"), (130, 192, 384, "nhwc")])
def test_gemm1x1_lds_tile(M, K, N, mode):
    """tile 26 (gemm1x1.hip): 1x1 GEMM / transposed-conv forward on the 128-row LDS tile, two-way fp16 split, against fp64 — plain rows,
    the (t,b) -> (b,t) row transpose (QEA_OUT_TBC) and the 2x2 stride-2 scatter into one half of a concat buffer (QEA_OUT_CONVT,
    /root/reference/models/model_unet.py:61-73), rows past M in the last tile, bias + ReLU, strided input rows"""
    from qea import _lib, ops
    prev = ops.set_mfma_mode("split_f16")
    try:
        g = torch.Generator().manual_seed(M + K)
        ldx = K + 64
        xfull = torch.randn(M, ldx, generator=g)
        x = xfull[:, :K]
        w = torch.randn(N, K, generator=g) / K ** 0.5
        relu = mode == "nhwc"
        xd, wd = xfull.cuda(), w.cuda()
        amax = torch.zeros(1, device="cuda")
        if mode == "convt":
            c = N // 4
            bias = torch.randn(c, generator=g)
            B, h, wd_ = {4 * 6 * 10: (4, 6, 10), 3 * 2 * 8: (3, 2, 8), 16 * 16 * 64: (16, 16, 64)}[M]
            ref = (x.double() @ w.double().t()).view(B, h, wd_, 2, 2, c) + bias.double()
            ref = ref.permute(0, 1, 3, 2, 4, 5).reshape(B, 2 * h, 2 * wd_, c)
            y = torch.full((B, 2 * h, 2 * wd_, 2 * c), float("nan"), device="cuda")
            d = dict(B=B, H=h, W=wd_, OH=h, OW=wd_, ldy=2 * c, out_mode=ops.OUT_CONVT)
            pick = lambda t: t[..., :c]
        else:
            bias = torch.randn(N, generator=g)
            ref = x.double() @ w.double().t() + bias.double()
            if relu:
                ref = ref.relu()
            y = torch.full((M, N + 32), float("nan"), device="cuda")
            d = dict(B=1, H=1, W=M, OH=1, OW=M, ldy=N + 32, out_mode=ops.OUT_NHWC)
            pick = lambda t: t[:, :N]
            if mode == "tbc":                                   # M = T * Bt rows (t, b) -> output row b * T + t
                T, Bt = 7, M // 7
                ref = ref.view(T, Bt, N).permute(1, 0, 2).reshape(M, N)
                d = dict(B=T, H=1, W=Bt, OH=1, OW=Bt, ldy=N + 32, out_mode=ops.OUT_TBC)
        desc = _lib.ConvDesc(x=None, w=None, y=None, scale=None, bias=None, mask=None, B=d["B"], H=d["H"], W=d["W"], Cin=K, OH=d["OH"], OW=d["OW"], N=N,
                             KH=1, KW=1, pad_h=0, pad_w=0, stride_h=1, stride_w=1, ldx=ldx, ldy=d["ldy"], ldmask=0, relu=int(relu), accumulate=0,
                             out_mode=d["out_mode"], tile=0, x_planes=None, w_planes=None, stats=None, w_frag_planes=None)
        assert _lib.lib().qea_conv_igemm_wants_frag_planes(C.byref(desc)) == 2          # the automatic choice is tile 26
        ops.prof_enable(ops.PROF_CONV_IGEMM, True)
        ops.prof_reset()
        ops.conv_igemm(xd, wd, y, Cin=K, N=N, KH=1, KW=1, ldx=ldx, bias=bias.cuda(), relu=relu, y_amax=amax, **d)
        torch.cuda.synchronize()
        assert ops.prof_read_tagged(ops.PROF_CONV_IGEMM, 26)["launches"] == 1
        ops.prof_enable(ops.PROF_CONV_IGEMM, False)
        got = pick(y).cpu().double()
        assert torch.isnan(y[..., (N // 4 if mode == "convt" else N):]).all()           # nothing written outside the output columns
        scale = ref.abs().max().item()
        err = (got - ref).abs().max().item()
        assert err <= 2e-6 * max(scale, 1.0), (err, scale)                                 # fp32-class: the split drops < 2^-22 per product
        assert abs(amax.item() - got.abs().max().item()) <= 1e-6 * scale
    finally:
        ops.set_mfma_mode(prev)


@pytest.mark.parametrize("ratio", [1.0, 2.0 ** 10, 2.0 ** 16])
def test_filter_scale_from_the_whole_parameter_buffer(ratio):
    """The fp16 planes of every filter of a model are scaled from ONE number, the abs-max of the model's flat parameter buffer
    (ops.filter_absmax).  A filter whose own maximum is `ratio` times smaller than that bound must keep fp32-class accuracy: the
    h + l pair carries 22 bits down to 2^-17 of the bound.  3x3 layer (tile 24) and 1x1 GEMM (tile 26), against fp64."""
    from qea import ops
    from qea.params import FlatState, flat_state_of

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            g = torch.Generator().manual_seed(5)
            self.w3 = torch.nn.Parameter((torch.randn(128, 64, 3, 3, generator=g) / 24.0).to(memory_format=torch.channels_last))
            self.w1 = torch.nn.Parameter(torch.randn(256, 128, generator=g) / 11.0)
            self.big = torch.nn.Parameter(torch.full((4,), 1.0))

    prev = ops.set_mfma_mode("split_f16")
    try:
        m = M().cuda()
        with torch.no_grad():
            m.big.fill_(float(m.w3.abs().max()) * ratio)
        fs = FlatState(m)
        ops.bump_weight_epoch()
        assert flat_state_of(m.w3) is fs
        bound = ops.filter_absmax(("fwd", m.w3), m.w3, 9 * 64, 128, 9 * 64)
        assert abs(bound.item() - fs.data.abs().max().item()) == 0.0                  # the bound is the buffer's abs-max, not the filter's
        g = torch.Generator().manual_seed(6)
        x = torch.randn(3, 8, 32, 64, generator=g)
        ref = F.conv2d(x.permute(0, 3, 1, 2).double(), m.w3.detach().cpu().double(), padding=1).permute(0, 2, 3, 1)
        y = torch.empty(3, 8, 32, 128, device="cuda")
        ops.conv_igemm(x.cuda(), m.w3, y, B=3, H=8, W=32, Cin=64, OH=8, OW=32, N=128, KH=3, KW=3, pad=(1, 1), ldx=64, ldy=128, w_src=("fwd", m.w3))
        err = (y.cpu().double() - ref).abs().max().item()
        assert err <= 2e-6 * ref.abs().max().item(), (ratio, err, ref.abs().max().item())
        x1 = torch.randn(500, 128, generator=g)
        ref1 = x1.double() @ m.w1.detach().cpu().double().t()
        y1 = torch.empty(500, 256, device="cuda")
        ops.conv_igemm(x1.cuda(), m.w1, y1, B=1, H=1, W=500, Cin=128, OH=1, OW=500, N=256, KH=1, KW=1, ldx=128, ldy=256, w_src=("fwd", m.w1))
        err1 = (y1.cpu().double() - ref1).abs().max().item()
        assert err1 <= 2e-6 * ref1.abs().max().item(), (ratio, err1)
    finally:
        ops.set_mfma_mode(prev)


@pytest.mark.parametrize("Cin,Cout,H,W,B,kw", [(64, 128, 16, 64, 3, 2), (256, 256, 8, 32, 2, 1), (32, 32, 32, 128, 2, 2), (64, 64, 16, 64, 2, 2),
                                               (256, 256, 4, 16, 5, 2), (128, 128, 8, 32, 3, 2)])
def test_max_pool_in_the_conv_epilogue_is_bit_identical(Cin, Cout, H, W, B, kw):
    """qea_conv_desc.pool_y (ABI v7): the 2 x kw max-pool behind relu(conv + bias) written by the LDS-halo kernel's epilogue — y and
    pooled and both abs-max slots against the same launch without pool_y followed by qea_maxpool_fwd (bit for bit; NaN in the input;
    the small-image tile with a partly empty last tile: B = 5 images of 4x16)"""
    from qea import ops
    prev = ops.set_mfma_mode("split_f16")
    try:
        g = torch.Generator().manual_seed(Cin + Cout + H)
        x = torch.randn(B, H, W, Cin, generator=g).cuda()
        x[0, 1, 3, 5] = float("nan")
        w = (torch.randn(Cout, 3, 3, Cin, generator=g) / (9 * Cin) ** 0.5).cuda()
        bias = torch.randn(Cout, generator=g).cuda()
        assert ops.conv_can_pool(B=B, H=H, W=W, Cin=Cin, N=Cout, kw=kw, ldx=Cin, ldy=Cout)
        y0 = torch.empty(B, H, W, Cout, device="cuda")
        y1 = torch.empty_like(y0)
        p0 = torch.empty(B, H // 2, W // kw, Cout, device="cuda")
        p1 = torch.full_like(p0, -5.0)
        am = torch.zeros(4, device="cuda")
        kws = dict(B=B, H=H, W=W, Cin=Cin, OH=H, OW=W, N=Cout, KH=3, KW=3, pad=(1, 1), ldx=Cin, ldy=Cout, bias=bias, relu=True)
        ops.conv_igemm(x, w, y0, y_amax=am[0:1], **kws)
        ops.maxpool_fwd(y0, Cout, p0, Cout, B, H, W, Cout, 2, kw, amax=am[1:2])
        ops.conv_igemm(x, w, y1, y_amax=am[2:3], pool=(p1, Cout, kw, am[3:4]), **kws)
        torch.cuda.synchronize()
        assert torch.equal(y0.view(torch.int32), y1.view(torch.int32))
        assert torch.equal(p0.view(torch.int32), p1.view(torch.int32))
        assert am[0].item() == am[2].item() and am[1].item() == am[3].item()
        assert not ops.conv_can_pool(B=B, H=H, W=W, Cin=Cin, N=Cout, kw=3, ldx=Cin, ldy=Cout)
    finally:
        ops.set_mfma_mode(prev)
