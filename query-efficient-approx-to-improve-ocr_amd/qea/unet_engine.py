"""UNet cleaner on the HIP library: explicit forward / backward kernel schedules.

Mirrors reference models/model_unet.py:49-76 (forward) and its autograd.  Everything between
the [B,1,H,W] input and the [B,1,H,W] sigmoid output stays NHWC in HBM:

  * every conv3x3 of a `_block` (model_unet.py:78-109) is one implicit-GEMM MFMA launch; its
    BatchNorm is a statistics pass (fp64 sums) + one apply(+ReLU) pass;
  * a decoder level's torch.cat((up, skip), 1) (model_unet.py:63-74) never copies: the encoder's
    second BN/ReLU writes its output into channels [c, 2c) of the level's concat buffer and the
    transposed conv scatters into channels [0, c);
  * the backward walks the same graph in reverse with dgrad = the same implicit GEMM on
    flipped/transposed filters, wgrad = the pixel-reduction MFMA kernel, accumulating straight
    into the flat gradient buffer (qea/params.py).
"""
import torch

from . import ops
from .params import ensure_flat

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
FUSE_EVAL_BN = True   # tests flip this to compare the fused inference epilogue with the two-pass form
FUSE_BN_BWD_SUMS = True   # tests flip this: BatchNorm1's backward reductions from the producing dgrad's epilogue (qea_conv_desc.bst_y)
FUSE_BN_POOL = True   # tests flip this: the encoder's BatchNorm apply + ReLU and its 2x2 max-pool in one pass (qea_bn_apply_pool)
FUSE_POOL_BWD = True  # tests flip this: the encoder's pool backward inside the BatchNorm2 backward (qea_bn_bwd_pool), no pass of its own


class _Block:
    """Names + channel counts of one conv-BN-ReLU x2 block."""

    def __init__(self, mod, name, cin, cout):
        self.mod, self.name, self.cin, self.cout = mod, name, cin, cout

    def key(self, i, leaf):
        kind = "conv" if leaf == "w" else "norm"
        suffix = {"w": "weight", "gamma": "weight", "beta": "bias", "rm": "running_mean", "rv": "running_var"}[leaf]
        return f"{self.mod}.{self.name}{kind}{i}.{suffix}"


class UNetEngine:
    def __init__(self, module, features=32):
        self.m = module
        f = features
        self.f = f
        self.enc = [_Block("encoder1", "enc1", 1, f), _Block("encoder2", "enc2", f, 2 * f), _Block("encoder3", "enc3", 2 * f, 4 * f),
                    _Block("encoder4", "enc4", 4 * f, 8 * f)]
        self.bott = _Block("bottleneck", "bottleneck", 8 * f, 16 * f)
        self.dec = {l: _Block(f"decoder{l}", f"dec{l}", 2 * c, c) for l, c in ((4, 8 * f), (3, 4 * f), (2, 2 * f), (1, f))}

    # ------------------------------------------------------------------ helpers
    def _tensors(self):
        P = dict(self.m.named_parameters())
        Bf = dict(self.m.named_buffers())
        return P, Bf

    def _conv_bn_relu(self, P, Bf, blk, i, x, ldx, cin, B, H, W, out, ldo, training, saved, groups=1, x_amax=None, out_amax=None, pool=None):
        """x [B,H,W,cin] (pixel stride ldx) -> conv -> BN -> ReLU -> out (pixel stride ldo).
        groups > 1 (train mode): batch statistics per group of B / groups consecutive images, running statistics updated once
        per group in order — what `groups` sequential forward calls of the reference do (one document per call,
        train_nn_patch.py:318-321).
        pool = (pooled [B*(H/2)*(W/2)][cout], its abs-max slot): the 2x2 max-pool of the block output (model_unet.py:52-59) leaves with
        the BatchNorm apply in ONE pass (FUSE_BN_POOL) where there is such a pass; returns True when it did, else the caller pools."""
        def apply(yv, ov, rows, sc, sh, b0, nb):
            if pool is not None and FUSE_BN_POOL:
                pv = pool[0][b0 * (H // 2) * (W // 2):(b0 + nb) * (H // 2) * (W // 2)]
                ops.bn_apply_pool(yv, cout, ov, ldo, pv, cout, nb, H, W, cout, sc, sh, 2, 2, relu=True, amax=out_amax, pooled_amax=pool[1])
            else:
                ops.bn_apply(yv, cout, ov, ldo, rows, cout, sc, sh, relu=True, amax=out_amax)
            return pool is not None and FUSE_BN_POOL

        dev = x.device
        cout = blk.cout
        M = B * H * W
        w = P[blk.key(i, "w")]
        if not training and saved is None and cin != 1 and FUSE_EVAL_BN:
            # inference (Phase A's cleaner pass, validation): eval-mode BatchNorm + ReLU ride in the conv epilogue — the
            # same fused multiply-add bn_apply evaluates, so the result is bit-identical to the two-pass form
            coef = torch.empty(4, cout, device=dev)
            ops.bn_eval_coeff(cout, P[blk.key(i, "gamma")], P[blk.key(i, "beta")], Bf[blk.key(i, "rm")], Bf[blk.key(i, "rv")], BN_EPS,
                              None, coef[0], coef[1], coef[2], coef[3])
            fused_pool = None
            if pool is not None and FUSE_BN_POOL and ops.conv_can_pool(B=B, H=H, W=W, Cin=cin, N=cout, kw=2, ldx=ldx, ldy=ldo):
                fused_pool = (pool[0], cout, 2, pool[1])         # ... and so does the 2x2 max-pool behind the block
            ops.conv_igemm(x, w, out, B=B, H=H, W=W, Cin=cin, OH=H, OW=W, N=cout, KH=3, KW=3, pad=(1, 1), ldx=ldx, ldy=ldo,
                           scale=coef[2], bias=coef[3], relu=True, w_src=("fwd", w), x_amax=x_amax, y_amax=out_amax, pool=fused_pool)
            return fused_pool is not None
        y = torch.empty(M, cout, device=dev)
        fused = None
        if cin == 1:
            ops.conv_c1_fwd(x, w, None, y, cout, B, H, W, cout, relu=False)
        else:
            # train-mode BatchNorm: the conv's epilogue also leaves per-block fp64 column sums of y (no second pass over y)
            # (per-group statistics take the separate pass: a statistics block of the generic tile may straddle two images)
            fused = ops.conv_igemm(x, w, y, B=B, H=H, W=W, Cin=cin, OH=H, OW=W, N=cout, KH=3, KW=3, pad=(1, 1), ldx=ldx, ldy=cout,
                                   w_src=("fwd", w), want_stats=training and groups == 1, x_amax=x_amax)
        gamma, beta = P[blk.key(i, "gamma")], P[blk.key(i, "beta")]
        rm, rv = Bf[blk.key(i, "rm")], Bf[blk.key(i, "rv")]
        if training and groups > 1:
            Mg = M // groups
            coef = torch.empty(groups, 4, cout, device=dev)
            stat64 = torch.empty(groups, 2, cout, device=dev, dtype=torch.float64) if saved is not None else None
            for g in range(groups):
                yg = y[g * Mg:(g + 1) * Mg]
                ops.bn_train_stats(yg, cout, Mg, cout, gamma, beta, BN_EPS, BN_MOMENTUM, rm, rv, coef[g, 0], coef[g, 1], coef[g, 2], coef[g, 3],
                                   stat64[g] if stat64 is not None else None)
                pooled_here = apply(yg, out[g * Mg:(g + 1) * Mg], Mg, coef[g, 2], coef[g, 3], g * (B // groups), B // groups)
            if saved is not None:
                saved.append((y, coef, stat64))
            return pooled_here
        coef = torch.empty(4, cout, device=dev)  # mean, invstd, scale, shift
        stat64 = None
        if training:
            stat64 = torch.empty(2, cout, device=dev, dtype=torch.float64) if saved is not None else None
            if fused is not None:
                ops.bn_train_stats_from_partials(fused[0], fused[1], M, cout, gamma, beta, BN_EPS, BN_MOMENTUM, rm, rv, coef[0], coef[1],
                                                 coef[2], coef[3], stat64)
            else:
                ops.bn_train_stats(y, cout, M, cout, gamma, beta, BN_EPS, BN_MOMENTUM, rm, rv, coef[0], coef[1], coef[2], coef[3], stat64)
        else:
            ops.bn_eval_coeff(cout, gamma, beta, rm, rv, BN_EPS, None, coef[0], coef[1], coef[2], coef[3])
        pooled_here = apply(y, out, M, coef[2], coef[3], 0, B)
        if saved is not None:
            saved.append((y, coef, stat64))
        return pooled_here

    # ------------------------------------------------------------------ forward
    def forward(self, x, training, need_grad, groups=1):
        """x: [B,1,H,W] contiguous CUDA fp32.  Returns (out [B,1,H,W], ctx or None).  groups: see _conv_bn_relu."""
        fs = ensure_flat(self.m)
        P, Bf = self._tensors()
        B, _, H, W = x.shape
        if H % 16 or W % 16:
            raise ValueError(f"UNet input {H}x{W} must be a multiple of 16 in both dimensions")
        dev = x.device
        f = self.f
        groups = int(groups) if training else 1
        if groups < 1 or B % groups:
            raise ValueError(f"batch {B} is not a multiple of bn_groups={groups}")
        ctx = {"x": x, "B": B, "H": H, "W": W, "training": training, "groups": groups, "blocks": {}} if need_grad else None

        # producer-carried abs-max slots of every tensor a split-fp16 conv / wgrad launch will consume (None: that split is off)
        pool = ops.amax_pool(dev)
        slot = (lambda: pool.slot()) if pool is not None else (lambda: None)

        def run_block(blk, xin, ldx, cin, h, w, out, ldo, xin_amax, out_amax, pool=None):
            saved = [] if need_grad else None
            a1 = torch.empty(B * h * w, blk.cout, device=dev)
            a1_amax = slot()
            self._conv_bn_relu(P, Bf, blk, 1, xin, ldx, cin, B, h, w, a1, blk.cout, training, saved, groups, xin_amax, a1_amax)
            pooled_here = self._conv_bn_relu(P, Bf, blk, 2, a1, blk.cout, blk.cout, B, h, w, out, ldo, training, saved, groups, a1_amax, out_amax, pool)
            if need_grad:
                ctx["blocks"][blk.mod] = {"xin": xin, "ldx": ldx, "cin": cin, "h": h, "w": w, "a1": a1, "out": out, "ldo": ldo,
                                          "y1": saved[0][0], "coef1": saved[0][1], "st1": saved[0][2], "y2": saved[1][0],
                                          "coef2": saved[1][1], "st2": saved[1][2], "xin_amax": xin_amax, "a1_amax": a1_amax}
            return pooled_here

        # encoder: level l has c = f*2^(l-1) channels at (H,W)/2^(l-1); its output goes to cat_l[:, c:2c]
        cats, cat_amax = {}, {}
        xin, ldx, cin, xin_amax = x, 1, 1, None
        h, w = H, W
        for l, blk in enumerate(self.enc, start=1):
            c = blk.cout
            cat = torch.empty(B * h * w, 2 * c, device=dev)
            cats[l] = (cat, h, w, c)
            cat_amax[l] = slot()                               # shared by the skip half (BatchNorm apply) and the up half (transposed conv)
            skip = cat[:, c:]
            pooled = torch.empty(B * (h // 2) * (w // 2), c, device=dev)
            xin_in, xin_amax = xin_amax, slot()
            if not run_block(blk, xin, ldx, cin, h, w, skip, 2 * c, xin_in, cat_amax[l], pool=(pooled, xin_amax)):
                ops.maxpool_fwd(skip, 2 * c, pooled, c, B, h, w, c, 2, 2, amax=xin_amax)
            xin, ldx, cin = pooled, c, c
            h, w = h // 2, w // 2
        d = torch.empty(B * h * w, self.bott.cout, device=dev)
        d_amax = slot()                                        # the transposed conv that consumes d is a split GEMM too
        run_block(self.bott, xin, ldx, cin, h, w, d, self.bott.cout, xin_amax, d_amax)
        dcin = self.bott.cout
        ups = {}
        for l in (4, 3, 2, 1):
            cat, hh, ww, c = cats[l]
            wup = P[f"upconv{l}.weight"]                       # [2c][2][2][c] physical (IOHW channels_last)
            wT = ops.transposed(wup, dcin, 4 * c)
            ops.conv_igemm(d, wT, cat, B=B, H=h, W=w, Cin=dcin, OH=h, OW=w, N=4 * c, KH=1, KW=1, ldx=dcin, ldy=2 * c,
                           bias=P[f"upconv{l}.bias"], out_mode=ops.OUT_CONVT, w_src=("T", wup), x_amax=d_amax, y_amax=cat_amax[l])
            ups[l] = (d, dcin, h, w, d_amax)
            h, w = hh, ww
            dnew = torch.empty(B * h * w, c, device=dev)
            d_amax = slot()
            run_block(self.dec[l], cat, 2 * c, 2 * c, h, w, dnew, c, cat_amax[l], d_amax)
            d, dcin = dnew, c
        out = torch.empty(B, 1, H, W, device=dev)
        ops.head_fwd(d, f, P["conv.weight"], P["conv.bias"], out, B * H * W, f)
        if training:
            fs.ibuf.add_(groups)                               # all num_batches_tracked counters at once
        if need_grad:
            ctx.update(cats=cats, ups=ups, d1=d, out=out)
        return out, ctx

    # ------------------------------------------------------------------ backward
    def backward(self, ctx, dout):
        """dout: gradient w.r.t. the sigmoid output [B,1,H,W].  Parameter gradients are accumulated
        into the flat gradient buffer; returns None (the input image needs no gradient)."""
        fs = ensure_flat(self.m)
        fs.attach_grads()
        P = dict(self.m.named_parameters())
        G = {n: p.grad for n, p in P.items()}
        B, H, W = ctx["B"], ctx["H"], ctx["W"]
        dev = dout.device
        f = self.f
        training = ctx["training"]
        dout = dout.contiguous()
        side = self.__dict__.get("_side")
        if side is None or side.side is not None and side.side.device != dev:
            side = ops.SideStream(dev)
            self._side = side

        NG = ctx.get("groups", 1)
        pool = ops.amax_pool(dev)
        slot = (lambda: pool.slot()) if pool is not None else (lambda: None)

        def bn_bwd(da, ldda, y, coef, st, i, dy, M, cout, blk, amax=None, partials=None, pooled=None):
            """BatchNorm(+ReLU) backward of conv i of a block; per statistics group when the forward ran with bn_groups.
            pooled = (dpool, ld, h, w): the gradient of the block's 2x2-pooled output still has to be routed to the winners and added to
            da — done inside the two passes of this backward (qea_bn_bwd_pool)."""
            if pooled is not None:
                ops.bn_bwd_pool(da, ldda, pooled[0], pooled[1], 2, y, cout, B, pooled[2], pooled[3], cout, P[blk.key(i, "gamma")], coef[0], coef[1],
                                training, G[blk.key(i, "gamma")], G[blk.key(i, "beta")], dy, cout, accumulate=True, stat64=st,
                                relu_scale=coef[2], relu_shift=coef[3], amax=amax)
                return
            if coef.dim() == 2:
                ops.bn_bwd(da, ldda, None, 0, y, cout, M, cout, P[blk.key(i, "gamma")], coef[0], coef[1], training, G[blk.key(i, "gamma")],
                           G[blk.key(i, "beta")], dy, cout, accumulate=True, stat64=st, relu_scale=coef[2], relu_shift=coef[3], amax=amax,
                           partials=partials)
                return
            Mg = M // NG
            for g in range(NG):
                sl = slice(g * Mg, (g + 1) * Mg)
                ops.bn_bwd(da[sl], ldda, None, 0, y[sl], cout, Mg, cout, P[blk.key(i, "gamma")], coef[g, 0], coef[g, 1], training,
                           G[blk.key(i, "gamma")], G[blk.key(i, "beta")], dy[sl], cout, accumulate=True,
                           stat64=st[g] if st is not None else None, relu_scale=coef[g, 2], relu_shift=coef[g, 3], amax=amax)

        last_amax = [None]                                        # abs-max slot of the tensor block_bwd returned last

        def block_bwd(blk, da2, ldda, pooled=None):
            """da2: grad w.r.t. the block output (pixel stride ldda).  Returns grad w.r.t. the block input
            as a fresh [M][cin] tensor, or None for the first encoder block.  pooled: see bn_bwd."""
            s = ctx["blocks"][blk.mod]
            h, w, cin, cout = s["h"], s["w"], s["cin"], blk.cout
            M = B * h * w
            dy2 = torch.empty(M, cout, device=dev)
            # ReLU mask recomputed from y with the forward's scale/shift: the activation is not re-read
            dy2_amax, dy1_amax = slot(), slot()
            bn_bwd(da2, ldda, s["y2"], s["coef2"], s["st2"], 2, dy2, M, cout, blk, dy2_amax, pooled=pooled)
            w2 = P[blk.key(2, "w")]
            side.run(lambda: ops.conv_wgrad(dy2, s["a1"], G[blk.key(2, "w")], B=B, PH=h, PW=w, QH=h, QW=w, R=cout, Cc=cout, KH=3, KW=3,
                                            pad=(1, 1), ldp=cout, ldq=cout, accumulate=True, p_amax=dy2_amax, q_amax=s.get("a1_amax")), dy2)
            w2t = ops.flip_transposed(w2, cout, cout, 3, 3)
            da1 = torch.empty(M, cout, device=dev)
            # the two reductions of BatchNorm1's backward come with da1 from the dgrad's epilogue (FUSE_BN_BWD_SUMS; train mode, one
            # statistics group, fp16 form) instead of a pass of their own over da1 and y1
            bst = None
            if FUSE_BN_BWD_SUMS and training and s["coef1"].dim() == 2 and s["st1"] is not None:
                bst = (s["y1"], cout, s["st1"], s["coef1"][2], s["coef1"][3])
            part = ops.conv_igemm(dy2, w2t, da1, B=B, H=h, W=w, Cin=cout, OH=h, OW=w, N=cout, KH=3, KW=3, pad=(1, 1), ldx=cout, ldy=cout,
                                  w_src=("flipT", w2), x_amax=dy2_amax, bwd_stats=bst)
            dy1 = torch.empty(M, cout, device=dev)       # (dy2 may still be read by its wgrad on the side stream)
            bn_bwd(da1, cout, s["y1"], s["coef1"], s["st1"], 1, dy1, M, cout, blk, dy1_amax, partials=part if bst is not None else None)
            if cin == 1:
                side.run(lambda: ops.conv_c1_wgrad(s["xin"], dy1, cout, G[blk.key(1, "w")], None, B, h, w, cout, accumulate=True), dy1)
                return None
            side.run(lambda: ops.conv_wgrad(dy1, s["xin"], G[blk.key(1, "w")], B=B, PH=h, PW=w, QH=h, QW=w, R=cout, Cc=cin, KH=3, KW=3,
                                            pad=(1, 1), ldp=cout, ldq=s["ldx"], accumulate=True, p_amax=dy1_amax, q_amax=s.get("xin_amax")), dy1)
            w1 = P[blk.key(1, "w")]
            w1t = ops.flip_transposed(w1, cout, cin, 3, 3)
            dxin = torch.empty(M, cin, device=dev)
            last_amax[0] = slot()                               # a decoder block's input gradient feeds the transposed conv's dgrad GEMM
            ops.conv_igemm(dy1, w1t, dxin, B=B, H=h, W=w, Cin=cout, OH=h, OW=w, N=cin, KH=3, KW=3, pad=(1, 1), ldx=cout, ldy=cin,
                           w_src=("flipT", w1), x_amax=dy1_amax, y_amax=last_amax[0])
            return dxin

        # head
        d1 = ctx["d1"]
        M0 = B * H * W
        dd = torch.empty(M0, f, device=dev)
        ops.head_bwd(d1, f, ctx["out"], dout, P["conv.weight"], dd, f, G["conv.weight"], G["conv.bias"], M0, f, accumulate=True)

        # decoder levels 1..4: block backward gives d(cat_l); split into up-conv grad and skip grad
        dcats = {}
        for l in (1, 2, 3, 4):
            dcat = block_bwd(self.dec[l], dd, self.dec[l].cout)
            cat, hh, ww, c = ctx["cats"][l]
            dcats[l] = dcat
            dprev, dcin, h, w, dprev_amax = ctx["ups"][l]  # input of upconv_l: [B*h*w][dcin]
            def up_grads(dcat=dcat, dprev=dprev, l=l, c=c, hh=hh, ww=ww, h=h, w=w, dcin=dcin, pa=dprev_amax, qa=last_amax[0]):
                ops.colsum(dcat, 2 * c, B * hh * ww, c, G[f"upconv{l}.bias"], accumulate=True)
                ops.conv_wgrad(dprev, dcat, G[f"upconv{l}.weight"], B=B, PH=h, PW=w, QH=hh, QW=ww, R=dcin, Cc=c, KH=2, KW=2,
                               stride=(2, 2), ldp=dcin, ldq=2 * c, accumulate=True, p_amax=pa, q_amax=qa)
            side.run(up_grads, dcat)
            dd = torch.empty(B * h * w, dcin, device=dev)
            # (the GEMM reads the up half of dcat, whose abs-max the slot of the whole tensor bounds)
            ops.conv_igemm(dcat, P[f"upconv{l}.weight"], dd, B=B, H=hh, W=ww, Cin=c, OH=h, OW=w, N=dcin, KH=2, KW=2, stride=(2, 2),
                           ldx=2 * c, ldy=dcin, w_src=("fwd", P[f"upconv{l}.weight"]), x_amax=last_amax[0])
        # bottleneck, then encoder 4..1: skip grad (dcat[:, c:]) + pool backward of the deeper level
        dpool = block_bwd(self.bott, dd, self.bott.cout)
        for l in (4, 3, 2, 1):
            cat, hh, ww, c = ctx["cats"][l]
            dskip = dcats[l][:, c:]
            if FUSE_POOL_BWD and ctx["blocks"][self.enc[l - 1].mod]["coef2"].dim() == 2:
                # the pool backward rides in the two passes of BatchNorm2's backward: dskip is read, never rewritten
                dpool = block_bwd(self.enc[l - 1], dskip, 2 * c, pooled=(dpool, c, hh, ww))
            else:
                ops.maxpool_bwd(cat[:, c:], 2 * c, dpool, c, dskip, 2 * c, B, hh, ww, c, 2, 2, relu_mask=False, accumulate=True)
                dpool = block_bwd(self.enc[l - 1], dskip, 2 * c)
        side.join()
        return None
