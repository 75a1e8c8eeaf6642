"""CRNN proxy on the HIP library: explicit forward / backward kernel schedules.

Mirrors reference models/model_crnn.py:16-28 (CRNN.forward), :47-56 (Convolutional.forward) and
their autograd, including the NaN scrub of CRNN.backward_hook (:30-32).

  x [B,1,32,W] -> conv1..conv7 (NHWC, bias/ReLU fused in the GEMM epilogue, pools as separate
  HBM passes) -> conv7 writes straight into the [T,B,512] sequence layout (map_to_sequence)
  -> 2 x bidirectional LSTM (input projection = one GEMM per direction, recurrence = fused
  MFMA step kernel) -> Linear (GEMM, vocab padded to 96 columns) -> log_softmax.
"""
import torch

from . import ops
from .params import ensure_flat

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
FUSE_POOL = True      # tests flip this: conv1 + ReLU + pool and BatchNorm2 apply + ReLU + pool as one pass each (same bits)
HID = 256

# (name, cin, cout, relu fused in the conv epilogue, pool after (kh,kw) or None)
CONVS = (("conv2", 64, 128, True, (2, 2)), ("conv3", 128, 256, True, None), ("conv4", 256, 256, True, (2, 1)))


FUSE_POOL_BWD = True  # tests flip this: conv6's (2,1) pool backward inside BatchNorm2's backward (qea_bn_bwd_pool)
FUSE_C1_BWD = True    # tests flip this: conv1 -> ReLU -> pool backward from the pooled gradient and x alone (qea_conv_c1_pool_bwd)
KEEP_A1 = False       # tests set this: keep conv1's full-resolution activation (2.1 GB at B = 2048) although nothing in the product reads it

class CRNNEngine:
    def __init__(self, module, vocab):
        self.m = module
        self.vocab = vocab
        self.vpad = (vocab + 31) // 32 * 32
        # `convo.module.` when the backbone sits in the (never used) nn.DataParallel wrapper
        self.cp = "convo.module." if any(n.startswith("convo.module.") for n, _ in module.named_parameters()) else "convo."

    # ------------------------------------------------------------------ forward
    def forward(self, x, bn_training, need_grad, groups=1):
        """groups > 1: the batch is `groups` jitter replicas of the same strips stacked along N (replica-major).
        Batch-statistic BatchNorm is then evaluated PER REPLICA GROUP and the running statistics are updated once
        per group in order, which is exactly what `groups` sequential forward passes of the reference do
        (sharing the statistics across replicas changes the gradients by 4e-2)."""
        fs = ensure_flat(self.m)
        P = dict(self.m.named_parameters())
        Bf = dict(self.m.named_buffers())
        dev = x.device
        B, _, H, W = x.shape
        if H != 32 or W % 4:
            raise ValueError(f"CRNN input must be [B,1,32,W] with W % 4 == 0, got {H}x{W}")
        c = self.cp
        # groups: an int (equal groups) or a sequence of per-group SAMPLE counts (ragged groups, round 4: the strips of several
        # documents in one pass, each document its own BatchNorm batch as in the reference's one-document-per-call loop)
        if isinstance(groups, int):
            if B % groups:
                raise ValueError(f"batch {B} is not a multiple of groups={groups}")
            sizes = [B // groups] * groups
        else:
            sizes = [int(v) for v in groups]
            if sum(sizes) != B or any(v <= 0 for v in sizes):
                raise ValueError(f"group sizes {sizes} do not partition the batch of {B}")
            groups = len(sizes)
        starts = [sum(sizes[:i]) for i in range(groups)]
        ctx = {"x": x, "B": B, "H": H, "W": W, "bn_training": bn_training, "groups": groups, "group_sizes": sizes} if need_grad else None

        # producer-carried abs-max of every tensor a split-fp16 conv / wgrad launch consumes (amx[name]; None when that split is off)
        pool_ = ops.amax_pool(dev)
        slot = (lambda: pool_.slot()) if pool_ is not None else (lambda: None)
        amx = {}
        # conv1 (C_in = 1) + ReLU, pool 2x2
        # (round 4: the backward rebuilds this activation from x, qea_conv_c1_pool_bwd — it is written only where something still reads it)
        keep_a1 = KEEP_A1 or (need_grad and not (FUSE_POOL and FUSE_C1_BWD)) or H % 2 or W % 8
        a1 = torch.empty(B * H * W, 64, device=dev) if keep_a1 else None
        h, w = H // 2, W // 2
        p1 = torch.empty(B * h * w, 64, device=dev)
        amx["p1"] = slot()
        # conv1 + ReLU + max-pool in one pass where the shape allows (FUSE_POOL; the same bits as the two launches)
        if not (FUSE_POOL and ops.conv_c1_fwd_pool(x, P[c + "conv1.weight"], P[c + "conv1.bias"], a1, 64, p1, 64, B, H, W, 64, relu=True,
                                                   pooled_amax=amx["p1"])):
            if a1 is None:
                a1 = torch.empty(B * H * W, 64, device=dev)
            ops.conv_c1_fwd(x, P[c + "conv1.weight"], P[c + "conv1.bias"], a1, 64, B, H, W, 64, relu=True)
            ops.maxpool_fwd(a1, 64, p1, 64, B, H, W, 64, 2, 2, amax=amx["p1"])
        acts = {"a1": a1, "p1": p1}                       # a1 None: not kept (the backward rebuilds it from x; tests: ctx["conv1_params"])
        cur, ccur, cur_name = p1, 64, "p1"
        dims = {"conv1": (H, W)}
        for name, cin, cout, relu, pool in CONVS:
            a = torch.empty(B * h * w, cout, device=dev)
            amx["a" + name[-1]] = slot()
            fused_pool = None
            if pool:
                ph, pw = h // pool[0], w // pool[1]
                pt = torch.empty(B * ph * pw, cout, device=dev)
                amx["p" + name[-1]] = slot()
                # the pool behind relu(conv) leaves with the conv's epilogue where the kernel has the instance (same bits)
                if FUSE_POOL and pool[0] == 2 and h % 2 == 0 and ops.conv_can_pool(B=B, H=h, W=w, Cin=cin, N=cout, kw=pool[1], ldx=cin, ldy=cout):
                    fused_pool = (pt, cout, pool[1], amx["p" + name[-1]])
            ops.conv_igemm(cur, P[c + name + ".weight"], a, B=B, H=h, W=w, Cin=cin, OH=h, OW=w, N=cout, KH=3, KW=3, pad=(1, 1),
                           ldx=cin, ldy=cout, bias=P[c + name + ".bias"], relu=relu, w_src=("fwd", P[c + name + ".weight"]),
                           x_amax=amx[cur_name], y_amax=amx["a" + name[-1]], pool=fused_pool)
            acts["a" + name[-1]] = a
            dims[name] = (h, w)
            cur, ccur, cur_name = a, cout, "a" + name[-1]
            if pool:
                if fused_pool is None:
                    ops.maxpool_fwd(a, cout, pt, cout, B, h, w, cout, pool[0], pool[1], amax=amx["p" + name[-1]])
                acts["p" + name[-1]] = pt
                cur, cur_name = pt, "p" + name[-1]
                h, w = ph, pw
        # conv5 + BN1 + ReLU, conv6 + BN2 + ReLU (bias stays in the conv: y = conv + b is what BN sees)
        for name, bn, cin in (("conv5", "batchnorm1", 256), ("conv6", "batchnorm2", 512)):
            M = B * h * w
            y = torch.empty(M, 512, device=dev)
            ops.conv_igemm(cur, P[c + name + ".weight"], y, B=B, H=h, W=w, Cin=cin, OH=h, OW=w, N=512, KH=3, KW=3, pad=(1, 1),
                           ldx=cin, ldy=512, bias=P[c + name + ".bias"], w_src=("fwd", P[c + name + ".weight"]), x_amax=amx[cur_name])
            amx["a" + name[-1]] = slot()
            G = groups if bn_training else 1
            gb = [(starts[gi], sizes[gi]) for gi in range(G)] if bn_training else [(0, B)]     # (first sample, samples) of every group
            coef = torch.empty(G, 4, 512, device=dev)
            stat64 = torch.empty(G, 2, 512, device=dev, dtype=torch.float64) if (bn_training and need_grad) else None
            a = torch.empty(M, 512, device=dev)
            for gi in range(G):
                b0, bg = gb[gi]
                Mg = bg * h * w
                yg, ag = y[b0 * h * w:(b0 + bg) * h * w], a[b0 * h * w:(b0 + bg) * h * w]
                if bn_training:
                    ops.bn_train_stats(yg, 512, Mg, 512, P[c + bn + ".weight"], P[c + bn + ".bias"], BN_EPS, BN_MOMENTUM,
                                       Bf[c + bn + ".running_mean"], Bf[c + bn + ".running_var"], coef[gi, 0], coef[gi, 1], coef[gi, 2],
                                       coef[gi, 3], stat64[gi] if stat64 is not None else None)
                else:
                    ops.bn_eval_coeff(512, P[c + bn + ".weight"], P[c + bn + ".bias"], Bf[c + bn + ".running_mean"],
                                      Bf[c + bn + ".running_var"], BN_EPS, None, coef[gi, 0], coef[gi, 1], coef[gi, 2], coef[gi, 3])
                if name == "conv6" and FUSE_POOL:
                    # BatchNorm apply + ReLU + the (2,1) max-pool behind it in one pass (bit-identical; p6 is allocated here)
                    if gi == 0:
                        p6 = torch.empty(B * (h // 2) * w, 512, device=dev)
                        amx["p6"] = slot()
                    ops.bn_apply_pool(yg, 512, ag, 512, p6[b0 * (h // 2) * w:(b0 + bg) * (h // 2) * w], 512, bg, h, w, 512, coef[gi, 2],
                                      coef[gi, 3], 2, 1, relu=True, amax=amx["a" + name[-1]], pooled_amax=amx["p6"])
                else:
                    ops.bn_apply(yg, 512, ag, 512, Mg, 512, coef[gi, 2], coef[gi, 3], relu=True, amax=amx["a" + name[-1]])
            acts["y" + name[-1]], acts["coef" + name[-1]], acts["a" + name[-1]], acts["st" + name[-1]] = y, coef, a, stat64
            dims[name] = (h, w)
            cur, cur_name = a, "a" + name[-1]
        if bn_training:
            fs.ibuf.add_(groups)                                   # num_batches_tracked: one per group, as sequential calls
        amx["seq"] = slot()
        if not FUSE_POOL:
            p6 = torch.empty(B * (h // 2) * w, 512, device=dev)
            amx["p6"] = slot()
            ops.maxpool_fwd(cur, 512, p6, 512, B, h, w, 512, 2, 1, amax=amx["p6"])
        acts["p6"] = p6
        h6, w6 = h // 2, w
        T = w6 - 1
        if h6 != 2:
            raise ValueError("CRNN backbone must reduce the height to 2 before conv7")
        # conv7 2x2 pad 0 -> [B,1,T,512], written as [T][B][512]
        seq = torch.empty(T, B, 512, device=dev)
        ops.conv_igemm(p6, P[c + "conv7.weight"], seq, B=B, H=h6, W=w6, Cin=512, OH=1, OW=T, N=512, KH=2, KW=2, ldx=512, ldy=512,
                       bias=P[c + "conv7.bias"], out_mode=ops.OUT_TBC, w_src=("fwd", P[c + "conv7.weight"]), x_amax=amx["p6"], y_amax=amx["seq"])
        dims["conv7"] = (h6, w6)

        # BiLSTM x2
        xin = seq
        lstm = []
        for layer in (0, 1):
            # the layer input's abs-max for its two projection GEMMs: conv7 carried it for layer 0; the BiLSTM output of layer 0
            # gets one pass of its own (shared by both directions)
            xin_amax = amx["seq"] if layer == 0 else lstm[0]["y_amax"]
            gates = torch.empty(T, B, 2 * 4 * HID, device=dev)
            whf, whr = P[f"lstm.weight_hh_l{layer}"], P[f"lstm.weight_hh_l{layer}_reverse"]

            pf, pb, split = ops.lstm_packs(whf, whr)               # W_hh in per-lane MFMA fragment order, both directions, fwd + bwd forms
            for d, suf in enumerate(("", "_reverse")):
                b_ih, b_hh = P[f"lstm.bias_ih_l{layer}{suf}"], P[f"lstm.bias_hh_l{layer}{suf}"]
                bias = ops.weight_cached("lstm_bias_sum", b_ih, lambda b_ih=b_ih, b_hh=b_hh: (b_ih + b_hh).detach(), also=(b_hh,))
                ops.conv_igemm(xin, P[f"lstm.weight_ih_l{layer}{suf}"], gates[:, :, d * 4 * HID:], B=1, H=1, W=T * B, Cin=512, OH=1,
                               OW=T * B, N=4 * HID, KH=1, KW=1, ldx=512, ldy=2 * 4 * HID, bias=bias,
                               w_src=("fwd", P[f"lstm.weight_ih_l{layer}{suf}"]), x_amax=xin_amax)
            cst = torch.empty(T, B, 2 * HID, device=dev)
            y = torch.empty(T, B, 2 * HID, device=dev)
            # the layer output's abs-max: the next layer's projections / the Linear GEMM and, in the backward, the weight gradients
            # that read y take it from here — left by the one-launch layer kernel itself, else one pass
            y_amax = slot()
            if not ops.lstm_layer_fwd_any(gates, cst, y, pf, split, T, B, y_amax=y_amax):
                y_amax = ops.absmax(y, 512, T * B, 512) if pool_ is not None else None
            lstm.append({"x": xin, "gates": gates, "c": cst, "y": y, "pb": pb, "split": split, "x_amax": xin_amax, "y_amax": y_amax})
            xin = y
        # Linear + log_softmax (vocab padded to a multiple of 32 columns; pad columns stay 0)
        vp = self.vpad
        logits = torch.zeros(T * B, vp, device=dev)
        ops.conv_igemm(xin, P["linear.weight"], logits, B=1, H=1, W=T * B, Cin=512, OH=1, OW=T * B, N=self.vocab, KH=1, KW=1, ldx=512,
                       ldy=vp, bias=P["linear.bias"], w_src=("fwd", P["linear.weight"]), x_amax=lstm[1]["y_amax"])
        lp = torch.zeros(T * B, vp, device=dev)
        ops.log_softmax_fwd(logits, vp, lp, vp, T * B, self.vocab)
        out = lp.view(T, B, vp)[:, :, :self.vocab]
        if need_grad:
            ctx.update(acts=acts, dims=dims, lstm=lstm, lp=lp, T=T, h6=h6, w6=w6, amx=amx, conv1_params=(P[c + "conv1.weight"], P[c + "conv1.bias"]))
        return out, ctx

    # ------------------------------------------------------------------ backward
    def backward(self, ctx, dlp, nan_scrub, need_dx, param_grads=True):
        """dlp: gradient w.r.t. the returned log-probs [T,B,vocab].  Returns dx [B,1,H,W] or None."""
        fs = ensure_flat(self.m)
        fs.attach_grads()
        P = dict(self.m.named_parameters())
        G = {n: p.grad for n, p in P.items()}
        dev = dlp.device
        full = None
        if ctx.get("grad_group", -1) >= 0 and ctx["groups"] > 1:
            # only ONE replica group receives a gradient (CRNN.forward(backward_group=g) detached the others): run the whole
            # backward on that group's samples.  Batch-major activations are row slices; the [T][B][..] sequence buffers are copied.
            full = (ctx["B"], ctx["x"])
            ctx, dlp = self._group_slice(ctx, dlp)
        B, H, W, T = ctx["B"], ctx["H"], ctx["W"], ctx["T"]
        acts, dims = ctx["acts"], ctx["dims"]
        vp, V = self.vpad, self.vocab
        c = self.cp
        TB = T * B
        side = self.__dict__.get("_side")
        if side is None or side.side is not None and side.side.device != dev:
            side = ops.SideStream(dev)
            self._side = side

        g = dlp.contiguous().view(TB, V)
        dlogits = torch.empty(TB, vp, device=dev)
        ops.log_softmax_bwd(g, V, ctx["lp"], vp, dlogits, vp, TB, V, vp, nan_scrub)
        pool_ = ops.amax_pool(dev)                                # abs-max slots of this pass (None: the fp16 split is off)
        slot = (lambda: pool_.slot()) if pool_ is not None else (lambda: None)
        f16 = pool_ is not None                                   # the split-fp16 GEMMs below want their operands' abs-max
        dl_amax = ops.absmax(dlogits, vp, TB, vp) if f16 else None

        # Linear
        y1 = ctx["lstm"][1]["y"]
        if param_grads:
            def linear_grads():
                dwl = torch.empty(vp, 512, device=dev)
                ops.conv_wgrad(dlogits, y1, dwl, B=1, PH=1, PW=TB, QH=1, QW=TB, R=vp, Cc=512, KH=1, KW=1, ldp=vp, ldq=512,
                               p_amax=dl_amax, q_amax=ctx["lstm"][1].get("y_amax"))
                G["linear.weight"].add_(dwl[:V])
                dbl = torch.empty(vp, device=dev)
                ops.colsum(dlogits, vp, TB, vp, dbl)
                G["linear.bias"].add_(dbl[:V])
            side.run(linear_grads, dlogits)
        wlin = P["linear.weight"]

        def padded_T():
            wpad = torch.zeros(vp, 512, device=dev)
            wpad[:V].copy_(wlin)
            out = torch.empty(512, vp, device=dev)
            ops.transpose2d(wpad, out, vp, 512)
            return out
        wlT = ops.weight_cached(("padT", vp), wlin, padded_T)
        dy = torch.empty(T, B, 512, device=dev)
        ops.conv_igemm(dlogits, wlT, dy, B=1, H=1, W=TB, Cin=vp, OH=1, OW=TB, N=512, KH=1, KW=1, ldx=vp, ldy=512, w_src=(("padT", vp), wlin),
                       x_amax=dl_amax)

        # BiLSTM layers, top first
        dseq_bt = None
        for layer in (1, 0):
            s = ctx["lstm"][layer]
            gates, cst, yl, xin = s["gates"], s["c"], s["y"], s["x"]
            dc = None if s["split"] == "seq" else torch.empty(B, 2 * HID, device=dev)    # the one-launch kernels keep dc in registers
            # ONE abs-max of the gate gradients serves the four weight-gradient GEMMs and the input-gradient GEMM of the layer (a bound
            # over both directions and all steps scales every slice of the tensor): from the one-launch kernel, else one pass
            g_amax = slot()
            if not ops.lstm_layer_bwd_any(gates, cst, dy, s["pb"], s["split"], dc, T, B, g_amax=g_amax):      # gates now hold dgates
                g_amax = ops.absmax(gates, 8 * HID, TB, 8 * HID) if f16 else None
            if param_grads:
                def lstm_grads(layer=layer, gates=gates, xin=xin, yl=yl, g_amax=g_amax, x_amax=s.get("x_amax"), y_amax=s.get("y_amax")):
                    for d, suf in enumerate(("", "_reverse")):
                        dg = gates[:, :, d * 4 * HID:]
                        ops.conv_wgrad(dg, xin, G[f"lstm.weight_ih_l{layer}{suf}"], B=1, PH=1, PW=TB, QH=1, QW=TB, R=4 * HID, Cc=512,
                                       KH=1, KW=1, ldp=8 * HID, ldq=512, accumulate=True, p_amax=g_amax, q_amax=x_amax)
                        db = torch.empty(4 * HID, device=dev)                  # b_ih and b_hh enter the gates as a sum: one column
                        ops.colsum(dg, 8 * HID, TB, 4 * HID, db)               # sum (a full pass over the gate gradients) serves both
                        G[f"lstm.bias_ih_l{layer}{suf}"].add_(db)
                        G[f"lstm.bias_hh_l{layer}{suf}"].add_(db)
                        if T > 1:
                            n = (T - 1) * B
                            if d == 0:
                                pg, qh = gates[1:, :, :4 * HID], yl[:T - 1, :, :HID]
                            else:
                                pg, qh = gates[:T - 1, :, 4 * HID:], yl[1:, :, HID:]
                            ops.conv_wgrad(pg, qh, G[f"lstm.weight_hh_l{layer}{suf}"], B=1, PH=1, PW=n, QH=1, QW=n, R=4 * HID, Cc=HID,
                                           KH=1, KW=1, ldp=8 * HID, ldq=2 * HID, accumulate=True, p_amax=g_amax, q_amax=y_amax)
                side.run(lstm_grads)
            wf, wr = P[f"lstm.weight_ih_l{layer}"], P[f"lstm.weight_ih_l{layer}_reverse"]

            def cat_T(wf=wf, wr=wr):
                wcat = torch.cat((wf, wr), 0)                                                            # [2048][512]
                out = torch.empty(512, 8 * HID, device=dev)
                ops.transpose2d(wcat, out, 8 * HID, 512)
                return out
            wT = ops.weight_cached("catT", wf, cat_T, also=(wr,))
            if layer == 1:
                dxl = torch.empty(T, B, 512, device=dev)
                ops.conv_igemm(gates, wT, dxl, B=1, H=1, W=TB, Cin=8 * HID, OH=1, OW=TB, N=512, KH=1, KW=1, ldx=8 * HID, ldy=512,
                               x_amax=g_amax, w_src=("catT", wf, (wr,)))      # (planes of the concatenated filter: cached with BOTH directions' weights)
                dy = dxl
            else:
                # rows (t,b) -> output row b*T + t : the gradient of conv7's output in its own [B,1,T,512] order
                dseq_bt = torch.empty(B, T, 512, device=dev)
                dseq_amax = slot()
                ops.conv_igemm(gates, wT, dseq_bt, B=T, H=1, W=B, Cin=8 * HID, OH=1, OW=B, N=512, KH=1, KW=1, ldx=8 * HID, ldy=512,
                               out_mode=ops.OUT_TBC, x_amax=g_amax, y_amax=dseq_amax, w_src=("catT", wf, (wr,)))

        # conv7 (2x2, pad 0) backward
        h6, w6 = ctx["h6"], ctx["w6"]
        p6 = acts["p6"]
        if param_grads:
            def conv7_grads():
                ops.colsum(dseq_bt, 512, B * T, 512, G[c + "conv7.bias"], accumulate=True)
                ops.conv_wgrad(dseq_bt, p6, G[c + "conv7.weight"], B=B, PH=1, PW=T, QH=h6, QW=w6, R=512, Cc=512, KH=2, KW=2, ldp=512,
                               ldq=512, accumulate=True, p_amax=dseq_amax, q_amax=ctx.get("amx", {}).get("p6"))
            side.run(conv7_grads, dseq_bt)
        w7t = ops.flip_transposed(P[c + "conv7.weight"], 512, 512, 2, 2)
        dp6 = torch.empty(B * h6 * w6, 512, device=dev)
        ops.conv_igemm(dseq_bt, w7t, dp6, B=B, H=1, W=T, Cin=512, OH=h6, OW=w6, N=512, KH=2, KW=2, pad=(1, 1), ldx=512, ldy=512,
                       w_src=("flipT", P[c + "conv7.weight"]), x_amax=dseq_amax)
        # pool (2,1) backward -> grad of a6 (ReLU handled by bn_bwd's mask)
        h, w = dims["conv6"]
        da = None
        if not FUSE_POOL_BWD:
            da = torch.empty(B * h * w, 512, device=dev)
            ops.maxpool_bwd(acts["a6"], 512, dp6, 512, da, 512, B, h, w, 512, 2, 1, relu_mask=False)
        bn_training = ctx["bn_training"]
        amx = ctx.get("amx", {})                                  # forward tensors' abs-max (a replica-group slice keeps its tensor's bound)
        for name, bn, cin, src in (("conv6", "batchnorm2", 512, "a5"), ("conv5", "batchnorm1", 256, "p4")):
            M = B * h * w
            k = name[-1]
            coef = acts["coef" + k]
            dy_ = torch.empty(M, 512, device=dev)
            dy_amax = slot()
            NG = coef.shape[0]
            gsz = ctx.get("group_sizes") if NG > 1 else None
            if gsz is None or len(gsz) != NG:
                gsz = [B // NG] * NG
            for gi in range(NG):
                b0 = sum(gsz[:gi])
                Mg = gsz[gi] * h * w
                sl = slice(b0 * h * w, b0 * h * w + Mg)
                st = acts["st" + k]
                if da is None:
                    # conv6's output went through the (2,1) pool only: its backward rides in the two passes of BatchNorm2's (qea_bn_bwd_pool)
                    Bg = gsz[gi]
                    psl = slice(b0 * (h // 2) * w, (b0 + Bg) * (h // 2) * w)
                    ops.bn_bwd_pool(None, 0, dp6[psl], 512, 1, acts["y" + k][sl], 512, Bg, h, w, 512, P[c + bn + ".weight"], coef[gi, 0],
                                    coef[gi, 1], bn_training, G[c + bn + ".weight"] if param_grads else None,
                                    G[c + bn + ".bias"] if param_grads else None, dy_[sl], 512, accumulate=True,
                                    stat64=st[gi] if st is not None else None, relu_scale=coef[gi, 2], relu_shift=coef[gi, 3], amax=dy_amax)
                    continue
                ops.bn_bwd(da[sl], 512, None, 0, acts["y" + k][sl], 512, Mg, 512, P[c + bn + ".weight"], coef[gi, 0],
                           coef[gi, 1], bn_training, G[c + bn + ".weight"] if param_grads else None,
                           G[c + bn + ".bias"] if param_grads else None, dy_[sl], 512, accumulate=True,
                           stat64=st[gi] if st is not None else None, relu_scale=coef[gi, 2], relu_shift=coef[gi, 3], amax=dy_amax)
            if param_grads:
                def bn_conv_grads(dy_=dy_, name=name, src=src, cin=cin, M=M, dy_amax=dy_amax):
                    # (the bias gradient = column sums of dy rides with the weight gradient's staging waves where the kernel has it)
                    ops.conv_wgrad(dy_, acts[src], G[c + name + ".weight"], B=B, PH=h, PW=w, QH=h, QW=w, R=512, Cc=cin, KH=3, KW=3,
                                   pad=(1, 1), ldp=512, ldq=cin, accumulate=True, p_amax=dy_amax, q_amax=amx.get(src),
                                   dbias=G[c + name + ".bias"])
                side.run(bn_conv_grads, dy_)
            wt = ops.flip_transposed(P[c + name + ".weight"], 512, cin, 3, 3)
            da = torch.empty(M, cin, device=dev)
            ops.conv_igemm(dy_, wt, da, B=B, H=h, W=w, Cin=512, OH=h, OW=w, N=cin, KH=3, KW=3, pad=(1, 1), ldx=512, ldy=cin,
                           w_src=("flipT", P[c + name + ".weight"]), x_amax=dy_amax)
        # da = grad of p4 [B,4,W/4,256]
        # conv4 (+ReLU, pool (2,1)), conv3 (+ReLU), conv2 (+ReLU, pool (2,2))
        dcur, dcur_amax = da, None
        for name, cin, cout, relu, pool in reversed(CONVS):
            h, w = dims[name]
            M = B * h * w
            a = acts["a" + name[-1]]
            if pool:
                dyc = torch.empty(M, cout, device=dev)
                dyc_amax = slot()
                ops.maxpool_bwd(a, cout, dcur, cout, dyc, cout, B, h, w, cout, pool[0], pool[1], relu_mask=True, amax=dyc_amax)
            else:
                dyc, dyc_amax = dcur, dcur_amax             # already masked by the producer's epilogue (which also left its abs-max)
            src = {"conv4": "a3", "conv3": "p2", "conv2": "p1"}[name]
            if param_grads:
                def conv_grads(dyc=dyc, name=name, src=src, cin=cin, cout=cout, M=M, h=h, w=w, dyc_amax=dyc_amax):
                    ops.conv_wgrad(dyc, acts[src], G[c + name + ".weight"], B=B, PH=h, PW=w, QH=h, QW=w, R=cout, Cc=cin, KH=3, KW=3,
                                   pad=(1, 1), ldp=cout, ldq=cin, accumulate=True, p_amax=dyc_amax, q_amax=amx.get(src),
                                   dbias=G[c + name + ".bias"])
                side.run(conv_grads, dyc)
            wt = ops.flip_transposed(P[c + name + ".weight"], cout, cin, 3, 3)
            dnext = torch.empty(M, cin, device=dev)
            # conv4's input a3 is a bare ReLU output (no pool in between): fuse its mask here
            mask = acts["a3"] if name == "conv4" else None
            dnext_amax = slot() if name == "conv4" else None   # conv4's input gradient IS conv3's output gradient (no pool in between)
            ops.conv_igemm(dyc, wt, dnext, B=B, H=h, W=w, Cin=cout, OH=h, OW=w, N=cin, KH=3, KW=3, pad=(1, 1), ldx=cout, ldy=cin,
                           mask=mask, ldmask=cin if mask is not None else 0, w_src=("flipT", P[c + name + ".weight"]),
                           x_amax=dyc_amax, y_amax=dnext_amax)
            dcur, dcur_amax = dnext, dnext_amax
        # dcur = grad of p1 [B,16,W/2,64]; conv1
        dx = None
        if FUSE_C1_BWD and H % 2 == 0 and W % 8 == 0 and (param_grads or need_dx):
            # conv1 -> ReLU -> pool backward from the pooled gradient and the 1-channel input: the activation is rebuilt (nine multiply-adds
            # per element), neither it nor its 2.1 GB gradient is read or written (dcur is overwritten with its masked values)
            if need_dx:
                dx = torch.empty(B, 1, H, W, device=dev)
            ops.conv_c1_pool_bwd(ctx["x"], P[c + "conv1.weight"], P[c + "conv1.bias"], dcur, 64, G[c + "conv1.weight"] if param_grads else None,
                                 G[c + "conv1.bias"] if param_grads else None, dx, B, H, W, 64, accumulate=True)
        else:
            a1 = acts["a1"]
            if a1 is None:                                  # (not kept by the forward: rebuild it)
                a1 = torch.empty(B * H * W, 64, device=dev)
                ops.conv_c1_fwd(ctx["x"], P[c + "conv1.weight"], P[c + "conv1.bias"], a1, 64, B, H, W, 64, relu=True)
            dy1 = torch.empty(B * H * W, 64, device=dev)
            ops.maxpool_bwd(a1, 64, dcur, 64, dy1, 64, B, H, W, 64, 2, 2, relu_mask=True)
            if param_grads:
                side.run(lambda: ops.conv_c1_wgrad(ctx["x"], dy1, 64, G[c + "conv1.weight"], G[c + "conv1.bias"], B, H, W, 64, accumulate=True),
                         dy1)
            if need_dx:
                dx = torch.empty(B, 1, H, W, device=dev)
                ops.conv_c1_dgrad(dy1, 64, P[c + "conv1.weight"], dx, B, H, W, 64)
        side.join()
        if full is not None and dx is not None:
            Bf, g, k = full[0], ctx["_g"], B
            dxf = torch.zeros(Bf, 1, H, W, device=dev)
            dxf[g * k:(g + 1) * k].copy_(dx)
            dx = dxf
        return dx

    @staticmethod
    def _group_slice(ctx, dlp):
        """The saved context and the output gradient restricted to replica group g = ctx["grad_group"]."""
        g, R, B = ctx["grad_group"], ctx["groups"], ctx["B"]
        k = B // R
        b0 = g * k
        T = ctx["T"]
        dims = ctx["dims"]
        sub = dict(ctx)
        sub.update(B=k, groups=1, group_sizes=[k], grad_group=-1, _g=g, x=ctx["x"][b0:b0 + k])
        acts = {}
        for name, t in ctx["acts"].items():
            if t is None:
                acts[name] = None
            elif name.startswith("coef") or name.startswith("st"):
                acts[name] = t[g:g + 1] if t.shape[0] == R else t        # per-group BN coefficients / fp64 statistics
            else:
                rows = t.shape[0] // B                                   # pixels per sample at this layer
                acts[name] = t[b0 * rows:(b0 + k) * rows]
        sub["acts"] = acts
        sub["lstm"] = [{key: (v[:, b0:b0 + k].contiguous() if torch.is_tensor(v) and v.dim() == 3 and v.shape[:2] == (T, B) else v)
                        for key, v in s.items()} for s in ctx["lstm"]]
        lp = ctx["lp"]
        sub["lp"] = lp.view(T, B, -1)[:, b0:b0 + k].contiguous().view(T * k, -1)
        return sub, dlp[:, b0:b0 + k].contiguous()
