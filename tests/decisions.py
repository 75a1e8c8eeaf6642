"""TEST INFRASTRUCTURE — the decision-conditioned gradient gate.

A ReLU / max-pool network's gradient is a discontinuous function of its forward activations: a pre-activation within
rounding distance of 0 (or two pool candidates within rounding of each other) is decided by the last bit of the forward,
and two correct fp32 implementations may decide it differently; every gradient upstream of such a flip then differs by
1e-3..1e-2 although both are exact derivatives of (slightly different) piecewise-linear selections.  The gate therefore
splits the question in two:

  1. GRADIENT: the decisions the HIP forward took (ReLU masks and pool winners of every layer, read from the activations
     the engines saved for their own backward) are imposed on the fp64 oracle (oracle.model_oracle.Trace.force).  The
     oracle's backward is then the exact gradient of the very function the HIP backward differentiates, and EVERY
     gradient tensor of EVERY candidate must be within the plain full-tensor 1e-4 of it — no candidate list, no
     statistics over runs.
  2. DECISIONS: every decision that differs from the free-running fp64 oracle's must be explainable by forward rounding:
     |fp64 pre-activation| (resp. the fp64 gap between the window maximum and the element HIP picked) at most
     FLIP_UNITS fp32 rounding units of the layer (unit = 2^-23 x rms of the layer's pre-activation / pool input),
     and the per-layer forward error is printed next to the reference's own fp32 figures (the ladder fixtures).
"""
import torch
import torch.nn.functional as F

from oracle import model_oracle as mo

UNET_BLOCKS = (("encoder1", "enc1"), ("encoder2", "enc2"), ("encoder3", "enc3"), ("encoder4", "enc4"), ("bottleneck", "bottleneck"),
               ("decoder4", "dec4"), ("decoder3", "dec3"), ("decoder2", "dec2"), ("decoder1", "dec1"))


def _nchw(t, B, h, w, start=0):
    """[>= (start+B)*h*w][C] NHWC rows (possibly a strided column slice) -> samples start .. start+B as [B,C,h,w] on the CPU"""
    return t.detach()[start * h * w:(start + B) * h * w].reshape(B, h, w, t.shape[-1]).permute(0, 3, 1, 2).contiguous().cpu()


def saved_of(t):
    """the context an engine kept for its backward, found on the autograd node (UNetFn / CRNNFn) that produced tensor t —
    directly or through the slices / cat the model wraps around it (CRNN.forward(backward_group=...))"""
    todo, seen = [t.grad_fn], set()
    while todo:
        fn = todo.pop(0)
        if fn is None or fn in seen:
            continue
        seen.add(fn)
        if isinstance(getattr(fn, "saved", None), dict):
            return fn.saved
        todo.extend(f for f, _ in fn.next_functions)
    raise RuntimeError("no engine context behind this tensor (was the backward already run?)")


def hip_unet_trace(saved, first=None):
    """saved = the context UNetEngine.forward kept for its backward (img.grad_fn.saved).
    -> (force, taps): the decisions the HIP forward took, and its layer outputs under the oracle's site names
    (post-ReLU activations under '<relu site>').  first = n: only the first n samples of the batch."""
    B = first or saved["B"]
    force, taps = {}, {}
    for mod, name in UNET_BLOCKS:
        s = saved["blocks"][mod]
        h, w = s["h"], s["w"]
        for i, (y, a) in enumerate(((s["y1"], s["a1"]), (s["y2"], s["out"])), start=1):
            taps[f"{mod}.{name}conv{i}"] = _nchw(y, B, h, w)
            act = _nchw(a, B, h, w)
            taps[f"{mod}.{name}relu{i}"] = act
            force[f"{mod}.{name}relu{i}"] = act > 0
    for l in (1, 2, 3, 4):
        cat, h, w, c = saved["cats"][l]
        skip = _nchw(cat[:, c:], B, h, w)
        force[f"pool{l}"] = F.max_pool2d(skip, 2, return_indices=True)[1]
        taps[f"upconv{l}"] = _nchw(cat[:, :c], B, h, w)
    taps["img"] = saved["out"].detach()[:B].cpu()
    return force, taps


def hip_crnn_trace(saved, first=None, start=0):
    """the same for CRNNEngine.forward's context (lp.grad_fn.saved); samples start .. start + first of the batch."""
    B, H, W, T = first or saved["B"], saved["H"], saved["W"], saved["T"]
    b0 = start
    acts, dims = saved["acts"], saved["dims"]
    force, taps = {}, {}
    for k in "123456":
        h, w = dims["conv" + k]
        a = acts["a" + k]
        if a is None and k == "1":
            # conv1's activation is not kept by the product (its backward rebuilds it from x): rebuild it the same way for the trace
            from qea import ops
            wt, bs = saved["conv1_params"]
            xin = saved["x"]
            a = torch.empty(xin.shape[0] * H * W, 64, device=xin.device)
            ops.conv_c1_fwd(xin, wt.detach(), bs.detach(), a, 64, xin.shape[0], H, W, 64, relu=True)
        act = _nchw(a, B, h, w, b0)
        taps["convo.relu" + k] = act
        force["convo.relu" + k] = act > 0
        if "convo.pool" + k in mo.POOL_SITES:
            force["convo.pool" + k] = F.max_pool2d(act, mo.POOL_SITES["convo.pool" + k], return_indices=True)[1]
    for k in "56":
        h, w = dims["conv" + k]
        taps["convo.conv" + k] = _nchw(acts["y" + k], B, h, w, b0)
    seq = saved["lstm"][0]["x"]                                     # conv7's output, written as [T][B][512]
    taps["convo.conv7"] = seq.detach()[:, b0:b0 + B].permute(1, 2, 0).unsqueeze(2).contiguous().cpu()
    taps["lstm0"] = saved["lstm"][0]["y"].detach()[:, b0:b0 + B].cpu()
    taps["lstm1"] = saved["lstm"][1]["y"].detach()[:, b0:b0 + B].cpu()
    return force, taps


def post_relu(rec, site):
    """oracle record -> the tensor comparable with a HIP tap of the same name"""
    if "relu" in site:
        head, leaf = site.rsplit(".", 1)
        if head == "convo":
            pre = {"5": "convo.batchnorm1", "6": "convo.batchnorm2"}.get(leaf[-1], "convo.conv" + leaf[-1])
        else:
            pre = head + "." + leaf.replace("relu", "norm")
        return F.relu(rec[pre])
    return rec[site]


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return (a - b).norm().item() / max(b.norm().item(), 1e-300)


def ladder(taps, rec):
    """{site: l2-relative error of the HIP layer output against the free-running fp64 oracle}"""
    return {site: rel_l2(t, post_relu(rec, site)) for site, t in taps.items()}


def flip_report(force, rec):
    """Decisions of the HIP forward that differ from the free-running fp64 oracle's own (rec = its Trace record).
    -> {site: (n_flips, n_decisions, worst margin in fp32 rounding units of the layer)} where the margin of a flipped ReLU
    is |fp64 pre-activation| and that of a flipped pool window is (fp64 window maximum - fp64 value of the element the HIP
    path kept); unit = 2^-23 * rms of the layer's tensor."""
    own = mo.own_decisions(rec)
    out = {}
    pre_of = {mo.relu_site(s): s for s in rec if mo.is_preactivation(s)}
    for site, dec in force.items():
        if site in mo.POOL_SITES:
            x = rec[site].double()
            unit = 2.0 ** -23 * x.pow(2).mean().sqrt().item()
            flat = x.flatten(2)
            mine = flat.gather(2, dec.flatten(2))
            best = flat.gather(2, own[site].flatten(2))
            diff = (dec != own[site]).flatten(2)
            gap = (best - mine)[diff]
            # windows whose candidates are exactly equal in fp64 (all-zero ReLU outputs) may legitimately name another element
            out[site] = (int((gap > 0).sum()), dec.numel(), (gap.max().item() / unit) if gap.numel() else 0.0)
        else:
            z = rec[pre_of[site]].double()
            unit = 2.0 ** -23 * z.pow(2).mean().sqrt().item()
            diff = dec != own[site]
            m = z[diff].abs()
            out[site] = (int(diff.sum()), dec.numel(), (m.max().item() / unit) if m.numel() else 0.0)
    return out
