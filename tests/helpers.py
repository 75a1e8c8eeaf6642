"""Shared helpers for the parity tests (synthetic inputs identical to tests/golden/make_golden.py)."""
import os

import numpy as np
import torch

CHAR_SET = ['`', ' ', '!', '"', '#', '$', '%', '&', "'", '(', ')', '*', '+', ',', '-', '.'] + list("0123456789") + \
    [':', ';', '<', '=', '>', '?', '@'] + [chr(c) for c in range(ord('A'), ord('Z') + 1)] + ['[', ']', '^'] + \
    [chr(c) for c in range(ord('a'), ord('z') + 1)] + ['{', '|', '~', '€', '}', '\\', '/']
assert len(CHAR_SET) == 95
C2I = {c: i for i, c in enumerate(CHAR_SET)}
I2C = {i: c for i, c in enumerate(CHAR_SET)}


def golden(name):
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name)
    return np.load(path, allow_pickle=False)


def synth_images(b, seed, h=32, w=128):
    g = torch.Generator().manual_seed(seed)
    m = (torch.rand(b, 1, h, w, generator=g) < 0.12).float()
    ink = torch.rand(b, 1, h, w, generator=g) * 0.7 + 0.3
    return (1 - m * ink + 0.02 * torch.randn(b, 1, h, w, generator=g)).clamp(0, 1)


def synth_labels(b, seed, lo=1, hi=12):
    rng = np.random.RandomState(seed)
    return ["".join(CHAR_SET[i] for i in rng.randint(1, 95, rng.randint(lo, hi + 1))) for _ in range(b)]


def encode(labels):
    y = torch.tensor([C2I[c] for c in "".join(labels)], dtype=torch.int)
    return y, torch.tensor([len(l) for l in labels], dtype=torch.int)


def check_grad_summary(fix, prefix, named_grads, rtol=2e-4, atol_frac=2e-5, atol=1e-7):
    """Compare gradients with the |sum/|abs/|l2/|head summaries stored by make_golden.grad_summary.
    The sum of n terms is compared with a tolerance relative to the abs-sum (its condition);
    `atol` is a per-element noise floor (e.g. a conv bias in front of a batch-stat BN has an
    exactly-zero gradient that both sides compute as 1e-10 rounding noise)."""
    bad = []
    for name, g in named_grads:
        g = g.detach().double().flatten().cpu()
        ref_abs = float(fix[f"{prefix}{name}|abs"])
        ref_l2 = float(fix[f"{prefix}{name}|l2"])
        tol = rtol * ref_abs + atol * g.numel()
        if abs(g.abs().sum().item() - ref_abs) > tol:
            bad.append((name, "abs", g.abs().sum().item(), ref_abs))
        if abs(g.sum().item() - float(fix[f"{prefix}{name}|sum"])) > tol:
            bad.append((name, "sum", g.sum().item(), float(fix[f"{prefix}{name}|sum"])))
        if abs(g.norm().item() - ref_l2) > rtol * ref_l2 + atol * g.numel() ** 0.5:
            bad.append((name, "l2", g.norm().item(), ref_l2))
        head = torch.from_numpy(fix[f"{prefix}{name}|head"])
        n = head.numel()
        scale = ref_l2 / max(1.0, g.numel() ** 0.5)
        if (g[:n] - head).abs().max().item() > rtol * head.abs().max().item() + atol_frac * scale + atol:
            bad.append((name, "head", g[:n].tolist()[:4], head.tolist()[:4]))
    assert not bad, bad[:6]


def check_tensor_summary(fix, prefix, named, rtol=1e-5, atol=0.0):
    bad = []
    for name, t in named:
        v = t.detach().double().flatten().cpu()
        ref_abs = float(fix[f"{prefix}{name}|abs"])
        if abs(v.abs().sum().item() - ref_abs) > rtol * ref_abs + 1e-9 + atol * v.numel():
            bad.append((name, "abs", v.abs().sum().item(), ref_abs))
        if abs(v.sum().item() - float(fix[f"{prefix}{name}|sum"])) > rtol * ref_abs + 1e-9 + atol * v.numel():
            bad.append((name, "sum"))
        head = torch.from_numpy(fix[f"{prefix}{name}|head"])
        if (v[:head.numel()] - head).abs().max().item() > rtol * max(head.abs().max().item(), 1e-3) + atol:
            bad.append((name, "head"))
    assert not bad, bad[:6]


def sample_index(n):
    """Same subset rule as tests/golden/make_golden.py."""
    if n <= 512:
        return torch.arange(n)
    return torch.randperm(n, generator=torch.Generator().manual_seed(1234 + n))[:512]


def robust_rel_err(got, ref, frac=0.005):
    """l2-relative error of `got` against `ref` after discarding the q = max(1, ceil(frac*n)) largest
    element errors, plus the largest discarded error relative to max|ref|.
    Why discard: a ReLU / max-pool decision taken on a pre-activation within ~1e-6 of zero can
    legitimately come out differently in two correct fp32 implementations (measured: ONE such flip
    among 262 144 activations of the B=2 UNet fixture); it moves a handful of gradient elements by
    their full magnitude while every other element agrees to ~1e-6.  Per-kernel tests compare every
    element, so a genuinely wrong region cannot hide behind this allowance."""
    got = got.detach().double().flatten().cpu()
    ref = ref.detach().double().flatten().cpu()
    e = (got - ref).abs()
    q = max(1, int(-(-frac * e.numel() // 1)))
    if e.numel() > q:
        top = e.topk(q)
        e2 = e.clone()
        e2[top.indices] = 0
        worst = top.values[0].item()
    else:
        e2, worst = torch.zeros_like(e), e.max().item()
    den = max(ref.norm().item(), 1e-300)
    return e2.norm().item() / den, worst / max(ref.abs().max().item(), 1e-300)


def check_grad_vs64(fix, prefix, named_grads, rtol=1e-4, k=3.0, atol=2e-7, report=None):
    """Conditioning-aware gradient check for an independent fp32 implementation.

    The fixture holds, per parameter, the reference's gradient computed in fp64 (|s64, a fixed
    512-element sample, and |l264) and |dev = the l2-relative deviation of the reference's OWN
    fp32 run from that fp64 run.  On the small-batch fixtures that deviation reaches 1-2 % for
    the UNet (batch-statistic BatchNorm over near-constant channels amplifies rounding), so
    "within 1e-4 of the fp32 CPU numbers" is not a property any independent fp32 path can have
    there.  |cond is the movement of the reference's fp64 gradient when the input image is perturbed
    by 4e-6 relative noise (the size of an fp32 forward's accumulated rounding): it also captures
    ReLU / max-pool decisions that flip, which a single fp32-vs-fp64 sample (|dev) may miss.
    Required:
        || hip - ref64 || / || ref64 ||  <=  max(rtol, k * max(dev, cond))
    i.e. the HIP result is as close to the exact gradient as the reference's own fp32 arithmetic
    can be expected to be (k = 3 margin), and within rtol = 1e-4 wherever the problem is well
    conditioned."""
    bad = []
    for name, g in named_grads:
        g = g.detach().double().flatten().cpu()
        s64 = torch.from_numpy(fix[f"{prefix}{name}|s64"]).double()
        l264 = float(fix[f"{prefix}{name}|l264"])
        dev = float(fix[f"{prefix}{name}|dev"])
        if f"{prefix}{name}|cond" in fix:
            dev = max(dev, float(fix[f"{prefix}{name}|cond"]))
        idx = sample_index(g.numel())
        den = s64.norm().item()
        floor = atol * idx.numel() ** 0.5
        tol = max(rtol, k * dev) if den > floor else float("inf")
        err, worst = robust_rel_err(g[idx], s64)
        if report is not None:
            report[name] = (err, dev)
        if err * den > tol * den + floor:
            bad.append((name, "sample", err, dev, worst))
        if abs(g.norm().item() - l264) > (max(rtol, k * dev) if l264 > atol * g.numel() ** 0.5 else 1.0) * l264 + atol * g.numel() ** 0.5:
            bad.append((name, "l2", g.norm().item(), l264, dev))
    assert not bad, bad[:8]


# ---------------------------------------------------------------------------------------------------------------
# round-2 knife-edge-free fixtures (tests/golden/cond_b*.npz): the oracle evaluated in fp64 on the fixture's inputs
def _state64(shapes, seed):
    from oracle import model_oracle as mo
    return {k: (v.double() if v.is_floating_point() else v) for k, v in mo.default_init_state(shapes, seed).items()}


def sample_index_small(n, keep=128):
    """Same subset rule as tests/golden/make_golden.py::sample_index_small (round-2 fixtures)."""
    if n <= keep:
        return torch.arange(n)
    return torch.randperm(n, generator=torch.Generator().manual_seed(4321 + n))[:keep]


def _encode_case(fx, c):
    labels, labels_a = [str(s) for s in fx[c + "labels"]], [str(s) for s in fx[c + "labels_a"]]
    return encode(labels), encode(labels_a)


def oracle_cond_phase_b(fx, c="", force=None, record=False):
    """Phase B of candidate `c` ("c0|", ...) of a cond_b*.npz pack on the CPU oracle in fp64: loss, activations, FULL gradient
    tensors.  force = ReLU / max-pool decisions to impose (oracle.model_oracle.Trace); record adds the per-site tensors."""
    import torch.nn.functional as F
    from oracle import model_oracle as mo
    ws = int(fx["ws"])
    x = torch.as_tensor(fx[c + "x"]).double()
    (y, ysz), _ = _encode_case(fx, c)
    Pu, Bu = mo.split_state(_state64(mo.unet_state_shapes(), ws))
    Pc, Bc = mo.split_state(_state64(mo.crnn_state_shapes(), ws + 1))
    tr = mo.Trace(force, record) if (force is not None or record) else None
    img = mo.unet_forward(Pu, Bu, x, training=True, trace=tr)
    lp = mo.crnn_forward(Pc, Bc, img, bn_training=False, trace=tr)
    ins = torch.full((x.shape[0],), lp.shape[0], dtype=torch.int)
    loss = F.ctc_loss(lp, y, ins, ysz) + F.mse_loss(img, torch.ones_like(img))
    loss.backward()
    return dict(loss=loss.item(), img=img.detach(), lp=lp.detach(), g_prep={k: p.grad for k, p in Pu.items()},
                g_crnn={k: p.grad for k, p in Pc.items()}, buf_prep=Bu, rec=tr.rec if tr is not None else None)


def oracle_cond_phase_a(fx, c="", force=None, record=False):
    """Phase A (CRNN in train-mode BN, gradient w.r.t. the input too) of candidate `c`, fp64."""
    import torch.nn.functional as F
    from oracle import model_oracle as mo
    ws = int(fx["ws"])
    x = torch.as_tensor(fx[c + "x"]).double()
    _, (ya, ysa) = _encode_case(fx, c)
    Pc2, Bc2 = mo.split_state(_state64(mo.crnn_state_shapes(), ws + 1))
    tr = mo.Trace(force, record) if (force is not None or record) else None
    xa = x.clone().requires_grad_()
    lpa = mo.crnn_forward(Pc2, Bc2, xa, bn_training=True, trace=tr)
    ins = torch.full((x.shape[0],), lpa.shape[0], dtype=torch.int)
    la = F.ctc_loss(lpa, ya, ins, ysa)
    la.backward()
    return dict(loss=la.item(), lp=lpa.detach(), dx=xa.grad, g_crnn={k: p.grad for k, p in Pc2.items()}, buf_crnn=Bc2,
                rec=tr.rec if tr is not None else None)


def oracle_cond_case(fx, c="", record=False):
    """-> (Phase-B result, Phase-A result) of the free-running CPU oracle in fp64."""
    return oracle_cond_phase_b(fx, c, None, record), oracle_cond_phase_a(fx, c, None, record)


def oracle_tracking_case(fx, batches, weights, c="c0|", force=None, sample_wise=False):
    """weighted_ctc_loss (tracking_utils.py:59-75) on the oracle CRNN (train-mode BN), fp64; force = decisions to impose.
    sample_wise: the non-decaying branch (:69-73) — per-sample CTC (reduction none) times weights[img_indices, i], mean."""
    import torch.nn.functional as F
    from oracle import model_oracle as mo
    Pc, Bc = mo.split_state(_state64(mo.crnn_state_shapes(), int(fx["ws"]) + 1))
    x = torch.from_numpy(fx[c + "x"]).double()
    lp = mo.crnn_forward(Pc, Bc, x, bn_training=True, trace=mo.Trace(force, False) if force is not None else None)
    total = 0
    for i, (t, ts, idx) in enumerate(batches):
        ins = torch.full((len(idx),), lp.shape[0], dtype=torch.int)
        if sample_wise:
            per = F.ctc_loss(lp[:, list(idx), :], t, ins, ts, reduction="none")
            total = total + torch.mean(weights[list(idx), i].double() * per)
        else:
            total = total + float(weights[i]) * F.ctc_loss(lp[:, list(idx), :], t, ins, ts)
    total.backward()
    return dict(loss=total.item(), lp=lp.detach(), g_crnn={k: p.grad for k, p in Pc.items()})


def full_rel_err(got, ref):
    """plain || got - ref || / || ref || over the FULL tensor (no discard, no conditioning term)."""
    got, ref = got.detach().double().flatten().cpu(), ref.detach().double().flatten().cpu()
    return (got - ref).norm().item() / max(ref.norm().item(), 1e-300)


def f3_setup(fx, method, device):
    """the product's host logic of the non-decaying label-history branch, armed with the fixture's history and (for
    self_attention) the fixture's HistoryAttention parameters"""
    import json
    import types
    from label_tracking.tracking_methods import weightgenerator_factory
    names = [str(s) for s in fx["names"]]
    args = types.SimpleNamespace(window_size=int(fx["window"]), query_dim=int(fx["query_dim"]), emb_dim=int(fx["emb_dim"]),
                                 attn_activation="sigmoid")
    wg = weightgenerator_factory(method)(args, device, C2I)
    if method == "self_attention":
        wg.attention_model.load_state_dict({k[4:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("att|")})
        wg.attention_model.to(device)
    self = types.SimpleNamespace(char_to_index=C2I, window_size=int(fx["window"]), weightgen_method=method,
                                 tracked_labels=json.loads(str(fx["history_json"])), device=device)
    return names, wg, self
