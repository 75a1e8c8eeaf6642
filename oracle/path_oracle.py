"""TEST INFRASTRUCTURE ONLY — CPU oracle for everything on the hot path that is not a network:
CTC loss/gradient, jitter, TopKCER, crop+pad, greedy decode, CER, Adam, and the Phase-A /
Phase-B step choreography.  numpy float64 for the arithmetic kernels, plain Python for the
host logic.  Each function cites the reference lines it follows; the CTC recursion lives in
ATen (torch==1.8.0 pinned by requirements.txt:94, torch 2.10 CPU in the build container) and
is restated from its published algorithm (Graves et al. 2006, eq. 6-16, as implemented by
aten/src/ATen/native/LossCTC.cpp) and pinned against torch's CPU ctc_loss by
tests/golden/ctc_cases.npz.
"""
import math

import numpy as np

NEG_INF = -np.inf


# ----------------------------------------------------------------------------- CTC
def _lse(a, b):
    m = np.maximum(a, b)
    with np.errstate(invalid="ignore"):
        r = m + np.log(np.exp(a - m) + np.exp(b - m))
    return np.where(np.isneginf(m), NEG_INF, r)


def _lse3(a, b, c):
    m = np.maximum(np.maximum(a, b), c)
    with np.errstate(invalid="ignore"):
        r = m + np.log(np.exp(a - m) + np.exp(b - m) + np.exp(c - m))
    return np.where(np.isneginf(m), NEG_INF, r)


def _shift_right(a, k):
    out = np.full_like(a, NEG_INF)
    if a.shape[0] > k:
        out[k:] = a[:-k]
    return out


def ctc_sample(lp, target, T, blank=0):
    """One sample.  lp: [T_max, C] log-probs, target: int sequence (len L), T: input length.
    Returns (nll, raw) where raw[t, c] = logsumexp_{s: l'_s = c} (alpha_t(s) + beta_t(s))
    (-inf where no state carries c) — the quantity ATen calls `res` before its
    `(exp(lp) - exp(res + nll - lp)) * grad_out` step.
    CTCLoss call sites: train_nn_patch.py:143,178,294; train_nn_area.py:146-147,265."""
    lp = np.asarray(lp, dtype=np.float64)
    L = len(target)
    S = 2 * L + 1
    ext = np.full(S, blank, dtype=np.int64)
    ext[1::2] = np.asarray(target, dtype=np.int64)
    # a skip (s-2 -> s) is allowed onto a non-blank that differs from the previous non-blank
    skip = np.zeros(S, dtype=bool)
    skip[2:] = (ext[2:] != blank) & (ext[2:] != ext[:-2])

    alpha = np.full((T, S), NEG_INF)
    alpha[0, 0] = lp[0, blank]
    if S > 1:
        alpha[0, 1] = lp[0, ext[1]]
    for t in range(1, T):
        a0 = alpha[t - 1]
        a1 = _shift_right(a0, 1)
        a2 = np.where(skip, _shift_right(a0, 2), NEG_INF)
        alpha[t] = _lse3(a0, a1, a2) + lp[t, ext]
    ll = alpha[T - 1, S - 1] if S == 1 else _lse(alpha[T - 1, S - 1], alpha[T - 1, S - 2])
    nll = -float(ll)

    beta = np.full((T, S), NEG_INF)
    beta[T - 1, S - 1] = lp[T - 1, blank]
    if S > 1:
        beta[T - 1, S - 2] = lp[T - 1, ext[S - 2]]
    skip_fwd = np.zeros(S, dtype=bool)      # from s may jump to s+2
    if S > 2:
        skip_fwd[:-2] = skip[2:]
    for t in range(T - 2, -1, -1):
        b0 = beta[t + 1]
        b1 = _shift_right(b0[::-1], 1)[::-1]
        b2 = np.where(skip_fwd, _shift_right(b0[::-1], 2)[::-1], NEG_INF)
        beta[t] = _lse3(b0, b1, b2) + lp[t, ext]

    C = lp.shape[1]
    raw = np.full((lp.shape[0], C), NEG_INF)
    ab = alpha + beta
    for s in range(S):
        raw[:T, ext[s]] = _lse(raw[:T, ext[s]], ab[:, s])
    return nll, raw


def ctc_loss(lp, targets, input_lengths, target_lengths, reduction="mean", blank=0):
    """lp [T,N,C]; targets 1-D concatenated.  Returns (loss, grad_lp) with grad_lp the gradient of
    the REDUCED loss w.r.t. lp exactly as ATen forms it:
        g[t,n,c] = (exp(lp) - exp(raw + nll_n - lp)) * grad_out_n,   0 for t >= input_length,
    grad_out_n = 1/(N*max(len_n,1)) for 'mean', 1 for 'none'/'sum'.  zero_infinity=False: an
    infeasible target gives nll = +inf and NaN gradient entries (the reference then relies on the
    CRNN NaN-scrub hook, models/model_crnn.py:30-32)."""
    lp = np.asarray(lp, dtype=np.float64)
    T, N, C = lp.shape
    nll = np.zeros(N)
    grad = np.zeros_like(lp)
    off = 0
    for n in range(N):
        Ln = int(target_lengths[n])
        Tn = int(input_lengths[n])
        tg = np.asarray(targets[off:off + Ln])
        off += Ln
        nll[n], raw = ctc_sample(lp[:, n, :], tg, Tn, blank)
        go = 1.0 / (N * max(Ln, 1)) if reduction == "mean" else 1.0
        with np.errstate(invalid="ignore", over="ignore"):
            g = (np.exp(lp[:Tn, n, :]) - np.exp(raw[:Tn] + nll[n] - lp[:Tn, n, :])) * go
        grad[:Tn, n, :] = g
    if reduction == "mean":
        tl = np.maximum(np.asarray(target_lengths, dtype=np.float64), 1.0)
        return float(np.mean(nll / tl)), grad
    if reduction == "sum":
        return float(nll.sum()), grad
    return nll, grad


def logsoftmax_bwd_scrub(grad_lp, lp):
    """log_softmax backward followed by the reference's NaN scrub (models/model_crnn.py:30-32):
    dlogits = g - exp(lp) * sum_c g ; NaN -> 0."""
    g = np.asarray(grad_lp, dtype=np.float64)
    with np.errstate(invalid="ignore"):
        d = g - np.exp(lp) * g.sum(axis=2, keepdims=True)
    d[np.isnan(d)] = 0.0
    return d


# ----------------------------------------------------------------------------- jitter / selection / crops
def jitter(image, noise, noise_coef=1.0):
    """AddGaussianNoice.__call__ with the drawn noise supplied (transform_helper.py:33-45):
    out = clamp(image - coef*noise, 0, 1)."""
    return np.clip(np.asarray(image, dtype=np.float32) - np.float32(noise_coef) * np.asarray(noise, dtype=np.float32), 0.0, 1.0)


def jitter_sigma(std, stochastic_draw=None):
    """sigma of transform_helper.py:34-38: randint(0..std)/100 if stochastic else std/100, + 1e-13."""
    base = (stochastic_draw if stochastic_draw is not None else std) / 100.0
    return base + 0.0000000000001


def num_bb_samples(n, prop):
    """k of train_nn_patch.py:255-256 / train_nn_area.py:220-223."""
    return max(1, math.ceil(n * (1 - prop)))


def topk_desc_stable(cers, k):
    """Indices of the k largest CERs, descending, ties in ascending index order (the stable
    reading of selection_utils.py:144-151; see SURVEY.md F4 on torch's unstable default)."""
    c = np.asarray(cers, dtype=np.float32)
    order = np.argsort(-c, kind="stable")
    return order[:k].astype(np.int64)


def topk_query(cers_dict, names, k):
    """TopKCERSampler.query index logic (selection_utils.py:144-151): names missing from the dict
    are skipped BEFORE ranking, so returned indices address the compacted list."""
    vals = [cers_dict[n] for n in names if n in cers_dict]
    return topk_desc_stable(vals, k)


def padder(crop, h, w):
    """utils.py:118-125 — centre a [C,ch,cw] crop on a white (1.0) h x w canvas.  ConstantPad2d with a NEGATIVE pad
    crops: left = floor((w - cw) / 2) (Python floor division), so an oversize crop loses |left| columns on the left and
    the rest on the right; out[y, x] = crop[y - top, x - left] wherever that index exists, else 1."""
    _, ch, cw = crop.shape
    left = (w - cw) // 2
    top = (h - ch) // 2
    out = np.ones((crop.shape[0], h, w), dtype=crop.dtype)
    ys, xs = np.arange(h) - top, np.arange(w) - left
    oky, okx = (ys >= 0) & (ys < ch), (xs >= 0) & (xs < cw)
    out[:, np.ix_(oky, okx)[0], np.ix_(oky, okx)[1]] = crop[:, ys[oky]][:, :, xs[okx]]
    return out


def text_stack(image, boxes, size):
    """utils.py:128-141 — crop each box out of a [C,H,W] image, pad, stack."""
    crops, labels = [], []
    for b in boxes:
        crops.append(padder(image[:, b["y_min"]:b["y_max"], b["x_min"]:b["x_max"]], *size))
        labels.append(b["label"])
    return np.stack(crops), labels


# ----------------------------------------------------------------------------- decode / CER
def greedy_decode(scores, index_to_char):
    """utils.py:74-92 — argmax per step, collapse repeats, drop blank(0).  scores [T,B,C]."""
    idx = np.argmax(np.asarray(scores), axis=2)     # first max wins, like torch.argmax on CPU
    out = []
    for b in range(idx.shape[1]):
        s, prev = "", None
        for t in range(idx.shape[0]):
            k = int(idx[t, b])
            # reference quirk (:85-88): the first emitted char ignores the previous index
            if k != 0 and (len(s) == 0 or k != prev):
                s += index_to_char[k]
            prev = k
        out.append(s)
    return out


def levenshtein(a, b):
    """Unit-cost edit distance (python-Levenshtein==0.12.0 `distance`, call site utils.py:106)."""
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def compare_labels(preds, labels):
    """utils.py:95-110 -> (exact-match count, sum of distance/max(1,len(label)))."""
    correct, total = 0, 0.0
    for p, l in zip(preds, labels):
        correct += int(p == l)
        total += levenshtein(l, p) / max(1, len(l))
    return correct, total


# ----------------------------------------------------------------------------- Adam
def adam_update(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    """torch.optim.Adam single-tensor math (train_nn_patch.py:146-152: L2 folded into the grad).
    float64 numpy; `step` is the 1-based step count AFTER increment.  Returns new (p, m, v)."""
    p = np.asarray(p, dtype=np.float64)
    g = np.asarray(g, dtype=np.float64)
    if weight_decay:
        g = g + weight_decay * p
    m = beta1 * np.asarray(m, dtype=np.float64) + (1 - beta1) * g
    v = beta2 * np.asarray(v, dtype=np.float64) + (1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = np.sqrt(v) / math.sqrt(bc2) + eps
    return p - (lr / bc1) * m / denom, m, v
