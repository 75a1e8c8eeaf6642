"""TEST INFRASTRUCTURE ONLY — CPU oracle of the training-step choreography.

Phase A / Phase B of the preprocessor-training inner loop, area flavour
(reference train_nn_area.py:212-287; patch flavour train_nn_patch.py:225-345 differs in
where backward sits — SURVEY.md F6), on the functional oracle networks, torch CPU autograd,
ATen's CPU ctc_loss and torch.optim.Adam.  Pinned by tests/golden/step_area_b4.npz.
"""
import torch
import torch.nn.functional as F

from . import model_oracle as mo
from . import path_oracle as po

CHAR_SET_LEN = 95


def encode_labels(labels, char_to_index):
    """train_nn_patch.py:170-174 — concatenated int32 targets + lengths (CPU)."""
    y = torch.tensor([char_to_index[c] for c in "".join(labels)], dtype=torch.int)
    return y, torch.tensor([len(l) for l in labels], dtype=torch.int)


def ctc_mean(lp, y, ysz):
    ins = torch.full((lp.shape[1],), lp.shape[0], dtype=torch.int)
    return F.ctc_loss(lp, y, ins, ysz, blank=0, reduction="mean", zero_infinity=False)


class OracleTrainer:
    """Holds oracle UNet/CRNN states + two Adam optimisers, exposes phase_a / phase_b."""

    def __init__(self, unet_state, crnn_state, char_to_index, lr_crnn=1e-4, lr_prep=5e-5, weight_decay=0.0,
                 scalar=1.0):
        self.Pu, self.Bu = mo.split_state(unet_state)
        self.Pc, self.Bc = mo.split_state(crnn_state)
        self.c2i = char_to_index
        self.scalar = scalar
        self.opt_c = torch.optim.Adam(list(self.Pc.values()), lr=lr_crnn, weight_decay=weight_decay)
        self.opt_p = torch.optim.Adam(list(self.Pu.values()), lr=lr_prep, weight_decay=weight_decay)

    def zero(self):
        for p in list(self.Pu.values()) + list(self.Pc.values()):
            p.grad = None

    def phase_a(self, x, labels_for, names, cers, prop, noises, backward_every_replica=False):
        """UNet eval fwd -> TopKCER -> jitter replicas (explicit noise tensors) -> label source
        `labels_for(noisy_imgs, selected_indices)` (the black box) -> CRNN(train BN) -> CTC ->
        backward (last replica only unless backward_every_replica: area vs patch, F6) -> Adam."""
        self.zero()
        with torch.no_grad():
            preds_all = mo.unet_forward(self.Pu, self.Bu, x, training=False)
        k = po.num_bb_samples(x.shape[0], prop)
        idx = torch.from_numpy(po.topk_query(cers, names, k))
        preds = preds_all[idx]
        losses = []
        loss = None
        for noise in noises:
            noisy = (preds - noise).clamp(0, 1)
            labels = labels_for(noisy, idx)
            lp = mo.crnn_forward(self.Pc, self.Bc, noisy, bn_training=True)
            y, ysz = encode_labels(labels, self.c2i)
            loss = ctc_mean(lp, y, ysz)
            losses.append(loss.item())
            if backward_every_replica:
                loss.backward()
        if not backward_every_replica and loss is not None:
            loss.backward()
        self.opt_c.step()
        return idx, losses

    def phase_b(self, x, labels, step_crnn=False):
        """UNet(train BN) -> CRNN(train, BN eval) -> CTC(GT) + scalar*MSE(img, 1) -> backward ->
        Adam(UNet) [+ Adam(CRNN) if --update_CRNN]."""
        self.zero()
        img = mo.unet_forward(self.Pu, self.Bu, x, training=True)
        lp = mo.crnn_forward(self.Pc, self.Bc, img, bn_training=False)
        y, ysz = encode_labels(labels, self.c2i)
        loss = ctc_mean(lp, y, ysz) + F.mse_loss(img, torch.ones_like(img)) * self.scalar
        loss.backward()
        if step_crnn:
            self.opt_c.step()
        self.opt_p.step()
        return loss.item(), img.detach(), lp.detach()
