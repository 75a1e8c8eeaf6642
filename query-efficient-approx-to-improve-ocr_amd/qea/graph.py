"""hipGraph capture of a whole training step (torch.cuda.CUDAGraph drives hipStreamBeginCapture on ROCm).

The C-ABI kernels are launched on torch's current stream, so they are recorded like any other work; the weight-gradient
side stream forks from and joins the capturing stream through events and becomes part of the graph.  What a step must
satisfy to be capturable: static shapes and device-resident inputs (CTC targets / lengths as CUDA tensors with
`CTCLoss.max_target_length` set), `FusedAdam(capturable=True)` (step count on the device), no `.item()` / `.cpu()`.
At B >= 512 the step is GPU-bound and a graph buys nothing; at the reference's real batch sizes (tens of strips) the
step is bound by ~11 us of host time per launch and the replay removes it.
"""
import torch


class GraphedStep:
    def __init__(self, fn, warmup=3):
        """fn(): one full step on static tensors; its return value (tensor or tuple of tensors) is static too."""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                    # warm-up off the default stream, as torch's graph recipe asks
            for _ in range(warmup):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        from . import ops
        self.graph = torch.cuda.CUDAGraph()
        ops.CAPTURE["token"] = object()                  # derived weight forms: built once per weight epoch inside the graph
        try:
            with torch.cuda.graph(self.graph):
                self.out = fn()
        finally:
            ops.CAPTURE["token"] = None

    def __call__(self):
        self.graph.replay()
        # a captured FusedAdam writes the weights through raw pointers and runs no Python on replay, so neither the tensors'
        # _version nor the optimiser's epoch bump moves: every derived weight form cached by an EARLIER eager pass
        # (qea.ops.weight_cached: dgrad filters, planes, W_hh packs) is stale from here on
        from . import ops
        ops.bump_weight_epoch()
        return self.out


class PhaseBGraphs:
    """Phase B of the area trainer (train_nn_area.py:273-287) as one hipGraph per (batch size, width, target-length cap).

    trainer: a TrainerCore on the HIP backend whose optimisers were built with capturable=True.  The first `eager_steps` steps of
    a shape run eagerly (they create every lazily allocated buffer: workspaces, the weight-gradient side stream, Adam's flat
    moments and device step counter); the next one is captured and every later one replays it.  Inputs are copied into static
    buffers (images on the device already; targets padded to batch x cap int32), outputs (loss, log-probs, cleaned images) are static
    tensors valid until the next replay.  A batch whose longest label exceeds the cap of its graph gets a new graph with a
    larger cap."""

    def __init__(self, trainer, eager_steps=2):
        self.t = trainer
        self.eager_steps = eager_steps
        self.seen = {}
        self.graphs = {}

    def _cap(self, labels):
        need = max(1, max(len(l) for l in labels))
        return ((need + 7) // 8) * 8

    def step(self, X, labels):
        """Returns (loss, scores, img_preds) of one Phase-B step on X [B,1,H,W] (device) with ground-truth `labels`, or None when the
        step has to run eagerly (shape still warming up)."""
        t = self.t
        B, _, Hh, W = X.shape
        shape = (B, Hh, W)
        n = self.seen.get(shape, 0)
        self.seen[shape] = n + 1
        if n < self.eager_steps:
            return None
        cap = self._cap(labels)
        lr = float(t.optimizer_prep.param_groups[0]["lr"])          # baked into the captured Adam launch
        key = None
        for k in self.graphs:
            if k[:3] == shape and k[4] == lr and k[3] >= cap and (key is None or k[3] < key[3]):
                key = k
        if key is None:
            key = shape + (cap, lr)
            self.graphs[key] = self._capture(X, key)
        g = self.graphs[key]
        g["X"].copy_(X)
        L = key[3]
        y = torch.zeros(B * L, dtype=torch.int32)
        ysz = torch.tensor([len(l) for l in labels], dtype=torch.int32)
        flat = [t.char_to_index[c] for c in "".join(labels)]
        y[:len(flat)] = torch.tensor(flat, dtype=torch.int32)
        g["y"].copy_(y, non_blocking=False)
        g["ysz"].copy_(ysz, non_blocking=False)
        return g["step"]()

    def _capture(self, X, key):
        t = self.t
        B, Hh, W, L, _lr = key
        dev = X.device
        st = {"X": torch.empty_like(X), "y": torch.zeros(B * L, dtype=torch.int32, device=dev),
              "ysz": torch.ones(B, dtype=torch.int32, device=dev)}
        ctc = type(t.primary_loss_fn)()
        ctc.max_target_length = L

        def fn():
            t._set_phase_b()
            img = t.prep_model(st["X"])
            scores = t.crnn_model(img)
            pred = torch.full((B,), scores.shape[0], dtype=torch.int32, device=dev)
            loss = ctc(scores, st["y"], pred, st["ysz"]) + t.secondary_loss_fn(img, torch.ones_like(img)) * t.sec_loss_scalar
            loss.backward()
            t._step_prep()
            return loss.detach(), scores.detach(), img.detach()

        st["step"] = GraphedStep(fn, warmup=0)
        return st


class PhaseAGraphs:
    """The CRNN side of Phase A of the area trainer (train_nn_area.py:237-275: CRNN forward on the R jittered copies of the k picked strips,
    CTC against the black box's labels of the LAST copy, backward of that copy, Adam(CRNN)) as one hipGraph per (k, R, width, target cap).
    The jitter (its Philox seed and call counter are kernel arguments) and the black box (a host call) stay outside: the graph starts from
    the noisy strips.  Same warm-up rule and capturable-Adam requirement as PhaseBGraphs."""

    def __init__(self, trainer, eager_steps=2):
        self.t = trainer
        self.eager_steps = eager_steps
        self.seen = {}
        self.graphs = {}

    def step(self, noisy, labels_last, R):
        """noisy [R*k,1,H,W] (device, replica-major), labels_last: the k black-box labels of the last replica.  Returns the loss (device
        scalar) after the graph has run the backward and the CRNN update, or None when the step has to run eagerly."""
        t = self.t
        RK, _, Hh, W = noisy.shape
        shape = (RK, R, Hh, W)
        n = self.seen.get(shape, 0)
        self.seen[shape] = n + 1
        if n < self.eager_steps:
            return None
        k = RK // R
        cap = ((max(1, max(len(l) for l in labels_last)) + 7) // 8) * 8
        lr = float(t.optimizer_crnn.param_groups[0]["lr"])          # baked into the captured Adam launch: a scheduler step means a new graph
        key = None
        for kk in self.graphs:
            if kk[:4] == shape and kk[5] == lr and kk[4] >= cap and (key is None or kk[4] < key[4]):
                key = kk
        if key is None:
            key = shape + (cap, lr)
            self.graphs[key] = self._capture(noisy, key)
        g = self.graphs[key]
        g["X"].copy_(noisy)
        L = key[4]
        y = torch.zeros(k * L, dtype=torch.int32)
        flat = [t.char_to_index[c] for c in "".join(labels_last)]
        y[:len(flat)] = torch.tensor(flat, dtype=torch.int32)
        g["y"].copy_(y)
        g["ysz"].copy_(torch.tensor([len(l) for l in labels_last], dtype=torch.int32))
        return g["step"]()

    def _capture(self, noisy, key):
        t = self.t
        RK, R, Hh, W, L, _lr = key
        k = RK // R
        dev = noisy.device
        st = {"X": torch.empty_like(noisy), "y": torch.zeros(k * L, dtype=torch.int32, device=dev), "ysz": torch.ones(k, dtype=torch.int32, device=dev)}
        ctc = type(t.primary_loss_fn)()
        ctc.max_target_length = L

        def fn():
            t.crnn_model.zero_grad()
            scores = t.crnn_model(st["X"], replica_groups=R, backward_group=R - 1)
            pred = torch.full((k,), scores.shape[0], dtype=torch.int32, device=dev)
            loss = ctc(scores[:, (R - 1) * k:, :], st["y"], pred, st["ysz"])
            loss.backward()
            t._step_crnn()
            return loss.detach()

        st["step"] = GraphedStep(fn, warmup=0)
        return st
