"""Shared machinery of the two preprocessor trainers (train_nn_area.py / train_nn_patch.py).

What the reference keeps in two near-identical 400-line scripts — model/optimiser set-up, the
`_call_model` / `_get_loss` helpers, validation, checkpoints and experiment-directory files — lives
here once; the two trainer classes add only their data flow and Phase-A/Phase-B loops.

`Backend` is the seam between host logic and arithmetic: the default one is the HIP path
(models.*, qea.loss.CTCLoss, qea.optim.FusedAdam on cuda:LOCAL_RANK) and has NO CPU variant; tests
inject a backend built on the CPU oracle to exercise the host logic without a GPU.
"""
import json
import math
import os
import shutil

import torch

import properties
from qea import dist as qdist
from utils import (compare_labels, create_dirs, get_char_maps, get_ocr_helper, pred_to_string, save_all_jsons, set_bn_eval,
                   set_random_seeds)


class Backend:
    def __init__(self, unet_cls, crnn_cls, ctc_cls, adam_cls, device, gpu_jitter):
        self.UNet, self.CRNN, self.CTCLoss, self.Adam = unet_cls, crnn_cls, ctc_cls, adam_cls
        self.device, self.gpu_jitter = device, gpu_jitter


def hip_backend():
    if not torch.cuda.is_available():
        raise RuntimeError("the product trainers run on the MI355X HIP path only (torch.cuda is not available); "
                           "there is no CPU implementation — tests inject the CPU oracle explicitly")
    from models.model_crnn import CRNN
    from models.model_unet import UNet
    from qea import _lib
    from qea.loss import CTCLoss
    from qea.optim import FusedAdam
    _lib.lib()                                                   # fail now, loudly, if libqea_hip.so is missing
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    return Backend(UNet, CRNN, CTCLoss, FusedAdam, torch.device("cuda", local), gpu_jitter=True)


class _NullLogger:
    """wandb is optional (reference default: mode 'disabled', wandb_config.json:1-5)."""

    def log(self, *_a, **_k):
        pass

    def save(self, *_a, **_k):
        pass

    def summary_update(self, *_a, **_k):
        pass


def _logger():
    try:
        import wandb
        if wandb.run is None:
            return _NullLogger()

        class _W(_NullLogger):
            def log(self, d):
                wandb.log(d)

            def save(self, p):
                wandb.save(p)

            def summary_update(self, d):
                wandb.run.summary.update(d)
        return _W()
    except ImportError:
        return _NullLogger()


class GraphedLoss:
    """A loss whose backward and optimiser step already ran inside a hipGraph replay (qea.graph.PhaseAGraphs): .item() reads it,
    .backward() and multiplication by the data-parallel share are no-ops the caller may still perform."""
    graphed = True

    def __init__(self, value):
        self.value = value

    def item(self):
        return float(self.value.item())

    def backward(self):
        return None

    def __mul__(self, _):
        return self


class TrainerCore:
    """Everything the two TrainNNPrep classes share.  Attribute names follow the reference's."""

    def _setup_common(self, args, backend, ocr, weight_decay):
        self.backend = backend or hip_backend()
        self.device = self.backend.device
        self.lr_crnn, self.lr_prep = args.lr_crnn, args.lr_prep
        self.max_epochs, self.warmup_epochs = args.epoch, args.warmup_epochs
        self.inner_limit, self.inner_limit_skip = args.inner_limit, args.inner_limit_skip
        self.sec_loss_scalar, self.ocr_name = args.scalar, args.ocr
        self.std, self.is_random_std = args.std, args.random_std
        self.start_epoch = args.start_epoch
        self.selection_method = args.minibatch_subset
        self.train_subset_size, self.val_subset_size = args.train_subset_size, args.val_subset_size
        self.window_size, self.weightgen_method = args.window_size, args.weightgen_method
        create_dirs(self, args)
        set_random_seeds(args.random_seed)
        self.world = qdist.init_from_env(self.device)
        self.rank = qdist.rank()
        self.log = _logger()

        self.train_batch_prop = 1
        if args.minibatch_subset_prop is not None and self.selection_method:
            self.train_batch_prop = args.minibatch_subset_prop
        self.cers, self.selected_samples = None, dict()
        if args.cers_ocr_path:
            with open(args.cers_ocr_path, "r") as f:
                self.cers = json.load(f)
            self.selected_samples = {k: [False] * self.max_epochs for k in self.cers}
        self.tracked_labels = {name: [] for name in self.cers} if self.cers else {}

        self.char_to_index, self.index_to_char, self.vocab_size = get_char_maps(properties.char_set)
        self.input_size = properties.input_size
        self.ocr = ocr if ocr is not None else get_ocr_helper(self.ocr_name)

        B = self.backend
        self.crnn_model = (B.CRNN(self.vocab_size, False) if self.crnn_model_path is None
                           else torch.load(self.crnn_model_path, weights_only=False)).to(self.device)
        self.crnn_model.register_backward_hook(self.crnn_model.backward_hook)
        self.prep_model = (B.UNet() if self.prep_model_path is None
                           else torch.load(self.prep_model_path, weights_only=False)).to(self.device)
        if self.world > 1:                                       # identical start on every rank
            for m in (self.crnn_model, self.prep_model):
                for t in list(m.parameters()) + list(m.buffers()):
                    torch.distributed.broadcast(t.data, 0)
            from qea import ops
            ops.bump_weight_epoch()                              # raw writes through .data: derived weight forms are stale

        from label_tracking import tracking_methods
        self.loss_wghts_gnrtr = tracking_methods.weightgenerator_factory(args.weightgen_method)(args, self.device, self.char_to_index)
        self.primary_loss_fn = B.CTCLoss().to(self.device)
        self.primary_loss_fn_sample_wise = B.CTCLoss(reduction="none").to(self.device)
        self.secondary_loss_fn = torch.nn.MSELoss().to(self.device)
        # [new] --graph: Phase B replayed as a hipGraph (qea.graph.PhaseBGraphs); needs Adam's step count on the device
        self.phase_b_graphs = None
        self.phase_a_graphs = None
        adam_kw = {}
        if getattr(args, "graph", False) and self.world == 1 and self.device.type == "cuda" and self.backend.gpu_jitter:
            adam_kw = {"capturable": True}
        self.optimizer_crnn = B.Adam(self.crnn_model.parameters(), lr=self.lr_crnn, weight_decay=weight_decay, **adam_kw)
        self.optimizer_prep = B.Adam(self.prep_model.parameters(), lr=self.lr_prep, weight_decay=weight_decay, **adam_kw)
        if adam_kw:
            from qea.graph import PhaseAGraphs, PhaseBGraphs
            self.phase_b_graphs = PhaseBGraphs(self)
            self.phase_a_graphs = PhaseAGraphs(self)

    def _make_sampler(self, needs_cers):
        from selection_utils import datasampler_factory
        if not self.selection_method:
            self.sampler = None
            return
        cls = datasampler_factory(self.selection_method)
        self.sampler = cls(self.cers) if needs_cers else cls()

    # ---- reference helpers (train_nn_patch.py:158-191) ----
    def _call_model(self, images, labels):
        scores = self.crnn_model(images.to(self.device))
        out_size = torch.tensor([scores.shape[0]] * images.shape[0], dtype=torch.int)
        y_size = torch.tensor([len(l) for l in labels], dtype=torch.int)
        y = torch.tensor([self.char_to_index[c] for c in "".join(labels)], dtype=torch.int)
        return scores, y, out_size, y_size

    def _get_loss(self, scores, y, pred_size, y_size, img_preds):
        pri = self.primary_loss_fn(scores, y, pred_size, y_size)
        sec = self.secondary_loss_fn(img_preds, torch.ones(img_preds.shape, device=img_preds.device)) * self.sec_loss_scalar
        return pri + sec

    def _jitter(self, imgs, noiser):
        """One jittered copy of every image.  HIP: one Philox kernel for the whole stack (images stay in
        HBM); injected CPU backend: the reference's per-image loop (train_nn_patch.py:187-191)."""
        if self.backend.gpu_jitter and imgs.is_cuda:
            out, _ = noiser.batch(imgs, replicas=1)
            return out
        res = [noiser(img) for img in imgs]
        return torch.stack([r[0] if isinstance(r, tuple) else r for r in res])

    def _replica_losses(self, imgs, noiser, R, last_only=False):
        """The R jittered CRNN passes of Phase A (train_nn_patch.py:288-294 repeated R times).
        HIP path: ONE Philox launch makes all R*k noisy strips, ONE black-box call labels them, ONE CRNN pass with
        per-replica-group BatchNorm scores them; the per-replica CTC losses come back as a list (same values,
        same running-stat updates as R sequential passes — tests/test_models_gpu.py::test_replica_groups...).
        Injected CPU backend: the reference's sequential loop."""
        k = imgs.shape[0]
        if R <= 0:
            return [], 0
        if self.backend.gpu_jitter and imgs.is_cuda and R > 1:
            noisy, _ = noiser.batch(imgs, replicas=R)
            ocr_labels = self.ocr.get_labels(noisy.cpu())
            if last_only and self.phase_a_graphs is not None and self.world == 1:
                # [new] --graph: CRNN forward on the R copies, CTC of the last copy, its backward and Adam(CRNN) as ONE hipGraph replay; the
                # caller sees a loss that is already back-propagated and applied (GraphedLoss)
                done = self.phase_a_graphs.step(noisy, ocr_labels[(R - 1) * k:], R)
                if done is not None:
                    return [GraphedLoss(done)], R * k
            # last_only (area flow): only the last replica's loss is back-propagated -> its samples are the only ones the backward visits
            scores = self.crnn_model(noisy, replica_groups=R, backward_group=R - 1 if last_only else None)
            out_size = torch.tensor([scores.shape[0]] * k, dtype=torch.int)
            losses = []
            for r in range(R):
                labels = ocr_labels[r * k:(r + 1) * k]
                y = torch.tensor([self.char_to_index[c] for c in "".join(labels)], dtype=torch.int)
                y_size = torch.tensor([len(l) for l in labels], dtype=torch.int)
                losses.append(self.primary_loss_fn(scores[:, r * k:(r + 1) * k, :], y, out_size, y_size))
            return losses, R * k
        losses = []
        for _ in range(R):
            noisy = self._jitter(imgs, noiser)
            ocr_labels = self.ocr.get_labels(noisy.cpu())
            scores, y, pred_size, y_size = self._call_model(noisy, ocr_labels)
            losses.append(self.primary_loss_fn(scores, y, pred_size, y_size))
        return losses, R * k

    def _replica_losses_docs(self, crops_list, noiser, R):
        """[new] Phase A of SEVERAL documents in one CRNN pass (--docs_per_step): the R jittered copies of every document's strips,
        document-major and replica-minor, with one BatchNorm group per (document, replica) — ragged groups, CRNN.forward(group_sizes=...)
        — so that values, gradients and running statistics are those of the reference's sequential loop (train_nn_patch.py:288-303 per
        document, one document after the other).  Returns ([per-document list of R losses], black-box calls).  HIP path only."""
        noisy = torch.cat([noiser.batch(c, replicas=R)[0] for c in crops_list])
        ocr_labels = self.ocr.get_labels(noisy.cpu())
        sizes = [c.shape[0] for c in crops_list for _ in range(R)]
        scores = self.crnn_model(noisy, group_sizes=sizes)
        out, a = [], 0
        for c in crops_list:
            k = c.shape[0]
            out_size = torch.tensor([scores.shape[0]] * k, dtype=torch.int)
            losses = []
            for _ in range(R):
                labels = ocr_labels[a:a + k]
                y = torch.tensor([self.char_to_index[ch] for ch in "".join(labels)], dtype=torch.int)
                y_size = torch.tensor([len(l) for l in labels], dtype=torch.int)
                losses.append(self.primary_loss_fn(scores[:, a:a + k, :], y, out_size, y_size))
                a += k
            out.append(losses)
        return out, noisy.shape[0]

    def _num_bb_samples(self, n):
        return max(1, math.ceil(n * (1 - self.train_batch_prop)))

    def _set_phase_a(self):
        self.crnn_model.train()
        self.prep_model.eval()
        self.prep_model.zero_grad()
        self.crnn_model.zero_grad()

    def _set_phase_b(self):
        self.prep_model.train()
        self.crnn_model.train()
        self.crnn_model.apply(set_bn_eval)
        self.prep_model.zero_grad()
        self.crnn_model.zero_grad()

    def _step_crnn(self):
        qdist.allreduce_module_grads(self.crnn_model)
        self.optimizer_crnn.step()

    def _step_prep(self, also_crnn=False):
        qdist.allreduce_module_grads(self.prep_model)
        if also_crnn:
            qdist.allreduce_module_grads(self.crnn_model)
            self.optimizer_crnn.step()
        self.optimizer_prep.step()

    # ---- epoch end: checkpoints in the reference layout (train_nn_patch.py:440-464) ----
    def _save_checkpoints(self, epoch, ocr_accuracy, best, save_optim):
        if self.rank != 0:
            return best
        prep_ckpt = os.path.join(self.ckpt_base_path, f"Prep_model_{epoch}_{ocr_accuracy * 100:.2f}")
        torch.save(self.prep_model, prep_ckpt)
        torch.save(self.crnn_model, os.path.join(self.ckpt_base_path, "CRNN_model_" + str(epoch)))
        if save_optim:
            torch.save(self.optimizer_prep.state_dict(), os.path.join(self.ckpt_base_path, "optim_prep_latest"))
            torch.save(self.optimizer_crnn.state_dict(), os.path.join(self.ckpt_base_path, "optim_crnn_latest"))
        best_acc, best_epoch = best
        if ocr_accuracy > best_acc:
            best_acc, best_epoch = ocr_accuracy, epoch
            best_path = os.path.join(self.ckpt_base_path, "Prep_model_best")
            shutil.copyfile(prep_ckpt, best_path)
            self.log.save(best_path)
            self.log.summary_update({"best_val_acc": best_acc, "best_val_epoch": best_epoch})
        return best_acc, best_epoch

    def _epoch_jsons(self, epoch):
        if self.selection_method and self.rank == 0:
            save_all_jsons(self, epoch)

    def _update_cers(self, scores, labels, names):
        """decode -> per-sample CER -> sampler.update_cer (train_nn_patch.py:330-342).  On the GPU the decode and
        the edit distances stay on the device (no [T,B,C] copy, no Python loop with an .item() per (b,t))."""
        if not (self.selection_method and len(names)):
            return
        if scores.is_cuda and scores.shape[0] <= 128:
            from utils import batch_cers
            cers = batch_cers(scores, labels, self.char_to_index)
        else:
            preds = pred_to_string(scores, labels, self.index_to_char)
            cers = [compare_labels([p], [l])[1] for p, l in zip(preds, labels)]
        self.sampler.update_cer(cers, names)
