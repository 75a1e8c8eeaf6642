"""Text-area preprocessor trainer — MI355X-native drop-in for the reference's train_nn_area.py.

`TrainNNPrep(args).train()` keeps the reference's constructor contract (an argparse Namespace with the
area_cli flags), its two-phase minibatch loop (train_nn_area.py:212-304), the experiment-directory
files and the whole-module checkpoints (:391-410), and returns (best_val_acc, best_val_epoch).

  Phase A: UNet(eval, no grad) -> TopKCER pick -> `inner_limit` x (jitter on the GPU -> black-box OCR
           labels -> CRNN(train BN) -> CTC); backward of the LAST replica only;
           [RCCL all-reduce of the flat CRNN gradient]; Adam(CRNN)
  Phase B: UNet(train) -> CRNN(train, BN eval) -> CTC(GT) + scalar*MSE(img, 1) -> backward;
           [RCCL all-reduce of the flat UNet gradient]; Adam(UNet); greedy decode -> CER -> sampler

Under torch.distributed every rank runs this loop on its own shard of the dataset (additive:
a single process behaves exactly like the reference).
"""
import os

import torch

import properties
from qea import dist as qdist
from qea.trainer_core import TrainerCore
from tracking_utils import add_labels_to_history, call_crnn, generate_ctc_target_batches, weighted_ctc_loss
from transform_helper import AddGaussianNoice
from utils import compare_labels, pred_to_string, save_img


class TrainNNPrep(TrainerCore):
    def __init__(self, args, backend=None, train_set=None, val_set=None, ocr=None):
        self.batch_size = args.batch_size
        self._setup_common(args, backend, ocr, weight_decay=0)
        self.train_batch_size = self.batch_size
        self._make_sampler(needs_cers=self.selection_method in ("topKCER", "rangeCER"))

        if train_set is None or val_set is None:
            train_set, val_set = self._default_datasets(args)
        self.train_set, self.validation_set = train_set, val_set
        if not self.train_subset_size:
            self.train_subset_size = len(train_set)
        if not self.val_subset_size:
            self.val_subset_size = len(val_set)
        tr_idx = torch.randperm(len(train_set))[: self.train_subset_size]
        widths = getattr(train_set, "widths", None)
        self.per_shard_topk = bool(getattr(args, "per_shard_topk", False))
        self.rebalance_topk = not bool(getattr(args, "no_rebalance_topk", False))
        self.select_before_clean = bool(getattr(args, "select_before_clean", False))
        if widths is not None:                                # [new] variable-width lines: one width bucket per batch
            # the bucket batches are formed over the WHOLE (rank-identical) list and dealt round-robin, so every rank
            # runs the same number of steps (qea.dist.deal_batches)
            from datasets.bucketing import BucketBatchSampler, bucket_collate
            sub = torch.utils.data.Subset(train_set, tr_idx.tolist())
            self.loader_train = torch.utils.data.DataLoader(
                sub, collate_fn=bucket_collate, batch_sampler=BucketBatchSampler([widths[i] for i in tr_idx.tolist()], self.batch_size,
                                                                                seed=args.random_seed))
        else:
            # data parallel: cut the (identically seeded) permutation to whole global batches, one group of batch_size per rank
            # and step: every rank runs the same number of optimiser steps (a rank with fewer would strand the others in RCCL)
            tr_idx = qdist.equal_shards(tr_idx, self.batch_size)
            self.loader_train = torch.utils.data.DataLoader(train_set, batch_size=self.batch_size, drop_last=True,
                                                            sampler=torch.utils.data.SubsetRandomSampler(tr_idx))
        va_idx = torch.randperm(len(val_set))[: self.val_subset_size]
        self.loader_validation = torch.utils.data.DataLoader(val_set, batch_size=self.batch_size, drop_last=True,
                                                             sampler=torch.utils.data.SubsetRandomSampler(va_idx))
        self.train_set_size, self.val_set_size = len(train_set), len(val_set)
        self.lr_scheduler = args.lr_scheduler
        if self.lr_scheduler == "cosine":
            self.scheduler_crnn = torch.optim.lr_scheduler.CosineAnnealingLR(self.optimizer_crnn, T_max=self.max_epochs)

    def _default_datasets(self, args):
        n = getattr(args, "synthetic_size", None)
        if n:
            from datasets.synthetic import SyntheticTextAreas
            return (SyntheticTextAreas(n, seed=1, include_name=True, include_index=True),
                    SyntheticTextAreas(max(self.batch_size, n // 4), seed=2, include_name=True))
        from datasets.img_dataset import ImgDataset
        from transform_helper import PadWhite
        tf = lambda img: _to_tensor(PadWhite(self.input_size)(img))
        return (ImgDataset(os.path.join(args.data_base_path, properties.vgg_text_dataset_train), transform=tf, include_name=True,
                           include_index=True),
                ImgDataset(os.path.join(args.data_base_path, properties.vgg_text_dataset_dev), transform=tf, include_name=True))

    def train(self):
        noiser = AddGaussianNoice(std=self.std, is_stochastic=self.is_random_std, return_noise=True)
        print(f"Batch size is {self.batch_size}")
        total_bb_calls, best = 0, (0, 0)
        self.crnn_model.zero_grad()
        for epoch in range(self.start_epoch, self.max_epochs):
            epoch_bb_calls, step, training_loss, CRNN_training_loss = 0, 0, 0.0, 0.0
            for images, labels, names, indices in self.loader_train:
                labels, names = list(labels), list(names)
                X_var = images.to(self.device)
                # ---------------- Phase A ----------------
                self._set_phase_a()
                select_first = (self.select_before_clean and self.selection_method and epoch >= self.warmup_epochs
                                and getattr(self.sampler, "content_free", False))
                if not select_first:
                    with torch.no_grad():
                        img_preds_all = self.prep_model(X_var)
                else:
                    img_preds_all = X_var                        # [new] pick on the inputs' names, clean only the picked images below
                share = 1.0                                      # this rank's weight in the data-parallel mean of Phase A
                rebalance = False
                if self.selection_method and epoch >= self.warmup_epochs:
                    if self.world > 1 and not self.per_shard_topk and hasattr(self.sampler, "query_global"):
                        # whole-minibatch TopKCER over the shards (train_nn_area.py:220-225 ranks the whole minibatch); the
                        # averaged gradient equals the global-batch mean when rank r weighs its mean loss by world * n_r / k
                        k = self._num_bb_samples(img_preds_all.shape[0] * self.world)
                        img_preds, labels_gt, bb_idx, k, counts = self.sampler.query_global(img_preds_all, labels, k, names, with_counts=True)
                        rebalance = self.rebalance_topk and not self.inner_limit_skip    # label histories are keyed by the owner's names
                        if not rebalance:
                            share = self.world * img_preds.shape[0] / max(1, k)
                    else:
                        k = self._num_bb_samples(img_preds_all.shape[0])
                        img_preds, labels_gt, bb_idx = self.sampler.query(img_preds_all, labels, k, names)
                    if select_first and img_preds.shape[0]:
                        with torch.no_grad():                    # eval-mode UNet: each image's output is independent of its batch
                            img_preds = self.prep_model(img_preds.contiguous())
                    img_preds = img_preds.detach()
                    img_preds_names = [names[i] for i in bb_idx.tolist()]
                    for name in img_preds_names:                 # bookkeeping stays with the rank that OWNS the strip
                        if name in self.selected_samples:
                            self.selected_samples[name][epoch] = True
                    if rebalance:
                        # [new] the k global winners are dealt out again in equal slices: no rank runs all of Phase A
                        # while the others wait in the all-reduce.  Phase A needs the cleaned images only (its labels come from
                        # the black box on the noisy copies), so images are all that travels.
                        img_preds = qdist.rebalance_rows(img_preds.contiguous(), counts)
                        img_preds_names = [None] * img_preds.shape[0]
                        share = self.world * img_preds.shape[0] / max(1, k)
                else:
                    img_preds, img_preds_names = img_preds_all.detach(), names
                loss = None
                n_skip = 1 if (self.inner_limit_skip and self.inner_limit > 0) else 0
                if img_preds.shape[0] == 0:                      # data parallel: none of the global winners lives in this shard
                    if self.inner_limit:
                        self._step_crnn()                        # still joins the all-reduce (with zero gradients)
                    n_skip = -1
                if n_skip > 0:                                   # iteration 0 without noise: label tracking (:247-259)
                    ocr_labels = self.ocr.get_labels(img_preds.cpu())
                    loss_weights = self.loss_wghts_gnrtr.gen_weights(self.tracked_labels, img_preds_names)
                    add_labels_to_history(self, img_preds_names, ocr_labels)
                    target_batches = generate_ctc_target_batches(self, img_preds_names)
                    scores, pred_size = call_crnn(self, img_preds)
                    loss = weighted_ctc_loss(self, scores, pred_size, target_batches, loss_weights)
                    total_bb_calls += len(ocr_labels)
                    epoch_bb_calls += len(ocr_labels)
                if n_skip >= 0:
                    rep_losses, calls = self._replica_losses(img_preds, noiser, self.inner_limit - n_skip, last_only=True)
                    total_bb_calls += calls
                    epoch_bb_calls += calls
                    if rep_losses:
                        loss = rep_losses[-1]
                    if self.inner_limit:
                        CRNN_training_loss = loss.item() / max(1, self.inner_limit)
                        if not getattr(loss, "graphed", False):    # ([new] --graph: backward and Adam(CRNN) ran inside the replay)
                            (loss * share if share != 1.0 else loss).backward()   # the last replica only, as the reference (:269-271)
                            self._step_crnn()
                # ---------------- Phase B ----------------
                replayed = self.phase_b_graphs.step(X_var, labels) if self.phase_b_graphs is not None else None
                if replayed is not None:                         # [new] --graph: the same step as one hipGraph replay
                    loss, scores, img_preds = replayed
                else:
                    self._set_phase_b()
                    img_preds = self.prep_model(X_var)
                    scores, y, pred_size, y_size = self._call_model(img_preds, labels)
                    loss = self._get_loss(scores, y, pred_size, y_size, img_preds)
                    loss.backward()
                    self._step_prep()
                self._update_cers(scores, labels, names)
                training_loss += loss.item()
                if step % 100 == 0:
                    print(f"Epoch: {epoch}, Iteration: {step} => {loss.item()}")
                step += 1
            self._epoch_jsons(epoch)
            if self.lr_scheduler:
                self.scheduler_crnn.step()
            val = self._validate(epoch)
            val.update({"Epoch": epoch + 1, "train_loss": training_loss / max(1, step), "Total Black-Box Calls": total_bb_calls,
                        "Black-Box Calls": epoch_bb_calls, "CRNN_loss": CRNN_training_loss / max(1, epoch_bb_calls)})
            self.log.log(val)
            print("Epoch: %d/%d => Training loss: %f | Validation loss: %f" % (epoch + 1, self.max_epochs, val["train_loss"], val["val_loss"]))
            best = self._save_checkpoints(epoch, val[f"{self.ocr_name}_accuracy"], best, save_optim=False)
        print("Training Completed.")
        return best

    def _validate(self, epoch=0):
        self.prep_model.eval()
        self.crnn_model.eval()
        cnt = dict(crnn=0, ocr=0, match=0)
        cer = dict(crnn=0.0, ocr=0.0, match=0.0)
        val_loss, nb, img_preds, images = 0.0, 0, None, None
        with torch.no_grad():
            for images, labels, names in self.loader_validation:
                labels = list(labels)
                img_preds = self.prep_model(images.to(self.device))
                scores, y, pred_size, y_size = self._call_model(img_preds, labels)
                val_loss += self._get_loss(scores, y, pred_size, y_size, img_preds).item()
                preds = pred_to_string(scores, labels, self.index_to_char)
                ocr_labels = self.ocr.get_labels(img_preds.cpu())
                for key, a, b in (("crnn", preds, labels), ("ocr", ocr_labels, labels), ("match", preds, ocr_labels)):
                    c, e = compare_labels(a, b)
                    cnt[key] += c
                    cer[key] += e
                nb += 1
        if self.rank == 0 and img_preds is not None:
            save_img(img_preds.cpu(), "out_" + str(epoch), self.img_out_path, 8)          # train_nn_area.py:373
            if epoch == 0:
                save_img(images.cpu(), "out_original", self.img_out_path, 8)              # :374-375
        n = max(1, nb * self.batch_size)
        return {"CRNN_accuracy": cnt["crnn"] / n, f"{self.ocr_name}_accuracy": cnt["ocr"] / n, "CRNN_CER": cer["crnn"] / n,
                f"{self.ocr_name}_cer": cer["ocr"] / n, "CRNN_OCR_Matching_ACC": cnt["match"] / n,
                "CRNN_OCR_Matching_CER": cer["match"] / n, "val_loss": val_loss / max(1, nb)}


def _to_tensor(pil_img):
    import numpy as np
    a = np.asarray(pil_img.convert("L"), dtype=np.float32) / 255.0
    return torch.from_numpy(a)[None]
