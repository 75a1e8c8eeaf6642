"""POS-patch preprocessor trainer — MI355X-native drop-in for the reference's train_nn_patch.py.

Same contract as the reference class (`TrainNNPrep(args, optuna_trial=None).train()`,
train_nn_patch.py:35-467).  Data flow: the UNet runs on a WHOLE padded document image
[1,1,400,512]; text strips are cropped out of its output and white-padded to 32x128 (a HIP gather
with a scatter-add backward) before the CRNN; the loader batch is one document.

  Phase A per document: UNet(eval) -> crops -> TopKCER pick -> `inner_limit` x (jitter -> OCR labels ->
           CRNN(train BN) -> CTC -> backward, gradients ACCUMULATE over the replicas: SURVEY F6);
           [all-reduce CRNN grads]; Adam(CRNN)
  Phase B per document: UNet(train) -> crops -> CRNN(BN eval) -> CTC(GT) + scalar*MSE(document, 1) ->
           backward; [all-reduce UNet (+CRNN if --update_CRNN) grads]; Adam
"""
import os

import torch

import properties
from qea import dist as qdist
from qea.trainer_core import TrainerCore
from tracking_utils import add_labels_to_history, call_crnn, generate_ctc_target_batches, weighted_ctc_loss
from transform_helper import AddGaussianNoice
from utils import compare_labels, get_text_stack, handle_optuna_trial, pred_to_string, save_img


class TrainNNPrep(TrainerCore):
    def __init__(self, args, optuna_trial=None, backend=None, train_set=None, val_set=None, ocr=None):
        self.optuna_trial = optuna_trial
        self.batch_size = max(1, int(getattr(args, "docs_per_step", 1) or 1))   # documents per step (reference: 1, train_nn_patch.py:37)
        self.random_seed = args.random_seed
        self.weight_decay = args.weight_decay
        self.update_CRNN = args.update_CRNN
        self._setup_common(args, backend, ocr, weight_decay=args.weight_decay)
        self._make_sampler(needs_cers=True)
        if args.optim_crnn_path:
            self.optimizer_crnn.load_state_dict(torch.load(args.optim_crnn_path, weights_only=False))
        if args.optim_prep_path:
            self.optimizer_prep.load_state_dict(torch.load(args.optim_prep_path, weights_only=False))

        if train_set is None or val_set is None:
            train_set, val_set = self._default_datasets(args)
        self.dataset, self.validation_set = train_set, val_set
        collate = getattr(type(train_set), "collate", None)
        if not self.train_subset_size:
            self.train_subset_size = len(train_set)
        if not self.val_subset_size:
            self.val_subset_size = len(val_set)
        idx = torch.randperm(len(train_set))[: self.train_subset_size]
        idx = qdist.equal_shards(idx, self.batch_size)          # data parallel: the same number of steps on every rank
        self._train_idx = idx
        self._collate = collate
        self.loader_train = torch.utils.data.DataLoader(train_set, batch_size=self.batch_size, drop_last=True, collate_fn=collate,
                                                        sampler=torch.utils.data.SubsetRandomSampler(idx))
        self.train_set_size, self.val_set_size = len(idx), len(val_set)
        self.num_subset_images = int(args.image_prop * self.train_set_size) if args.image_prop else None
        if self.cers:
            self.all_cers = {name: [] for name in self.cers}

    def _default_datasets(self, args):
        n = getattr(args, "synthetic_size", None)
        if n:
            from datasets.synthetic import SyntheticPatches
            return SyntheticPatches(n, seed=1), SyntheticPatches(max(1, n // 4), seed=2, include_name=False)
        from datasets.patch_dataset import PatchDataset
        return (PatchDataset(os.path.join(args.data_base_path, properties.patch_dataset_train), pad=True, include_name=True),
                PatchDataset(os.path.join(args.data_base_path, properties.patch_dataset_dev), pad=True, num_subset=self.val_subset_size))

    @staticmethod
    def _strip_names(labels, name):
        folder, file_name = name.split("/")[-2:]
        file_name = file_name.split(".")[0]
        return [f"{j}_{labels[j]}_{folder}_{file_name}" for j in range(len(labels))]

    def train(self):
        noiser = AddGaussianNoice(std=self.std, is_stochastic=self.is_random_std)
        step, total_bb_calls, best = 0, 0, (0, 0)
        for epoch in range(self.start_epoch, self.max_epochs):
            if self.selection_method and "global" in self.selection_method:
                self.sampler.select_samples()
            training_loss, epoch_bb_calls, CRNN_training_loss = 0.0, 0, 0.0
            if self.num_subset_images:
                # a fresh subset of THIS rank's documents per epoch (train_nn_patch.py:209-218); the draw is rank-identical
                # and every shard has the same length, so the step counts stay equal
                sub = self._train_idx[torch.randperm(self.train_set_size)[: self.num_subset_images]]
                self.loader_train = torch.utils.data.DataLoader(self.dataset, batch_size=self.batch_size, drop_last=True, collate_fn=self._collate,
                                                                sampler=torch.utils.data.SubsetRandomSampler(sub))
            for images, labels_dicts, names in self.loader_train:
                # ---------------- Phase A ----------------
                self._set_phase_a()
                strip_names = []
                n_docs = len(labels_dicts)
                X_all = (images if torch.is_tensor(images) else torch.stack(list(images))).to(self.device)
                with torch.no_grad():                            # eval-mode BatchNorm: a document's output does not depend on its batch
                    preds_all = self.prep_model(X_all) if n_docs > 1 else None
                # [new] the CRNN side of Phase A for the N documents in one pass (ragged BatchNorm groups) where nothing per document has to
                # happen in between: HIP path, more than one replica, no label-history pass (--inner_limit_skip runs per document)
                batched_a = (n_docs > 1 and self.backend.gpu_jitter and self.inner_limit > 1 and not self.inner_limit_skip
                             and self.device.type == "cuda" and getattr(self, "_batch_phase_a", True))
                pending = []
                for i in range(n_docs):
                    with torch.no_grad():
                        pred = preds_all[i] if preds_all is not None else self.prep_model(X_all[i:i + 1])[0]
                        crops_all, labels = get_text_stack(pred, labels_dicts[i], self.input_size)
                    n_strips = crops_all.shape[0]
                    strip_names = self._strip_names(labels, names[i])
                    local = self.selection_method and epoch >= self.warmup_epochs and "global" not in self.selection_method
                    if local:
                        k = self._num_bb_samples(n_strips)
                        crops, labels_gt, bb_idx = self.sampler.query(crops_all, labels, k, strip_names)
                        bb_idx = bb_idx[: crops.shape[0]]
                        crop_names = [strip_names[j] for j in bb_idx.tolist()]
                        for nm in crop_names:
                            if nm in self.selected_samples:
                                self.selected_samples[nm][epoch] = True
                    else:
                        crops, crop_names = crops_all, strip_names
                    crops = crops.detach()
                    approx_loss = 0.0
                    n_skip = 1 if (self.inner_limit_skip and self.inner_limit > 0) else 0
                    if n_skip:                                   # iteration 0 without noise: label tracking (:280-287)
                        ocr_labels = self.ocr.get_labels(crops.cpu())
                        loss_weights = self.loss_wghts_gnrtr.gen_weights(self.tracked_labels, crop_names)
                        add_labels_to_history(self, crop_names, ocr_labels)
                        target_batches = generate_ctc_target_batches(self, crop_names)
                        scores, pred_size = call_crnn(self, crops)
                        loss = weighted_ctc_loss(self, scores, pred_size, target_batches, loss_weights)
                        approx_loss += loss.item()
                        loss.backward()
                        total_bb_calls += crops.shape[0]
                        epoch_bb_calls += crops.shape[0]
                    if batched_a:                                # [new] all documents' replicas go through the CRNN together below
                        pending.append(crops)
                        continue
                    rep_losses, calls = self._replica_losses(crops, noiser, self.inner_limit - n_skip)
                    total_bb_calls += calls
                    epoch_bb_calls += calls
                    if rep_losses:
                        # every replica is back-propagated and the gradients accumulate (:301-303): backward of the sum
                        total = rep_losses[0]
                        for l in rep_losses[1:]:
                            total = total + l
                        approx_loss += total.item()
                        total.backward()
                    CRNN_training_loss += approx_loss / max(1, self.inner_limit)
                if pending:
                    # [new] --docs_per_step N: ONE jitter + black-box + CRNN pass for the N documents, one BatchNorm group per
                    # (document, replica) in the order of the sequential loop; the sum of all losses is back-propagated once
                    doc_losses, calls = self._replica_losses_docs(pending, noiser, self.inner_limit)
                    total_bb_calls += calls
                    epoch_bb_calls += calls
                    total = None
                    for losses in doc_losses:
                        for l in losses:
                            total = l if total is None else total + l
                    total.backward()
                    CRNN_training_loss += sum(float(l.item()) for losses in doc_losses for l in losses) / max(1, self.inner_limit)
                if self.inner_limit:
                    self._step_crnn()
                # ---------------- Phase B ----------------
                self._set_phase_b()
                if n_docs > 1:
                    # [new] --docs_per_step N: ONE cleaner pass over the N documents with per-document BatchNorm statistics (the
                    # values of N sequential passes), ONE CRNN pass over all their strips (BatchNorm in eval mode: batch-independent);
                    # the N per-document losses of :327-328 are summed and back-propagated once = the accumulated gradients of :329
                    img_all = self.prep_model(X_all, bn_groups=n_docs)
                    stacks = [get_text_stack(img_all[i], labels_dicts[i], self.input_size) for i in range(n_docs)]
                    crops = torch.cat([c for c, _ in stacks])
                    scores = self.crnn_model(crops)
                    total, a = None, 0
                    for i, (c, labels) in enumerate(stacks):
                        b = a + c.shape[0]
                        sc = scores[:, a:b, :]
                        y = torch.tensor([self.char_to_index[ch] for ch in "".join(labels)], dtype=torch.int)
                        pred_size = torch.tensor([sc.shape[0]] * c.shape[0], dtype=torch.int)
                        y_size = torch.tensor([len(l) for l in labels], dtype=torch.int)
                        loss = self._get_loss(sc, y, pred_size, y_size, img_all[i])
                        total = loss if total is None else total + loss
                        self._update_cers(sc, labels, self._strip_names(labels, names[i]))
                        training_loss += loss.item()
                        if step % 100 == 0:
                            print("Iteration: %d => %f" % (step, loss.item()))
                        step += 1
                        a = b
                    total.backward()
                else:
                    img_out = self.prep_model(X_all)[0]
                    crops, labels = get_text_stack(img_out, labels_dicts[0], self.input_size)
                    scores, y, pred_size, y_size = self._call_model(crops, labels)
                    loss = self._get_loss(scores, y, pred_size, y_size, img_out)
                    loss.backward()
                    self._update_cers(scores, labels, self._strip_names(labels, names[0]))
                    training_loss += loss.item()
                    if step % 100 == 0:
                        print("Iteration: %d => %f" % (step, loss.item()))
                    step += 1
                self._step_prep(also_crnn=self.update_CRNN)
            self._epoch_jsons(epoch)
            print(f"Epoch BB calls - {epoch_bb_calls}")
            val = self._validate(epoch)
            val.update({"Epoch": epoch + 1, "train_loss": training_loss / max(1, self.train_set_size),
                        "Total Black-Box Calls": total_bb_calls, "Black-Box Calls": epoch_bb_calls,
                        "CRNN_loss": CRNN_training_loss / max(1, epoch_bb_calls)})
            self.log.log(val)
            best = self._save_checkpoints(epoch, val[f"{self.ocr_name}_accuracy"], best, save_optim=True)
            handle_optuna_trial(self.optuna_trial, val[f"{self.ocr_name}_accuracy"], epoch)
        print("Training Completed.")
        return best

    def _validate(self, epoch):
        self.prep_model.eval()
        self.crnn_model.eval()
        cnt = dict(crnn=0, ocr=0, match=0)
        cer = dict(crnn=0.0, ocr=0.0, match=0.0)
        val_loss, n_strips, last = 0.0, 0, None
        with torch.no_grad():
            for k in range(len(self.validation_set)):
                item = self.validation_set[k]
                image, boxes = item[0], item[1]
                if not boxes:
                    continue
                img_out = self.prep_model(image.unsqueeze(0).to(self.device))[0]
                crops, labels = get_text_stack(img_out, boxes, self.input_size)
                scores, y, pred_size, y_size = self._call_model(crops, labels)
                val_loss += self._get_loss(scores, y, pred_size, y_size, img_out).item()
                preds = pred_to_string(scores, labels, self.index_to_char)
                ocr_labels = self.ocr.get_labels(crops.cpu())
                for key, a, b in (("crnn", preds, labels), ("ocr", ocr_labels, labels), ("match", preds, ocr_labels)):
                    c, e = compare_labels(a, b)
                    cnt[key] += c
                    cer[key] += e
                n_strips += len(labels)
                last = img_out
        if self.rank == 0 and last is not None:
            save_img(last.unsqueeze(0).cpu(), "out_" + str(epoch), self.img_out_path, 1)
        n = max(1, n_strips)
        return {"CRNN_accuracy": cnt["crnn"] / n, f"{self.ocr_name}_accuracy": cnt["ocr"] / n, "CRNN_CER": cer["crnn"] / n,
                f"{self.ocr_name}_cer": cer["ocr"] / n, "CRNN_OCR_Matching_ACC": cnt["match"] / n,
                "CRNN_OCR_Matching_CER": cer["match"] / n, "val_loss": val_loss / max(1, len(self.validation_set))}
