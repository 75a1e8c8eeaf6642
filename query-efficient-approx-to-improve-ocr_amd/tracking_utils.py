"""Label-history ("label tracking") helpers of the `--inner_limit_skip` branch — same functions and
argument meaning as the reference's root tracking_utils.py (:5-81); the CRNN forward and every CTC
evaluation run on the HIP path."""
import torch


def call_crnn(self, images):
    scores = self.crnn_model(images.to(self.device))
    out_size = torch.tensor([scores.shape[0]] * images.shape[0], dtype=torch.int)
    return scores, out_size


def str_to_tensor(self, words):
    """history words -> [window_size, max_char_len] index tensor, padded with the out-of-vocabulary index
    len(char_set) (characters inside a word, and whole missing words) — reference tracking_utils.py:13-31."""
    import properties
    pad = len(properties.char_set)
    rows = [[self.char_to_index[c] for c in w] + [pad] * max(0, properties.max_char_len - len(w)) for w in words]
    rows += [[pad] * properties.max_char_len] * max(0, self.window_size - len(words))
    return torch.tensor(rows).to(self.device)


def generate_ctc_label(self, labels):
    y_size = torch.tensor([len(l) for l in labels], dtype=torch.int)
    y = torch.tensor([self.char_to_index[c] for c in "".join(labels)], dtype=torch.int)
    return y, y_size


def generate_ctc_target_batches(self, img_names):
    """For history depth i = 0..window-1: the i-th most recent OCR label of every strip that has one."""
    batches = []
    for depth in range(self.window_size):
        picked = [(j, self.tracked_labels[n][-(depth + 1)]) for j, n in enumerate(img_names) if depth < len(self.tracked_labels[n])]
        if picked:
            target, target_size = generate_ctc_label(self, [l for _, l in picked])
            batches.append([target, target_size, [j for j, _ in picked]])
    return batches


def weighted_ctc_loss(self, scores, pred_size, target_batches, loss_weights):
    losses = []
    for i in range(min(len(target_batches), self.window_size)):
        target, target_size, idx = target_batches[i]
        sub = scores[:, idx, :]
        if self.weightgen_method == "decaying":
            losses.append(loss_weights[i] * self.primary_loss_fn(sub, target, pred_size[idx], target_size))
        else:
            per = self.primary_loss_fn_sample_wise(sub, target, pred_size[idx], target_size)
            losses.append(torch.mean(loss_weights[idx, i] * per))
    return sum(losses)


def add_labels_to_history(self, image_keys, ocr_labels):
    for i, name in enumerate(image_keys):
        self.tracked_labels.setdefault(name, []).append(ocr_labels[i])
