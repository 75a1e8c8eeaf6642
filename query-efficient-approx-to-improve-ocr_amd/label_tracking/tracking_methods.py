"""Loss-weight generators for the label-history CTC — same classes, constructor and `gen_weights(tracked_labels,
img_names)` protocol as the reference's label_tracking/tracking_methods.py:

  * decaying       (:105-115)  weights decay^i per history depth                      -> 1-D tensor [window]
  * levenshtein    (:63-101)   agreement of each remembered label with the others     -> [len(names), window+1]
  * self_attention (:26-59)    HistoryAttention scores of the remembered labels       -> [len(names), window+1]

All of it is host-side logic over at most `window_size` short strings per strip; the CTC evaluations the weights
multiply run on the HIP path (tracking_utils.weighted_ctc_loss)."""
import torch

import properties


class LossWeightGenerator:
    def __init__(self, tracking_args, device, char_to_index=None):
        self.window_size = tracking_args.window_size
        self.device = device
        self.char_to_index = char_to_index

    def print_debug_statements(self):
        pass

    def _recent(self, tracked_labels, name):
        """labels of `name` inside the window, most recent first"""
        return tracked_labels[name][-self.window_size:][::-1] if name in tracked_labels else []

    def _table(self, n):
        w = torch.zeros(n, self.window_size + 1)
        w[:, 0] = 1                                              # the label of the current epoch always counts fully
        return w


class DecayingWeightGenerator(LossWeightGenerator):
    def __init__(self, tracking_args, device, char_to_index=None):
        super().__init__(tracking_args, device, char_to_index)
        self.decay_factor = tracking_args.decay_factor

    def gen_weights(self, training_obj, img_names):
        return torch.tensor([self.decay_factor ** i for i in range(self.window_size)]).to(self.device)


class LevenshteinWeightGenerator(LossWeightGenerator):
    HIST_MULTIPLIER = 0.5

    def gen_weights(self, tracked_labels, img_names):
        from utils import levenshtein
        w = self._table(len(img_names))
        for row, name in enumerate(img_names):
            hist = self._recent(tracked_labels, name)
            others = max(len(hist) - 1, 1)
            for i, word in enumerate(hist):
                mean_dist = sum(levenshtein(word, o) for j, o in enumerate(hist) if j != i) / others
                n_chars = max(1, len(word))
                w[row, i + 1] = self.HIST_MULTIPLIER * (1 - min(mean_dist, n_chars) / n_chars)
        return w.to(self.device)


class AttentionWeightGenerator(LossWeightGenerator):
    def __init__(self, tracking_args, device, char_to_index):
        super().__init__(tracking_args, device, char_to_index)
        from models.model_attention import HistoryAttention
        # area_cli.py has no --query_dim / --emb_dim / --attn_activation (the reference's area trainer raises
        # AttributeError here); fall back to patch_cli's defaults
        self.query_dim, self.emb_dim = getattr(tracking_args, "query_dim", 32), getattr(tracking_args, "emb_dim", 256)
        self.attn_activation = getattr(tracking_args, "attn_activation", "sigmoid")
        self.attention_model = HistoryAttention(len(properties.char_set), self.emb_dim, self.query_dim, self.window_size,
                                                self.attn_activation).to(self.device)

    def gen_weights(self, tracked_labels, img_names):
        from tracking_utils import str_to_tensor
        w = self._table(len(img_names)).to(self.device)
        for row, name in enumerate(img_names):
            hist = self._recent(tracked_labels, name)
            if hist:
                with torch.no_grad():
                    scores = self.attention_model(str_to_tensor(self, hist))
                w[row, 1:len(hist) + 1] = scores[:len(hist)]
        return w


_GENERATORS = {"self_attention": AttentionWeightGenerator, "levenshtein": LevenshteinWeightGenerator, "decaying": DecayingWeightGenerator}


def weightgenerator_factory(method):
    return _GENERATORS[method]
