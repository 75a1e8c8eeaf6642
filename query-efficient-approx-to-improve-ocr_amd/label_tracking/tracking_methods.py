"""Loss-weight generators for the label-history CTC (reference label_tracking/tracking_methods.py).
Only the default `decaying` generator (:105-115) is on the hot path; `levenshtein` and
`self_attention` (:26-101) are host-side O(window^2) string logic declared out of scope in SURVEY.md §2.1."""
import torch


class DecayingWeightGenerator:
    def __init__(self, tracking_args, device, char_to_index=None):
        self.decay_factor = tracking_args.decay_factor
        self.window_size = tracking_args.window_size
        self.device = device

    def print_debug_statements(self):
        pass

    def gen_weights(self, training_obj, img_names):
        return torch.tensor([self.decay_factor ** i for i in range(self.window_size)]).to(self.device)


def weightgenerator_factory(method):
    if method == "decaying":
        return DecayingWeightGenerator
    if method in ("self_attention", "levenshtein"):
        raise NotImplementedError(f"weightgen_method={method!r} is outside the MI355X hot-path scope (SURVEY.md §2.1); use 'decaying'")
    raise KeyError(method)
