"""HistoryAttention — the tiny self-attention scorer of the `self_attention` label-history weight generator
(reference models/model_attention.py:7-38; same parameter / buffer names: embedding, Wq, loss_coef_layer,
positional_encodings).  It sees at most `window_size` (~5) words of <= 100 characters and is never optimised in the
reference (its parameters are in no optimizer), so it is plain torch on whatever device it is given — it is not
part of the HIP hot path and is kept only so that `--weightgen_method self_attention` works."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class HistoryAttention(nn.Module):
    def __init__(self, char_vocab_size, emb_size, Dq, window_size, activation="sigmoid", is_emb_train=False):
        super().__init__()
        self.Dq, self.activation = Dq, activation
        table = torch.normal(0, 1, (char_vocab_size + 1, emb_size))
        if is_emb_train:
            self.embedding = nn.Parameter(table, requires_grad=True)
        else:
            table[char_vocab_size, :] = 0                        # the padding symbol embeds to zero
            self.register_buffer("embedding", table)
        self.Wq = nn.Linear(emb_size, Dq)
        self.loss_coef_layer = nn.Linear(window_size, 1)
        self.positional_encodings = nn.Parameter(torch.zeros(window_size, emb_size), requires_grad=True)

    def forward(self, char_indices):
        words = self.embedding[char_indices].mean(dim=1) + self.positional_encodings      # [window, emb]
        q = self.Wq(words)
        attn = F.softmax(q @ q.T / math.sqrt(self.Dq), dim=1)
        z = self.loss_coef_layer(attn)
        if self.activation == "sigmoid":
            w = torch.sigmoid(z)
        elif self.activation == "softmax":
            w = F.softmax(z, dim=0)
        elif self.activation == "relu":
            w = F.relu(z)
            w = w / (w.sum() + 0.000001)
        else:
            raise ValueError(f"unknown attn_activation {self.activation!r}")
        return w.squeeze(dim=1)
