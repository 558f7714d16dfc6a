"""CNN sequence classifier (Kim 2014) - ADJACENT component, not on the north-star hot path.

Counterpart of the reference's models/classifier.py:15-60.  The reference constructs it inside RNN_VAE but never
trains it; it is reached only through q_c='classifier'.  The torch modules below are parameter containers
(same checkpoint keys `classifier.conv_layers.{0,1,2}.*`, `classifier.fc.1.*`, same RNG consumption at construction).
  * token inputs (`forward_tokens`, used by RNN_VAE.forward_classifier for id batches): HIP path, forward and backward - the
    convolutions collapse to token-table look-ups (csrc/classifier.hip), logits and gradients (of the classifier's own
    parameters, of the embedding, and of the reconstruction loss through c = softmax(classifier(x))) pinned to the reference's
    autograd by tests/golden/classifier_A.npz;
  * embedding / soft inputs (`forward`): plain torch, kept for completeness, no claim.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def build_classifier(classifier_type, emb_dim, **C_args):
    if classifier_type != 'cnn':
        raise ValueError('Please use CNN classifier')
    return CNNClassifier(emb_dim, **C_args)


class CNNClassifier(nn.Module):
    def __init__(self, emb_dim, min_filter_width, max_filter_width, num_filters, dropout):
        super().__init__()
        self.max_filter_width = max_filter_width
        widths = range(min_filter_width, max_filter_width + 1)
        self.conv_layers = nn.ModuleList([nn.Conv2d(1, num_filters, (w, emb_dim)) for w in widths])
        self.fc = nn.Sequential(nn.Dropout(dropout), nn.Linear(num_filters * len(widths), 2))

    def forward_tokens(self, ids, emb_weight):
        """ids int64 [mbsize, seq_len] -> class logits [mbsize, 2]; differentiable in the classifier's parameters and in
        emb_weight (the reference's q_c='classifier' path keeps this gradient, models/model.py:186-188)."""
        from cpg import ops
        B, T = ids.shape
        V, E = emb_weight.shape
        F_ = self.conv_layers[0].out_channels
        widths = [c.kernel_size[0] for c in self.conv_layers]
        assert widths == list(range(widths[0], widths[0] + len(widths))), 'consecutive filter widths expected'
        assert T >= widths[-1], 'Current classifier arch needs at least seqlen {}'.format(widths[-1])
        tabs = []
        for conv in self.conv_layers:
            w = conv.kernel_size[0]
            wmat = conv.weight.view(F_, w * E)                      # [F, w*E]: filter tap dw occupies columns dw*E..(dw+1)*E
            for dw in range(w):
                tabs.append(ops.LinearFn.apply(emb_weight, wmat[:, dw * E:(dw + 1) * E], None))    # [V, F]
        tabs = torch.cat(tabs, 0)                                    # [sum_w * V, F]: layers and taps back to back
        bias = torch.stack([c.bias for c in self.conv_layers])
        pooled = ops.CnnPoolFn.apply(ids, tabs, bias, V, widths[0])
        if self.training and self.fc[0].p > 0:
            pooled = F.dropout(pooled, self.fc[0].p, True)
        return ops.LinearFn.apply(pooled, self.fc[1].weight, self.fc[1].bias)

    def forward(self, x):
        """x: embeddings [mbsize, seq_len, emb_dim] -> class logits [mbsize, 2]."""
        x = x.unsqueeze(1)
        assert x.size(2) >= self.max_filter_width, \
            'Current classifier arch needs at least seqlen {}'.format(self.max_filter_width)
        pooled = [F.relu(conv(x)).squeeze(3).max(dim=2)[0] for conv in self.conv_layers]
        return self.fc(torch.cat(pooled, dim=1))
