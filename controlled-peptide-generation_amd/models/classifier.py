"""CNN sequence classifier (Kim 2014) - ADJACENT component, not on the north-star hot path.

Counterpart of the reference's models/classifier.py:15-60.  The reference constructs it inside RNN_VAE but never
trains it; it is reached only through q_c='classifier'.  It is kept here so that checkpoints keep their
keys (`classifier.conv_layers.{0,1,2}.*`, `classifier.fc.1.*`) and model construction consumes the RNG exactly like the
reference.  Its forward is plain torch (MIOpen conv) - SURVEY 8f rank 2 lists a HIP version as a "next" row; no parity or
performance claim is made for it.
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

    def forward(self, x):
        """x: embeddings [mbsize, seq_len, emb_dim] -> class logits [mbsize, 2]."""
        x = x.unsqueeze(1)
        assert x.size(2) >= self.max_filter_width, \
            'Current classifier arch needs at least seqlen {}'.format(self.max_filter_width)
        pooled = [F.relu(conv(x)).squeeze(3).max(dim=2)[0] for conv in self.conv_layers]
        return self.fc(torch.cat(pooled, dim=1))
