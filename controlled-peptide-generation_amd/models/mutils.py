"""Special-token ids and small model helpers (counterpart of the reference's models/mutils.py:5-14,31-45)."""
import os

import torch

UNK_IDX, PAD_IDX, START_IDX, EOS_IDX = 0, 1, 2, 3


def save_model(model, fn):
    """Checkpoint = plain state_dict with the reference's key names (models/mutils.py:11-14)."""
    d = os.path.dirname(fn)
    if d:
        os.makedirs(d, exist_ok=True)
    torch.save(model.state_dict(), fn)
    print('Saved model to ' + fn)


def onehot_embed(hard_ix, vocab_size):
    assert hard_ix.dim() == 1, 'expecting 1D tensor: minibatch of indices.'
    out = torch.zeros(hard_ix.size(0), vocab_size, device=hard_ix.device)
    out.scatter_(1, hard_ix.unsqueeze(1), 1.0)
    return out


def soft_embed(embed, soft_ix):
    """[mbsize, vocab] soft one-hots -> [mbsize, emb_dim] through the MFMA engine (reference: softIx @ embed.weight)."""
    from cpg import ops
    w = embed.weight
    flat = soft_ix.reshape(-1, soft_ix.shape[-1]).contiguous()
    out = torch.empty(flat.shape[0], w.shape[1], device=flat.device, dtype=torch.float32)
    ops.call("cpg_matmul_nn", ops._p(flat), flat.shape[1], ops._p(w), w.shape[1], ops._p(out), w.shape[1],
             flat.shape[0], w.shape[1], flat.shape[1], 0, ops._stream())
    return out.reshape(*soft_ix.shape[:-1], w.shape[1])
