"""clip_grad_norm_ + Adam as the reference's train_vae drives them (oracle; test infrastructure only).

train_vae.py:15,39-42:  Adam(model.vae_params(), lr) ; clip_grad_norm_(model.vae_params(), clip) ; step().
`vae_params()` yields word_emb.weight TWICE (models/model.py:88-94: once via word_emb.parameters(), once via
decoder.parameters() because decoder.emb is the same nn.Embedding) - SURVEY F6.  Consequences restated here:
  * the total norm counts the embedding gradient twice;
  * when clipping is active the embedding gradient is scaled by coef twice (coef**2);
  * Adam performs two sequential updates of the embedding per iteration (its step counter advances by 2),
    both with the same (already clipped) gradient.
Third-party arithmetic: torch.optim.Adam / torch.nn.utils.clip_grad_norm_ (pinned pytorch=1.7.1, amp_gen.yml:8;
fixtures generated with torch 2.10 single-tensor CPU path, which loops per parameter like 1.7.1 does).
"""
import numpy as np

F32 = np.float32

DUP_KEY = "word_emb.weight"


def vae_param_order(P):
    """Order/multiplicity in which vae_params() yields parameters: emb, encoder.*, decoder.* (emb again first)."""
    enc = [k for k in P if k.startswith("encoder.")]
    dec = [k for k in P if k.startswith("decoder.") and k != "decoder.emb.weight"]
    return [DUP_KEY] + enc + [DUP_KEY] + dec


class AdamClip:
    def __init__(self, P, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, max_norm=5.0):
        self.lr, self.b1, self.b2, self.eps, self.max_norm = lr, betas[0], betas[1], eps, max_norm
        self.order = vae_param_order(P)
        self.m = {k: np.zeros_like(P[k]) for k in set(self.order)}
        self.v = {k: np.zeros_like(P[k]) for k in set(self.order)}
        self.t = {k: 0 for k in set(self.order)}

    def step(self, P, G):
        """In-place update of P from gradients G (dict).  Returns the pre-clip total norm."""
        G = {k: G[k].astype(F32).copy() for k in set(self.order)}
        total = np.sqrt(sum(float(np.sum(G[k].astype(np.float64) ** 2)) for k in self.order))
        coef = min(self.max_norm / (total + 1e-6), 1.0)
        for k in self.order:  # duplicates are multiplied twice, exactly like the in-place foreach/loop mul_
            G[k] = (G[k] * F32(coef)).astype(F32)
        for k in self.order:
            g = G[k]
            self.t[k] += 1
            t = self.t[k]
            self.m[k] = (self.b1 * self.m[k] + (1 - self.b1) * g).astype(F32)
            self.v[k] = (self.b2 * self.v[k] + (1 - self.b2) * g * g).astype(F32)
            bc1 = 1 - self.b1 ** t
            bc2 = 1 - self.b2 ** t
            denom = np.sqrt(self.v[k]) / np.sqrt(bc2) + self.eps
            P[k] = (P[k] - (self.lr / bc1) * self.m[k] / denom).astype(F32)
        if "decoder.emb.weight" in P:
            P["decoder.emb.weight"] = P[DUP_KEY]
        return total
