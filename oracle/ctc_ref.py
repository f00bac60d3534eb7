"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the CTC loss lines of the reference's recognition trainer.

    recognition_model.py:96-101
        pred = F.log_softmax(model(...), 2)
        pred = pad_sequence(decollate_tensor(pred, lengths))            # utterance-major, time first
        loss = F.ctc_loss(pred, y, lengths, text_int_lengths, blank=n_chars)     # reduction='mean'

F.ctc_loss is ATen's alpha-beta recursion (Graves et al. 2006, eq. 6-16): over the blank-extended label string
l' (2S+1 states), alpha_t(s) = logp_t(l'_s) + logsumexp(alpha_{t-1}(s), alpha_{t-1}(s-1), alpha_{t-1}(s-2) if
l'_s != blank and l'_s != l'_{s-2}); nll = -logsumexp(alpha_{T-1}(2S), alpha_{T-1}(2S-1)); 'mean' divides each nll
by max(S, 1) and averages over utterances.  Gradient w.r.t. the (pre-softmax) logits:
softmax - exp(logsumexp_{s: l'_s = c}(alpha_t(s) + beta_t(s)) + nll - logp_t(c)), scaled the same way.
Pinned against tests/golden/ctc_*.npz (made by running the reference's lines on torch's CPU ATen).
Everything here is float64 numpy on the packed (frames, V) layout.
"""
import numpy as np


def _logsumexp(xs):
    m = np.max(xs)
    if not np.isfinite(m):
        return m
    return m + np.log(np.sum(np.exp(np.asarray(xs) - m)))


def ctc_utterance(logp, target, blank):
    """logp (T, V) log-probabilities of one utterance -> (nll, d nll / d logits (T, V))."""
    T, V = logp.shape
    S = len(target)
    ext = np.full(2 * S + 1, blank, dtype=np.int64)
    ext[1::2] = target
    SP = len(ext)
    NEG = -np.inf
    alpha = np.full((T, SP), NEG)
    beta = np.full((T, SP), NEG)
    alpha[0, 0] = logp[0, blank]
    if SP > 1:
        alpha[0, 1] = logp[0, ext[1]]
    for t in range(1, T):
        for s in range(SP):
            c = [alpha[t - 1, s]]
            if s >= 1:
                c.append(alpha[t - 1, s - 1])
            if s >= 2 and ext[s] != blank and ext[s] != ext[s - 2]:
                c.append(alpha[t - 1, s - 2])
            alpha[t, s] = _logsumexp(c) + logp[t, ext[s]]
    beta[T - 1, SP - 1] = logp[T - 1, blank]
    if SP > 1:
        beta[T - 1, SP - 2] = logp[T - 1, ext[SP - 2]]
    for t in range(T - 2, -1, -1):
        for s in range(SP):
            c = [beta[t + 1, s]]
            if s + 1 < SP:
                c.append(beta[t + 1, s + 1])
            if s + 2 < SP and ext[s] != blank and ext[s] != ext[s + 2]:
                c.append(beta[t + 1, s + 2])
            beta[t, s] = _logsumexp(c) + logp[t, ext[s]]
    ll = _logsumexp([alpha[T - 1, SP - 1]] + ([alpha[T - 1, SP - 2]] if SP > 1 else []))
    nll = -ll
    grad = np.exp(logp)
    with np.errstate(invalid='ignore', over='ignore'):
        for t in range(T):
            ab = alpha[t] + beta[t]
            for c in range(V):
                sel = ab[ext == c]
                lcab = _logsumexp(sel) if len(sel) else NEG
                grad[t, c] -= np.exp(lcab + nll - logp[t, c])
    return nll, grad


def ctc_loss_packed(logits, lengths, targets, blank):
    """logits (rows, row_len, V) packed frames (utterances back to back, data_utils.py:159-179) ->
    (loss, dlogits of the same shape, nll per utterance); semantics of recognition_model.py:96-101."""
    logits = np.asarray(logits, dtype=np.float64)
    shp = logits.shape
    V = shp[-1]
    flat = logits.reshape(-1, V)
    m = flat.max(1, keepdims=True)
    logp = flat - (m + np.log(np.exp(flat - m).sum(1, keepdims=True)))
    d = np.zeros_like(flat)
    nlls = []
    off = 0
    n = len(lengths)
    for T, tgt in zip(lengths, targets):
        tgt = np.asarray(tgt, dtype=np.int64)
        nll, g = ctc_utterance(logp[off:off + T], tgt, blank)
        d[off:off + T] = g / (max(len(tgt), 1) * n)
        nlls.append(nll)
        off += T
    nlls = np.asarray(nlls)
    loss = float(np.mean(nlls / np.maximum([len(t) for t in targets], 1)))
    return loss, d.reshape(shp), nlls
