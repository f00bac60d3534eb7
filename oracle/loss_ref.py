"""Oracle: dtw_loss (fp32, torch CPU).  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates reference transduction_model.py:98-157 and data_utils.py:158-178:
  silent utterance : costs = cdist(pred, y) + lambda * -(log_softmax(aux)[:, y_phone])   (:116-124)
                     alignment = DTW(costs.T)  (align.py:16-34)                          (:126)
                     loss = sum_k costs[alignment[k], k]                                 (:128)
  voiced utterance : loss = sum_t ||y_t - pred_t + 1e-6||_2 + lambda * CE_sum(aux, y_phone) (:141-145)
  batch            : sum(losses) / sum(T2)                                               (:157)
cdist is evaluated directly (sqrt of summed squared differences); the reference's torch.cdist
switches to a matmul expansion above 25 rows -- same value to f32 rounding, so loss parity with the
reference is tolerance-based while the alignment for a GIVEN cost matrix is bit-exact.
"""
import numpy as np
import torch
import torch.nn.functional as F
from .dtw_ref import align_from_distances_c, align_from_distances_numpy


def combine_fixed_length(tensor_list, length):
    """data_utils.py:158-167."""
    total = sum(t.shape[0] for t in tensor_list)
    pad = (-total) % length
    parts = list(tensor_list)
    if pad:
        parts.append(torch.zeros((pad,) + tuple(parts[0].shape[1:]), dtype=parts[0].dtype))
    flat = torch.cat(parts, 0)
    return flat.view((flat.shape[0] // length, length) + tuple(flat.shape[1:]))


def decollate_tensor(tensor, lengths):
    """data_utils.py:169-178."""
    b, s, d = tensor.shape
    flat = tensor.reshape(b * s, d)
    out, idx = [], 0
    for n in lengths:
        assert idx + n <= b * s
        out.append(flat[idx:idx + n])
        idx += n
    return out


def silent_costs(pred, y, pred_phone, y_phone, lam):
    diff = pred[:, None, :] - y[None, :, :]
    dists = torch.sqrt((diff * diff).sum(-1))
    lp = F.log_softmax(pred_phone, -1)
    return dists + lam * -(lp[:, y_phone])


def dtw_loss_ref(predictions, phoneme_predictions, example, lam=0.5, use_c=True, return_alignments=False):
    """predictions (B,T,80), phoneme_predictions (B,T,48); example: reference batch dict
    (keys lengths, audio_features, phonemes, silent).  Returns (loss, phoneme_acc[, alignments])."""
    preds = decollate_tensor(predictions, example['lengths'])
    phs = decollate_tensor(phoneme_predictions, example['lengths'])
    losses, correct, total, aligns = [], 0, 0, []
    for pred, y, pp, yp, silent in zip(preds, example['audio_features'], phs, example['phonemes'], example['silent']):
        if silent:
            costs = silent_costs(pred, y, pp, yp, lam)
            cT = costs.detach().numpy().T
            al = align_from_distances_c(cT) if use_c else align_from_distances_numpy(cT)
            aligns.append(al)
            loss = costs[al, range(len(al))].sum()
            correct += int((pp.argmax(-1)[al] == yp).sum())
        else:
            assert y.shape[0] == pred.shape[0]
            d = torch.sqrt(((y - pred + 1e-6) ** 2).sum(-1))
            loss = d.sum() + lam * F.cross_entropy(pp, yp, reduction='sum')
            aligns.append(None)
            correct += int((pp.argmax(-1) == yp).sum())
        losses.append(loss)
        total += y.shape[0]
    out = (sum(losses) / total, correct / total)
    return out + (aligns,) if return_alignments else out
