"""Drop-in for the training path of the reference's recognition_model.py ("next" row N1, BASELINE cfg5):

    ctc_loss(logits, example, blank)         the three loss lines :96-101 fused on the packed layout
    greedy_decode(logits, lengths, blank)    best-path CTC decode (argmax, collapse repeats, drop blanks)
    test(model, testset, device)             :30-58 with the greedy decoder -> WER
    train_model(trainset, devset, device)    :61-117 (AdamW, warm-up, x2 gradient accumulation, MultiStepLR)

The encoder is the same MI355X engine as the transduction trainer (Model without the aux head).  The CTC
alpha/beta recursion and its gradient run in csrc/ctc.hip straight on the packed (rows*200, V) logits, so the
decollate + pad_sequence copies and the (T_max, N, V) log-prob tensor of the reference never exist.
The reference's KenLM beam search (ctcdecode + lm.binary, :33-35) is a third-party C++ dependency that is not
vendored and needs a language-model file; it is out of scope (parity unpinned) -- test() reports greedy WER.
"""
import logging
import os
import string

import numpy as np
import torch

from . import _lib, staging
from .pipeline import SizeAwareSampler
from .architecture import Model
from .flags import FLAGS
from .optim import FusedAdamW
from .transduction_model import prepare_batch

_L = _lib.lib
_p = _lib.ptr


class TextTransform(object):
    """data_utils.py:243-258 without the unidecode/jiwer dependencies (ASCII punctuation stripping + lower-casing)."""

    def __init__(self):
        self.chars = string.ascii_lowercase + string.digits + ' '

    def clean_text(self, text):
        return ''.join(c for c in text.lower() if c not in string.punctuation)

    def text_to_int(self, text):
        return [self.chars.index(c) for c in self.clean_text(text)]

    def int_to_text(self, ints):
        return ''.join(self.chars[i] for i in ints)


class _CtcPlan(object):
    """Utterance descriptors for ss_ctc_loss (include/silent_speech_hip.h)."""

    def __init__(self, lengths, targets, rows_total, device):
        lengths = [int(n) for n in lengths]
        tl = [int(t.shape[0]) for t in targets]
        assert sum(lengths) <= rows_total
        assert len(tl) == len(lengths)
        f0 = np.concatenate([[0], np.cumsum(lengths)])
        g0 = np.concatenate([[0], np.cumsum(tl)])
        ws = np.concatenate([[0], np.cumsum([n * (2 * s + 1) for n, s in zip(lengths, tl)])])
        desc = np.stack([f0[:-1], lengths, g0[:-1], tl, ws[:-1]], 1).astype(np.int64) if lengths else np.zeros((0, 5), np.int64)
        self.n, self.max_s, self.ws_floats = len(lengths), max(tl) if tl else 0, int(ws[-1])
        # both tables cross PCIe in ONE pinned copy on the current stream (staging.upload): two pageable .to(device) calls cost a blocking
        # copy each -- more than the CTC kernels of the batch (bench.py ctc.ctc_loss.hip_ms 0.34 ms with them against 0.18 ms of kernels)
        if all(torch.is_tensor(t) and t.device.type == 'cpu' for t in targets) or not targets:
            flat = np.concatenate([t.reshape(-1).numpy().astype(np.int32) for t in targets]) if sum(tl) else np.zeros(1, dtype=np.int32)
            self.desc, self.targets = staging.upload([desc if desc.size else np.zeros((1, 5), np.int64), flat], device)
            self.desc = self.desc[:len(lengths)]
        else:                                                            # labels that already live on the device
            self.desc, = staging.upload([desc if desc.size else np.zeros((1, 5), np.int64)], device)
            self.desc = self.desc[:len(lengths)]
            self.targets = torch.cat([t.reshape(-1).to(device=device, dtype=torch.int32) for t in targets]).contiguous() if sum(tl) else torch.zeros(1, dtype=torch.int32, device=device)
        self.lengths, self.tlens = lengths, tl


def ctc_loss(pred, example, blank=None, *, return_plan=False):
    """recognition_model.py:96-101 in one call.  pred: (rows, 200, V) raw model outputs (NOT log-softmaxed);
    example: the collate_raw batch dict ('lengths', 'text_int').  Returns the 0-dim mean loss
    (per-utterance nll / max(target length, 1), averaged), attached to autograd."""
    B, T, V = pred.shape
    blank = V - 1 if blank is None else int(blank)
    logits = pred.reshape(B * T, V).float().contiguous()
    plan = _CtcPlan(example['lengths'], example['text_int'], B * T, pred.device)
    loss, _, nll, amax = torch.ops.silent_speech.ctc_loss(logits, plan.desc, plan.targets, plan.n, plan.max_s, plan.ws_floats, V, blank)
    plan.nll, plan.argmax = nll.detach()[:plan.n], amax
    loss = loss[0]
    return (loss, plan) if return_plan else loss


def _collapse(path, blank):
    out, prev = [], -1
    for c in path:
        if c != prev and c != blank:
            out.append(int(c))
        prev = c
    return out


def greedy_decode(pred, lengths, blank=None):
    """Best-path decode of packed logits (rows, T, V): per-frame argmax on the device (ss_frame_lse), then
    collapse-repeats / drop-blanks per utterance on the host.  Returns a list of int lists."""
    B, T, V = pred.shape
    blank = V - 1 if blank is None else int(blank)
    logits = pred.reshape(B * T, V).float().contiguous()
    M = B * T
    lse = torch.empty(M, dtype=torch.float32, device=pred.device)
    amax = torch.empty(M, dtype=torch.int32, device=pred.device)
    _lib.check(_L().ss_frame_lse(_p(logits), V, 0, V, M, _p(lse), _p(amax), _lib.stream_of(logits)), 'ss_frame_lse')
    path = amax.cpu().numpy()
    out, off = [], 0
    for n in lengths:
        out.append(_collapse(path[off:off + int(n)], blank))
        off += int(n)
    return out


def _edit_distance(a, b):
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i]
        for j, y in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = cur
    return prev[-1]


def wer(references, predictions):
    """jiwer.wer (:58): total word edit distance / total reference words."""
    errs = sum(_edit_distance(r.split(), p.split()) for r, p in zip(references, predictions))
    words = sum(len(r.split()) for r in references)
    return errs / max(words, 1)


def test(model, testset, device, *, batch_size=1):
    """:30-58.  Default (batch_size=1) = the reference: eval-mode forward of ONE WHOLE utterance at a time (:37-43), so the convolutions
    and the +-99-frame attention band span the utterance (the banded attention kernels take any T).  batch_size > 1 packs the utterances
    into 200-frame rows like training does -- faster, but context is cut at the row boundaries, so its WER is not the reference's number.
    Decoding is greedy best-path in both cases (the reference's KenLM beam search is third-party C++ needing lm.binary: out of scope)."""
    model.eval()
    tt = testset.text_transform
    blank = len(tt.chars)
    references, predictions = [], []
    with torch.no_grad():
        if batch_size == 1:
            for i in range(len(testset)):
                ex = testset[i]
                X = ex['emg'].to(device=device, dtype=torch.float32).unsqueeze(0)
                X_raw = ex['raw_emg'].to(device=device, dtype=torch.float32).unsqueeze(0)
                sess = ex['session_ids'].to(device=device).unsqueeze(0)
                pred = model(X, X_raw, sess)                                   # (1, T, V) logits
                predictions.append(tt.int_to_text(greedy_decode(pred, [pred.shape[1]], blank)[0]))
                references.append(tt.int_to_text(torch.as_tensor(ex['text_int']).tolist()))
        else:
            dataloader = torch.utils.data.DataLoader(testset, batch_size=batch_size, collate_fn=testset.collate_raw)
            for batch in dataloader:
                X, X_raw, sess = prepare_batch(batch, device, loss_plan=False)
                pred = model(X, X_raw, sess)
                for ints, tgt in zip(greedy_decode(pred, batch['lengths'], blank), batch['text_int']):
                    predictions.append(tt.int_to_text(ints))
                    references.append(tt.int_to_text(tgt.tolist()))
    model.train()
    return wer(references, predictions)


def train_model(trainset, devset, device, n_epochs=200, *, compute_dtype=torch.bfloat16, f32_matmul='exact', max_steps=None):
    """:61-117 on the MI355X: batches under a 128 000-sample budget, AdamW (lr 3e-4 in the reference's flags), linear
    warm-up, an optimiser step every SECOND batch (gradients accumulate in the flat .grad arena), MultiStepLR."""
    dataloader = torch.utils.data.DataLoader(trainset, collate_fn=devset.collate_raw, num_workers=0, batch_sampler=SizeAwareSampler(trainset, 128000))
    n_chars = len(devset.text_transform.chars)
    model = Model(devset.num_features, n_chars + 1, compute_dtype=compute_dtype, f32_matmul=f32_matmul).to(device)
    # flag defaults of recognition_model.py:20-28 (they differ from the transduction trainer's)
    lr0, warmup, l2 = FLAGS.lookup('learning_rate', 3e-4), FLAGS.lookup('learning_rate_warmup', 1000), FLAGS.lookup('l2', 0.0)
    out_dir, start = FLAGS.lookup('output_directory', 'output'), FLAGS.lookup('start_training_from', None)
    if start is not None:
        model.load_state_dict(torch.load(start, map_location=torch.device(device)), strict=False)
    optim = FusedAdamW(model, lr=lr0, weight_decay=l2)
    lr_sched = torch.optim.lr_scheduler.MultiStepLR(optim, milestones=[125, 150, 175], gamma=.5)

    def set_lr(new_lr):
        for param_group in optim.param_groups:
            param_group['lr'] = new_lr

    target_lr = lr0

    def schedule_lr(iteration):
        iteration = iteration + 1
        if iteration <= warmup:
            set_lr(iteration * target_lr / warmup)

    batch_idx = 0
    optim.zero_grad()
    for epoch_idx in range(n_epochs):
        losses = []
        for batch in dataloader:
            schedule_lr(batch_idx)
            X, X_raw, sess = prepare_batch(batch, device, loss_plan=False)
            pred = model(X, X_raw, sess)
            loss = ctc_loss(pred, batch, blank=n_chars)
            losses.append(loss.detach())
            loss.backward()
            if (batch_idx + 1) % 2 == 0:
                optim.step()
                optim.zero_grad()
            batch_idx += 1
            if max_steps is not None and batch_idx >= max_steps:
                break
        train_loss = float(torch.stack(losses).mean()) if losses else float('nan')
        val = test(model, devset, device)
        lr_sched.step()
        logging.info(f'finished epoch {epoch_idx+1} - training loss: {train_loss:.4f} validation WER: {val*100:.2f}')
        os.makedirs(out_dir, exist_ok=True)
        torch.save(model.state_dict(), os.path.join(out_dir, 'model.pt'))
        if max_steps is not None and batch_idx >= max_steps:
            break
    return model
