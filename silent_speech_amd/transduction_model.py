"""Drop-in for the hot-path entry points of the reference's transduction_model.py:

    dtw_loss(predictions, phoneme_predictions, example, phoneme_eval=False, phoneme_confusion=None)   (:98-157)
    test(model, testset, device)                                                                       (:33-55)
    train_model(trainset, devset, device, save_sound_outputs=True)                                     (:159-227)

dtw_loss keeps everything on the GPU: cost matrices are produced directly in the DTW kernel's strip
layout, DTW + backtrace run on device, and loss/gradient touch only the aligned pairs -- no D2H copy
of a T1 x T2 matrix and no per-utterance synchronisation (the reference blocks on both at :126).
"""
import logging
import os

import numpy as np
import torch

from . import _lib, ops, torch_ops  # noqa: F401  (torch_ops registers torch.ops.silent_speech.*)
from .align import _workspace_layout
from .architecture import Model
from . import staging
from .data_utils import PackJob, _host_pack, combine_fixed_length, phoneme_inventory
from .flags import FLAGS
from .optim import FusedAdamW

_L = _lib.lib
_p = _lib.ptr


class _LossPlan(object):
    """Host-side index arithmetic for one batch (the packed-row <-> utterance bookkeeping that decollate_tensor + zip do in the
    reference, transduction_model.py:101-111), kept PER UTTERANCE: one row of 8 int64 per utterance (`utt`) and one DTW descriptor
    per silent utterance (`desc`).  The per-frame index tables the loss kernels read are expanded on the device
    (`ss_loss_index_tables`).  The plan belongs to ONE dtw_loss call: a training loop sees new tensors every step, nothing here is
    cached across calls."""

    def __init__(self, example, rows_total):
        lengths = [int(n) for n in example['lengths']]
        silent = [bool(s) for s in example['silent']]
        audio = example['audio_features']
        t2 = [int(a.shape[0]) for a in audio]
        assert sum(lengths) <= rows_total                                           # data_utils.py:175
        n = len(lengths)
        pred_off = np.zeros(n + 1, dtype=np.int64)
        tgt_off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(lengths, out=pred_off[1:])
        np.cumsum(t2, out=tgt_off[1:])
        self.total_length = int(tgt_off[-1])
        utt = np.zeros((max(n, 1), 8), dtype=np.int64)
        desc, shapes = [], []
        res_tot = vo = si = 0
        for u, (n1, n2, s) in enumerate(zip(lengths, t2, silent)):
            assert audio[u].dim() == 2
            utt[u] = (n1, n2, int(s), pred_off[u], tgt_off[u], res_tot, vo, si)
            if s:
                shapes.append((n2, n1))
                desc.append([n2, n1, pred_off[u], tgt_off[u], 0, 0, 0, 0, res_tot, 0])
                res_tot += n2
                si += n2
            else:
                assert n2 == n1, 'voiced utterance: audio features (%d) and predictions (%d) differ in length' % (n2, n1)   # :139
                vo += n1
        self.n_utt, self.n_silent, self.shapes = n, len(shapes), shapes
        layout, ws_bytes = _workspace_layout(shapes) if shapes else ([], 0)
        for row, (sk, dr, bd) in zip(desc, layout):
            row[5], row[6], row[7] = sk, dr, bd
        self.ws_bytes, self.res_total = ws_bytes, res_tot
        self.n_voiced, self.n_silent_frames = vo, si
        self.utt_host = utt
        self.desc_host = np.asarray(desc, dtype=np.int64).reshape(-1, 10) if desc else np.zeros((1, 10), dtype=np.int64)
        self.lengths, self.silent, self.t2 = lengths, silent, t2
        self.pred_off, self.tgt_off = pred_off, tgt_off
        self.results = self.argmax = None
        self.signature = None

    def host_tables(self):
        return [self.utt_host, self.desc_host]

    def bind(self, utt_dev, desc_dev, Y, phones):
        """Device side: the two uploaded tables, the concatenated targets, and the per-frame index tables (one launch)."""
        dev = Y.device
        self.Y, self.phones = Y, phones
        self.desc = desc_dev
        nv, ns = max(self.n_voiced, 1), max(self.n_silent_frames, 1)
        idx = self.idx = torch.empty(2 * nv + 3 * ns, dtype=torch.int32, device=dev)
        self.vo_pred, self.vo_tgt = idx[:nv], idx[nv:2 * nv]
        self.si_tgt, self.si_base, self.si_res = idx[2 * nv:2 * nv + ns], idx[2 * nv + ns:2 * nv + 2 * ns], idx[2 * nv + 2 * ns:]
        _lib.check(_L().ss_loss_index_tables(_p(utt_dev), self.n_utt, _p(self.vo_pred), _p(self.vo_tgt), _p(self.si_tgt), _p(self.si_base), _p(self.si_res),
                                             _lib.stream_of(Y)), 'ss_loss_index_tables')
        return self


def _target_jobs(example, device):
    """The two concatenations of dtw_loss's targets (all utterances' audio features / phoneme labels back to back) either as gather
    jobs over device tensors or, for a batch that arrives in host memory, as ONE pinned host concatenation + upload each."""
    audio, phones = example['audio_features'], example['phonemes']
    on_device = _lib.kernels_can_read(audio[0])
    if on_device and audio[0].is_cuda:                # a mixed batch (some targets still on the host): the gather kernel only takes device pointers
        audio = [a if a.is_cuda else a.to(device) for a in audio]
        phones = [q if q.is_cuda else q.to(device) for q in phones]
    if on_device:
        audio = [a if a.dtype == torch.float32 else a.float() for a in audio]
        phones = [q if q.dtype == torch.int64 else q.long() for q in phones]
        return PackJob(audio), PackJob(phones), None, None
    pin = torch.cuda.is_available()
    Y = _host_pack([a.float() for a in audio], None, pin).to(device, non_blocking=True)
    ph = _host_pack([q.long() for q in phones], None, pin).to(device, non_blocking=True)
    return None, None, Y, ph


def _build_loss_plan(example, rows_total, device):
    plan = _LossPlan(example, rows_total)
    ja, jp, Y, ph = _target_jobs(example, device)
    tabs = plan.host_tables() + ([ja.table, jp.table] if ja is not None else [])
    up = staging.upload(tabs, device)
    if ja is not None:
        Y, ph = ja.launch(up[2]), jp.launch(up[3])
    return plan.bind(up[0], up[1], Y, ph)


class _Prepared(object):
    """What prepare_batch leaves for the dtw_loss call of the same step: keyed by the IDENTITY of the example dict and consumed by
    the first dtw_loss on it (one-shot: no entry outlives its step, nothing is keyed on tensor addresses or version counters)."""
    example = None
    plan = None
    rows = 0


def _target_signature(example):
    """What the prepared loss plan snapshotted of a batch: address and version counter of every target tensor, and the silent flags (about 90
    cheap attribute reads for a reference-size batch).  A caller that edits or replaces targets between prepare_batch and dtw_loss gets a
    fresh plan instead of stale targets."""
    return (tuple((t.data_ptr(), t._version) for t in example['audio_features']), tuple((t.data_ptr(), t._version) for t in example['phonemes']),
            tuple(bool(x) for x in example['silent']))


def _loss_plan(example, rows_total, device):
    if _Prepared.example is example and _Prepared.plan is not None and _Prepared.rows == rows_total:
        plan, _Prepared.example, _Prepared.plan = _Prepared.plan, None, None
        if plan.lengths == [int(n) for n in example['lengths']] and plan.Y.device == device and plan.signature == _target_signature(example):
            return plan
    return _build_loss_plan(example, rows_total, device)


def _fused_head(predictions, phoneme_predictions, M, n_mel, n_ph):
    """[pred | phoneme logits] per packed frame as ONE f32 matrix.  Model.forward hands out the two as column slices of exactly that
    matrix (the plan's `head` buffer): it is then used as it is (no concatenation, and the gradient lands in it directly)."""
    base = predictions._base
    if (base is not None and base is phoneme_predictions._base and base.dim() == 2 and base.dtype == torch.float32 and base.is_contiguous()
            and base.shape[0] == M and base.shape[1] % 4 == 0 and base.shape[1] >= n_mel + n_ph
            and predictions.storage_offset() == base.storage_offset() and phoneme_predictions.storage_offset() == base.storage_offset() + n_mel
            and predictions.stride(-1) == 1 and phoneme_predictions.stride(-1) == 1
            and predictions.stride(-2) == base.shape[1] and phoneme_predictions.stride(-2) == base.shape[1]):
        return base
    ld = (n_mel + n_ph + 3) // 4 * 4
    parts = [predictions.reshape(M, n_mel).float(), phoneme_predictions.reshape(M, n_ph).float()]
    if ld > n_mel + n_ph:
        parts.append(torch.zeros(M, ld - n_mel - n_ph, device=predictions.device))
    return torch.cat(parts, 1)


def _dtw_loss_plan(predictions, phoneme_predictions, example, lam, total_length=None):
    """Shared by dtw_loss and get_aligned_prediction: builds the fused head, runs the loss kernels, returns (loss, correct, plan)."""
    B, T, n_mel = predictions.shape
    n_ph = phoneme_predictions.shape[2]
    if n_mel % 4:
        raise ValueError('number of mel bins must be a multiple of 4')
    M = B * T
    head = _fused_head(predictions, phoneme_predictions, M, n_mel, n_ph)
    plan = _loss_plan(example, M, predictions.device)
    total = plan.total_length if total_length is None else total_length
    mx_n = max((a for a, _ in plan.shapes), default=0)
    mx_m = max((b for _, b in plan.shapes), default=0)
    loss, correct, _, results, amax = torch.ops.silent_speech.dtw_loss(
        head, plan.Y, plan.phones, plan.idx, plan.desc, n_mel, n_ph, float(lam), 1.0 / float(total), plan.n_voiced, plan.n_silent_frames,
        plan.n_silent, int(plan.ws_bytes), int(plan.res_total), mx_n, mx_m, float(sum(a * b for a, b in plan.shapes)),
        float(sum(a + b for a, b in plan.shapes)))
    plan.results, plan.argmax = (results if plan.n_silent else None), amax        # the plan is this call's own object (evaluation extras)
    return loss[0], correct, plan


def dtw_loss(predictions, phoneme_predictions, example, phoneme_eval=False, phoneme_confusion=None, *, phoneme_loss_weight=None,
             total_length=None):
    """Returns (loss, phoneme accuracy) like transduction_model.py:98-157.  The loss is a 0-dim tensor attached
    to autograd; the accuracy is a Python float when phoneme_eval=True (needs a device sync, as the reference's
    .item() calls do) and a 0-dim device tensor otherwise (the training loop discards it, :206).
    total_length: override of the normaliser sum(T2) (data-parallel ranks pass the GLOBAL frame count)."""
    lam = FLAGS.phoneme_loss_weight if phoneme_loss_weight is None else phoneme_loss_weight
    loss, correct, plan = _dtw_loss_plan(predictions, phoneme_predictions, example, lam, total_length)
    if not phoneme_eval:
        return loss, correct[0].float() / plan.total_length
    # ---- evaluation extras: the confusion matrix (transduction_model.py:130-137,147-152)
    if isinstance(phoneme_confusion, DeviceConfusion):
        # accumulated on the device (one launch, integer atomics), no synchronisation: test() reads the matrix back once per epoch,
        # and the accuracy stays a device scalar until then
        phoneme_confusion.add(plan)
        return loss, correct[0].float() / plan.total_length
    acc = float(correct.item()) / plan.total_length
    if phoneme_confusion is not None:
        # a numpy matrix handed in by a caller written against the reference: filled through the same device kernel, one copy back
        dc = DeviceConfusion(phoneme_confusion.shape[0], predictions.device)
        dc.add(plan)
        phoneme_confusion += dc.numpy().astype(phoneme_confusion.dtype)
    return loss, acc


class DeviceConfusion(object):
    """The phoneme confusion matrix of an evaluation pass, kept on the device (int32, [predicted][target]); dtw_loss(..., phoneme_eval=True,
    phoneme_confusion=<this>) adds a batch with one kernel launch and without a host synchronisation."""

    def __init__(self, n_phone, device):
        self.n = int(n_phone)
        self.mat = torch.zeros(self.n * self.n + 1, dtype=torch.int32, device=device)      # + 1: frames with a label outside the inventory (see numpy())

    def add(self, plan):
        L = _lib.lib()
        res = plan.results if plan.results is not None else None
        _lib.check(L.ss_phoneme_confusion(_lib.ptr(plan.argmax), _lib.ptr(plan.phones), _lib.ptr(res), _lib.ptr(plan.vo_pred), _lib.ptr(plan.vo_tgt),
                                          plan.n_voiced, _lib.ptr(plan.si_tgt), _lib.ptr(plan.si_base), _lib.ptr(plan.si_res),
                                          plan.n_silent_frames if res is not None else 0, _lib.ptr(self.mat), self.n, _lib.stream_of(self.mat)),
                   'ss_phoneme_confusion')

    def numpy(self):
        """The matrix on the host (one read-back).  Raises IndexError when a prediction or target label fell outside [0, n_phone) in any batch added since
        construction -- the reference's `phoneme_confusion[p, t] += 1` (transduction_model.py:134-137, :150-152) raises on such a label; silently dropping the
        frame would hand back a matrix whose total is smaller than the number of target frames."""
        host = self.mat.cpu().numpy()
        if int(host[-1]) != 0:
            raise IndexError('DeviceConfusion: %d frame(s) carried a predicted or target phoneme label outside [0, %d)' % (int(host[-1]), self.n))
        return host[:-1].reshape(self.n, self.n)


class EnsembleModel(torch.nn.Module):
    """evaluate.py:22-34: averages the mel predictions and phoneme logits of several models."""

    def __init__(self, models):
        super().__init__()
        self.models = torch.nn.ModuleList(models)

    def forward(self, x, x_raw, sess):
        ys, ps = [], []
        for model in self.models:
            y, p = model(x, x_raw, sess)
            ys.append(y)
            ps.append(p)
        return torch.stack(ys, 0).mean(0), torch.stack(ps, 0).mean(0)


def predict_utterance(model, datapoint, device):
    """The model half of save_output (transduction_model.py:57-66): eval-mode forward of ONE whole utterance
    (un-chunked: T is the utterance length, the banded attention kernel skips everything beyond +-99 frames).
    Returns the (T, n_mel) prediction on the device; vocoding (:68-72) is outside the hot path."""
    was_training = model.training
    model.eval()
    with torch.no_grad():
        sess = datapoint['session_ids'].to(device=device).unsqueeze(0)
        X = datapoint['emg'].to(dtype=torch.float32, device=device).unsqueeze(0)
        X_raw = datapoint['raw_emg'].to(dtype=torch.float32, device=device).unsqueeze(0)
        pred, _ = model(X, X_raw, sess)
    model.train(was_training)
    return pred.squeeze(0)


def get_aligned_prediction(model, datapoint, device, audio_normalizer):
    """transduction_model.py:75-96.  Silent utterances are aligned to the parallel voiced audio features by DTW on the
    plain Euclidean cost (no phoneme term): cost matrix, DTW and backtrace all stay on the device (the reference
    copies the T1 x T2 cdist matrix to the host, :87-88); only the T2 indices are used to gather."""
    was_training = model.training
    model.eval()
    with torch.no_grad():
        silent = datapoint['silent']
        sess = datapoint['session_ids'].to(device).unsqueeze(0)
        X = datapoint['emg'].to(device).unsqueeze(0)
        X_raw = datapoint['raw_emg'].to(device).unsqueeze(0)
        y = datapoint['parallel_voiced_audio_features' if silent else 'audio_features'].to(device)
        pred, aux = model(X, X_raw, sess)                       # (1, seq, dim)
        if silent:
            ex = dict(lengths=[pred.shape[1]], silent=[True], audio_features=[y],
                      phonemes=[torch.zeros(y.shape[0], dtype=torch.int64, device=y.device)])
            _, _, plan = _dtw_loss_plan(pred, aux, ex, 0.0)     # lambda = 0: cost = ||pred_q - y_k||_2 exactly (torch.cdist, :87)
            pred_aligned = pred.squeeze(0)[plan.results[:y.shape[0]].long()]
        else:
            pred_aligned = pred.squeeze(0)
        pred_aligned = audio_normalizer.inverse(pred_aligned.cpu())
    model.train(was_training)
    return pred_aligned


def prepare_batch(batch, device, seq_len=200, loss_plan=True):
    """The three combine_fixed_length calls of a step (transduction_model.py:198-200, :42-44) AND the host bookkeeping of the dtw_loss that
    follows, done together BEFORE the forward pass is enqueued: every table the step needs (three pack tables, the two target
    concatenations, utterance rows, DTW descriptors) crosses PCIe in ONE pinned copy, then six small launches.  The loss plan waits in
    `_Prepared` for the dtw_loss call on this same `batch` dict.  A batch that lives in host memory (the DataLoader case) is packed on
    the host into pinned buffers and uploaded with one copy per field instead of one per utterance."""
    device = torch.device(device)
    fields = ('emg', 'raw_emg', 'session_ids')
    lens = (seq_len, seq_len * 8, seq_len)
    first = batch['raw_emg'][0]
    on_device = _lib.kernels_can_read(first)
    jobs, packed = [], []
    if on_device:
        jobs = [PackJob(batch[f], n) for f, n in zip(fields, lens)]
    else:
        pin = torch.cuda.is_available()
        packed = [_host_pack(batch[f], n, pin).to(device, non_blocking=True) for f, n in zip(fields, lens)]
    want_plan = loss_plan and 'audio_features' in batch and 'phonemes' in batch and 'silent' in batch
    tabs = [j.table for j in jobs]
    plan = ja = None
    if want_plan:
        rows_total = (sum(int(n) for n in batch['lengths']) + seq_len - 1) // seq_len * seq_len
        plan = _LossPlan(batch, rows_total)
        ja, jp, Y, ph = _target_jobs(batch, device)
        tabs += plan.host_tables() + ([ja.table, jp.table] if ja is not None else [])
    up = staging.upload(tabs, device) if tabs else []
    if jobs:
        packed = [j.launch(t) for j, t in zip(jobs, up)]
    if want_plan:
        k = len(jobs)
        if ja is not None:
            Y, ph = ja.launch(up[k + 2]), jp.launch(up[k + 3])
        plan.signature = _target_signature(batch)
        _Prepared.example, _Prepared.plan, _Prepared.rows = batch, plan.bind(up[k], up[k + 1], Y, ph), rows_total
    return tuple(packed)


def _pack_batch(batch, device, seq_len=200):
    return prepare_batch(batch, device, seq_len)


def _lookahead(iterable):
    """(item, next item or None): the training loop looks one batch ahead so that data-parallel bookkeeping of the next step can
    overlap the current one."""
    it = iter(iterable)
    try:
        cur = next(it)
    except StopIteration:
        return
    for nxt in it:
        yield cur, nxt
        cur = nxt
    yield cur, None


def test(model, testset, device):
    """transduction_model.py:33-55: eval-mode forward on batches of 32 utterances; returns
    (mean loss, mean phoneme accuracy, confusion matrix)."""
    model.eval()
    dataloader = torch.utils.data.DataLoader(testset, batch_size=32, collate_fn=testset.collate_raw)
    losses, accuracies = [], []
    confusion = DeviceConfusion(len(phoneme_inventory), device)      # accumulated on the device: ONE read-back per epoch
    with torch.no_grad():
        for batch in dataloader:
            X, X_raw, sess = _pack_batch(batch, device)
            pred, phoneme_pred = model(X, X_raw, sess)
            loss, phon_acc = dtw_loss(pred, phoneme_pred, batch, True, confusion)
            losses.append(loss.detach().reshape(1))
            accuracies.append(phon_acc.reshape(1))
    phoneme_confusion = confusion.numpy().astype(np.float64)          # the reference's matrix is np.zeros(...) float64
    losses = torch.cat(losses).cpu().numpy() if losses else np.zeros(0)
    accuracies = torch.cat(accuracies).cpu().numpy() if accuracies else np.zeros(0)
    model.train()
    return np.mean(losses), np.mean(accuracies), phoneme_confusion


def train_model(trainset, devset, device, save_sound_outputs=True, *, compute_dtype=torch.bfloat16, f32_matmul='exact', max_steps=None, data_parallel=None):
    """transduction_model.py:159-227 on the MI355X.  trainset/devset follow the reference's EMGDataset protocol
    (collate_raw, num_features, num_speech_features, size-aware batches -- see synthetic.SyntheticEMGDataset).
    The vocoder / ASR tail (:175-176,218-225) is outside the hot path: save_sound_outputs is accepted for
    signature compatibility and ignored with a warning."""
    if save_sound_outputs:
        logging.warning('save_sound_outputs: HiFi-GAN vocoding / DeepSpeech evaluation are outside the MI355X hot path; skipped')
    from .pipeline import SizeAwareSampler
    n_epochs = FLAGS.epochs
    training_subset = trainset if FLAGS.data_size_fraction >= 1 else trainset.subset(FLAGS.data_size_fraction)
    dp = data_parallel if (data_parallel is not None and data_parallel.world > 1) else None
    # transduction_model.py:166: SizeAwareSampler(training_subset, 256000); data-parallel ranks take every world-th batch of one shared shuffle
    sampler = SizeAwareSampler(training_subset, 256000, rank=dp.rank, world=dp.world, seed=0) if dp is not None else SizeAwareSampler(training_subset, 256000)
    dataloader = torch.utils.data.DataLoader(training_subset, collate_fn=devset.collate_raw, num_workers=0, batch_sampler=sampler)
    n_phones = len(phoneme_inventory)
    model = Model(devset.num_features, devset.num_speech_features, n_phones, compute_dtype=compute_dtype, f32_matmul=f32_matmul).to(device)
    if FLAGS.start_training_from is not None:
        model.load_state_dict(torch.load(FLAGS.start_training_from), strict=False)
    if data_parallel is not None:
        data_parallel.attach(model)
    main_rank = dp is None or dp.rank == 0
    optim = FusedAdamW(model, weight_decay=FLAGS.l2)
    lr_sched = torch.optim.lr_scheduler.ReduceLROnPlateau(optim, 'min', 0.5, patience=FLAGS.learning_rate_patience)

    def set_lr(new_lr):
        for param_group in optim.param_groups:
            param_group['lr'] = new_lr

    target_lr = FLAGS.learning_rate

    def schedule_lr(iteration):
        iteration = iteration + 1
        if iteration <= FLAGS.learning_rate_warmup:
            set_lr(iteration * target_lr / FLAGS.learning_rate_warmup)

    batch_idx = 0
    for epoch_idx in range(n_epochs):
        losses = []
        sampler.set_epoch(epoch_idx)
        for batch, upcoming in _lookahead(dataloader):
            optim.zero_grad()
            schedule_lr(batch_idx)
            X, X_raw, sess = _pack_batch(batch, device)
            if data_parallel is not None:
                # the host-side exchange of the NEXT batch's (row, frame) counts starts now and is picked up by the next begin_step
                nxt = None if upcoming is None else ((sum(int(n) for n in upcoming['lengths']) + 199) // 200 * 200, data_parallel.local_target_frames(upcoming))
                data_parallel.begin_step(X_raw.shape[0] * (X_raw.shape[1] // 8), data_parallel.local_target_frames(batch), next_counts=nxt)
            pred, phoneme_pred = model(X, X_raw, sess)
            total = data_parallel.global_total(batch) if data_parallel is not None else None
            loss, _ = dtw_loss(pred, phoneme_pred, batch, total_length=total)
            losses.append(loss.detach())
            loss.backward()
            if data_parallel is not None:
                data_parallel.sync_gradients(model)
            optim.step()
            batch_idx += 1
            if max_steps is not None and batch_idx >= max_steps:
                break
        train_loss = float(torch.stack(losses).mean()) if losses else float('nan')      # one sync per epoch instead of per step (:207)
        val, phoneme_acc, _ = test(model, devset, device)                                 # every rank evaluates (identical weights) ...
        if data_parallel is not None:                                                      # ... and rank 0's figures decide: kernels with f32 atomics may differ in the
            val, phoneme_acc = data_parallel.broadcast_scalars(val, phoneme_acc)           # last bits between ranks, a plateau decision must not
        lr_sched.step(val)
        if main_rank:
            logging.info(f'finished epoch {epoch_idx+1} - validation loss: {val:.4f} training loss: {train_loss:.4f} phoneme accuracy: {phoneme_acc*100:.2f}')
            out_dir = FLAGS.output_directory
            os.makedirs(out_dir, exist_ok=True)
            torch.save(model.state_dict(), os.path.join(out_dir, 'model.pt'))             # :217
        if max_steps is not None and batch_idx >= max_steps:
            break
    return model
