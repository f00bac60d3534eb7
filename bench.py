#!/usr/bin/env python3
"""Benchmark of the EMG->mel transduction TRAINING step (BASELINE.json metric: EMG frames/s).

One "step" = everything reference transduction_model.py:196-212 does for one batch, with the batch's
tensors already resident in HBM: pack utterances into 200-frame rows (combine_fixed_length) -> Model
forward (shift augmentation, 3 ResBlocks, 6 relative-position encoder layers, heads; dropout 0.2) ->
dtw_loss (cost matrices + on-device DTW for the silent utterances) -> backward -> (N>1: RCCL all-reduce of
the flat gradient arena + BatchNorm statistic sums) -> fused AdamW with the reference's warm-up schedule.
Workload (BASELINE config 2): full 768-d / 6-layer model, bf16 MFMA compute, one synthetic
reference-size batch per GPU (256 000 raw-sample budget => ~40 utterances, ~22 k frames, ~110 rows),
25 % silent utterances.  Weak scaling: every rank gets its own batch of that size.

  python bench.py --gpus N --steps K --warmup W        (N>1: launched under torch.distributed.run)
Prints ONE JSON line on rank 0:
  value / ms_per_step   the K timed steps (barrier + synchronize on both sides, max over ranks)
  roofline              the kernel with the largest share of the step + a `kernels` list (MFMA- and HBM-bound ones):
                        HIP events around every launch on every 10th timed step (those steps run serially: no side stream)
  cpu_baseline          the oracle (CPU restatement of the reference step) on >= 16 packed rows of the same batch, this node's cores
  parity                mel-L1 of the HIP model (bf16 and exact-f32 kernels) against the oracle on those rows, same weights
  dtw / mel             BASELINE configs[2] (64 and 256 x 1000^2 DTW, HIP vs the oracle's C twin on 1 core and on all cores) and the
                        mel-target extraction (frames/s vs the numpy oracle)
  fp32_mode             the same step in exact-f32 kernels (the mode whose mel-L1 meets north_star's 1e-4), driver-timed
  ctc                   BASELINE configs[4]: the recognition step (768-d / 6-layer encoder + CTC, 128 000-sample batches, x2 accumulation)
  eval                  validation (`test()`, packed rows) and whole-utterance inference (`predict_utterance`, T up to ~1000: per-tile attention)
  pipeline              the per-batch device loader (DeviceBatchBuilder: raw recordings + audio -> batch dict), frames/s vs the host chain
--scaling strong deals ONE reference-size batch over the ranks by length (default: weak, a batch per rank).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBPS = 8000.0         # HBM3E spec (6.3 TB/s is what a streaming copy reaches)


def _median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def cpu_baseline(batch_cpu, rows_limit, warm, timed, model_sd, dev, want_pred=True):
    """The oracle (CPU restatement of the reference step, torch fp32 + compiled C DTW) timed on this node's host cores on a
    bounded sample of the same workload: the first utterances of the batch up to `rows_limit` packed rows.  Also returns the
    oracle's eval-free forward (dropout 0, shift 3) on that sample for the parity entry."""
    import subprocess
    subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'oracle')])
    from oracle import loss_ref, model_ref
    # 32 threads is the fastest setting on the GPU node's 2x64-core host for this small-batch fp32 step (probed:
    # 16 thr 0.41 s, 32 thr 0.34 s, 64 thr 0.76 s, 128 thr 1.70 s, 256 thr >100 s per fwd+bwd of 762 frames)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    n, frames = 0, 0
    while n < len(batch_cpu['lengths']) and (frames + batch_cpu['lengths'][n] + 199) // 200 <= rows_limit:
        frames += batch_cpu['lengths'][n]
        n += 1
    n = max(n, 1)
    sub = {k: v[:n] for k, v in batch_cpu.items()}
    frames = sum(sub['lengths'])
    sd = {k: v.detach().cpu().clone() for k, v in model_sd.items()}
    params = [v.requires_grad_(True) for k, v in sd.items() if v.dtype == torch.float32 and 'running' not in k and 'relative_positional' not in k]
    opt = torch.optim.AdamW(params, lr=1e-5, weight_decay=1e-7)
    g = torch.Generator().manual_seed(0)
    x_raw = loss_ref.combine_fixed_length(sub['raw_emg'], 1600)
    ref_pred = None
    if want_pred:
        with torch.no_grad():
            ref_pred, _ = model_ref.model_forward({k: v.detach().clone() for k, v in sd.items()}, x_raw.clone(), training=True, shift_r=3, running_out={})

    def step():
        opt.zero_grad()
        xr = loss_ref.combine_fixed_length(sub['raw_emg'], 1600)
        B, T = xr.shape[0], 200
        masks = []
        for _ in range(6):
            masks.append({'attn': (torch.rand(B, 8, T, T, generator=g) >= 0.2).float(), 'res1': (torch.rand(B, T, 768, generator=g) >= 0.2).float(),
                          'ffn': (torch.rand(B, T, 3072, generator=g) >= 0.2).float(), 'res2': (torch.rand(B, T, 768, generator=g) >= 0.2).float()})
        pred, aux = model_ref.model_forward(sd, xr, training=True, shift_r=3, running_out={}, layer_masks=masks, dropout_p=0.2)
        loss, _ = loss_ref.dtw_loss_ref(pred, aux, sub)
        loss.backward()
        opt.step()

    for _ in range(warm):
        step()
    ts = []
    for _ in range(timed):
        t0 = time.perf_counter()
        step()
        ts.append(time.perf_counter() - t0)
    dt = _median(ts)
    rows = (frames + 199) // 200
    out = {'value': frames / dt, 'unit': 'frames/s', 'cores': torch.get_num_threads(), 'kind': 'port',
           'sample': '%d utterances = %d frames (%d packed rows of 200) of the same synthetic batch, full 768-d/6-layer fp32 step '
                     '(fwd+dtw_loss+bwd+AdamW), median of %d timed steps after %d warm-up, %.2f s/step' % (n, frames, rows, timed, warm, dt)}
    return out, sub, ref_pred


def parity_entry(sub, ref_pred, model_sd, dev, dropout_p=0.2):
    """mel-L1 (mean |pred - oracle pred| in normalised log-mel units) of the HIP model on the CPU baseline's sample, same weights,
    training-mode statistics, shift 3: bf16 kernels (what the bench times) and exact-f32 kernels, with dropout off AND with the
    dropout the bench runs (p = 0.2; the oracle replays the kernels' masks, restated in oracle/dropout_ref.py).  The full-batch
    version with every gradient tensor is tests/test_fullsize.py."""
    from oracle import dropout_ref, loss_ref, model_ref
    from silent_speech_amd import _lib
    from silent_speech_amd.architecture import Model
    from silent_speech_amd.data_utils import combine_fixed_length

    class _R(object):
        @staticmethod
        def randrange(n):
            return 3
    out = {'rows': int(ref_pred.shape[0]), 'dropout': dropout_p,
           'reference': 'oracle/model_ref.py (fp32 torch restatement pinned to the reference by tests/golden), dropout masks replayed by oracle/dropout_ref.py'}
    sd_cpu = {k: v.detach().cpu().clone() for k, v in model_sd.items()}
    for name, dt, mm in (('bf16', torch.bfloat16, 'exact'), ('fp32', torch.float32, 'exact'), ('fp32_bf16x3', torch.float32, 'bf16x3')):
        for p in (0.0, dropout_p):
            m = Model(112, 80, 48, model_size=768, num_layers=6, dropout=p, compute_dtype=dt, f32_matmul=mm)
            m.load_state_dict(model_sd, strict=True)
            m.to(dev)
            m.shift_rng = _R
            m.train()
            X_raw = combine_fixed_length([t.to(dev) for t in sub['raw_emg']], 1600)
            with torch.no_grad():
                pred, _ = m(None, X_raw, None)
            want = ref_pred
            if p > 0:
                resident = m.attention_mask_family(200)
                B = int(X_raw.shape[0])
                masks = dropout_ref.layer_masks(m.last_seed, 6, B, 200, 768, 8, 3072, p, resident)
                with torch.no_grad():
                    want, _ = model_ref.model_forward(sd_cpu, loss_ref.combine_fixed_length(sub['raw_emg'], 1600), training=True, shift_r=3,
                                                      running_out={}, layer_masks=masks, dropout_p=p)
            out['mel_l1_%s%s_vs_oracle' % (name, '_dropout' if p > 0 else '')] = float((pred.float().cpu() - want).abs().mean())
            del m
    return out


def dtw_leg(dev, nb=64, n=1000):
    """BASELINE configs[2]: a batch of 64 cost matrices of 1000 x 1000 f32: the HIP wavefront kernel (cost skew + recurrence +
    backtrace, one launch) vs the oracle's C twin of align.py on 1 core (the reference's behaviour) and on all cores."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from oracle import dtw_ref
    from silent_speech_amd import align
    rng = np.random.default_rng(0)
    host = rng.random((nb, n, n), dtype=np.float32)
    flat = torch.from_numpy(host).to(dev)
    shapes, offs, strides = [(n, n)] * nb, [i * n * n for i in range(nb)], [(n, 1)] * nb
    job = align.DtwBatch(shapes, offs, strides, dev)        # descriptors / workspace once: the timed region is the two kernels
    res, res_offs = job.run(flat.view(-1))
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 5
    a.record()
    for _ in range(iters):
        job.run(flat.view(-1))
    b.record()
    torch.cuda.synchronize()
    t_gpu = a.elapsed_time(b) * 1e-3 / iters
    got = res.cpu().numpy()
    t0 = time.perf_counter()
    want0 = dtw_ref.align_from_distances_c(host[0])
    t1 = time.perf_counter() - t0
    t_1core = t1 * nb
    ok = bool((got[res_offs[0]:res_offs[0] + n] == np.asarray(want0)).all())
    cores = min(os.cpu_count() or 1, nb)
    with ThreadPoolExecutor(cores) as ex:        # ctypes releases the GIL: one matrix per core
        t0 = time.perf_counter()
        alls = list(ex.map(dtw_ref.align_from_distances_c, [host[i] for i in range(nb)]))
        t_all = time.perf_counter() - t0
    ok = ok and all(bool((got[res_offs[i]:res_offs[i] + n] == np.asarray(alls[i])).all()) for i in range(nb))
    byts = 8.0 * n * n * nb
    return {'workload': 'configs[2]: %d cost matrices %d x %d f32' % (nb, n, n), 'hip_ms': t_gpu * 1e3, 'matrices_per_s': nb / t_gpu,
            'bit_exact_vs_oracle': ok,
            'roofline': {'bound': 'hbm', 'achieved': byts / t_gpu / 1e9, 'peak': PEAK_HBM_GBPS, 'unit': 'GB/s', 'frac': byts / t_gpu / 1e9 / PEAK_HBM_GBPS,
                         'algorithmic_bytes': byts, 'note': '8 N M bytes per matrix (f32 cost in + f32 cumulative out, SURVEY 8d); latency-bound: N+M-1 dependent wavefronts'},
            'cpu_1core_ms': t_1core * 1e3, 'cpu_allcores_ms': t_all * 1e3, 'cpu_cores': cores,
            'speedup_vs_1core': t_1core / t_gpu, 'speedup_vs_allcores': t_all / t_gpu,
            'cpu_sample': 'oracle/dtw_ref.c (-O3): 1 core = %d x the time of one matrix; all cores = %d matrices on %d threads' % (nb, nb, cores)}


def mel_leg(dev, n_utt=32, seconds=6.0, saturating=True):
    """Mel-target extraction (data_utils.py:39-62) on the device -- one kernel from the signals to the log-mel values (csrc/mel.hip: an LDS radix-8 FFT per
    frame, sparse filterbank; rounds 1-5: STFT as two f32-MFMA GEMMs + magnitude + mel GEMM) -- vs the numpy oracle."""
    import numpy as np
    from oracle import mel_ref
    from silent_speech_amd.data_utils import mel_spectrogram
    L = int(22050 * seconds) // 256 * 256
    rng = np.random.default_rng(1)
    y = np.clip(0.1 * rng.standard_normal((n_utt, L)), -1, 1).astype(np.float32)
    yd = torch.from_numpy(y).to(dev)
    out = mel_spectrogram(yd, 1024, 80, 22050, 256, 1024, 0, 8000)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 20
    a.record()
    for _ in range(iters):
        mel_spectrogram(yd, 1024, 80, 22050, 256, 1024, 0, 8000)
    b.record()
    torch.cuda.synchronize()
    t_gpu = a.elapsed_time(b) * 1e-3 / iters
    frames = out.shape[0] * out.shape[2]
    byts = frames * (1024.0 + 320.0)          # 1024 B of unique audio in + 320 B out per frame
    res = {'workload': '%d utterances x %.1f s @ 22.05 kHz -> 80-bin log-mel' % (n_utt, seconds), 'frames': int(frames), 'hip_ms': t_gpu * 1e3,
           'frames_per_s': frames / t_gpu,
           'roofline': {'bound': 'hbm', 'achieved': byts / t_gpu / 1e9, 'peak': PEAK_HBM_GBPS, 'unit': 'GB/s', 'frac': byts / t_gpu / 1e9 / PEAK_HBM_GBPS,
                        'note': '1344 B per frame algorithmic; one launch per call: at this size a call is ~30 us of kernel behind ~40 us of Python / dispatcher work '
                                '(the `saturating` entry is the kernel at a size where the launch no longer matters); the kernel is VALU-issue-bound (~750 instructions '
                                'per frame and wave, tools/mel_probe.py), the dense-DFT formulation of rounds 1-5 ran at 0.5 % of this roofline'}}
    if saturating:
        t0 = time.perf_counter()
        ref = mel_ref.mel_spectrogram_ref(y[:4])
        t_cpu = (time.perf_counter() - t0) * n_utt / 4
        res.update({'cpu_frames_per_s': frames / t_cpu, 'cpu_kind': 'oracle/mel_ref.py (numpy rfft, 1 process)',
                    'mean_abs_err_vs_oracle': float(np.abs(out[:4].cpu().numpy() - ref).mean())})
        sat = mel_leg(dev, n_utt=512, seconds=8.0, saturating=False)
        res['saturating'] = {k: sat[k] for k in ('workload', 'frames', 'hip_ms', 'frames_per_s')}
        res['saturating']['roofline_frac'] = sat['roofline']['frac']
    return res


def csrc_fingerprint():
    """sha256 over the kernel sources: stamped into profiles/*_pmc_traffic.json when the counters are collected (tools/pmc_summary.py)
    and compared here, so that `traffic` figures measured on other kernels are reported as stale instead of silently reused."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'silent_speech_amd', 'csrc')
    for fn in sorted(os.listdir(d)):
        if fn.endswith('.hip') or fn.endswith('.h'):
            h.update(fn.encode()); h.update(open(os.path.join(d, fn), 'rb').read())
    return h.hexdigest()[:16]


def make_step(model, optim, batches, dp, it):
    """One training step per call, on batches[it % len(batches)]: the timed loop ROTATES distinct reference-size batches, so every step
    hands the path tensors it has not seen the step before -- the per-batch host work (pack tables, loss plan, their upload) is paid
    every step exactly as reference transduction_model.py:196-212 pays it for every DataLoader batch.  (Nothing in the package caches
    per-batch state across steps any more; a single-element list reproduces the old same-batch loop for the `same_batch` figure.)"""
    from silent_speech_amd.transduction_model import dtw_loss, prepare_batch
    if isinstance(batches, dict):
        batches = [batches]

    def step():
        batch = batches[it[0] % len(batches)]
        optim.zero_grad()
        i = it[0] + 1
        if i <= 500:
            for gp in optim.param_groups:
                gp['lr'] = i * 1e-3 / 500                                       # transduction_model.py:186-189
        X, X_raw, sess = prepare_batch(batch, batch['raw_emg'][0].device)       # the three combine_fixed_length calls + the loss plan, one upload
        if dp is not None:
            dp.begin_step(X_raw.shape[0] * 200, dp.local_target_frames(batch))
        pred, aux = model(X, X_raw, sess)
        total = dp.global_total(batch) if dp is not None else None
        loss, _ = dtw_loss(pred, aux, batch, phoneme_loss_weight=0.5, total_length=total)
        loss.backward()
        if dp is not None:
            dp.sync_gradients(model)
        optim.step()
        it[0] += 1
        return loss
    return step


def _time_steps(step, warm, timed):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(timed):
        loss = step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / timed, loss


def fp32_mode_leg(batch, init_sd, dev, frames, warm=2, timed=4, f32_matmul='exact'):
    """The SAME step with f32 activations / weights between the kernels (compute_dtype=float32), timed by the driver like the bf16 line:
    f32_matmul='exact' = f32 MFMA 16x16x4 (the mode of parity.mel_l1_fp32*), 'bf16x3' = every GEMM / attention product on three bf16
    MFMAs over operands split hi + lo in registers (parity.mel_l1_fp32_bf16x3*: the parity-grade FAST mode, still inside north_star's
    1e-4 mel-L1 of the reference)."""
    from silent_speech_amd.architecture import Model
    from silent_speech_amd.optim import FusedAdamW
    model = Model(112, 80, 48, model_size=768, num_layers=6, dropout=0.2, compute_dtype=torch.float32, f32_matmul=f32_matmul)
    model.load_state_dict(init_sd, strict=True)
    model.to(dev).train()
    optim = FusedAdamW(model, weight_decay=1e-7)
    dt, loss = _time_steps(make_step(model, optim, batch, None, [0]), warm, timed)
    if f32_matmul == 'bf16x3':
        return {'dtype': 'fp32 storage, bf16x3 MFMA', 'ms_per_step': dt * 1e3, 'frames_per_s': frames / dt, 'steps': timed, 'warmup': warm, 'final_loss': float(loss.detach()),
                'roofline_note': '3 bf16 MFMAs per product: 3 x 398 MFLOP/frame issued => %.0f %% of the bf16 MFMA peak' % (100.0 * frames * 3 * 398e6 / dt / 1e12 / PEAK_BF16_TFLOPS)}
    return {'dtype': 'fp32', 'ms_per_step': dt * 1e3, 'frames_per_s': frames / dt, 'steps': timed, 'warmup': warm, 'final_loss': float(loss.detach()),
            'roofline_note': 'f32 MFMA peak is %.1f TFLOP/s: 398 MFLOP/frame => %.0f %% of it' % (PEAK_F32_TFLOPS, 100.0 * frames * 398e6 / dt / 1e12 / PEAK_F32_TFLOPS)}


def ctc_leg(dev, warm=2, timed=6):
    """BASELINE configs[4]: recognition_model.py:92-107 -- shared encoder (768-d, 6 layers, bf16) with a 38-way output, CTC loss on the
    packed logits, an optimiser step every SECOND batch of a 128 000-sample budget.  One timed unit = two batches (forward, CTC,
    backward each) + the AdamW step.  The loss lines (:96-101) alone are timed too, next to the reference's own CPU path for them
    (torch CPU log_softmax + pad_sequence + F.ctc_loss on the same logits)."""
    import torch.nn.functional as F
    from silent_speech_amd.architecture import Model
    from silent_speech_amd.optim import FusedAdamW
    from silent_speech_amd.recognition_model import ctc_loss
    from silent_speech_amd.synthetic import reference_size_batch
    from silent_speech_amd.transduction_model import prepare_batch
    torch.manual_seed(1)
    model = Model(112, 38, model_size=768, num_layers=6, dropout=0.2, compute_dtype=torch.bfloat16).to(dev)
    model.train()
    optim = FusedAdamW(model, lr=3e-4, weight_decay=0.0)
    batches = [reference_size_batch(seed=s, budget=128000, device=dev) for s in (11, 12)]
    frames = sum(sum(b['lengths']) for b in batches)

    def unit():
        optim.zero_grad()
        for b in batches:
            X, X_raw, sess = prepare_batch(b, dev, loss_plan=False)
            loss = ctc_loss(model(X, X_raw, sess), b, blank=37)
            loss.backward()
        optim.step()
        return loss
    dt, loss = _time_steps(unit, warm, timed)
    # the loss lines alone, on the logits of the first batch
    b = batches[0]
    with torch.no_grad():
        X, X_raw, sess = prepare_batch(b, dev, loss_plan=False)
        pred = model(X, X_raw, sess).float()
    logits = pred.detach().clone().requires_grad_(True)
    ctc_loss(logits, b, blank=37).backward()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 10
    e0.record()
    for _ in range(iters):
        logits.grad = None
        gl = ctc_loss(logits, b, blank=37)
        gl.backward()
    e1.record()
    torch.cuda.synchronize()
    t_call = e0.elapsed_time(e1) * 1e-3 / iters
    # the device time of the op itself (lse + alpha/beta + gradient kernels): the utterance tables built once, the launches back to back -- the loop above is bound by
    # the host (numpy tables, one pinned upload, dispatcher, autograd) at this size: ~0.35 ms per call against ~0.18 ms of kernels
    from silent_speech_amd.recognition_model import _CtcPlan
    plan = _CtcPlan(b['lengths'], b['text_int'], int(logits.shape[0] * logits.shape[1]), dev)
    flat = logits.detach().reshape(-1, 38).contiguous()
    for _ in range(3):
        torch.ops.silent_speech.ctc_loss(flat, plan.desc, plan.targets, plan.n, plan.max_s, plan.ws_floats, 38, 37)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        torch.ops.silent_speech.ctc_loss(flat, plan.desc, plan.targets, plan.n, plan.max_s, plan.ws_floats, 38, 37)
    e1.record()
    torch.cuda.synchronize()
    t_loss = e0.elapsed_time(e1) * 1e-3 / 20
    # the reference's own lines on the host
    from silent_speech_amd.data_utils import decollate_tensor
    cpu_logits = pred.cpu().requires_grad_(True)
    lens = b['lengths']
    tgt = torch.nn.utils.rnn.pad_sequence([t.cpu() for t in b['text_int']], batch_first=True)
    tl = [int(t.shape[0]) for t in b['text_int']]
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    t0 = time.perf_counter()
    lp = F.log_softmax(cpu_logits, 2)
    lp = torch.nn.utils.rnn.pad_sequence(decollate_tensor(lp, lens), batch_first=False)
    ref = F.ctc_loss(lp, tgt, lens, tl, blank=37)
    ref.backward()
    t_cpu = time.perf_counter() - t0
    M, V = int(pred.shape[0] * pred.shape[1]), 38
    byts = M * V * 4.0 * 3 + M * 4.0 * 2                  # logits read twice (lse, gradient) + gradient written; lse / argmax
    derr = float((logits.grad.cpu() - cpu_logits.grad).abs().max() / (cpu_logits.grad.abs().max() + 1e-30))
    return {'workload': 'configs[4]: recognition step, 768-d / 6-layer encoder bf16 + CTC (38 classes), 2 batches of a 128 000-sample budget per optimiser step, dropout 0.2',
            'frames_per_unit': frames, 'utterances': [len(x['lengths']) for x in batches], 'ms_per_unit': dt * 1e3, 'frames_per_s': frames / dt,
            'final_loss': float(loss.detach()),
            'ctc_loss': {'hip_ms': t_loss * 1e3, 'call_ms': t_call * 1e3, 'frames': int(sum(lens)), 'cpu_ms': t_cpu * 1e3, 'cpu_kind': 'torch CPU log_softmax + pad_sequence + F.ctc_loss + backward (the reference\'s own lines, recognition_model.py:96-101), %d threads' % torch.get_num_threads(),
                         'loss_hip': float(gl.detach()), 'loss_cpu': float(ref.detach()), 'grad_max_err_over_max': derr,
                         'roofline': {'bound': 'hbm', 'achieved': byts / t_loss / 1e9, 'peak': PEAK_HBM_GBPS, 'unit': 'GB/s', 'frac': byts / t_loss / 1e9 / PEAK_HBM_GBPS,
                                      'note': 'latency-bound: T dependent alpha/beta columns per utterance; one workgroup per (utterance, direction), its waves a pipeline over the states + a loader wave (csrc/ctc.hip, round 6); hip_ms = device time of the op (its three kernels, tables built once, 20 launches back to back; rounds 3-5 timed the whole call here: 0.785 ms, kernel-bound then); call_ms = the whole ctc_loss(...).backward() call incl. building and uploading the utterance tables (host-bound at this size)'}}}


def eval_leg(dev):
    """Row N2: `test()` (transduction_model.py:33-55: eval-mode forward on packed rows + dtw_loss with the confusion matrix) over a synthetic
    dev set, and whole-utterance inference (`predict_utterance`, :57-66) where T runs up to ~1000 frames and the per-tile banded attention
    kernels run instead of the LDS-resident ones."""
    from silent_speech_amd.architecture import Model
    from silent_speech_amd.synthetic import SyntheticEMGDataset
    from silent_speech_amd.transduction_model import predict_utterance, test
    torch.manual_seed(2)
    model = Model(112, 80, 48, model_size=768, num_layers=6, dropout=0.2, compute_dtype=torch.bfloat16).to(dev)
    ds = SyntheticEMGDataset(64, seed=5, min_frames=200, max_frames=860)
    frames = sum(int(ds[i]['emg'].shape[0]) for i in range(len(ds)))
    test(model, ds, dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss, acc, _ = test(model, ds, dev)
    torch.cuda.synchronize()
    t_test = time.perf_counter() - t0
    long_ds = SyntheticEMGDataset(16, seed=6, min_frames=600, max_frames=1000, silent_fraction=0.0)
    lframes = sum(int(long_ds[i]['emg'].shape[0]) for i in range(len(long_ds)))
    for i in range(2):
        predict_utterance(model, long_ds[i], dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(len(long_ds)):
        out = predict_utterance(model, long_ds[i], dev)
    torch.cuda.synchronize()
    t_long = time.perf_counter() - t0
    return {'test': {'utterances': len(ds), 'frames': frames, 'ms': t_test * 1e3, 'frames_per_s': frames / t_test, 'loss': float(loss), 'phoneme_acc': float(acc),
                     'note': 'eval-mode forward on batches of 32 utterances packed into 200-frame rows + dtw_loss(eval) incl. one DTW per silent utterance and the phoneme confusion matrix accumulated on the device (one read-back per test())'},
            'whole_utterance': {'utterances': len(long_ds), 'frames': lframes, 'ms': t_long * 1e3, 'frames_per_s': lframes / t_long, 'finite': bool(torch.isfinite(out).all()),
                                'note': 'predict_utterance: B = 1, T = 600..1000 (un-chunked): per-tile banded attention kernels (cost linear in T)'}}


def pipeline_leg(dev, n_utt=24):
    """Row N3: DeviceBatchBuilder -- raw 1 kHz recordings (+ filter context) and 22.05 kHz audio -> the collate_raw batch dict on the device --
    against the same chain on the host (the oracle's numpy restatement of scipy filtfilt / np.interp / the STFT-mel path), per frame."""
    import numpy as np
    from oracle import filter_ref, mel_ref
    from silent_speech_amd.pipeline import DeviceBatchBuilder
    rng = np.random.default_rng(4)
    recs = []
    for _ in range(n_utt):
        T = int(rng.integers(200, 861))
        n = int(T * 8 / 0.68906) + 40
        x = np.cumsum(rng.standard_normal((n + 400, 8)), 0) + 40.0 * np.sin(2 * np.pi * 60.0 * np.arange(n + 400) / 1000.0)[:, None] + rng.standard_normal((n + 400, 8)) * 30.0
        recs.append({'raw_emg': x[200:200 + n], 'raw_emg_before': x[:200], 'raw_emg_after': x[200 + n:], 'silent': False,
                     'audio': np.clip(0.1 * rng.standard_normal(256 * (T + 2)), -1, 1).astype(np.float32), 'text_int': np.zeros(3, dtype=np.int64)})
    builder = DeviceBatchBuilder(dev)
    for _ in range(10):                                    # steady state: EVERY slot of the pinned staging ring (8 slots, visited in turn by the batch's
        b = builder.build(recs)                            # three uploads) has grown to the size of the large one -- a pinned allocation of 32 MB costs 1-2 ms
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        b = builder.build(recs)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    t_dev = _median(ts)
    frames = int(sum(b['lengths']))
    k = 3
    t0 = time.perf_counter()
    for r in recs[:k]:
        e689, e516 = filter_ref.condition(r['raw_emg'], r['raw_emg_before'], r['raw_emg_after'])
        mel_ref.mel_spectrogram_ref(r['audio'][None])
    t_cpu = (time.perf_counter() - t0) * n_utt / k
    return {'workload': '%d recordings (%d frames): 8-filter zero-phase IIR cascade + resample + soft clip, batched STFT / mel / normalise -> batch dict' % (n_utt, frames),
            'hip_ms': t_dev * 1e3, 'frames_per_s': frames / t_dev, 'includes': 'H2D of the raw recordings and audio (host arrays in, one pinned upload per kind); ONE ragged filter / resample launch sequence and one DFT GEMM for the whole batch, the audio half on a side stream; median of 5 builds after 10 warm ones (the 8-slot pinned ring at its final size)',
            'cpu_frames_per_s': frames / t_cpu, 'cpu_kind': 'oracle/filter_ref.py + oracle/mel_ref.py (numpy, 1 process; %d of %d recordings timed, scaled)' % (k, n_utt)}


def n1_comparison(args, local_rank):
    """The same command on ONE GPU (this rank's), run by rank 0 after the process group is gone: the N = 1 figure the N-GPU line is read against,
    measured on the same node in the same call.  No legs, no CPU baseline, no per-launch events."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'MASTER_PORT', 'GROUP_RANK', 'ROLE_RANK',
                                                           'TORCHELASTIC_RUN_ID', 'SS_BENCH_LAUNCHER', 'SS_BENCH_SCHEDULE')}
    env['HIP_VISIBLE_DEVICES'] = env.get('HIP_VISIBLE_DEVICES', '').split(',')[local_rank] if env.get('HIP_VISIBLE_DEVICES') else str(local_rank)
    if os.environ.get('SS_BENCH_BACKEND', 'nccl') != 'nccl':
        env.pop('HIP_VISIBLE_DEVICES', None)          # gloo ranks shared the GPUs that exist
    cmd = [sys.executable, os.path.abspath(__file__), '--gpus', '1', '--steps', str(args.steps), '--warmup', str(args.warmup), '--dtype', args.dtype,
           '--batches', str(args.batches), '--no-legs', '--no-profile', '--no-same', '--cpu-rows', '0']
    try:
        outp = subprocess.check_output(cmd, env=env, timeout=900).decode(errors='replace')
        d = json.loads([l for l in outp.splitlines() if l.startswith('{')][-1])
        return {'value': d['value'], 'unit': d['unit'], 'ms_per_step': d['ms_per_step'], 'steps': d['steps'], 'n_gpus': 1,
                'note': 'the same loop on one GPU of this node, run by rank 0 after the N-rank measurement (no legs, no per-launch events)'}
    except Exception as e:          # noqa: BLE001
        return {'value': None, 'error': repr(e)[:300]}


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no torchrun around it: this process becomes the launcher.  It starts N copies of this
    script, one rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / a free MASTER_PORT in the environment, exactly what
    torch.distributed.run would export), forwards rank 0's JSON line and returns the worst exit code.  Backend: RCCL ('nccl') when the node
    has a GPU per rank, otherwise gloo ranks sharing the GPUs that exist (a functional run of the N > 1 path; the line says which).
    Should the two-communicator schedule (gradient buckets on their own RCCL communicator, BatchNorm exchanges on the default one) fail
    on this RCCL build, the run is repeated ONCE with SS_DP_SINGLE_GROUP=1 and the line records which schedule produced the number."""
    import socket
    import subprocess
    n = args.gpus
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if ndev == 0:
        raise SystemExit('bench.py needs an MI355X (no CPU fallback exists for the product path)')
    backend = os.environ.get('SS_BENCH_BACKEND') or ('nccl' if ndev >= n else 'gloo')
    schedules = [('two_communicators', {})]
    if os.environ.get('SS_DP_SINGLE_GROUP', '0') == '1':
        schedules = [('single_communicator', {'SS_DP_SINGLE_GROUP': '1'})]
    elif backend == 'nccl':
        schedules.append(('single_communicator', {'SS_DP_SINGLE_GROUP': '1'}))
    limit = float(os.environ.get('SS_BENCH_LAUNCH_TIMEOUT', '1500'))
    failures = []
    for name, extra in schedules:
        s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
        procs = []
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                       SS_BENCH_BACKEND=backend, SS_BENCH_SCHEDULE=name, SS_BENCH_LAUNCHER='self', SS_BENCH_FAILED_SCHEDULES=';'.join(failures), **extra)
            env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
        t0, rc, out0 = time.time(), None, b''
        import threading
        def drain():                                  # rank 0's stdout is read while it runs (a full pipe would block it)
            nonlocal out0
            out0 = procs[0].stdout.read()
        th = threading.Thread(target=drain, daemon=True); th.start()
        while True:
            codes = [p.poll() for p in procs]
            if any(c not in (None, 0) for c in codes):
                rc = next(c for c in codes if c not in (None, 0))
                break
            if all(c == 0 for c in codes):
                rc = 0
                break
            if time.time() - t0 > limit:
                rc = 124
                break
            time.sleep(0.2)
        for p in procs:                               # exactly the processes started above
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill(); p.wait()
        th.join(timeout=5)
        lines = [l for l in out0.decode(errors='replace').splitlines() if l.startswith('{')]
        if rc == 0 and lines:
            print(lines[-1], flush=True)
            return 0
        failures.append('%s: rc %s' % (name, rc))
        sys.stderr.write('bench.py launcher: schedule %s failed (rc %s)%s\n' % (name, rc, '; retrying with one communicator' if name != schedules[-1][0] else ''))
    return 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32', 'fp32x3'], help="fp32x3: f32 storage, every matmul product on three bf16 MFMAs (Model(f32_matmul='bf16x3'))")
    ap.add_argument('--cpu-rows', type=int, default=16, help='packed rows of the batch given to the CPU baseline (0 = skip baseline and parity)')
    ap.add_argument('--cpu-steps', type=int, default=5)
    ap.add_argument('--cpu-warmup', type=int, default=3)
    ap.add_argument('--batches', type=int, default=4, help='distinct reference-size batches rotated through warm-up and the timed loop (1 = the same batch every step)')
    ap.add_argument('--no-same', action='store_true', help='skip the same-batch comparison loop after the timed loop (profiling runs)')
    ap.add_argument('--no-profile', action='store_true', help='no per-launch HIP events (roofline entry omitted)')
    ap.add_argument('--no-legs', action='store_true', help='skip the DTW (configs[2]), mel, fp32-mode, CTC (configs[4]), eval and pipeline legs')
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'], help='weak: one reference-size batch per rank; strong: ONE such batch dealt over the ranks by length')
    ap.add_argument('--cpu-full', type=int, default=1, help='also time the CPU baseline on the FULL batch (1 warm-up + 2 timed steps, ~1 min); 0 = sample only')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world and world > 1:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(launch_ranks(args))          # plain `python bench.py --gpus N`: start the N ranks from here
    if args.gpus > 1 and world == 1:
        raise SystemExit('--gpus %d but WORLD_SIZE=1' % args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback exists for the product path)')
    if os.environ.get('SS_BENCH_BACKEND', 'nccl') != 'nccl':
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = os.environ.get('SS_BENCH_BACKEND', 'nccl')          # 'gloo' lets two ranks share one GPU (functional check of the N>1 path)
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    from silent_speech_amd import _lib, engine, ops, staging
    from silent_speech_amd.architecture import Model
    from silent_speech_amd.data_utils import combine_fixed_length
    from silent_speech_amd.distributed import DataParallel
    from silent_speech_amd.optim import FusedAdamW
    from silent_speech_amd.synthetic import reference_size_batch
    from silent_speech_amd.transduction_model import dtw_loss

    dt = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    torch.manual_seed(0)
    model = Model(112, 80, 48, model_size=768, num_layers=6, dropout=0.2, compute_dtype=dt, f32_matmul='bf16x3' if args.dtype == 'fp32x3' else 'exact').to(dev)
    model.train()
    init_sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    dp = DataParallel() if world > 1 else None
    if dp is not None:
        dp.attach(model)
    optim = FusedAdamW(model, weight_decay=1e-7)
    def to_dev(b):
        return {k: ([t.to(dev) for t in v] if isinstance(v, list) and len(v) and torch.is_tensor(v[0]) else v) for k, v in b.items()}
    nb = max(1, args.batches)
    if args.scaling == 'strong' and world > 1:
        # ONE global batch per rotation slot, utterances dealt round-robin in order of decreasing length: frames (and DTW problems) per rank stay balanced
        batches_cpu = []
        for j in range(nb):
            g = reference_size_batch(seed=100 * j)
            order = sorted(range(len(g['lengths'])), key=lambda i: -g['lengths'][i])[rank::world]
            batches_cpu.append({k: [v[i] for i in order] for k, v in g.items()})
    else:
        batches_cpu = [reference_size_batch(seed=rank + 100 * j) for j in range(nb)]      # slot 0 = the batch of the earlier rounds (seed = rank)
    batch_cpu = batches_cpu[0]
    batches = [to_dev(b) for b in batches_cpu]
    batch = batches[0]
    frames_of = [sum(b['lengths']) for b in batches]
    frames = frames_of[0]
    rows = (frames + 199) // 200
    it = [0]
    step = make_step(model, optim, batches, dp, it)

    for _ in range(args.warmup):
        loss = step()
    torch.cuda.synchronize()
    if not torch.isfinite(loss.detach()).item():
        raise SystemExit('non-finite loss after warm-up')
    plan = engine.plan_binding(model)
    L = _lib.lib()
    prof = None
    if not args.no_profile:
        prof = ops.LaunchProfiler()
        ops.PROFILER = prof
    if dp is not None:
        dp.measure_exposed = True                  # events on both sides of sync_gradients' waits: the un-hidden part of the gradient all-reduce
        dp._exposed = []
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host_enqueue = ring_wait = 0.0
    pstride = max(10, args.steps // 2)             # per-launch events on one step in ten of the timed region (two steps of a long run): they run serially (no side stream)
    timed_frames = 0
    for i in range(args.steps):
        timed_frames += frames_of[it[0] % len(batches)]
        profiled = prof is not None and i % pstride == 0
        if prof is not None:
            # every 10th timed step carries the per-launch HIP events (roofline numerator); those steps run serially (no side stream):
            # a duration taken while a second stream shares the CUs is not a per-kernel quantity
            prof.enabled = profiled
            L.ss_plan_profile(plan.handle, int(profiled))
            engine.SIDE_STREAM_ENABLED = not profiled
        th, tw = time.perf_counter(), staging.WAIT_SECONDS[0]
        loss = step()
        if not profiled:
            host_enqueue += time.perf_counter() - th
            ring_wait += staging.WAIT_SECONDS[0] - tw
    engine.SIDE_STREAM_ENABLED = True
    L.ss_plan_profile(plan.handle, 0)
    n_plain = args.steps - (len(range(0, args.steps, pstride)) if prof is not None else 0)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ops.PROFILER = None
    final_loss = float(loss.detach())
    comm = None
    if dp is not None:
        exposed = dp.exposed_ms()
        dp.measure_exposed = False
        prop = torch.cuda.get_device_properties(dev)
        mine = {'rank': rank, 'local_rank': local_rank, 'device': dev.index, 'name': prop.name, 'pci_bus_id': getattr(prop, 'pci_bus_id', None),
                'exposed_ms': exposed, 'pid': os.getpid()}
        everyone = [None] * world
        dist.all_gather_object(everyone, mine, group=dp.host_group)          # host-side (gloo) group: no device traffic
        backend_used = DataParallel._backend_of(None)
        nbytes, ncoll = dp.bucket_bytes()
        rccl_version = None
        if backend_used == 'nccl':
            try:
                rccl_version = '.'.join(str(x) for x in torch.cuda.nccl.version())
            except Exception:          # noqa: BLE001
                pass
        comm = {'rccl': {'backend': backend_used + (' (= RCCL on ROCm)' if backend_used == 'nccl' else ' (ranks share the GPUs that exist: functional run of the N > 1 path, not a scaling figure)'),
                         'world_size': dist.get_world_size(), 'rccl_version': rccl_version, 'ranks': [{k: e[k] for k in ('rank', 'local_rank', 'device', 'name', 'pci_bus_id')} for e in everyone],
                         'distinct_devices': len({(e['device'], e['pci_bus_id']) for e in everyone}),
                         'communicators': {'gradient_buckets': 'own communicator' if dp.bucket_group is not dp.group else 'default communicator',
                                           'batchnorm_sums': 'default communicator', 'host_counts': 'gloo side group' if dp.host_group is not None else 'default group',
                                           'count': 2 if dp.bucket_group is not dp.group else 1},
                         'schedule': dp.schedule, 'launcher': os.environ.get('SS_BENCH_LAUNCHER', 'torch.distributed.run'),
                         'failed_schedules': [x for x in os.environ.get('SS_BENCH_FAILED_SCHEDULES', '').split(';') if x]},
                'allreduce': {'bytes_per_step': nbytes, 'buckets': ncoll, 'dtype': 'bf16' if dp.grad_dtype is not None else 'f32',
                              'layer_buckets': bool(dp.layer_buckets), 'batchnorm_collectives_per_step': 12,
                              'exposed_ms': max((e['exposed_ms'] or 0.0) for e in everyone), 'exposed_ms_per_rank': [e['exposed_ms'] for e in everyone],
                              'exposed_note': 'device time the main stream waits in sync_gradients (HIP events on both sides of the waits), mean over the timed steps, max over ranks'}}

    # the same loop on ONE batch (what rounds 1-3 timed), un-profiled: how much of a step is per-batch host work / uploads
    # Un-profiled comparison loops after the timed one (the timed loop carries per-launch events on two of its steps, which run
    # without the side stream): the SAME loop rotating the batches, and the loop rounds 1-3 reported (batch 0 every step).
    same_ms = rot_ms = float('nan')
    n_same, rot_frames = 0, 0
    if not args.no_same:
        n_same = min(max(args.steps, len(batches)), 12)

        def loop(fn, n):
            for _ in range(2):
                fn()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            ts = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            return (time.perf_counter() - ts) / n * 1e3
        L.ss_plan_profile(plan.handle, 0)
        rot_it = [it[0]]
        rot_ms = loop(make_step(model, optim, batches, dp, rot_it), n_same)
        rot_frames = sum(frames_of[(rot_it[0] - n_same + j) % len(batches)] for j in range(n_same)) / n_same
        same_ms = loop(make_step(model, optim, [batch], dp, [rot_it[0]]), n_same)

    stats = torch.tensor([elapsed, float(timed_frames), same_ms, rot_ms], dtype=torch.float64, device=dev)
    if world > 1:
        mx = stats.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed, total_frames, same_ms, rot_ms = float(mx[0]), float(sm[1]), float(mx[2]), float(mx[3])
    else:
        total_frames = float(timed_frames)

    if rank == 0:
        out = {
            'metric': 'EMG frames/s training (transduction_model.py)', 'value': total_frames / elapsed, 'unit': 'frames/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': args.scaling if world > 1 else 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': 'configs[1]: full transduction model (768-d, 6-layer rel-pos encoder, 3 ResBlocks) training step '
                                   '(pack+fwd+dtw_loss incl. on-device DTW+bwd+AdamW), dropout 0.2, synthetic 8-ch EMG',
                       'batch_rotation': '%d distinct reference-size batches (seeds rank + 100 j) resident in HBM, rotated through warm-up and the timed '
                                         'loop: every step packs and plans a batch it did not see the step before (no per-batch cache exists in the package)' % len(batches),
                       'frames_per_gpu_step': [int(f) for f in frames_of], 'rows_per_gpu_step': [(f + 199) // 200 for f in frames_of],
                       'utterances_per_gpu_step': [len(b['lengths']) for b in batches],
                       'silent_utterances': [int(sum(b['silent'])) for b in batches], 'parallelism': 'dp%d' % world, 'final_loss': final_loss,
                       'unprofiled': None if n_same == 0 else {
                           'steps': n_same, 'ms_per_step_rotated': rot_ms, 'frames_per_s_rotated': rot_frames / rot_ms * 1e3,
                           'ms_per_step_same_batch': same_ms, 'frames_per_s_same_batch': frames / same_ms * 1e3,
                           'note': 'after the timed loop, no per-launch events (the timed loop runs one step in ten serially for the roofline table): '
                                   'the rotating loop again, and batch 0 on every step (what rounds 1-3 timed)'},
                       'host_enqueue_ms_per_step': (host_enqueue - ring_wait) / max(n_plain, 1) * 1e3,
                       'host_ring_wait_ms_per_step': ring_wait / max(n_plain, 1) * 1e3,
                       'host_enqueue_note': 'host time to enqueue one un-profiled step (forward and backward are one native call each), EXCLUDING the time the host '
                                            'sat blocked on the staging ring (a host more than 8 batches ahead of the GPU waits for a slot: back-pressure, reported '
                                            'separately as host_ring_wait_ms_per_step)'},
        }
        if comm is not None:
            out.update(comm)
        if prof is not None:
            psteps = len(range(0, args.steps, pstride))
            rows_buf = (_lib.ProfileRow * 64)()
            nrows = L.ss_plan_profile_read(plan.handle, rows_buf, 64)
            table = {}
            for r in rows_buf[:nrows]:
                table[r.name.decode()] = dict(calls=int(r.calls), flops=float(r.flops), bytes=float(r.bytes), seconds=float(r.seconds))
            for k, v in prof.summary().items():
                table[k] = v
            pmc, pmc_src, pmc_stale = {}, None, None
            for fn in ('r06_pmc_traffic.json', 'r05_pmc_traffic.json', 'r04_pmc_traffic.json', 'r03_pmc_traffic.json', 'r02_pmc_traffic.json', 'r01_pmc_traffic.json'):
                try:        # HBM bytes per launch from the committed rocprofv3 PMC passes of this same command (tools/pmc_summary.py)
                    pmc = json.load(open(os.path.join(ROOT, 'profiles', fn)))
                    pmc_src = 'profiles/' + fn
                    break
                except Exception:
                    continue
            # the counters belong to the kernels they were collected on: the file carries a fingerprint of csrc/ (tools/pmc_summary.py);
            # when the sources have changed since, the figures are reported as stale and `traffic` stays null
            here = csrc_fingerprint()
            stamp = (pmc.get('_meta') or {}).get('csrc_fingerprint')
            pmc_stale = stamp != here
            if pmc_stale:
                pmc = {}
            hints = {'gemm8_kc_kernel (288x256)': 'gemm8_kc_kernel<unsigned short, 9', 'gemm8_kc_kernel (256x256)': 'gemm8_kc_kernel<unsigned short, 8',
                     'gemm_dw_grouped': 'gemm8_dwk_kernel', 'gemm_smallk_kernel (K <= 32, first conv)': 'gemm_smallk_kernel', 'gemm_w2_kernel (128|144 x 128)': 'gemm_w2_kernel', 'gemm_glds_kernel (128x128)': 'gemm_glds_kernel',
                     'attn_fwd': 'attn_t_fwd_kernel', 'attn_bwd': ('attn_t_bwd_kv_kernel', 'attn_t_bwd_q_kernel'), 'bn_stats': 'bn_partial_kernel', 'bn_apply': 'bn_apply_kernel',
                     'bn_bwd_sums': ('bn_bwd_partial_kernel', 'bn_bwd_finalize_kernel'), 'bn_bwd_apply': 'bn_bwd_apply_kernel', 'add_dropout_ln_fwd': 'add_dropout_ln_fwd_kernel',
                     'ln_bwd': ('ln_bwd2_kernel<', 'ln_bwd2_finalize_kernel'), 'adamw_kernel': 'adamw_kernel', 'dtw_kernel': 'dtw_kernel', 'silent_cost_skewed_kernel': 'silent_cost_skewed_kernel',
                     'colsum': 'colsum_partial_kernel', 'permute3d_batch (weight re-layout)': 'permute3d_batch_kernel', 'grad_unlayout': 'permute3d_batch_f32_kernel'}

            def traffic_of(name):              # a logical launch may be several kernels (attention backward = 3): their HBM bytes add up
                hs = hints.get(name)
                if not hs:
                    return None
                hs = (hs,) if isinstance(hs, str) else hs
                tot, hit = 0.0, False
                for h in hs:
                    for k, v in pmc.items():
                        if k != '_meta' and h in k:
                            tot += v.get('hbm_bytes_per_launch') or 0.0
                            hit = True
                            break
                return tot if hit else None
            kernels = []
            total_s = sum(v['seconds'] for v in table.values())
            for name, v in sorted(table.items(), key=lambda kv: -kv[1]['seconds']):
                if v['seconds'] <= 0 or v['calls'] == 0:
                    continue
                mfma = v['flops'] > 0
                f32_mfma = args.dtype == 'fp32'
                peak = (PEAK_F32_TFLOPS if f32_mfma else PEAK_BF16_TFLOPS) if mfma else PEAK_HBM_GBPS
                ach = v['flops'] / v['seconds'] / 1e12 if mfma else v['bytes'] / v['seconds'] / 1e9
                kernels.append({'kernel': name, 'bound': 'mfma' if mfma else 'hbm', 'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s' if mfma else 'GB/s',
                                'frac': ach / peak, 'ms_per_step': v['seconds'] / psteps * 1e3, 'launches_per_step': v['calls'] / psteps,
                                'avg_launch_us': v['seconds'] / v['calls'] * 1e6,
                                'algorithmic_per_launch': (v['flops'] if mfma else v['bytes']) / v['calls'], 'traffic': traffic_of(name)})
                k = kernels[-1]
                if k['traffic']:        # the measured HBM bytes of a launch over its duration: what share of the 8 TB/s the kernel actually drew
                    k['traffic_frac_of_hbm_peak'] = k['traffic'] / (k['avg_launch_us'] * 1e-6) / 1e9 / PEAK_HBM_GBPS
                if name in ('attn_fwd', 'attn_bwd'):
                    # the roof that binds at this shape is HBM: Q, K, V in + O out (forward), Q, K, V, dO, O in + dQ, dK, dV out (backward) -- the
                    # plan's byte count -- plus the saved probability image once per launch (bf16, 88 MB for the 22 000 frames x 8 heads of the reference batch)
                    alg = v['bytes'] / v['calls'] + 88.0e6 * (v['bytes'] / v['calls']) / ((4 if name == 'attn_fwd' else 8) * 22000 * 768 * 2.0)
                    k['hbm'] = {'bound': 'hbm', 'algorithmic_bytes_per_launch': alg, 'achieved': alg / (k['avg_launch_us'] * 1e-6) / 1e9, 'peak': PEAK_HBM_GBPS, 'unit': 'GB/s',
                                'frac': alg / (k['avg_launch_us'] * 1e-6) / 1e9 / PEAK_HBM_GBPS}
                    k['note'] = ('counted as MFMA work (band-limited flops: 3 products forward, 5 backward); transposed-score kernels of round 5 '
                                 '(csrc/attention_t.hip: S^T = K Q^T on 32x32x16 MFMAs, backward = query-major + key-major kernel on the saved probabilities, '
                                 'D folded into the query-major one).  At T = 200 / d_head = 96 with the saved image the forward moves ~75 flops per HBM byte '
                                 'against a machine balance of 312: traffic_frac_of_hbm_peak is the second figure to read (DESIGN.md section 4)')
            top = kernels[0]
            out['roofline'] = {'bound': top['bound'], 'achieved': top['achieved'], 'peak': top['peak'], 'unit': top['unit'], 'frac': top['frac'],
                               'traffic': top['traffic'], 'traffic_source': pmc_src if pmc and top['traffic'] is not None else None,
                               'traffic_stale': bool(pmc_stale), 'csrc_fingerprint': here, 'traffic_fingerprint': stamp, 'kernel': top['kernel'],
                               'launches_per_step': top['launches_per_step'], 'avg_launch_us': top['avg_launch_us'],
                               'algorithmic_per_launch': top['algorithmic_per_launch'], 'event_timed_steps': psteps,
                               'serial_kernel_ms_per_step': total_s / psteps * 1e3,
                               'timing': 'HIP events around every kernel launch (inside the native plan: ss_plan_profile; Python-launched kernels: '
                                         'torch events on the launch stream) on every 10th (runs of <= 20 steps) / every (steps/2)-th timed step; those steps run without the side stream '
                                         '(exclusive durations); rocprofv3 counterpart: profiles/r06_serial_kernel_stats.txt',
                               'kernels': kernels[:16]}
        if world == 1 and args.cpu_rows > 0:
            base, sub, ref_pred = cpu_baseline(batch_cpu, args.cpu_rows, args.cpu_warmup, args.cpu_steps, init_sd, dev)
            out['cpu_baseline'] = base
            out['parity'] = parity_entry(sub, ref_pred, init_sd, dev)
            if args.cpu_full:        # SURVEY 8d asks for the identical batch: the whole 110-row step is the headline CPU figure, the bounded
                full, _, _ = cpu_baseline(batch_cpu, 10 ** 9, 2, 5, init_sd, dev, want_pred=False)        # sample (faster per frame: it fits the caches) rides along
                full['bounded_sample'] = {k: base[k] for k in ('value', 'unit', 'cores', 'sample')}
                out['cpu_baseline'] = full
        if world == 1 and not args.no_legs:
            out['dtw'] = dtw_leg(dev)
            out['dtw']['saturating'] = {k: v for k, v in dtw_leg(dev, nb=256).items() if k in ('workload', 'hip_ms', 'matrices_per_s', 'roofline', 'bit_exact_vs_oracle')}
            out['mel'] = mel_leg(dev)
            out['fp32_mode'] = fp32_mode_leg(batch, init_sd, dev, frames)
            out['fp32_bf16x3_mode'] = fp32_mode_leg(batch, init_sd, dev, frames, warm=3, timed=8, f32_matmul='bf16x3')
            # the mode that meets north_star's precision bar (mel-L1 < 1e-4 of the reference function) at a usable speed, next to the bf16 headline:
            # f32 storage, every product on three bf16 MFMAs; mel_l1 = against the oracle on the SAME dropout masks (the `parity` object)
            x3 = out['fp32_bf16x3_mode']
            out['parity_grade'] = {'dtype': x3['dtype'], 'ms_per_step': x3['ms_per_step'], 'frames_per_s': x3['frames_per_s'],
                                   'mel_l1': (out.get('parity') or {}).get('mel_l1_fp32_bf16x3_dropout_vs_oracle'), 'mel_l1_bar': 1e-4,
                                   'vs_cpu_baseline': (x3['frames_per_s'] / out['cpu_baseline']['value']) if out.get('cpu_baseline') else None}
            out['ctc'] = ctc_leg(dev)
            out['eval'] = eval_leg(dev)
            out['pipeline'] = pipeline_leg(dev)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if world > 1 and os.environ.get('SS_BENCH_N1', '1') != '0':
            out['n1_comparison'] = n1_comparison(args, local_rank)
            if out['n1_comparison'].get('value'):
                out['n1_comparison']['speedup_of_this_line'] = out['value'] / out['n1_comparison']['value']
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
