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
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3


def cpu_baseline(batch_cpu, rows_limit, steps):
    """The oracle (CPU restatement of the reference step, torch fp32 + compiled C DTW) timed on this node's host
    cores on a bounded sample of the same workload: the first utterances of the batch up to `rows_limit` rows."""
    import subprocess
    subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'oracle')])
    from oracle import loss_ref, model_ref
    # 32 threads is the fastest setting on the GPU node's 2x64-core host for this small-batch fp32 step (probed:
    # 16 thr 0.41 s, 32 thr 0.34 s, 64 thr 0.76 s, 128 thr 1.70 s, 256 thr >100 s per fwd+bwd of 762 frames)
    torch.set_num_threads(min(32, os.cpu_count()))
    n, frames = 0, 0
    while n < len(batch_cpu['lengths']) and (frames + batch_cpu['lengths'][n] + 199) // 200 <= rows_limit:
        frames += batch_cpu['lengths'][n]
        n += 1
    n = max(n, 1)
    sub = {k: v[:n] for k, v in batch_cpu.items()}
    frames = sum(sub['lengths'])
    sd = model_ref.init_state_dict(768, 6, 80, 48, seed=0)
    params = [v.requires_grad_(True) for k, v in sd.items() if v.dtype == torch.float32 and 'running' not in k and 'relative_positional' not in k]
    opt = torch.optim.AdamW(params, lr=1e-5, weight_decay=1e-7)
    g = torch.Generator().manual_seed(0)

    def step():
        opt.zero_grad()
        x_raw = loss_ref.combine_fixed_length(sub['raw_emg'], 1600)
        B, T = x_raw.shape[0], 200
        masks = []
        for _ in range(6):
            masks.append({'attn': (torch.rand(B, 8, T, T, generator=g) >= 0.2).float(), 'res1': (torch.rand(B, T, 768, generator=g) >= 0.2).float(),
                          'ffn': (torch.rand(B, T, 3072, generator=g) >= 0.2).float(), 'res2': (torch.rand(B, T, 768, generator=g) >= 0.2).float()})
        pred, aux = model_ref.model_forward(sd, x_raw, training=True, shift_r=3, running_out={}, layer_masks=masks, dropout_p=0.2)
        loss, _ = loss_ref.dtw_loss_ref(pred, aux, sub)
        loss.backward()
        opt.step()

    step()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = (time.perf_counter() - t0) / steps
    return {'value': frames / dt, 'unit': 'frames/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': '%d utterances = %d frames (%d packed rows of 200) of the same synthetic batch, full 768-d/6-layer fp32 step '
                      '(fwd+dtw_loss+bwd+AdamW), %d timed steps after 1 warm-up, %.2f s/step' % (n, frames, (frames + 199) // 200, steps, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--cpu-rows', type=int, default=6, help='packed rows of the batch given to the CPU baseline (0 = skip)')
    ap.add_argument('--cpu-steps', type=int, default=2)
    ap.add_argument('--no-profile', action='store_true', help='do not bracket GEMM launches with HIP events')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world and world > 1:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit('for --gpus N>1 launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py --gpus N ...')
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

    from silent_speech_amd import engine, ops
    from silent_speech_amd.architecture import Model
    from silent_speech_amd.data_utils import combine_fixed_length
    from silent_speech_amd.distributed import DataParallel
    from silent_speech_amd.optim import FusedAdamW
    from silent_speech_amd.synthetic import reference_size_batch
    from silent_speech_amd.transduction_model import dtw_loss

    dt = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    torch.manual_seed(0)
    model = Model(112, 80, 48, model_size=768, num_layers=6, dropout=0.2, compute_dtype=dt).to(dev)
    model.train()
    dp = DataParallel() if world > 1 else None
    if dp is not None:
        dp.attach(model)
    optim = FusedAdamW(model, weight_decay=1e-7)
    batch_cpu = reference_size_batch(seed=rank)
    batch = {k: ([t.to(dev) for t in v] if isinstance(v, list) and len(v) and torch.is_tensor(v[0]) else v) for k, v in batch_cpu.items()}
    frames = sum(batch['lengths'])
    rows = (frames + 199) // 200
    it = [0]

    def step():
        optim.zero_grad()
        i = it[0] + 1
        if i <= 500:
            for gp in optim.param_groups:
                gp['lr'] = i * 1e-3 / 500                                       # transduction_model.py:186-189
        X = combine_fixed_length(batch['emg'], 200)
        X_raw = combine_fixed_length(batch['raw_emg'], 1600)
        sess = combine_fixed_length(batch['session_ids'], 200)
        if dp is not None:
            dp.begin_step(X_raw.shape[0] * 200)
        pred, aux = model(X, X_raw, sess)
        total = dp.global_total(batch) if dp is not None else None
        loss, _ = dtw_loss(pred, aux, batch, phoneme_loss_weight=0.5, total_length=total)
        loss.backward()
        if dp is not None:
            dp.sync_gradients(model)
        optim.step()
        it[0] += 1
        return loss

    for _ in range(args.warmup):
        loss = step()
    torch.cuda.synchronize()
    if not torch.isfinite(loss.detach()).item():
        raise SystemExit('non-finite loss after warm-up')
    prof = None
    if not args.no_profile:
        prof = ops.GemmProfiler()
        ops.PROFILER = prof
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if prof is not None:
            # every 5th timed step carries the per-launch HIP events (roofline numerator); on those steps the weight-gradient
            # GEMMs stay on the main stream: a duration taken while a second stream shares the CUs is not a per-kernel quantity
            prof.enabled = i % 5 == 0
            engine.SIDE_STREAM_ENABLED = not prof.enabled
        loss = step()
    engine.SIDE_STREAM_ENABLED = True
    host_enqueue = time.perf_counter() - t0               # host time to ENQUEUE the steps (the GPU runs behind, asynchronously)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ops.PROFILER = None
    final_loss = float(loss.detach())

    stats = torch.tensor([elapsed, float(frames)], dtype=torch.float64, device=dev)
    if world > 1:
        mx = stats.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed, total_frames = float(mx[0]), float(sm[1])
    else:
        total_frames = float(frames)

    if rank == 0:
        out = {
            'metric': 'EMG frames/s training (transduction_model.py)', 'value': total_frames * args.steps / elapsed, 'unit': 'frames/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': 'configs[1]: full transduction model (768-d, 6-layer rel-pos encoder, 3 ResBlocks) training step '
                                   '(pack+fwd+dtw_loss incl. on-device DTW+bwd+AdamW), dropout 0.2, synthetic 8-ch EMG',
                       'frames_per_gpu_step': frames, 'rows_per_gpu_step': rows, 'utterances_per_gpu_step': len(batch['lengths']),
                       'silent_utterances': int(sum(batch['silent'])), 'parallelism': 'dp%d' % world, 'final_loss': final_loss,
                       'host_enqueue_ms_per_step': host_enqueue / args.steps * 1e3},
        }
        if prof is not None:
            summ = prof.summary()
            psteps = len(range(0, args.steps, 5))                       # steps that carried the per-launch events
            key = max(summ, key=lambda k: summ[k]['seconds'])
            d = summ[key]
            peak = PEAK_BF16_TFLOPS if 'bfloat16' in key[0] else PEAK_F32_TFLOPS
            ach = d['flops'] / d['seconds'] / 1e12
            traffic, traffic_src = None, None
            try:        # HBM bytes per launch from the committed rocprofv3 PMC passes of this same command (tools/pmc_summary.py)
                pmc = json.load(open(os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')))
                ctype = {'torch.bfloat16': 'unsigned short', 'torch.float32': 'float'}
                names = {1: ['gemm_glds_kernel<%s, %s>' % (ctype[key[0]], ctype[key[1]])], 2: ['gemm_w2_kernel<%s, 144>' % ctype[key[1]], 'gemm_w2_kernel<%s, 128>' % ctype[key[1]]]}.get(key[4], [])
                names.append('gemm_kernel<%s, %s, %d, %d>' % (ctype[key[0]], ctype[key[1]], key[2], key[3]))
                for nm in names:
                    hit = [v for k, v in pmc.items() if nm in k]
                    if hit:
                        traffic, traffic_src = hit[0]['hbm_bytes_per_launch'], 'profiles/r01_pmc_traffic.json: ' + nm
                        break
            except Exception:
                pass
            out['roofline'] = {'bound': 'mfma', 'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s', 'frac': ach / peak, 'traffic': traffic,
                               'traffic_source': traffic_src,
                               'kernel': '%s<%s,%s,a_mode=%d,b_mode=%d>' % ((('gemm_kernel', 'gemm_glds_kernel', 'gemm_w2_kernel')[key[4]],) + tuple(key[:4])), 'launches_per_step': d['launches'] / psteps, 'event_timed_steps': psteps,
                               'timing': 'HIP events around every ss_gemm launch on every 5th timed step; those steps keep the dW GEMMs on the main stream (exclusive durations); rocprofv3 counterpart: profiles/*_serial_kernel_stats.txt (SS_AMD_SIDE_STREAM=0)',
                               'avg_launch_us': d['seconds'] / d['launches'] * 1e6, 'algorithmic_gflop_per_launch': d['flops'] / d['launches'] / 1e9,
                               'all_gemm_variants': {str(k): {'tflops': v['flops'] / v['seconds'] / 1e12, 'ms_per_step': v['seconds'] / psteps * 1e3,
                                                              'launches_per_step': v['launches'] / psteps} for k, v in summ.items()}}
        if world == 1 and args.cpu_rows > 0:
            out['cpu_baseline'] = cpu_baseline(batch_cpu, args.cpu_rows, args.cpu_steps)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
