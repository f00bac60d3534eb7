"""Data-parallel training across the 8 MI355X of a node: one process per GPU, torch.distributed with the
'nccl' backend (= RCCL over xGMI on ROCm).  The reference is single-process (transduction_model.py:246);
the contract here is "N-GPU step == 1-GPU step on the concatenated batch":

  * gradients : the flat f32 gradient arena (53 M floats = 213 MB) is all-reduced in FOUR buckets, in the order backward
    completes them (encoder + heads + w_raw_in = 176 MB first, then ResBlocks 2, 1, 0): the native plan raises a "gradients
    ready" event per bucket on its side stream and the collective is enqueued there, under the remaining backward kernels.
    The 8 GPUs are fully connected by 7 xGMI links each, so few large collectives let RCCL drive all links at once; there
    is no per-parameter bucket traffic.
  * BatchNorm : the reference's batch statistics span the whole batch (architecture.py:19,21,25), so the
    per-channel sums of every BatchNorm (forward: sum, sum-of-squares; backward: sum g, sum g*xhat) are
    all-reduced between the two phases of the HIP kernels (ss_bn_stats_sums/ss_bn_finalize, ss_bn_backward_*).
    12 small latency-bound collectives per step (bn1 + res_norm of a block share one in the forward, bn2 + res_norm in the
    backward), on a process group of their own: the gradient buckets use a second group (= a second RCCL communicator and stream),
    so a 176 MB bucket in flight never sits in front of a 12 KB BatchNorm exchange the main stream is waiting for.
  * loss      : sum(losses)/sum(T2) uses the GLOBAL frame count (transduction_model.py:157); each rank
    scales by it, so the summed gradients equal the single-process ones.
  * relative-position embeddings never receive a gradient (transformer.py:214-218) and are not in the arena.
"""
import random

import torch
import torch.distributed as dist


class DataParallel(object):
    """One instance per process.  Per step: begin_step(rows, frames) (ONE host collective for the two Python-side counts),
    forward / backward (BatchNorm sums all-reduced from inside the native plan; gradient buckets all-reduced from the plan's
    "gradients ready" events while backward is still running), sync_gradients() (waits for the buckets)."""

    def __init__(self, group=None, bucketed=True):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.bucketed = bucketed
        self._ratio, self._frames_total = 1.0, None
        self._works, self._covered, self._buckets, self._model = [], [], {}, None
        # Host-known scalars (row / frame counts) travel over a gloo group: an all-reduce on the GPU stream would need an
        # .item() per step, i.e. a full device sync that lets the GPU run dry while the host re-fills the launch queue.
        self.host_group = None
        if self.world > 1 and dist.get_backend(group) != 'gloo':
            try:
                self.host_group = dist.new_group(backend='gloo')
            except Exception as e:      # noqa: BLE001 -- no silent fallback to a per-step device sync
                raise RuntimeError('DataParallel: could not create the gloo side group for the host-side counts (%s); '
                                   'set MASTER_ADDR=127.0.0.1 / check that gloo can bind a local interface' % e)
        # gradient buckets on their own communicator: with one group the async 176 MB bucket and the blocking BatchNorm exchanges
        # share one collective stream, and the main stream's conv backward would stall behind the bucket
        self.bucket_group = group
        if self.world > 1 and bucketed:
            self.bucket_group = dist.new_group(ranks=None if group is None else dist.get_process_group_ranks(group), backend=dist.get_backend(group))

    def attach(self, model, shift_seed=0x5EED):
        self._model = model
        if self.world > 1:
            model._bn_reduce_fn = self._reduce_sums
            flat, _, n = model.flat_arenas()
            dist.broadcast(flat, 0, group=self.group)            # identical initial weights ...
            arena_lo, arena_hi = flat.data_ptr(), flat.data_ptr() + 4 * n
            for t in model.state_dict().values():               # ... BN buffers and the (never trained, randomly initialised)
                if not (arena_lo <= t.data_ptr() < arena_hi):   # relative-position embeddings, which live outside the arena
                    dist.broadcast(t, 0, group=self.group)
            model.mark_weights_updated()
            model.set_seed(model._seed_base + 7919 * self.rank)  # independent dropout streams per rank
            model.shift_rng = random.Random(shift_seed)          # the SAME shift r on every rank (N-GPU step == 1-GPU step on the concatenated batch)
            # gradient buckets in the order backward completes them: what = 0 encoder + heads + w_raw_in, 1..3 ResBlocks 2, 1, 0
            ranges = model.arena_ranges()

            def span(pred):       # up to the padded slot boundary (slots are 4-float aligned): the spans tile the arena without gaps
                sel = [(a, (b + 3) // 4 * 4) for nme, a, b in ranges if pred(nme)]
                return (min(a for a, _ in sel), max(b for _, b in sel)) if sel else None
            self._buckets = {0: span(lambda nme: not nme.startswith('conv_blocks.'))}
            for i in range(3):
                self._buckets[1 + (2 - i)] = span(lambda nme, i=i: nme.startswith('conv_blocks.%d.' % i))
            model._grad_ready_fn = self._on_grads_ready if self.bucketed else None
        return model

    # ---- the two host-side counts of a step in ONE collective
    def begin_step(self, local_rows_b_times_t, local_target_frames=None):
        """All ranks exchange (packed-row count x frames per row, target-frame count): BatchNorm normalises by the global row count,
        the loss by the global number of target frames (transduction_model.py:157)."""
        self._works, self._covered = [], []
        if self.world == 1:
            self._ratio = 1.0
            self._frames_total = float(local_target_frames) if local_target_frames is not None else None
            return
        t = torch.tensor([float(local_rows_b_times_t), float(local_target_frames or 0.0)], dtype=torch.float64)
        t = self._host_sum(t)
        self._ratio = float(t[0]) / float(local_rows_b_times_t)
        self._frames_total = float(t[1]) if local_target_frames is not None else None

    def _host_sum(self, t):
        """Sum of a small float64 vector over the ranks without touching the GPU stream."""
        if self.host_group is not None:
            dist.all_reduce(t, group=self.host_group)
        elif dist.get_backend(self.group) == 'gloo':
            dist.all_reduce(t, group=self.group)
        else:                                       # no host group available: fall back to the device collective (+ sync)
            t = t.to(torch.device('cuda', torch.cuda.current_device()))
            dist.all_reduce(t, group=self.group)
            t = t.cpu()
        return t

    def _reduce_sums(self, sums, n_local):
        dist.all_reduce(sums, group=self.group)
        return n_local * self._ratio

    @staticmethod
    def local_target_frames(batch):
        return float(sum(int(a.shape[0]) for a in batch['audio_features']))

    def global_total(self, batch):
        """Global sum of target frames (the loss normaliser); free when begin_step was given the local count."""
        if self._frames_total is not None:
            return self._frames_total
        local = self.local_target_frames(batch)
        if self.world == 1:
            return local
        return float(self._host_sum(torch.tensor([local], dtype=torch.float64))[0])

    # ---- gradients
    def _on_grads_ready(self, what, stream=0):
        """Called from inside ss_plan_backward when bucket `what` is final on `stream` (the raw hipStream_t the plan produced the
        gradients on: its side stream, or the main stream when the side stream is off): start the all-reduce behind exactly that
        stream, so it overlaps the rest of backward (xGMI is otherwise idle until the end of the step)."""
        rng = self._buckets.get(what)
        if rng is None or self.world == 1:
            return
        model = self._model
        _, gflat, _ = model.flat_arenas()
        a, b = rng
        if gflat.is_cuda and stream and stream != torch.cuda.current_stream(gflat.device).cuda_stream:
            with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=gflat.device)):
                w = dist.all_reduce(gflat[a:b], group=self.bucket_group, async_op=True)
        else:
            w = dist.all_reduce(gflat[a:b], group=self.bucket_group, async_op=True)
        self._works.append(w)
        self._covered.append((a, b))

    def sync_gradients(self, model):
        if self.world == 1:
            return
        _, gflat, n = model.flat_arenas()
        for w in self._works:
            w.wait()                                               # the current stream waits for the collective
        # whatever the events did not cover (bucketing off, a model variant without the hook): one more collective
        todo, pos = [], 0
        for a, b in sorted(self._covered):
            if a > pos:
                todo.append((pos, a))
            pos = max(pos, b)
        if pos < n:
            todo.append((pos, n))
        for a, b in todo:
            dist.all_reduce(gflat[a:b], group=self.group)           # losses are already divided by the GLOBAL frame count
        self._works, self._covered = [], []

    def broadcast_scalars(self, *values):
        """Rank 0's values on every rank (validation loss / accuracy before the LR scheduler: kernels with f32 atomics can differ
        in the last bits between ranks, and a plateau decision taken on one rank only would let the weights diverge)."""
        if self.world == 1:
            return values
        t = torch.tensor([float(v) if self.rank == 0 else 0.0 for v in values], dtype=torch.float64)
        return tuple(float(v) for v in self._host_sum(t))
