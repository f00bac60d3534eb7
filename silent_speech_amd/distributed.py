"""Data-parallel training across the 8 MI355X of a node: one process per GPU, torch.distributed with the
'nccl' backend (= RCCL over xGMI on ROCm).  The reference is single-process (transduction_model.py:246);
the contract here is "N-GPU step == 1-GPU step on the concatenated batch":

  * gradients : the flat f32 gradient arena (53 M floats = 213 MB) is all-reduced in TEN buckets, in the order backward completes
    them: encoder layers 5 .. 0 (28 MB each, fired layer by layer while the backward of the earlier layers still runs), heads +
    w_raw_in, then ResBlocks 2, 1, 0.  The native plan raises a "gradients ready" event per bucket on its side stream and the
    collective is enqueued there, under the remaining backward kernels.  The 8 GPUs are fully connected by 7 xGMI links each, so a
    collective of tens of MB drives all links at once; there is no per-parameter bucket traffic.  SS_DP_LAYER_BUCKETS=0 restores the
    single 176 MB encoder bucket of rounds 2-3 (fired after the whole encoder backward); `grad_dtype=torch.bfloat16` halves the bytes
    on the links (the bucket is cast to bf16, all-reduced in bf16 and copied back into the f32 arena: a ring / tree all-reduce rounds at EVERY hop,
    so the error grows with the world size -- 1.3e-3 of max|g| measured at 2 gloo ranks, unmeasured on RCCL at 8; off by default).
  * BatchNorm : the reference's batch statistics span the whole batch (architecture.py:19,21,25), so the
    per-channel sums of every BatchNorm (forward: sum, sum-of-squares; backward: sum g, sum g*xhat) are
    all-reduced between the two phases of the HIP kernels (ss_bn_stats_sums/ss_bn_finalize, ss_bn_backward_*).
    12 small latency-bound collectives per step (bn1 + res_norm of a block share one in the forward, bn2 + res_norm in the
    backward), on a process group of their own: the gradient buckets use a second group (= a second RCCL communicator and stream),
    so a bucket in flight never sits in front of a 12 KB BatchNorm exchange the main stream is waiting for.  Two communicators
    with collectives in flight at once rely on every rank enqueueing them in the same order per communicator (they do: both
    orders are fixed by the plan); SS_DP_SINGLE_GROUP=1 puts everything on one communicator should a RCCL build object.
  * loss      : sum(losses)/sum(T2) uses the GLOBAL frame count (transduction_model.py:157); each rank
    scales by it, so the summed gradients equal the single-process ones.
  * relative-position embeddings never receive a gradient (transformer.py:214-218) and are not in the arena.
"""
import os
import random

import torch
import torch.distributed as dist


class DataParallel(object):
    """One instance per process.  Per step: begin_step(rows, frames) (ONE host collective for the two Python-side counts),
    forward / backward (BatchNorm sums all-reduced from inside the native plan; gradient buckets all-reduced from the plan's
    "gradients ready" events while backward is still running), sync_gradients() (waits for the buckets)."""

    def __init__(self, group=None, bucketed=True, grad_dtype=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.bucketed = bucketed
        if grad_dtype not in (None, torch.float32, torch.bfloat16):
            raise ValueError('grad_dtype: None / torch.float32 (exact) or torch.bfloat16 (half the bytes on the links)')
        self.grad_dtype = None if grad_dtype == torch.float32 else grad_dtype
        self.layer_buckets = os.environ.get('SS_DP_LAYER_BUCKETS', '1') != '0'
        self._ratio, self._frames_total = 1.0, None
        self._works, self._covered, self._buckets, self._model = [], [], {}, None
        self._pending = None                        # (local counts, async work, tensor) of the NEXT step's host-side exchange
        # Host-known scalars (row / frame counts) travel over a gloo group: an all-reduce on the GPU stream would need an
        # .item() per step, i.e. a full device sync that lets the GPU run dry while the host re-fills the launch queue.
        self.host_group = None
        backend = self._backend_of(group) if self.world > 1 else None
        if self.world > 1 and backend != 'gloo':
            try:
                self.host_group = dist.new_group(backend='gloo')
            except Exception as e:      # noqa: BLE001 -- no silent fallback to a per-step device sync
                raise RuntimeError('DataParallel: could not create the gloo side group for the host-side counts (%s); '
                                   'set MASTER_ADDR=127.0.0.1 / check that gloo can bind a local interface' % e)
        # gradient buckets on their own communicator: with one group the async buckets and the blocking BatchNorm exchanges
        # share one collective stream, and the main stream's conv backward would stall behind a bucket
        self.bucket_group = group
        self.schedule = 'single_communicator'
        if self.world > 1 and bucketed and os.environ.get('SS_DP_SINGLE_GROUP', '0') != '1':
            self.bucket_group = dist.new_group(ranks=None if group is None else dist.get_process_group_ranks(group), backend=backend)
            self.schedule = 'two_communicators'
        # bench.py: time the main stream spends blocked in sync_gradients (events on both sides of the waits) = the part of the
        # gradient all-reduce that the backward did not hide
        self.measure_exposed = False
        self._exposed = []

    @staticmethod
    def _backend_of(group):
        """'nccl' (= RCCL) or 'gloo' of the group the tensors of this process travel on.  torch reports a composite string such as
        'cpu:gloo,cuda:nccl' for a default group created without an explicit backend: new groups are created with the plain name."""
        b = str(dist.get_backend(group))
        if 'nccl' in b and torch.cuda.is_available():
            return 'nccl'
        return 'gloo' if 'gloo' in b else b

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
            # gradient buckets in the order backward completes them (the plan's event codes): 4 + l = encoder layer l (last layer first),
            # 0 = heads + w_raw_in, 1..3 = ResBlocks 2, 1, 0.  A bucket is a list of arena spans, each padded up to the 4-float slot boundary.
            ranges = model.arena_ranges()

            def span(pred):
                sel = [(a, (b + 3) // 4 * 4) for nme, a, b in ranges if pred(nme)]
                return (min(a for a, _ in sel), max(b for _, b in sel)) if sel else None
            n_layers = len(model.transformer.layers)
            self._buckets = {}
            if self.layer_buckets:
                for l in range(n_layers):
                    self._buckets[4 + l] = [span(lambda nme, l=l: nme.startswith('transformer.layers.%d.' % l))]
                self._buckets[0] = [span(lambda nme: nme.startswith('w_raw_in.')), span(lambda nme: nme.startswith('w_out.') or nme.startswith('w_aux.'))]
            else:
                self._buckets[0] = [span(lambda nme: not nme.startswith('conv_blocks.'))]
            for i in range(3):
                self._buckets[1 + (2 - i)] = [span(lambda nme, i=i: nme.startswith('conv_blocks.%d.' % i))]
            self._buckets = {k: [x for x in v if x is not None] for k, v in self._buckets.items()}
            model._grad_ready_fn = self._on_grads_ready if self.bucketed else None
        return model

    # ---- the two host-side counts of a step in ONE collective
    def begin_step(self, local_rows_b_times_t, local_target_frames=None, next_counts=None):
        """All ranks exchange (packed-row count x frames per row, target-frame count): BatchNorm normalises by the global row count,
        the loss by the global number of target frames (transduction_model.py:157).  next_counts = the same pair for the NEXT batch
        (a training loop that looks one batch ahead): its exchange is started now, asynchronously on the host-side group, and is
        simply picked up by the next begin_step -- the per-step collective then never sits on the critical path."""
        self._works, self._covered = [], []
        local = (float(local_rows_b_times_t), float(local_target_frames or 0.0))
        if self.world == 1:
            self._ratio = 1.0
            self._frames_total = float(local_target_frames) if local_target_frames is not None else None
            return
        pend, self._pending = self._pending, None
        if pend is not None:
            # A prefetched exchange is ALWAYS consumed (every rank started it, so every rank waits for it) and never replaced by a
            # rank-local extra collective: whether the announced counts match is only known locally, and a rank that re-exchanged on its
            # own would issue a collective nobody joins while the others silently kept a sum containing its stale announcement.
            if pend[1] is not None:
                pend[1].wait()
            # whether the announcement matched is known per rank only: every rank learns it (one more tiny host-side collective, only on the
            # prefetch path) and ALL of them raise -- a rank that raised alone would leave the others hanging in the first BatchNorm exchange
            bad = self._host_sum(torch.tensor([1.0 if pend[0] != local else 0.0], dtype=torch.float64))
            if float(bad[0]) > 0.0 and pend[0] == local:
                raise RuntimeError('DataParallel.begin_step: %d rank(s) announced next_counts that do not match the counts of this step; the step is '
                                   'aborted on every rank (a mismatch is fatal for the whole job: the prefetched sums contain the stale announcement)' % int(bad[0]))
            if pend[0] != local:
                raise RuntimeError('DataParallel.begin_step: rank %d announced next_counts=%r for this step but was called with %r; next_counts '
                                   'must be exactly the (rows x frames, target frames) pair of the next begin_step on every rank (pass '
                                   'next_counts=None when the next batch is not known)' % (self.rank, pend[0], local))
            t = pend[2]
        else:
            t = self._host_sum(torch.tensor(local, dtype=torch.float64))
        self._ratio = float(t[0]) / float(local_rows_b_times_t)
        self._frames_total = float(t[1]) if local_target_frames is not None else None
        if next_counts is not None:
            nl = (float(next_counts[0]), float(next_counts[1] or 0.0))
            nt = torch.tensor(nl, dtype=torch.float64)
            if self.host_group is not None or self._backend_of(self.group) == 'gloo':      # a host-side group exists: the exchange costs no device sync
                grp = self.host_group if self.host_group is not None else self.group
                self._pending = (nl, dist.all_reduce(nt, group=grp, async_op=True), nt)

    def _host_sum(self, t):
        """Sum of a small float64 vector over the ranks without touching the GPU stream."""
        if self.host_group is not None:
            dist.all_reduce(t, group=self.host_group)
        elif self._backend_of(self.group) == 'gloo':
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
        spans = self._buckets.get(what)
        if not spans or self.world == 1:
            return
        model = self._model
        _, gflat, _ = model.flat_arenas()
        ext = gflat.is_cuda and stream and stream != torch.cuda.current_stream(gflat.device).cuda_stream

        def launch():
            for a, b in spans:
                if self.grad_dtype is None:
                    self._works.append((dist.all_reduce(gflat[a:b], group=self.bucket_group, async_op=True), None))
                else:                                   # half the bytes on the links: cast, all-reduce, add back in f32 after the wait
                    from . import ops
                    tmp = torch.empty(b - a, dtype=self.grad_dtype, device=gflat.device)
                    ops.cast_f32(gflat[a:b], tmp, b - a)
                    self._works.append((dist.all_reduce(tmp, group=self.bucket_group, async_op=True), (a, b, tmp)))
                self._covered.append((a, b))
        if ext:
            with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=gflat.device)):
                launch()
        else:
            launch()

    def sync_gradients(self, model):
        if self.world == 1:
            return
        _, gflat, n = model.flat_arenas()
        ev = None
        if self.measure_exposed and gflat.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for w, back in self._works:
            w.wait()                                               # the current stream waits for the collective
            if back is not None:
                a, b, tmp = back
                tmp.record_stream(torch.cuda.current_stream(gflat.device)) if tmp.is_cuda else None      # allocated under the producing (external) stream, read here
                gflat[a:b].copy_(tmp)                              # bf16 transport: the rank-sum back into the f32 arena
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
        if ev is not None:
            ev[1].record()
            self._exposed.append(ev)
        self._works, self._covered = [], []

    def exposed_ms(self):
        """Mean device time per step the main stream sat in sync_gradients (call after a device synchronize); None if never measured."""
        ms = [a.elapsed_time(b) for a, b in self._exposed]
        self._exposed = []
        return sum(ms) / len(ms) if ms else None

    def bucket_bytes(self):
        """(bytes all-reduced per step and rank-buffer, number of gradient collectives per step) as attach() laid the buckets out."""
        if not self._buckets:
            return 0, 0
        el = 2 if self.grad_dtype is not None else 4
        spans = [x for v in self._buckets.values() for x in v]
        return sum((b - a) * el for a, b in spans), len(spans)

    def broadcast_scalars(self, *values):
        """Rank 0's values on every rank (validation loss / accuracy before the LR scheduler: kernels with f32 atomics can differ
        in the last bits between ranks, and a plateau decision taken on one rank only would let the weights diverge)."""
        if self.world == 1:
            return values
        t = torch.tensor([float(v) if self.rank == 0 else 0.0 for v in values], dtype=torch.float64)
        return tuple(float(v) for v in self._host_sum(t))
