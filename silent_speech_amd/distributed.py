"""Data-parallel training across the 8 MI355X of a node: one process per GPU, torch.distributed with the
'nccl' backend (= RCCL over xGMI on ROCm).  The reference is single-process (transduction_model.py:246);
the contract here is "N-GPU step == 1-GPU step on the concatenated batch":

  * gradients : ONE all-reduce of the flat f32 gradient arena (53 M floats = 213 MB).  The 8 GPUs are fully
    connected by 7 xGMI links each, so a single large collective lets RCCL drive all links at once; there is
    no per-parameter bucket traffic.
  * BatchNorm : the reference's batch statistics span the whole batch (architecture.py:19,21,25), so the
    per-channel sums of every BatchNorm (forward: sum, sum-of-squares; backward: sum g, sum g*xhat) are
    all-reduced between the two phases of the HIP kernels (ss_bn_stats_sums/ss_bn_finalize, ss_bn_backward_*).
    18 tiny latency-bound collectives per step.
  * loss      : sum(losses)/sum(T2) uses the GLOBAL frame count (transduction_model.py:157); each rank
    scales by it, so the summed gradients equal the single-process ones.
  * relative-position embeddings never receive a gradient (transformer.py:214-218) and are not in the arena.
"""
import torch
import torch.distributed as dist


class DataParallel(object):
    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._rows_total = None
        # Host-known scalars (row / frame counts) travel over a gloo group: an all-reduce on the GPU stream would need an
        # .item() per step, i.e. a full device sync that lets the GPU run dry while the host re-fills the launch queue.
        self.host_group = None
        if self.world > 1 and dist.get_backend(group) != 'gloo':
            self.host_group = dist.new_group(backend='gloo')

    def attach(self, model):
        if self.world > 1:
            model._bn_reduce_fn = self._reduce_sums
            flat, _, n = model.flat_arenas()
            dist.broadcast(flat, 0, group=self.group)            # identical initial weights and BN buffers
            for b in model.buffers():
                dist.broadcast(b, 0, group=self.group)
            model.mark_weights_updated()
            model.set_seed(model._seed_base + 7919 * self.rank)  # independent dropout streams per rank
        return model

    # ---- BatchNorm statistic sums: sum over ranks; the row count scales by the (pre-agreed) global/local ratio
    def begin_step(self, local_rows_b_times_t):
        """All ranks exchange their packed-row counts once per step so BatchNorm can normalise by the global count."""
        if self.world == 1:
            self._ratio = 1.0
            return
        self._ratio = self._host_sum(float(local_rows_b_times_t)) / float(local_rows_b_times_t)

    def _host_sum(self, value):
        """Sum of a Python scalar over the ranks without touching the GPU stream."""
        t = torch.tensor([value], dtype=torch.float64)
        if self.host_group is not None:
            dist.all_reduce(t, group=self.host_group)
        elif dist.get_backend(self.group) == 'gloo':
            dist.all_reduce(t, group=self.group)
        else:                                       # no host group available: fall back to the device collective (+ sync)
            t = t.to(torch.device('cuda', torch.cuda.current_device()))
            dist.all_reduce(t, group=self.group)
        return float(t.item())

    def _reduce_sums(self, sums, n_local):
        dist.all_reduce(sums, group=self.group)
        return n_local * self._ratio

    def global_total(self, batch):
        """Global sum of target frames (the loss normaliser)."""
        local = float(sum(int(a.shape[0]) for a in batch['audio_features']))
        if self.world == 1:
            return local
        return self._host_sum(local)

    def sync_gradients(self, model):
        if self.world == 1:
            return
        _, gflat, _ = model.flat_arenas()
        dist.all_reduce(gflat, group=self.group)                 # losses are already divided by the GLOBAL frame count
