"""Worker for tests/test_distributed.py: one data-parallel rank (gloo, CPU tensors, host-emulator kernels)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)


class _R(object):
    @staticmethod
    def randrange(n):
        return 2


def make_batch(seed_list):
    from silent_speech_amd.synthetic import make_utterance, SyntheticEMGDataset
    items = []
    for sd, T, silent in seed_list:
        it = make_utterance(np.random.default_rng(sd), T, silent)
        items.append(it)
    return SyntheticEMGDataset.collate_raw(items)


UTTS = [(1, 24, False), (2, 48, True), (3, 48, False), (4, 24, True), (5, 48, False), (6, 24, True), (7, 24, False), (8, 48, True)]
UTTS = UTTS[:int(os.environ.get('SS_DP_UTTS', '8'))]      # 2-rank tests use the first four (voiced + silent on both ranks): half the emulator time
# whole rows of 24 frames -> per-rank rows == global rows; 8 utterances deal evenly to 1, 2 and 4 ranks


def run_step(model, batch, dp, seq_len=24, next_counts=None):
    from silent_speech_amd.data_utils import combine_fixed_length
    from silent_speech_amd.transduction_model import dtw_loss
    X_raw = combine_fixed_length(batch['raw_emg'], seq_len * 8)
    X = combine_fixed_length(batch['emg'], seq_len)
    sess = combine_fixed_length(batch['session_ids'], seq_len)
    zero_late = os.environ.get('SS_DP_ZERO_LATE') == '1'
    if not zero_late:
        model.zero_grad(set_to_none=True)
    if dp is not None:
        dp.begin_step(X_raw.shape[0] * seq_len, dp.local_target_frames(batch), next_counts=next_counts)
    pred, aux = model(X, X_raw, sess)
    total = dp.global_total(batch) if dp is not None else None
    loss, _ = dtw_loss(pred, aux, batch, phoneme_loss_weight=0.5, total_length=total)
    if zero_late:
        model.zero_grad(set_to_none=True)       # between forward and backward: .grad must be re-homed in the flat arena, not silently dropped
    loss.backward()
    if dp is not None:
        dp.sync_gradients(model)
    return loss


def build_model():
    from silent_speech_amd.architecture import Model
    torch.manual_seed(123)
    m = Model(112, 80, 48, model_size=16, num_layers=int(os.environ.get('SS_DP_LAYERS', '1')), dropout=0.0, compute_dtype=torch.float32)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.dim() == 1 and 'bias' not in n:
                p.add_(torch.randn(p.shape) * 0.1)
    m.shift_rng = _R
    m.train()
    return m


def train_model_mode(out, dev, world, rank):
    """Two steps of the drop-in train_model() with data_parallel=...: begin_step per batch, rank-sharded batches of one shared
    shuffle, checkpoint written by rank 0 only."""
    from silent_speech_amd import transduction_model as tm
    from silent_speech_amd.distributed import DataParallel
    from silent_speech_amd.flags import FLAGS
    from silent_speech_amd.synthetic import SyntheticEMGDataset
    FLAGS.model_size, FLAGS.num_layers, FLAGS.epochs, FLAGS.dropout = 16, 1, 1, 0.0
    FLAGS.output_directory = os.path.join(os.path.dirname(out), 'ckpt_rank%d' % rank)
    train = SyntheticEMGDataset(12, seed=1, min_frames=40, max_frames=80)
    devset = SyntheticEMGDataset(3, seed=2, min_frames=40, max_frames=60)
    for it in train.items:
        it['length_1k'] = 1000                                           # 4 utterances per batch of 4000 "samples"
    import silent_speech_amd.pipeline as pl
    orig = pl.SizeAwareSampler.__init__

    def small_budget(self, ds, max_len, **kw):
        orig(self, ds, 4000, **kw)
    pl.SizeAwareSampler.__init__ = small_budget
    dp = DataParallel() if world > 1 else None
    model = tm.train_model(train, devset, dev, save_sound_outputs=False, compute_dtype=torch.float32, max_steps=2, data_parallel=dp)
    flat, _, n = model.flat_arenas()
    torch.save({'flat': flat.clone().cpu(), 'saved': os.path.exists(os.path.join(FLAGS.output_directory, 'model.pt'))}, out + '.rank%d' % rank)


def main():
    out = sys.argv[1]
    from silent_speech_amd import _lib
    on_gpu = os.environ.get('SS_DP_DEVICE') == 'cuda'          # gpu tier: both ranks share the one MI355X, gloo moves the CUDA tensors
    if on_gpu:
        _lib.load()
    else:
        _lib.use_library_for_testing(os.path.join(ROOT, 'silent_speech_amd', 'lib', 'libsilent_speech_emu.so'))
    dev = torch.device('cuda', 0) if on_gpu else torch.device('cpu')
    from silent_speech_amd.distributed import DataParallel
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world > 1:
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%s' % os.environ['MASTER_PORT'], rank=rank, world_size=world)
    if os.environ.get('SS_DP_BAD_PREFETCH') == '1':
        # rank 1 announces counts for the next step that it then does not deliver: it must raise (and must NOT issue a collective of its own,
        # which nobody would join); rank 0, whose announcement was right, learns of the mismatch through the flag exchange and raises as well
        dp = DataParallel()
        dp.begin_step(48, 48.0, next_counts=(48 + (24 if rank == 1 else 0), 48.0))
        try:
            dp.begin_step(48, 48.0)
            verdict = 'ok'
        except RuntimeError as e:
            verdict = 'raised: ' + str(e)[:60]
        with open(out + '.rank%d' % rank, 'w') as f:
            f.write(verdict)
        return
    if os.environ.get('SS_DP_TRAIN_MODEL') == '1':
        train_model_mode(out, dev, world, rank)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    model = build_model().to(dev)
    dp = None
    if world > 1:
        torch.manual_seed(1000 + rank)               # ranks start from DIFFERENT weights / embeddings: attach() must make them equal to rank 0's
        with torch.no_grad():
            for prm in model.parameters():
                if rank > 0:
                    prm.add_(torch.randn(prm.shape, device=prm.device) * 0.01)
        dp = DataParallel(bucketed=os.environ.get('SS_DP_BUCKETED', '1') == '1', grad_dtype=torch.bfloat16 if os.environ.get('SS_DP_GRAD_BF16') == '1' else None)
        dp.attach(model)
        batch = make_batch(UTTS[rank::world])
    else:
        batch = make_batch([u for r in range(int(os.environ.get('SS_DP_ORDER_OF', '2'))) for u in UTTS[r::int(os.environ.get('SS_DP_ORDER_OF', '2'))]])
    batch = {k: ([t.to(dev) for t in v] if isinstance(v, list) and len(v) and torch.is_tensor(v[0]) else v) for k, v in batch.items()}
    if os.environ.get('SS_DP_PREFETCH') == '1' and dp is not None:
        # the step is run twice; the first begin_step already starts the exchange of the second step's counts (a loop that looks one batch ahead)
        model.shift_rng = _R                         # attach() installed the ranks' shared random shift: both steps must see the same augmentation here
        counts = (int(sum(batch['lengths'])), dp.local_target_frames(batch))
        run_step(model, batch, dp, next_counts=counts)
        assert dp._pending is not None
        model.zero_grad(set_to_none=True)
        loss = run_step(model, batch, dp)
        assert dp._pending is None
    else:
        loss = run_step(model, batch, dp)
    _, gflat, n = model.flat_arenas()
    emb = model.transformer.layers[0].self_attn.relative_positional.embeddings.detach().clone().cpu()
    res = {'loss': float(loss), 'grads': gflat.clone().cpu(), 'emb': emb, 'rm': model.conv_blocks[0].bn1.running_mean.clone().cpu(), 'rv': model.conv_blocks[2].bn2.running_var.clone().cpu()}
    if rank == 0:
        torch.save(res, out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
