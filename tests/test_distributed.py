"""Data-parallel contract, checked without GPUs: 2 gloo ranks (CPU tensors, host-emulator kernels), each with
half of the utterances, must reproduce the single-process step on the concatenated batch: same BatchNorm batch
statistics (all-reduced sums), same loss normaliser (global frame count), summed gradients equal."""
import os
import socket
import subprocess
import sys

import pytest
import torch

from tests.backend import _ensure_emu

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


_single_cache = {}


def _ranks_vs_single(tmp_path, extra_env, world=2, tol=2e-4, check_running=True):
    worker = os.path.join(ROOT, 'tests', 'dp_worker.py')
    env = dict(os.environ, OMP_NUM_THREADS='1', SS_DP_ORDER_OF=str(world), SS_DP_UTTS='4' if world == 2 else '8', **extra_env)
    # the single-process reference depends on the model and the utterance order only: computed once per (layers, world), next to the ranks
    key = (extra_env.get('SS_DP_LAYERS', '1'), world, extra_env.get('SS_DP_DEVICE', 'cpu'))
    single, p0 = _single_cache.get(key), None
    if single is None or not os.path.exists(single):
        import tempfile
        single = _single_cache[key] = os.path.join(tempfile.mkdtemp(prefix='ss_dp_single_'), 'single.pt')
        senv = {k: v for k, v in env.items() if k not in ('SS_DP_ZERO_LATE', 'SS_DP_GRAD_BF16', 'SS_DP_PREFETCH', 'SS_DP_LAYER_BUCKETS', 'SS_DP_BUCKETED')}
        p0 = subprocess.Popen([sys.executable, worker, single], env=dict(senv, WORLD_SIZE='1', RANK='0'))
    port = str(_free_port())
    multi = str(tmp_path / 'multi.pt')
    procs = [subprocess.Popen([sys.executable, worker, multi], env=dict(env, WORLD_SIZE=str(world), RANK=str(r), MASTER_PORT=port, MASTER_ADDR='127.0.0.1'))
             for r in range(world)]
    for p in ([p0] if p0 is not None else []) + procs:
        assert p.wait(timeout=900) == 0
    a, b = torch.load(single), torch.load(multi)
    # loss: each rank reports sum(local losses)/global frames; the single-process loss is the sum over ranks
    assert abs(b['loss']) < abs(a['loss'])
    ga, gb = a['grads'], b['grads']
    assert ga.shape == gb.shape
    assert float(ga.abs().max()) > 1e-3 and float(gb.abs().max()) > 1e-3, 'vacuous comparison: the gradient arena is empty'
    err = float((ga - gb).abs().max()) / (float(ga.abs().max()) + 1e-12)
    assert err < tol, err
    if check_running:                        # (a worker that runs the step twice has updated the running statistics twice)
        assert torch.allclose(a['rm'], b['rm'], rtol=1e-4, atol=1e-6)
        assert torch.allclose(a['rv'], b['rv'], rtol=1e-4, atol=1e-6)
    assert torch.equal(a['emb'], b['emb'])      # the never-trained relative-position embeddings were broadcast from rank 0 too
    return err


def _two_rank_vs_single(tmp_path, extra_env):
    _ranks_vs_single(tmp_path, extra_env, 2)


def test_two_rank_step_equals_single_process(tmp_path):
    _ensure_emu()
    _two_rank_vs_single(tmp_path, {})


def test_zero_grad_between_forward_and_backward_keeps_the_arena(tmp_path):
    """model.zero_grad(set_to_none=True) after the forward pass used to leave the fused optimiser / the all-reduce with a stale,
    all-zero arena: gradients are re-homed instead.  Run on a 2-layer encoder, so that a layer bucket is fired from INSIDE the backward
    (event 4 + 1 while layer 0 is still to come)."""
    _ensure_emu()
    _two_rank_vs_single(tmp_path, {'SS_DP_ZERO_LATE': '1', 'SS_DP_LAYERS': '2'})


def test_single_encoder_bucket_bf16_transport_and_count_prefetch(tmp_path):
    """SS_DP_LAYER_BUCKETS=0 is the one-bucket schedule of rounds 2-3; grad_dtype=bfloat16 halves the bytes on the links: the rank-sums are
    rounded once to bf16 (measured delta recorded below, bound 2^-7 of the largest gradient); next_counts starts the next step's host-side
    exchange during the current step."""
    _ensure_emu()
    err = _ranks_vs_single(tmp_path, {'SS_DP_LAYER_BUCKETS': '0', 'SS_DP_GRAD_BF16': '1', 'SS_DP_PREFETCH': '1'}, 2, tol=8e-3, check_running=False)
    assert err > 1e-6, 'the bf16 transport was not exercised'
    with open(str(tmp_path / 'bf16_transport_delta.txt'), 'w') as f:
        f.write('max |g_bf16 - g_f32| / max |g| = %.3e\n' % err)


def test_wrong_next_counts_raise_on_every_rank(tmp_path):
    """The prefetched count exchange is always consumed; a rank whose announced next_counts do not match what it is called with raises
    instead of issuing a rank-local extra collective (round-4 advisor finding: that collective had no partner and the job hung) -- and so does
    EVERY other rank (round-5 advisor finding: they would otherwise walk into the first BatchNorm exchange and hang there): the mismatch flag
    travels over the host-side group before anyone raises."""
    _ensure_emu()
    worker = os.path.join(ROOT, 'tests', 'dp_worker.py')
    port = str(_free_port())
    out = str(tmp_path / 'bad')
    env = dict(os.environ, OMP_NUM_THREADS='1', SS_DP_BAD_PREFETCH='1', SS_DP_DEVICE='none')
    procs = [subprocess.Popen([sys.executable, worker, out], env=dict(env, WORLD_SIZE='2', RANK=str(r), MASTER_PORT=port, MASTER_ADDR='127.0.0.1')) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    assert open(out + '.rank0').read().startswith('raised: DataParallel.begin_step: 1 rank(s) announced')
    assert open(out + '.rank1').read().startswith('raised: DataParallel.begin_step: rank 1 announced')


def test_four_rank_step_equals_single_process_unbucketed(tmp_path):
    _ensure_emu()
    _ranks_vs_single(tmp_path, {'SS_DP_BUCKETED': '0'}, 4)


def test_train_model_data_parallel_two_ranks(tmp_path):
    """train_model(..., data_parallel=DataParallel()) end to end on 2 ranks: it used to die in the first BatchNorm reduction (no
    begin_step) and every rank saw the same batches; now the ranks stay bit-identical and only rank 0 writes the checkpoint."""
    _ensure_emu()
    worker = os.path.join(ROOT, 'tests', 'dp_worker.py')
    port = str(_free_port())
    out = str(tmp_path / 'tm.pt')
    env = dict(os.environ, OMP_NUM_THREADS='1', SS_DP_TRAIN_MODEL='1')
    procs = [subprocess.Popen([sys.executable, worker, out], env=dict(env, WORLD_SIZE='2', RANK=str(r), MASTER_PORT=port, MASTER_ADDR='127.0.0.1')) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=900) == 0
    a, b = torch.load(out + '.rank0'), torch.load(out + '.rank1')
    assert torch.equal(a['flat'], b['flat']) and torch.isfinite(a['flat']).all()
    assert a['saved'] and not b['saved']


@pytest.mark.gpu
def test_two_rank_step_equals_single_process_on_the_gpu(tmp_path):
    """The same contract through the HIP kernels: two gloo ranks share the one MI355X of the test box (RCCL itself needs one
    GPU per rank and cannot be exercised here; the collectives are torch.distributed calls either way)."""
    _two_rank_vs_single(tmp_path, {'SS_DP_DEVICE': 'cuda'})
