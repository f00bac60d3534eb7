"""Offline EMG conditioning: oracle vs the reference's golden vectors, device kernels (host emulator on the
CPU tier, MI355X on the gpu tier) vs both.  Tolerances: the notch filters agree with scipy to f64 round-off; the 2 Hz third-order
drift filter has three poles at 0.99, scipy's own answer sits 1.4e-9 from an extended-precision evaluation on signals of this scale,
and the chunked scan lands within 1e-8 of scipy (measured 4e-9 at T = 20000) -- the bound below."""
import os

import numpy as np
import pytest
import torch

from oracle import filter_ref
from tests.backend import dev  # noqa: F401  (fixture: host emulator on the CPU tier, libsilent_speech_hip.so on the gpu tier)

GOLD = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'filters.npz'))
TOL_NOTCH = 1e-10
TOL_DRIFT = 1e-8


# ------------------------------------------------------------------ oracle pinned on the reference's outputs
@pytest.mark.parametrize('tag', ['short', 'mid', 'long'])
def test_oracle_matches_reference(tag):
    x = GOLD[tag + '/x']
    nh = filter_ref.notch_harmonics(x, 60, 1000)
    assert np.abs(nh - GOLD[tag + '/notch_harmonics']).max() < 1e-11
    chain = filter_ref.remove_drift(nh, 1000)
    assert np.abs(chain - GOLD[tag + '/chain']).max() < 1e-11
    assert np.array_equal(filter_ref.subsample(GOLD[tag + '/chain'], 689.06, 1000), GOLD[tag + '/emg_orig'])
    assert np.array_equal(filter_ref.subsample(GOLD[tag + '/chain'], 516.79, 1000), GOLD[tag + '/emg'])
    if tag != 'long':
        assert np.abs(filter_ref.notch(x, 60, 1000) - GOLD[tag + '/notch60']).max() < 1e-11
        assert np.abs(filter_ref.remove_drift(x, 1000) - GOLD[tag + '/remove_drift']).max() < 1e-11


def test_oracle_context_chain():
    a, b = filter_ref.condition(GOLD['mid/x'], GOLD['short/x'], GOLD['context/after'])
    assert a.shape == GOLD['context/emg_orig'].shape and b.shape == GOLD['context/emg'].shape
    assert np.abs(a - GOLD['context/emg_orig']).max() < 1e-11 and np.abs(b - GOLD['context/emg']).max() < 1e-11


def test_filter_design_closed_forms():
    import scipy.signal
    from silent_speech_amd import read_emg as R
    for h in range(1, 8):
        for got, want in zip(R.iirnotch_coeffs(60 * h, 30, 1000), scipy.signal.iirnotch(60 * h, 30, 1000)):
            assert np.array_equal(got, want)
        for got, want in zip(filter_ref.iirnotch(60 * h, 30, 1000), scipy.signal.iirnotch(60 * h, 30, 1000)):
            assert np.array_equal(got, want)
    for fs in (1000, 689.06):
        want = scipy.signal.butter(3, 2, 'highpass', fs=fs)
        for got, w in zip(R.butter_highpass_coeffs(3, 2, fs), want):
            assert np.abs(got - w).max() < 1e-14
        for got, w in zip(filter_ref.butter_highpass(3, 2, fs), want):
            assert np.abs(got - w).max() < 1e-14
        assert np.array_equal(R.lfilter_zi(*R.butter_highpass_coeffs(3, 2, fs)), scipy.signal.lfilter_zi(*R.butter_highpass_coeffs(3, 2, fs)))


# ------------------------------------------------------------------ the device path
def _check_device(R):
    for tag in ('short', 'mid', 'long'):
        x = GOLD[tag + '/x']
        nh = R.apply_to_all(R.notch_harmonics, x, 60, 1000)
        assert isinstance(nh, np.ndarray) and nh.dtype == np.float64
        assert np.abs(nh - GOLD[tag + '/notch_harmonics']).max() < TOL_NOTCH
        chain = R.apply_to_all(R.remove_drift, nh, 1000)
        assert np.abs(chain - GOLD[tag + '/chain']).max() < TOL_DRIFT
        assert np.array_equal(R.apply_to_all(R.subsample, GOLD[tag + '/chain'], 689.06, 1000), GOLD[tag + '/emg_orig'])
        assert np.array_equal(R.apply_to_all(R.subsample, GOLD[tag + '/chain'], 516.79, 1000), GOLD[tag + '/emg'])
        if tag != 'long':
            assert np.abs(R.apply_to_all(R.notch, x, 60, 1000) - GOLD[tag + '/notch60']).max() < TOL_NOTCH
            assert np.abs(R.apply_to_all(R.remove_drift, x, 1000) - GOLD[tag + '/remove_drift']).max() < TOL_DRIFT
            # one channel, 1-d in / 1-d out like the reference's per-channel calls
            one = R.notch(x[:, 3], 60, 1000)
            assert one.shape == (x.shape[0],) and np.abs(one - GOLD[tag + '/notch60'][:, 3]).max() < TOL_NOTCH
    a, b = R.condition_raw_emg_recording(GOLD['mid/x'], GOLD['short/x'], GOLD['context/after'])
    assert a.shape == GOLD['context/emg_orig'].shape and b.shape == GOLD['context/emg'].shape
    assert np.abs(a - GOLD['context/emg_orig']).max() < TOL_DRIFT and np.abs(b - GOLD['context/emg']).max() < TOL_DRIFT
    # too short for scipy's padding -> the same ValueError
    with pytest.raises(ValueError, match='padlen'):
        R.remove_drift(GOLD['short/x'][:12], 1000)
    # a python callable that is not one of ours still goes column by column
    assert np.array_equal(R.apply_to_all(lambda s, k: s * k, GOLD['short/x'], 2.0), GOLD['short/x'] * 2.0)


def test_device_filters(dev):
    from silent_speech_amd import read_emg as R
    _check_device(R)
    x = torch.from_numpy(GOLD['mid/x']).to(dev)                                # tensors in -> tensors out, on the same device
    y = R.remove_drift(R.notch_harmonics(x, 60, 1000), 1000)
    assert torch.is_tensor(y) and y.device.type == dev.type and y.dtype == torch.float64
    assert np.abs(y.cpu().numpy() - GOLD['mid/chain']).max() < TOL_DRIFT


def test_chunk_length_independence(dev):
    """ragged ends: every length around the 256 / 1024 chunk boundaries (+ 2 x padlen of extension) against the serial oracle"""
    from silent_speech_amd import read_emg as R
    rng = np.random.default_rng(5)
    base = rng.standard_normal((1300, 2)) * 30 + 100
    for T in (13, 232, 233, 238, 239, 240, 256, 257, 999, 1000, 1001, 1006, 1007, 1300):
        x = base[:T]
        if T > 9:
            assert np.abs(R.notch(x, 120, 1000) - filter_ref.notch(x, 120, 1000)).max() < TOL_NOTCH, T
        if T > 12:
            assert np.abs(R.remove_drift(x, 1000) - filter_ref.remove_drift(x, 1000)).max() < TOL_DRIFT, T
        for f in (689.06, 516.79, 1000.0, 2000.0):
            assert np.array_equal(R.subsample(x, f, 1000), filter_ref.subsample(x, f, 1000)), (T, f)


@pytest.mark.gpu
def test_long_recording_on_gpu():
    from silent_speech_amd import _lib
    _lib.load()
    """a 60 s, 8-channel recording: the serial oracle on the first channel pair, linearity and DC rejection on the rest"""
    from silent_speech_amd import read_emg as R
    rng = np.random.default_rng(9)
    T = 60000
    x = rng.standard_normal((T, 8)) * 40 + rng.uniform(-200, 200, (1, 8))
    y = R.condition_raw_emg_recording(x)[0]
    want = filter_ref.condition(x[:, :2])[0]
    assert np.abs(y[:, :2] - want).max() < TOL_DRIFT
    y2 = R.condition_raw_emg_recording(2.0 * x + 1000.0)[0]                    # linear, and a constant is removed entirely
    assert np.abs(y2 - 2.0 * y).max() < 1e-6


def test_ragged_batch_equals_per_recording(dev):
    """ss_iir_filtfilt_batch / ss_linear_resample_batch: all recordings of a batch through one launch sequence.  Chunk boundaries are
    relative to each recording's own extended signal, so the results are bit-identical to the one-recording entry points -- and
    therefore within the same bounds of the reference's golden chain."""
    from silent_speech_amd import read_emg as rd
    xs = [torch.from_numpy(GOLD[t + '/x'].astype(np.float64)).to(dev) for t in ['short', 'mid', 'short']]
    xs[-1] = xs[-1][: xs[-1].shape[0] - 7].contiguous()                 # a length none of the others has
    if dev.type != 'cpu':                                               # many chunks per recording, chunk counts that differ between recordings
        rng = np.random.default_rng(8)
        xs += [torch.from_numpy(np.cumsum(rng.standard_normal((n, 8)), 0) * 3.0).to(dev) for n in (20000, 5121, 1024)]
    filters = [rd.iirnotch_coeffs(60 * h, 30, 1000) for h in range(1, 8)] + [rd.butter_highpass_coeffs(3, 2, 1000)]
    ys = rd.filtfilt_cascade_batch(filters, xs)
    for x, y in zip(xs, ys):
        assert torch.equal(y, rd.filtfilt_cascade(filters, x))
    assert np.abs(ys[1].cpu().numpy() - GOLD['mid/chain']).max() < TOL_DRIFT
    rs = rd.subsample_batch(ys, 689.06, 1000)
    for y, r in zip(ys, rs):
        assert torch.equal(r, rd.subsample(y, 689.06, 1000))
    assert np.array_equal(rd.subsample_batch([torch.from_numpy(GOLD['mid/chain']).to(dev)], 689.06, 1000)[0].cpu().numpy(), GOLD['mid/emg_orig'])
    with pytest.raises(ValueError, match='padlen'):
        rd.filtfilt_cascade_batch(filters, [xs[0], xs[0][:9]])
