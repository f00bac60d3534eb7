"""Developer tool: the log-mel kernel (csrc/mel.hip, ss_stft_logmel_fft) alone on a large ragged-free batch: HIP-event time per launch, frames / s and the
fraction of the 1344 B / frame HBM roofline.  GPU only."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from silent_speech_amd import data_utils

dev = torch.device('cuda:0')
for n_utt, seconds in ((32, 6.0), (256, 8.0), (1024, 8.0)):
    L = int(22050 * seconds) // 256 * 256
    y = (0.1 * torch.randn(n_utt, L, device=dev)).clamp_(-1, 1)
    F = 1 + (L + 768 - 1024) // 256
    out = torch.empty(n_utt, 80, F, device=dev)
    run = lambda: data_utils._logmel_fft(y, None, None, L, n_utt, F, 384, False, 1024, 80, 22050, 256, 1024, 0, 8000, out, False)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    a.record()
    for _ in range(n):
        run()
    b.record(); torch.cuda.synchronize()
    t = a.elapsed_time(b) / n * 1e-3
    fr = n_utt * F
    print('%5d x %.0f s: %7d frames  %8.1f us  %7.1f M frames/s  %5.1f %% of the 1344 B/frame HBM roofline' % (n_utt, seconds, fr, t * 1e6, fr / t / 1e6, fr * 1344 / t / 8e12 * 100))
