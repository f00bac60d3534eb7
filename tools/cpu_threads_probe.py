import os, sys, time
sys.path.insert(0, '.')
import torch
from oracle import model_ref, loss_ref
from silent_speech_amd.synthetic import reference_size_batch
b = reference_size_batch(seed=0)
sub = {k: v[:1] for k, v in b.items()}
sd = model_ref.init_state_dict(768, 6, 80, 48, seed=0)
for v in sd.values():
    if v.dtype == torch.float32: v.requires_grad_(True)
print('cpu_count', os.cpu_count(), 'default threads', torch.get_num_threads(), flush=True)
for nt in (16, 32, 64, 128):
    torch.set_num_threads(nt)
    x = loss_ref.combine_fixed_length(sub['raw_emg'], 1600)
    t = time.perf_counter()
    pred, aux = model_ref.model_forward(sd, x, training=True, shift_r=0, running_out={})
    loss, _ = loss_ref.dtw_loss_ref(pred, aux, sub)
    loss.backward()
    print(nt, 'threads: %.2f s fwd+bwd for %d frames' % (time.perf_counter() - t, sum(sub['lengths'])), flush=True)
