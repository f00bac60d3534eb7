"""Loss of a FIXED reference-size batch under the full training step (bf16, dropout 0.2, warm-up schedule): must fall steadily.
A cheap end-to-end guard for kernel changes that parity tests at small sizes might miss."""
import sys, torch
sys.path.insert(0, '.')
from silent_speech_amd.architecture import Model
from silent_speech_amd.data_utils import combine_fixed_length
from silent_speech_amd.optim import FusedAdamW
from silent_speech_amd.synthetic import reference_size_batch
from silent_speech_amd.transduction_model import dtw_loss
dev = torch.device('cuda')
torch.manual_seed(0)
model = Model(112, 80, 48, model_size=768, num_layers=6, dropout=0.2, compute_dtype=torch.bfloat16).to(dev).train()
opt = FusedAdamW(model, weight_decay=1e-7)
batch = reference_size_batch(seed=0, device=dev)
losses = []
for it in range(40):
    opt.zero_grad()
    for g in opt.param_groups:
        g['lr'] = min(1.0, (it + 1) / 20) * 3e-4
    X = combine_fixed_length(batch['emg'], 200); X_raw = combine_fixed_length(batch['raw_emg'], 1600); sess = combine_fixed_length(batch['session_ids'], 200)
    pred, aux = model(X, X_raw, sess)
    loss, _ = dtw_loss(pred, aux, batch, phoneme_loss_weight=0.5)
    loss.backward()
    opt.step()
    losses.append(float(loss.detach()))
print(' '.join('%.3f' % l for l in losses))
assert all(l == l for l in losses), 'NaN'
assert losses[-1] < 0.8 * losses[0], 'loss did not fall'
print('ok: %.3f -> %.3f' % (losses[0], losses[-1]))
