"""Step time with and without dropout (same model / batch as bench.py): what the in-kernel RNG + masking costs."""
import sys, time, torch
sys.path.insert(0, '.')
from silent_speech_amd.architecture import Model
from silent_speech_amd.data_utils import combine_fixed_length
from silent_speech_amd.optim import FusedAdamW
from silent_speech_amd.synthetic import reference_size_batch
from silent_speech_amd.transduction_model import dtw_loss
dev = torch.device('cuda')
batch = reference_size_batch(seed=0, device=dev)
for p in (0.2, 0.0, 0.2, 0.0):
    torch.manual_seed(0)
    model = Model(112, 80, 48, model_size=768, num_layers=6, dropout=p, compute_dtype=torch.bfloat16).to(dev).train()
    opt = FusedAdamW(model, weight_decay=1e-7)
    def step():
        opt.zero_grad()
        X = combine_fixed_length(batch['emg'], 200); X_raw = combine_fixed_length(batch['raw_emg'], 1600); sess = combine_fixed_length(batch['session_ids'], 200)
        pred, aux = model(X, X_raw, sess)
        loss, _ = dtw_loss(pred, aux, batch, phoneme_loss_weight=0.5)
        loss.backward(); opt.step()
    for _ in range(3): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize(); print('dropout %.1f: %.2f ms/step' % (p, (time.perf_counter() - t) * 100), flush=True)
