"""Developer tool: which Python-side ops of one training step turn into device copies / small glue kernels (torch.profiler, with stacks)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from silent_speech_amd.architecture import Model
from silent_speech_amd.data_utils import combine_fixed_length
from silent_speech_amd.optim import FusedAdamW
from silent_speech_amd.synthetic import reference_size_batch
from silent_speech_amd.transduction_model import dtw_loss

dev = torch.device('cuda:0')
torch.manual_seed(0)
model = Model(112, 80, 48, model_size=768, num_layers=6, dropout=0.2, compute_dtype=torch.float32, f32_matmul='bf16x3').to(dev)
model.train()
optim = FusedAdamW(model, weight_decay=1e-7)
b = reference_size_batch(seed=0)
batch = {k: ([t.to(dev) for t in v] if isinstance(v, list) and len(v) and torch.is_tensor(v[0]) else v) for k, v in b.items()}

def step():
    optim.zero_grad()
    X = combine_fixed_length(batch['emg'], 200)
    X_raw = combine_fixed_length(batch['raw_emg'], 1600)
    sess = combine_fixed_length(batch['session_ids'], 200)
    pred, aux = model(X, X_raw, sess)
    loss, _ = dtw_loss(pred, aux, batch, phoneme_loss_weight=0.5)
    loss.backward()
    optim.step()

for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
ev = [e for e in prof.events() if e.device_type is not None and str(e.device_type).endswith('CUDA')]
import collections
agg = collections.defaultdict(lambda: [0, 0.0])
for e in ev:
    agg[e.name[:70]][0] += 1; agg[e.name[:70]][1] += e.device_time if hasattr(e, 'device_time') else e.cuda_time
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print('%-72s %4d %9.1f us' % (k, n, t))
print('---- CPU ops that launched Memcpy / copy kernels')
print(prof.key_averages(group_by_stack_n=6).table(sort_by='cuda_time_total', row_limit=25, max_name_column_width=60))
