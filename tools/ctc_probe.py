"""CTC loss of a reference-size recognition batch (128 000-sample budget): kernel-level timing of ss_ctc_loss (run under rocprofv3 --kernel-trace --stats)
and the device time of the whole op with the plan (descriptors, targets) built once.  Tuning aid (GPU only)."""
import torch
from silent_speech_amd.recognition_model import ctc_loss, _CtcPlan
from silent_speech_amd.synthetic import reference_size_batch

dev = torch.device('cuda')
b = reference_size_batch(seed=11, budget=128000, device=dev)
rows = (sum(b['lengths']) + 199) // 200
logits = torch.randn(rows, 200, 38, device=dev, requires_grad=True)
for _ in range(3):
    ctc_loss(logits, b, blank=37).backward()
torch.cuda.synchronize()
plan = _CtcPlan(b['lengths'], b['text_int'], rows * 200, dev)
flat = logits.detach().reshape(rows * 200, 38).contiguous()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 20
e0.record()
for _ in range(n):
    torch.ops.silent_speech.ctc_loss(flat, plan.desc, plan.targets, plan.n, plan.max_s, plan.ws_floats, 38, 37)
e1.record(); torch.cuda.synchronize()
print('utterances %d, frames %d, longest %d frames / %d labels: ss_ctc_loss (lse + alpha/beta + gradient kernels) %.3f ms per call' % (
    len(b['lengths']), sum(b['lengths']), max(b['lengths']), plan.max_s, e0.elapsed_time(e1) / n))
