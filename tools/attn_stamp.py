import math, os, sys, torch
sys.path.insert(0, '/root/repo')
from silent_speech_amd import ops
dev = torch.device('cuda')
B, H, T, dh, D = 110, 8, 200, 96, 100
dp, Tp = 96, 200
dt = torch.bfloat16
qkv = (torch.randn(B * T, 3 * H * dp, device=dev) * 0.5).to(dt)
qkvT = qkv.view(B, T, 3 * H * dp).transpose(1, 2).contiguous()
E = (torch.randn(H, 2 * D - 1, dp, device=dev) * 0.1).to(dt)
o = torch.empty(B * T, H * dp, device=dev, dtype=dt); lse = torch.zeros(B, H, T, device=dev)
for _ in range(3):
    ops.relpos_attention_forward(qkv, qkvT, E, o, lse, B, H, T, Tp, dp, D, 1 / math.sqrt(dh), p=0.0, seed=1, rng_stream=0)
torch.cuda.synchronize()
st = lse.view(-1)[:128].cpu().view(8, 16)
for w in range(8):
    print('wave', w, [int(x) for x in st[w].tolist()])
