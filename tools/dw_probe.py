import sys, torch
sys.path.insert(0, '.')
from silent_speech_amd import ops
from silent_speech_amd._lib import OP_KC, OP_OC
dev = torch.device('cuda'); dt = torch.bfloat16
R, N, K = 22016, 3072, 768
dy = torch.randn(R, N, device=dev).to(dt); x = torch.randn(R, K, device=dev).to(dt)
dW = torch.zeros(N, K, device=dev)
for _ in range(3):
    ops.gemm(dy, x, dW, N, K, R, ops.rowmap(N), ops.rowmap(K), ops.rowmap(K), a_mode=OP_OC, b_mode=OP_OC, mode=2, split_k=8)
a = torch.randn(R, K, device=dev).to(dt); b = torch.randn(N, K, device=dev).to(dt); c = torch.zeros(R, N, device=dev, dtype=dt)
for _ in range(3):
    ops.gemm(a, b, c, R, N, K, ops.rowmap(K), ops.rowmap(K), ops.rowmap(N))
torch.cuda.synchronize()
