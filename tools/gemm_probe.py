"""One KC x KC bf16 GEMM shape, repeated (for rocprofv3 --pmc A/B runs of the GEMM kernels).  usage: gemm_probe.py M N K"""
import sys, torch
sys.path.insert(0, '.')
from silent_speech_amd import ops
M, N, K = [int(x) for x in sys.argv[1:4]]
dev = torch.device('cuda')
a = torch.randn(M, K, device=dev).to(torch.bfloat16); b = torch.randn(N, K, device=dev).to(torch.bfloat16)
c = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
for _ in range(6):
    ops.gemm(a, b, c, M, N, K, ops.rowmap(K), ops.rowmap(K), ops.rowmap(N), a_mode=0, b_mode=0)
torch.cuda.synchronize()
