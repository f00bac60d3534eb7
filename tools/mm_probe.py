import torch, time
dev='cuda'
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n
for (M,N,K,tag) in [(3072,768,22016,'dW ffn1'),(768,3072,22016,'dW ffn2'),(768,768,22016,'dW lin'),(768,2304,22016,'dW conv200'),(768,2304,88064,'dW conv800'), (2304,768,22016,'dW qkv')]:
    a=torch.randn(K,M,device=dev,dtype=torch.bfloat16); b=torch.randn(K,N,device=dev,dtype=torch.bfloat16)
    t=timeit(lambda: torch.mm(a.t(), b))
    print(tag, 'TN bf16->bf16 %.1f us %.0f TF' % (t*1e6, 2.0*M*N*K/t/1e12))
    try:
        t=timeit(lambda: torch.mm(a.t(), b, out_dtype=torch.float32))
        print(tag, 'TN bf16->f32 %.1f us %.0f TF' % (t*1e6, 2.0*M*N*K/t/1e12))
    except Exception as e:
        print('out_dtype unsupported', str(e)[:80])
# forward-like NT gemm for reference
for (M,N,K,tag) in [(22016,3072,768,'ffn1'),(22016,768,3072,'ffn2'),(22016,768,768,'lin')]:
    a=torch.randn(M,K,device=dev,dtype=torch.bfloat16); w=torch.randn(N,K,device=dev,dtype=torch.bfloat16)
    t=timeit(lambda: torch.mm(a, w.t()))
    print(tag, 'NT %.1f us %.0f TF' % (t*1e6, 2.0*M*N*K/t/1e12))
