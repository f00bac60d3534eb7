"""Developer tool (GPU): the transposed-score attention forward against the closed form over a list of shapes, with the rows / tiles / (batch, head)
pairs that are off -- how the narrow-band and partial-tile failures of round 5 (asm loads under branches) were localised."""
import math, sys, torch
sys.path.insert(0, '.')
from oracle import model_ref
from silent_speech_amd import _lib, ops
_lib.load()
dev = torch.device('cuda')
def ref(q,k,v,E,D,dh):
    logits = torch.einsum('bhqa,bhka->bhqk', q, k)/math.sqrt(dh) + model_ref.relpos_logits(q, E[...,None], max_rel=D)
    P = torch.softmax(logits,-1)
    return torch.einsum('bhqk,bhka->bhqa',P,v), torch.logsumexp(logits,-1)
def run(B,H,T,dh,D,seed=1):
    dt=torch.bfloat16; dp=(dh+31)//32*32
    g=torch.Generator().manual_seed(seed)
    q,k,v=[(torch.randn(B,H,T,dh,generator=g)*0.8).to(dt).float() for _ in range(3)]
    E=(torch.randn(H,2*D-1,dh,generator=g)*dh**-0.5).to(dt).float()
    O,LSE=ref(q,k,v,E,D,dh)
    def pack(x):
        o=torch.zeros(B,T,H,dp); o[...,:dh]=x.permute(0,2,1,3); return o
    qkv=torch.cat([pack(t).reshape(B*T,H*dp) for t in (q,k,v)],1).to(dt).contiguous().to(dev)
    Ed=torch.zeros(H,2*D-1,dp,dtype=dt); Ed[...,:dh]=E.to(dt)
    out=torch.zeros(B*T,H*dp,dtype=dt,device=dev); lse_d=torch.zeros(B,H,T,device=dev)
    ops.relpos_attention_forward(qkv,None,Ed.to(dev),out,lse_d,B,H,T,(T+7)//8*8,dp,D,1/math.sqrt(dh))
    got=out.float().cpu().view(B,T,H,dp)[...,:dh].permute(0,2,1,3)
    err=(got-O).abs(); lerr=(lse_d.cpu()-LSE).abs()
    print('   lse err by row/8:', [round(float(x),2) for x in lerr[0,0].view(-1)[: (T//8)*8].view(-1,8).amax(1)]); print('   O err by row/8:', [round(float(x),2) for x in err[0,0].amax(-1)[: (T//8)*8].view(-1,8).amax(1)])
    print('   lse bad rows', int((lerr>0.05).sum()), 'first', (lerr>0.05).nonzero()[:3].tolist(), 'nan in out', int(torch.isnan(got).sum()))
    bad = (err.amax(-1) > 0.03).nonzero()
    print('B',B,'H',H,'T',T,'dh',dh,'D',D,'max err',float(err.max()),'bad rows',len(bad), 'first', bad[:4].tolist(), 'bad tiles', sorted(set((int(r[2])//32) for r in bad))[:8], 'bad b,h', sorted(set((int(r[0]),int(r[1])) for r in bad))[:6])
for args in [(1,1,200,96,17),(1,1,131,96,17),(1,1,131,96,100)]:
    run(*args)
