"""Numerical experiment behind DESIGN.md §4 "Arithmetic": emulate split-bf16 convolutions (3-term bf16x3 vs 6-term
bf16x6) inside the CPU oracle and compare logits / gradients with fp64 and with plain fp32.
  python tests/experiments/split_precision.py 3     # logits 6e-3, gradients 16 %  -> rejected
  python tests/experiments/split_precision.py 6     # logits 1e-5, gradients 0.2 % -> as good as fp32 (3.9 %)
(CPU only; not collected by pytest.)"""
import sys, numpy as np, torch, torch.nn.functional as F
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import pfst_oracle as O
from pfst_amd.synthetic import synth_batch, fill_state_dict
torch.set_num_threads(8)
NTERM = int(sys.argv[1]) if len(sys.argv)>1 else 3
def bf(x): return x.float().bfloat16().double()
def split(x, n):
    parts=[]; r=x
    for _ in range(n):
        p=bf(r); parts.append(p); r=r-p
    return parts
def pairs(n):  # which (i,j) products are kept: x3 -> (0,0),(0,1),(1,0); x6 -> all with i+j<=2
    lim = 1 if n==3 else 2
    return [(i,j) for i in range(lim+1) for j in range(lim+1) if i+j<=lim]
orig=F.conv2d
class SplitConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, stride, pad, dil, groups):
        ctx.save_for_backward(x, w); ctx.a=(stride,pad,dil,groups)
        if groups>1: return orig(x,w,None,stride,pad,dil,groups)
        xs, ws = split(x,3), split(w,3)
        return sum(orig(xs[i],ws[j],None,stride,pad,dil) for i,j in pairs(NTERM))
    @staticmethod
    def backward(ctx, dy):
        x,w=ctx.saved_tensors; stride,pad,dil,groups=ctx.a
        if groups>1:
            return (torch.nn.grad.conv2d_input(x.shape,w,dy,stride,pad,dil,groups), torch.nn.grad.conv2d_weight(x,w.shape,dy,stride,pad,dil,groups),None,None,None,None)
        ds, ws, xs = split(dy,3), split(w,3), split(x,3)
        dx=sum(torch.nn.grad.conv2d_input(x.shape,ws[j],ds[i],stride,pad,dil) for i,j in pairs(NTERM))
        dw=sum(torch.nn.grad.conv2d_weight(xs[i],w.shape,ds[j],stride,pad,dil) for i,j in pairs(NTERM))
        return dx,dw,None,None,None,None

def run(mode):
    sd = fill_state_dict(O.init_state_dict(6,3), 5)
    dt = torch.float32 if mode=='f32' else torch.float64
    sd = {k:(v.to(dt) if v.is_floating_point() else v) for k,v in sd.items()}
    pk = O.param_keys(sd)
    for k in pk: sd[k].requires_grad_(True)
    bt = synth_batch(2,64,6,seed=1234)
    if mode=='split':
        F.conv2d = lambda x,w,b=None,s=1,p=0,d=1,g=1: SplitConv.apply(x,w,s,p,d,g) + (0 if b is None else b.view(1,-1,1,1))
    try:
        losses, feats, logits, dec, _ = O.segmentor_forward_train(sd, bt['img'].to(dt), bt['gt_semantic_seg'], None)
        loss, log = O.parse_losses(losses); loss.backward()
    finally:
        F.conv2d = orig
    return {k: sd[k].grad.double() for k in pk}, logits.detach().double()
g64,l64 = run('f64'); g32,l32 = run('f32'); gs,ls = run('split')
ks=list(g64)
e32=[float((g32[k]-g64[k]).norm()/g64[k].norm()) for k in ks]
es=[float((gs[k]-g64[k]).norm()/g64[k].norm()) for k in ks]
print('terms',NTERM,'logits rel err: f32 %.2e split %.2e'%(float((l32-l64).norm()/l64.norm()), float((ls-l64).norm()/l64.norm())))
print('grad rel err vs f64: f32 median %.2e max %.2e | split median %.2e max %.2e'%(np.median(e32),max(e32),np.median(es),max(es)))
