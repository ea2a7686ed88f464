import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lrcn_amd
from lrcn_amd import lrcn as L
# usage: tools/lstm_step_bench.py B [E=H] [V] [T]   (LSTM-only training step: lossgradient + Adam on given features)
B=int(sys.argv[1]); E=H=int(sys.argv[2]) if len(sys.argv)>2 else 1000; V=int(sys.argv[3]) if len(sys.argv)>3 else 10640; T=int(sys.argv[4]) if len(sys.argv)>4 else 11
ctx = L.Context(E,H,H,V,max_B=B,max_T=T,lstm_dtype=lrcn_amd.LRCN_BF16)
ctx.set_option(lrcn_amd._lib.LRCN_OPT_FUSED_UPDATE, int(os.environ.get('LRCN_FUSED_UPDATE','0')))
param = L.initweights(ctx, seed=42); optim = L.initparams(param); grads = L.zeros_like_model(param)
feats = L.to_jl((np.random.default_rng(0).standard_normal((B,4096))*0.01).astype(np.float32))
toks = torch.as_tensor(np.random.default_rng(1).integers(3,V,size=(T,B)).astype(np.int32)).cuda()
for _ in range(5): L.train_step(ctx,param,optim,grads,feats,toks,pdrop=0.4,seed=1)
torch.cuda.synchronize()
n=30
t0=time.perf_counter()
for i in range(n): L.train_step(ctx,param,optim,grads,feats,toks,pdrop=0.4,seed=i)
t1=time.perf_counter()
torch.cuda.synchronize()
t2=time.perf_counter()
print("B=%d E=H=%d V=%d T=%d host enqueue %.3f ms/step, total %.3f ms/step = %.0f captions/s"%(B,E,V,T,(t1-t0)/n*1e3,(t2-t0)/n*1e3, B*n/(t2-t0)))
