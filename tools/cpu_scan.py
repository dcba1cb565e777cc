import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bench import make_case, MXINT_Q, flops
from oracle import lqer_oracle as O
M,K,N,r=2048,4096,4096,32
x,W,A,B=make_case(M,K,N,r,seed=0); x=x.half().float()
wq=O.mxint_quantize(W,width=4,block_size=[1,16],skip_first_dim=False)
for th in (8,16,32,64,128):
    torch.set_num_threads(th)
    for unf in (True, False):
        O.lqer_linear_forward(x,wq,None,A,B,MXINT_Q,weight_is_quantized=True,via_unfold=unf)
        t0=time.perf_counter(); O.lqer_linear_forward(x,wq,None,A,B,MXINT_Q,weight_is_quantized=True,via_unfold=unf); t=time.perf_counter()-t0
        print(th, "unfold" if unf else "reshape", f"{t*1e3:.0f} ms")
