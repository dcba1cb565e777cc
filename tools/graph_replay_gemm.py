"""Round 6 debugging aid: the split calls (lqer_quantize_act_xa + lqer_linear_gemm) of a multi-round int8 shape captured in a hipGraph and
replayed with other tokens, against the eager calls - how the memset-node problem of the atomicMax pre-pass was isolated (the fill is a kernel
now).   usage: python tools/graph_replay_gemm.py [tuning, e.g. 0x4000000] [f32|f16] [both|act|gemm: which call is captured]"""
import sys, os, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lqer_amd
from lqer_amd import _lib, ops
from bench import INT_Q, make_case
DEV="cuda:0"
M,K,N,r = 640,128,16384,32
dtype = {"f32":torch.float32,"f16":torch.float16}[sys.argv[2]] if len(sys.argv)>2 else torch.float32
tun = int(sys.argv[1],0) if len(sys.argv)>1 else 0
what = sys.argv[3] if len(sys.argv)>3 else "both"
x,W,A,B = make_case(M,K,N,r,seed=N+r+M,quantize_ab=False)
mod = lqer_amd.LinearFlexibleLqer(K,N,bias=False,q_config=INT_Q,l_config={"rank":r}); mod.load_state_dict({"weight":W,"A":A,"B":B}); mod=mod.to(DEV).to(dtype)
mod.tuning=_lib.TUNE_I8_ROWS_128|tun
xd=x.to(dtype).to(DEV); mod(xd)
L=_lib.lib(); desc=mod._desc(); p=mod._packed
dt=ops.dtype_code(xd)
a_t,a_limbs = mod._side_image(M, desc, dt)
wsb=ops.linear_sizes(desc,M).workspace
Kp,Mp,rp=L.lqer_padded_k(K),L.lqer_padded_m(M),L.lqer_padded_r(r)
act=L.lqer_act_image_bytes(C.byref(desc),M)
offx=act; offs=act+((Mp*rp*2+255)//256)*256
nscr=L.lqer_lowrank_xa_scratch_bytes(C.byref(desc),M); gscr=L.lqer_linear_gemm_scratch_bytes(C.byref(desc),M)
def qxa(xt,ws,st):
    rc=L.lqer_quantize_act_xa(C.byref(desc), xt.data_ptr(), dt, M, K, a_t, a_limbs, ws.data_ptr(), ws.data_ptr()+offx, ws.data_ptr()+offs, nscr, st); assert rc==0, rc
def gemm(y,ws,st):
    rc=L.lqer_linear_gemm(C.byref(desc), ws.data_ptr(), M, p["w"].data_ptr(), ws.data_ptr()+offx, p["b_t"].data_ptr(), p["b_limbs"], None, y.data_ptr(), dt, N, ws.data_ptr()+offs, gscr, st); assert rc==0, rc
xs=xd.clone(); yg=torch.empty(M,N,dtype=dtype,device=DEV); wsg=torch.zeros(wsb,dtype=torch.uint8,device=DEV)
ye=torch.empty(M,N,dtype=dtype,device=DEV); wse=torch.zeros(wsb,dtype=torch.uint8,device=DEV)
cur=lambda: torch.cuda.current_stream().cuda_stream
s=torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    qxa(xs,wsg,s.cuda_stream); gemm(yg,wsg,s.cuda_stream)
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g=torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    if what in ("both","act"): qxa(xs,wsg,cur())
    if what in ("both","gemm"): gemm(yg,wsg,cur())
for scale in (1.0,-0.5,3.0):
    xn=(x*scale).to(dtype).to(DEV)
    qxa(xn,wse,cur()); gemm(ye,wse,cur()); torch.cuda.synchronize()
    xs.copy_(xn)
    if what=="gemm": qxa(xs,wsg,cur())
    g.replay()
    if what=="act": gemm(yg,wsg,cur())
    torch.cuda.synchronize()
    d=(yg.float()-ye.float()).abs(); rows=(d.max(dim=1)[0]>0).nonzero().flatten().tolist()
    print(what, sys.argv[2] if len(sys.argv)>2 else "f32", hex(tun), "scale",scale,"y rows differing",len(rows),rows[:6])
