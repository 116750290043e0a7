import os, sys, time, ctypes, torch
sys.path.insert(0, os.getcwd())
from xfmamba_amd import _lib
from xfmamba_amd import ss2d as S2
dev="cuda"; dt=torch.bfloat16
Bt,D,H,N=64,96,56,1; L=H*H
x=torch.randn(Bt,D,L,device=dev).to(dt); dts=(0.5*torch.rand(Bt,4,D,L,device=dev)).to(dt)
Bs=torch.randn(Bt,4,N,L,device=dev).to(dt); Cs=torch.randn(Bt,4,N,L,device=dev).to(dt)
A=-torch.rand(4*D,N,device=dev)-0.1; Dp=torch.randn(4*D,device=dev); bias=0.1*torch.rand(4*D,device=dev)
plan=_lib.ScanPlan(); _lib.lib().xfm_ss2d_plan(Bt,D,H,H,N,2,ctypes.byref(plan))
chk=torch.empty((Bt,4,D,plan.n_chunks,N),dtype=torch.float32,device=dev); y=torch.empty((Bt,D,L),dtype=torch.float32,device=dev)
p=_lib.SS2DParams(); S2._fill(p,x,dts,A,Bs,Cs,Dp,bias,H,H,torch.float32,chk); p.y=y.data_ptr()
f=_lib.lib().xfm_ss2d_fwd; st=_lib.stream_ptr()
for _ in range(3): f(ctypes.byref(p),st)
torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(100): f(ctypes.byref(p),st)
t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
print(f"cpu enqueue {1e6*(t1-t0)/100:.1f} us/call, total incl gpu {1e6*(t2-t0)/100:.1f} us/call")
a=torch.randn(1024,device=dev); 
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(100): a.add_(1)
t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
print(f"torch add: cpu {1e6*(t1-t0)/100:.1f} us/call, total {1e6*(t2-t0)/100:.1f}")
