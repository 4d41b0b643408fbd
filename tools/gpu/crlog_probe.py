import ctypes as C, math, numpy as np, torch
L=C.CDLL('tests/gpu_kernels/libcrlog_harness.so'); dp=C.POINTER(C.c_double)
L.crlog_run.argtypes=[dp,dp,dp,C.c_long]
rng=np.random.RandomState(0); n=200000
x1=2*rng.random_sample(n)-1; x2=2*rng.random_sample(n)-1
r2=x1*x1+x2*x2; r2=np.ascontiguousarray(r2[(r2<1)&(r2>0)])
lg=np.zeros_like(r2); f=np.zeros_like(r2)
print("rc", L.crlog_run(r2.ctypes.data_as(dp), lg.ctypes.data_as(dp), f.ctypes.data_as(dp), len(r2)))
hl=np.array([math.log(v) for v in r2]); hf=np.array([math.sqrt(-2.0*math.log(v)/v) for v in r2])
print(len(r2), "device log_cr != glibc log:", int((lg!=hl).sum()), " device f != host f:", int((f!=hf).sum()))
# f from the HOST log but device division/sqrt?  check division and sqrt separately through torch
t=torch.from_numpy(r2).cuda(); num=torch.from_numpy(-2.0*hl).cuda()
print("torch div mismatch", int(((num/t).cpu().numpy()!=(-2.0*hl)/r2).sum()), "torch sqrt mismatch", int((torch.sqrt(num/t).cpu().numpy()!=np.sqrt((-2.0*hl)/r2)).sum()))
