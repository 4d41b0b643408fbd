import math, numpy as np, torch
rng = np.random.RandomState(0)
n = 400000
x1 = 2*rng.random_sample(n)-1; x2 = 2*rng.random_sample(n)-1
r2 = x1*x1+x2*x2; r2 = r2[(r2<1)&(r2>0)]
host = np.array([math.log(v) for v in r2])
dev = torch.log(torch.from_numpy(r2).cuda()).cpu().numpy()
npv = np.log(r2)
print("samples", len(r2), "device log != glibc log:", int((dev!=host).sum()), "numpy vector log != glibc:", int((npv!=host).sum()))
f_host = np.array([math.sqrt(-2.0*math.log(v)/v) for v in r2])
t = torch.from_numpy(r2).cuda()
f_dev = torch.sqrt(-2.0*torch.log(t)/t).cpu().numpy()
print("f = sqrt(-2 log(r2)/r2): device != host:", int((f_dev!=f_host).sum()))
d = np.abs(dev-host)/np.abs(host); print("max rel diff", d.max())
