import sys, os, time, warnings
sys.path.insert(0, "/root/repo")
import numpy as np, torch
warnings.simplefilter("ignore")
import fvgp_amd
from fvgp_amd import _lib
def synth(n, d):
    rng = np.random.default_rng(20240501); x = rng.random((n, d))
    return x, np.sin(3.0 * x.sum(axis=1)) + 0.1 * rng.standard_normal(n)
for n, d in ((200,1),(500, 1), (1000,3),(2000, 3), (4000,3), (8000, 3)):
    x, y = synth(n, d)
    ths = np.array([1.0] + [0.2 if d == 1 else 0.3] * d)
    gp = fvgp_amd.GP(x, y, init_hyperparameters=ths, noise_variances=np.full(n, 0.01), kernel_function="rbf_ard")
    ts=[]
    for i in range(8):
        torch.cuda.synchronize(); t0=time.perf_counter(); gp.log_likelihood(ths*(1.01+0.001*i)); torch.cuda.synchronize(); ts.append(time.perf_counter()-t0)
    H = gp._H
    npad=_lib.pad128(n)
    xd=H.to_device(x); ymd=H.zeros(npad,1); ymd[:n,0]=H.to_device(y-y.mean()); V=H.to_device(np.full(n,0.01)); KV=H.empty(npad,npad); al=H.empty(npad,1)
    tr=[]
    for i in range(8):
        torch.cuda.synchronize(); t0=time.perf_counter(); H.loglik(0,xd,ths*(1.01+0.001*i),V,ymd,KV,al); torch.cuda.synchronize(); tr.append(time.perf_counter()-t0)
    print(f"N {n} d {d}: facade {1e3*min(ts):.3f} ms, raw ABI call {1e3*min(tr):.3f} ms")
