"""Does the trailing-update kernel hold its burst rate when launched back to back for about a second (DVFS)?"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fvgp_amd import _lib

H = _lib.Handle(0)
g = torch.Generator(device="cuda"); g.manual_seed(0)
M, K = 24576, 2048
A = torch.randn(M, K, dtype=torch.float64, device="cuda", generator=g)
C = torch.randn(M, M, dtype=torch.float64, device="cuda", generator=g)
T = M // 128; fl = T * (T + 1) // 2 * 128 * 128 * 2.0 * K
H.gemm(0, 0, 1, M, M, K, -1.0, A, A, 1.0, C); torch.cuda.synchronize()
time.sleep(2.0)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(81)]
ev[0].record()
for i in range(80):
    H.gemm(0, 0, 1, M, M, K, -1e-6, A, A, 1.0, C)
    ev[i + 1].record()
torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(80)]
print(json.dumps({"first5_tflops": [round(fl / m / 1e9, 2) for m in ms[:5]], "last5_tflops": [round(fl / m / 1e9, 2) for m in ms[-5:]],
                  "every10": [round(fl / m / 1e9, 2) for m in ms[::10]], "total_ms": round(sum(ms), 1)}))
