import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fvgp_amd import _lib
H = _lib.Handle(0)
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best
out = H.empty(4096 * 256)
for blocks in (256, 512, 768, 1024, 2048):
    iters = 4000
    ms = timeit(lambda: H.mfma_peak(out, blocks, iters))
    fl = blocks * 4 * iters * 16 * 2048.0
    print(json.dumps({"test": "mfma_peak", "blocks": blocks, "ms": round(ms, 3), "tflops": round(fl / ms / 1e9, 2)}))
