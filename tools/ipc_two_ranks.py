"""Two ranks sharing ONE GPU, the row-sharded evaluation with the direct collectives of csrc/ipc.hip: what the all-gathers cost on the
chain stream beside the ranks' trailing updates (events around every call), the wall time of an evaluation, and the agreement with a
single-GPU evaluation.  Starts its two ranks itself.   python tools/ipc_two_ranks.py [N] [panel]"""
import json, os, socket, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("RANK") is None:
    n = sys.argv[1] if len(sys.argv) > 1 else "20000"
    panel = sys.argv[2] if len(sys.argv) > 2 else "1024"
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2",
                   HSA_ENABLE_IPC_MODE_LEGACY="0", FVGP_DEVICE="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), n, panel], env=env))
    sys.exit(max(p.wait(timeout=600) for p in procs))
sys.path.insert(0, ROOT)
import time
import numpy as np, torch, torch.distributed as dist
from fvgp_amd import _lib
from fvgp_amd.dist import ShardedGP
torch.cuda.set_device(0)
dist.init_process_group(backend="gloo")
n, panel = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(20240501)
x = rng.random((n, 3)); y = np.sin(3 * x.sum(1)) + 0.1 * rng.standard_normal(n)
th = np.array([1.0, 0.3, 0.3, 0.3])
gp = ShardedGP(x, y, np.full(n, 0.01), kernel="rbf_ard", panel=panel, collectives="ipc")
gp.log_likelihood(th)
gp.ops.set_option("profile", 1); gp.collective_summary()
dist.barrier(); torch.cuda.synchronize(); t0 = time.perf_counter()
ll = gp.log_likelihood(th * 1.01)[0]
torch.cuda.synchronize(); wall = time.perf_counter() - t0
prof = gp.collective_summary()
if dist.get_rank() == 0:
    H = _lib.Handle(0); npad = _lib.pad128(n)
    KV = H.empty(npad, npad); al = H.empty(npad, 1)
    ref = H.loglik(0, H.to_device(x), th * 1.01, H.to_device(np.full(n, 0.01)), H.to_device((y - y.mean()).reshape(n, 1)), KV, al)[0]
    print(json.dumps({"ranks_on_one_gpu": 2, "n": n, "panel": panel, "evaluation_ms": 1e3 * wall, "rel_diff_vs_single_gpu": abs(ll - ref) / abs(ref),
                      "all_gather": {"calls": prof["all_gather"][0], "bytes_from_peers": prof["all_gather"][1], "ms_on_chain_stream_beside_update": prof["all_gather"][2]}}), flush=True)
gp.close()
dist.destroy_process_group()
