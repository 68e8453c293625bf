cd $GRAFT_REPO_ROOT
python - <<'PY'
import torch, sys
sys.path.insert(0,'.')
from fvgp_amd import _lib
H=_lib.Handle(0)
g = torch.Generator(device="cuda"); g.manual_seed(3)
for (M, N, K) in ((512, 384, 176), (768, 128, 16), (256, 256, 2064), (512,256,32), (512,256,48), (512,256,64), (2048,2048,2048), (256,128,16)):
    A = torch.randn(M, K, dtype=torch.float64, device="cuda", generator=g)
    B = torch.randn(N, K, dtype=torch.float64, device="cuda", generator=g)
    C0 = torch.randn(M, N, dtype=torch.float64, device="cuda", generator=g)
    out={}
    for v in (0, 9000):
        H.set_option("gemm_probe", v)
        C = C0.clone()
        H.gemm(0, 0, 0, M, N, K, -0.75, A, B, 1.25, C)
        H.sync()
        out[v]=C
    d=(out[0]-out[9000]).abs()
    bad=(d>0).nonzero()
    print(M,N,K,"equal",torch.equal(out[0],out[9000]),"maxdiff",float(d.max()),"nbad",len(bad), bad[:3].tolist() if len(bad) else "")
PY
