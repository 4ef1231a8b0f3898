import sys
sys.path.insert(0, "dalle-mtf_amd")
import torch, dalle_hip as dh
DEV = "cuda"
M, I, J = 40960, 128, 50816
g = torch.Generator().manual_seed(0)
X = torch.zeros(M, I, dtype=torch.bfloat16, device=DEV)
X[30000:30064] = torch.randn(64, I, generator=g).to(torch.bfloat16).to(DEV)      # only rows beyond 2 GiB / (2 J) = 21130 contribute
Y = torch.randn(M, J, generator=g, dtype=torch.float32).to(torch.bfloat16).to(DEV) if False else (torch.randn(M, 1, device=DEV) * torch.ones(1, J, device=DEV)).to(torch.bfloat16)
w = torch.empty(int(dh.gemm_tn_workspace_bytes(M, I, J)) + 256, dtype=torch.uint8, device=DEV)
ref = X[30000:30064].float().t() @ Y[30000:30064].float()
for wide in (0, 1):
    dh.set_option("tn_wide", wide)
    dW = torch.full((I, J), float("nan"), device=DEV)
    dh.gemm_tn(X, I, Y, J, dW, M, I, J, w)
    torch.cuda.synchronize()
    print("tn_wide", wide, "max |dW|", float(dW.abs().max()), "max |ref|", float(ref.abs().max()), "max err", float((dW - ref).abs().max()))
dh.set_option("tn_wide", 1)
