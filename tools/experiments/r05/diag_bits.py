import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dalle-mtf_amd")]
import torch, numpy as np
import dalle_hip as dh
M, N, K = 300, 256, 128
g = torch.Generator().manual_seed(1)
A = torch.randn(M, K, generator=g).bfloat16().cuda(); Bt = (torch.randn(N, K, generator=g) * 0.2).bfloat16().cuda(); bias = torch.randn(N, generator=g).bfloat16().cuda()
h = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
bits = torch.full((dh.relu_bits_bytes(M, N),), 0xAA, dtype=torch.uint8, device="cuda")
dh.gemm_nt_relu_bits(A, K, Bt, K, h, N, M, N, K, bias, bits)
torch.cuda.synchronize()
hb = (h.float() > 0).cpu().numpy()
w = bits.cpu().numpy().view(np.uint16).reshape(N // 64, M, 4)
exp = np.zeros_like(w)
for grp in range(N // 64):
    for g16 in range(4):
        nst = grp * 64 + 8 * (((g16 & 1) << 1) | (g16 >> 1))
        for k in range(8):
            exp[grp, :, g16] |= (hb[:, nst + k].astype(np.uint16) << k) | (hb[:, nst + 32 + k].astype(np.uint16) << (8 + k))
bad = (w != exp)
print("producer words wrong:", int(bad.sum()), "of", w.size)
if bad.any():
    i = np.argwhere(bad)[0]
    print("first", i, hex(w[tuple(i)]), hex(exp[tuple(i)]))
    # which bit positions are wrong overall
    x = w ^ exp
    print("wrong-bit histogram", [int(((x >> b) & 1).sum()) for b in range(16)])
