"""CPU, numpy/torch only: how much larger is the round-off of fp32 Winograd F(4x4,3x3) than F(2x2,3x3) at the backbone's channel counts?
Transforms in fp32 (as a kernel would compute them), U = G g G^T rounded once to fp32, the channel sum in fp32 (torch's CPU einsum), the
reference = float64 direct convolution.  Informs DESIGN.md section 10 (F(4x4) would cut the Winograd layers' MFMA work 1.78x)."""
import numpy as np
import torch

torch.manual_seed(0)
BT4 = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=torch.float64)
G4 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64)
AT4 = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)
BT2 = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G2 = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float64)
AT2 = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def wino(x, w, BT, G, AT, m):
    """x [N,C,H,W] fp32 (H, W multiples of m), w [K,C,3,3] fp32 -> y [N,K,H,W] fp32, pad 1"""
    a = m + 2
    N, C, H, W = x.shape
    K = w.shape[0]
    xp = torch.nn.functional.pad(x, (1, 1, 1, 1))
    tiles = xp.unfold(2, a, m).unfold(3, a, m)                       # [N,C,th,tw,a,a]
    BTf, ATf = BT.float(), AT.float()
    V = torch.einsum("ij,ncyxjk,lk->ncyxil", BTf, tiles, BTf)        # fp32 transform
    U = torch.einsum("ij,kcjl,ml->kcim", G, w.double(), G).float()    # host: fp64, rounded once
    M = torch.einsum("kcim,ncyxim->nkyxim", U, V)                     # fp32 channel sum
    Y = torch.einsum("ij,nkyxjl,ml->nkyxim", ATf, M, ATf)             # [N,K,th,tw,m,m]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(N, K, H, W)


for C in (64, 128, 256, 512):
    x = torch.relu(torch.randn(2, C, 24, 24))
    w = torch.randn(64, C, 3, 3) * (2.0 / (9 * C)) ** 0.5
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    scale = ref.abs().max().item()
    e2 = (wino(x, w, BT2, G2, AT2, 2).double() - ref).abs().max().item() / scale
    e4 = (wino(x, w, BT4, G4, AT4, 4).double() - ref).abs().max().item() / scale
    ed = (torch.nn.functional.conv2d(x, w, padding=1).double() - ref).abs().max().item() / scale
    print(f"C={C:4d}  max error / output scale:  direct fp32 {ed:.2e}   F(2x2,3x3) {e2:.2e}   F(4x4,3x3) {e4:.2e}   ratio {e4 / e2:.1f}")
