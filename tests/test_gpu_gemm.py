"""fp32 MFMA GEMM (rsrgan_op_gemm) vs a plain PyTorch fp32/fp64 reference of the same op."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _eng():
    from rsrgan_amd.engine_hip import HipEngine
    return HipEngine(batch_size=2, max_frames=4, input_dim=9, output_dim=5, g_layers=1, g_cells=8, g_proj=8,
                     d_layers=1, d_cells=8, d_proj=4)


def pad4(n):
    return (n + 3) // 4 * 4


@pytest.mark.parametrize("M,N,K", [(128, 128, 16), (300, 257, 40), (1600, 3040, 280), (37, 1, 40), (560, 3040, 640),
                                   (257, 280, 1003), (5, 7, 3),
                                   (1000, 12, 143), (512, 32, 2464), (2500, 24, 20000), (256, 1, 77),       # N <= 32: 256x32 tiles
                                   (8200, 2050, 72), (4100, 4100, 40)])      # >= 512 big tiles: k_gemm_s (128 x 256 / 256 x 256, four self-loading waves)
@pytest.mark.parametrize("akc,bkc", [(True, False), (True, True), (False, False), (False, True)])
def test_gemm_variants(M, N, K, akc, bkc):
    eng = _eng()
    dev = eng.device
    g = torch.Generator(device="cpu").manual_seed(M * 31 + N * 7 + K)
    A = torch.randn(M, K, generator=g, dtype=torch.float64)
    # asymmetric B so a transposed C-write cannot pass (guide rule 16)
    B = torch.randn(K, N, generator=g, dtype=torch.float64) + torch.arange(N, dtype=torch.float64)[None, :] * 0.01
    bias = torch.randn(N, generator=g, dtype=torch.float64)
    ref = A @ B + bias
    ref = torch.maximum(ref, 0.3 * ref)

    def store(mat, kcontig):           # returns padded device buffer + logical view
        rows, cols = (mat.shape if kcontig else mat.t().shape)
        buf = torch.zeros(rows, pad4(cols), dtype=torch.float32, device=dev)
        buf[:, :cols] = (mat if kcontig else mat.t()).to(torch.float32)
        return buf
    Ad = store(A, akc)                 # akc: [M][K] ; else [K][M]
    Bd = store(B.t().contiguous(), bkc) if bkc else store(B.t().contiguous(), False)   # bkc: [N][K]; else [K][N]
    Cd = torch.full((M, pad4(N)), 7.0, dtype=torch.float32, device=dev)
    eng.op_gemm(Ad, akc, Bd, bkc, Cd, M, N, K, bias=bias.to(torch.float32).to(dev), act=1, alpha=0.3)
    torch.cuda.synchronize()
    got = Cd[:, :N].double().cpu()
    err = (got - ref).abs().max().item() / max(ref.abs().max().item(), 1.0)
    assert err < 2e-5, err                      # fp32 inputs/accumulate vs fp64: ~1e-6 expected
    if pad4(N) != N:
        assert torch.all(Cd[:, N:] == 7.0)      # padding columns untouched
    # accumulate
    eng.op_gemm(Ad, akc, Bd, bkc, Cd, M, N, K, bias=None, act=0, accumulate=True)
    torch.cuda.synchronize()
    got2 = Cd[:, :N].double().cpu()
    err2 = (got2 - (ref + A @ B)).abs().max().item() / max(ref.abs().max().item(), 1.0)
    assert err2 < 4e-5, err2
