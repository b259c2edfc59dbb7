import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sys, time
import torch
from rsrgan_amd.engine_hip import HipEngine
eng = HipEngine(batch_size=2, max_frames=4, input_dim=9, output_dim=5, g_layers=1, g_cells=8, g_proj=8, d_layers=1, d_cells=8, d_proj=4)
dev = eng.device
def run(M, N, K, akc, bkc, reps=20):
    A = torch.randn((M, K) if akc else (K, M), device=dev); B = torch.randn((N, K) if bkc else (K, N), device=dev)
    C = torch.empty(M, N, device=dev)
    for _ in range(3): eng.op_gemm(A, akc, B, bkc, C, M, N, K)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): eng.op_gemm(A, akc, B, bkc, C, M, N, K)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    Ar = A if akc else A.t(); Br = B.t() if bkc else B
    for _ in range(3): torch.matmul(Ar, Br)
    t0.record()
    for _ in range(reps): torch.matmul(Ar, Br)
    t1.record(); torch.cuda.synchronize()
    msr = t0.elapsed_time(t1) / reps
    print("M=%5d N=%5d K=%5d akc=%d bkc=%d : %.3f ms %.1f TF | rocBLAS/hipBLASLt (torch.matmul) %.3f ms %.1f TF" % (M, N, K, akc, bkc, ms, 2*M*N*K/ms/1e9, msr, 2*M*N*K/msr/1e9), flush=True)
for shape in [(4096, 4096, 4096), (6400, 3040, 280), (6400, 1024, 1024), (6400, 1024, 2828), (560, 3040, 6400), (1024, 1024, 6400)]:
    for akc, bkc in [(True, False), (True, True), (False, False)]:
        run(*shape, akc, bkc)
