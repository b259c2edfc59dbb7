"""Grid-barrier micro-benchmark driver (rsrgan_microbench kind 4)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from rsrgan_amd import _lib
lib = _lib.load()
torch.zeros(1, device="cuda")
us = C.c_float()
A = 64 * 560            # floats of one layer's [x_t | m_{t-1}] for 64 rows
for nwg in (256, 512):
    for name, v, wr, rd in (("flat barrier", 0, 0, 0), ("2-level barrier", 1, 0, 0),
                            ("flat + write 560 + read 143KB", 2, 560, A), ("2lvl + write 560 + read 143KB", 3, 560, A),
                            ("2lvl + write 560 + read 36KB", 3, 560, A // 4)):
        rc = lib.rsrgan_microbench(4, v, nwg, 0, wr, rd, 1, 2000, C.byref(us))
        print("nwg=%d %-34s rc=%d  %.2f us / iteration" % (nwg, name, rc, us.value), flush=True)
        if rc:
            print(lib.rsrgan_last_error())

print("flag exchange in groups (kind 5): per-iteration time")
for nwg, gsz in ((16, 4), (16, 8), (64, 8), (64, 4)):
    for name, v in (("members on different XCDs", 0), ("members on one XCD (id % 8)", 1)):
        for wr in (640, 2560):
            if v == 1 and nwg < 8 * gsz:
                continue
            rc = lib.rsrgan_microbench(5, v, nwg, gsz, wr, 0, 1, 2000, C.byref(us))
            print("nwg=%d group=%d wr=%d floats %-30s rc=%d  %.2f us / iteration" % (nwg, gsz, wr, name, rc, us.value), flush=True)
            if rc:
                print(lib.rsrgan_last_error())
