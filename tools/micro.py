import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, sys
from rsrgan_amd import _lib
lib = _lib.load()
names = {0: "production (pipelined, 8 waves)", 1: "no A loads", 2: "no B loads", 3: "no loads at all", 4: "no MFMA (loads only)",
         7: "empty (no loads, no MFMA)", 8: "un-pipelined 8 waves", 16: "un-pipelined 16 waves", 32: "paired k-blocks (full 128B lines)", 36: "paired, loads only"}
for rnd in range(2):
    for v in (0, 32, 4, 36):
        us = C.c_float()
        rc = lib.rsrgan_microbench(3, v, 64, 760, 280, 280, 3, 200, C.byref(us))
        print("bwd_b G-wave (3 layers, N=64,H=760,I=P=280) variant %2d %-34s rc=%d  %.2f us/launch" % (v, names[v], rc, us.value), flush=True)
