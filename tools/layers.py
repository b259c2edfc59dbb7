"""k_fwd_gates launch time of the merged forward wave vs number of generator layers (is a launch as long as its
busiest CU? 48 column blocks x 2 row blocks per 760-cell layer: 1 layer = 96 WGs, 2 = 192, 3 = 288, 4 = 384)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sys
from types import SimpleNamespace
import numpy as np
import torch
from rsrgan_amd import GAN_RNN

B, T = 64, 100
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.standard_normal((B, T, 257)).astype(np.float32)).cuda()
lab = torch.from_numpy(rng.standard_normal((B, T, 40)).astype(np.float32)).cuda()
ln = torch.full((B,), T, dtype=torch.int32).cuda()
for cells in (760,) if len(sys.argv) < 2 else [int(v) for v in sys.argv[1:]]:
    for L in (1, 2, 3, 4, 5):
        args = SimpleNamespace(batch_size=B, input_dim=257, output_dim=40, left_context=0, right_context=0, g_type="lstm",
                               keep_prob=1.0, batch_norm=False, num_gpu=1, save_dir=None, l2_scale=0.0, disc_updates=1, gen_updates=1,
                               init_mse_weight=10.0, init_disc_noise_std=0.0, d_learning_rate=1e-3, g_learning_rate=8e-5)
        m = GAN_RNN(None, args, ["gpu:0"], max_frames=T, net_overrides=dict(g_layers=L, g_cells=cells, flags=1))
        for _ in range(3):
            m.engine.d_backward(x, lab, ln, train=True, apply=False)
        m.engine.profile_begin()
        m.engine.d_backward(x, lab, ln, train=True, apply=False)
        n, us, fl = m.engine.profile_read()
        print("cells=%d layers=%d: %d gates launches, avg %.2f us (event-to-event), %.1f MFLOP/launch" % (cells, L, n, us / n, fl / n / 1e6), flush=True)
        del m
