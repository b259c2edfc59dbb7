"""Debug helper: R-CED weight gradients at the BASELINE configs[3] frame shape (W=257, splice 11) vs the fp64 oracle, per tensor,
with the error broken down by filter row / tap / channel.  Run on the MI355X box from the repo root."""
import sys
from types import SimpleNamespace

import numpy as np

sys.path.insert(0, ".")
from oracle import dnn_gan_oracle as DO
from oracle import rced_oracle as R
from tests.helpers import NET_G, rel_err, split_flat


def main(N=2, width=257, ctx=(5, 5)):
    from rsrgan_amd.trainer import DNNTrainer
    cfg = R.RcedCfg(input_dim=width, output_dim=5, left_context=ctx[0], right_context=ctx[1], d_units=18, d_hidden=2, filters_num=R.FILTERS_NUM)
    rng = np.random.default_rng(N)
    g = {k: v.astype(np.float32) for k, v in R.init_params(R.g_param_specs(cfg), rng).items()}
    for k in g:
        if k.endswith("biases"):
            g[k] = rng.normal(0.05, 0.1, g[k].shape).astype(np.float32)
    d = {k: (2.0 * v).astype(np.float32) for k, v in DO.init_params(DO.d_param_specs(cfg), rng, relu_init=True).items()}
    args = SimpleNamespace(batch_size=N, input_dim=cfg.input_dim, output_dim=cfg.output_dim, left_context=cfg.left_context,
                           right_context=cfg.right_context, g_type="rced", keep_prob=1.0, batch_norm=False, num_gpu=1, save_dir=None,
                           l2_scale=1e-3, g_learning_rate=1e-3, d_learning_rate=2e-3, init_mse_weight=10.0, disc_updates=1, gen_updates=1)
    ov = dict(g_layers=9, g_cells=32, d_layers=cfg.d_hidden, d_cells=cfg.d_units)
    m = DNNTrainer(None, args, ["gpu:0"], net_overrides=ov)
    o = R.GanRcedOracle(cfg, g, d, l2_scale=1e-3, g_learning_rate=float(np.float32(1e-3)), mse_lambda=1.0)
    o.supervised = True
    m.set_vars(g, d)
    x = rng.standard_normal((N, cfg.fed_dim)).astype(np.float32); lab = rng.standard_normal((N, cfg.output_dim)).astype(np.float32)
    print("forward max abs err", np.abs(m.forward(x) - o.forward(x)).max())
    got = m.engine.g_backward(x[:, None], lab[:, None], None, train=True, reuse=False, apply=False).cpu().numpy()
    want, wg, _ = o.g_tower(x, lab)
    print("losses", got, want)
    gr = split_flat(m.engine.get_grads(NET_G).cpu().numpy(), m.engine.tensor_table(NET_G))
    for k in wg:
        a = gr[k].reshape(wg[k].shape).astype(np.float64); b = wg[k]
        e = rel_err(a, b)
        print("%-28s %-18s rel_err %.3e" % (k, str(b.shape), e))
        if e > 2e-4 and b.ndim == 4:
            err = np.abs(a - b)
            print("   by dh :", np.array2string(err.max(axis=(1, 2, 3)), precision=2))
            print("   by dw :", np.array2string(err.max(axis=(0, 2, 3)), precision=2))
            print("   by c  :", np.array2string(err.max(axis=(0, 1, 3)), precision=2))
            print("   by co :", np.array2string(err.max(axis=(0, 1, 2)), precision=2))
            print("   |want| max", np.abs(b).max(), " worst idx", np.unravel_index(err.argmax(), err.shape), a.flat[err.argmax()], b.flat[err.argmax()])


if __name__ == "__main__":
    main(N=int(sys.argv[1]) if len(sys.argv) > 1 else 2)
