"""Placement of the jobs of a launch (rsrgan_amd/csrc/kernels.h Place: XCD groups, folded discriminator forward) changes which
workgroup computes which tile and, for the fold, the association of one product -- never the arithmetic of a tile.  The step must
be bit-identical with and without XCD groups, and agree to fp32 rounding with and without the fold."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import hashlib, json, sys
import numpy as np
sys.path.insert(0, %r)
from oracle import rsrgan_oracle as O
from tests.helpers import NET_D, NET_G, build_hip_pair, rand_batch
import os
_net = os.environ.get("RSRGAN_TEST_NET")
cfg = (O.NetCfg.res_lstm_l() if _net == "res_lstm_l" else O.NetCfg.res_lstm_l(g_type="res_lstm_base") if _net == "res_lstm_base" else
       O.NetCfg(g_type="res_lstm_base", g_layers=2, g_cells=512, g_proj=0) if _net == "noproj" else O.NetCfg())      # the reference's sizes: G 3x760/p280 (or the shipped 4x760/p257 residual stack), D 2x256/p40
B, T = int(os.environ.get("RSRGAN_TEST_B", "8")), int(os.environ.get("RSRGAN_TEST_T", "7"))
model, _ = build_hip_pair(cfg, B, T, seed=5, flags=3)
x, lab, ln = rand_batch(cfg, B, T, seed=6, ragged=True)
out = {}
side = None
if os.environ.get("RSRGAN_TEST_SIDELOAD") == "1":
    # a second stream keeps the chip busy with foreign kernels for the whole test: a single-workgroup spin kernel that holds one CU
    # for ~40 ms at a time, and streaming adds over 1 GB that take every CU in turn (what an RCCL kernel on the communication stream or
    # another tenant of the GPU does to the persistent launches' "every workgroup resident" assumption)
    import threading, torch
    stop = threading.Event()
    def load():
        st = torch.cuda.Stream()
        a = torch.ones(1 << 28, device="cuda"); b = torch.ones(1 << 28, device="cuda")
        with torch.cuda.stream(st):
            while not stop.is_set():
                torch.cuda._sleep(100_000_000)
                for _ in range(8):
                    a.add_(b)
                st.synchronize()
    side = threading.Thread(target=load, daemon=True); side.start()
    import time; time.sleep(0.3)
_reuse = os.environ.get("RSRGAN_TEST_REUSE", "1") != "0"      # 0: the G-run recomputes the generator's forward (a second G-run of gen_updates = 2)
for it in range(2):
    d = np.ravel(model.d_step(x, lab, ln)); g = np.ravel(model.g_step(x, lab, ln, reuse_g_forward=_reuse))
    out["d%%d" %% it] = [float(v) for v in d]; out["g%%d" %% it] = [float(v) for v in g]
for i, Ti in enumerate(int(v) for v in os.environ.get("RSRGAN_TEST_TSEQ", "").split(",") if v):
    # batches of other lengths on the same handle (the outer loop's buckets): the ring positions of the persistent launches carry on
    # from wherever the previous launch stopped
    xi, labi, lni = rand_batch(cfg, B, Ti, seed=20 + i, ragged=True)
    d = np.ravel(model.d_step(xi, labi, lni)); g = np.ravel(model.g_step(xi, labi, lni, reuse_g_forward=True))
    out["sd%%d" %% i] = [float(v) for v in d]; out["sg%%d" %% i] = [float(v) for v in g]
_async = int(os.environ.get("RSRGAN_TEST_ASYNC", "0"))
if _async:
    # steps enqueued without waiting for their results (what bench.py and train_one_iteration do): the host runs ahead of the device
    import torch
    xs = [tuple(torch.from_numpy(v).cuda() for v in rand_batch(cfg, B, T, seed=40 + i, ragged=True)) for i in range(3)]
    if os.environ.get("RSRGAN_TEST_LEN64") == "1":
        # the common dtype of a device `lengths` tensor, and a strided label: their conversion to int32 / contiguous fp32 is a kernel
        xs = [(x_, torch.stack([l_, l_], -1)[..., 0], n_.to(torch.int64)) for (x_, l_, n_) in xs]
    torch.cuda.synchronize()
    with model.engine.on_stream():
        for i in range(_async):
            xi, labi, lni = xs[i %% 3]
            model.d_step(xi, labi, lni, sync=False, gather=False)
            last = model.g_step(xi, labi, lni, reuse_g_forward=True, sync=False, gather=False)
    torch.cuda.synchronize()
    out["async_last"] = [float(v) for v in np.ravel(last.cpu().numpy())]
model.engine.profile_begin()
model.d_step(x, lab, ln); model.g_step(x, lab, ln, reuse_g_forward=_reuse)
out["gp_n"] = int(model.engine.profile_read_kind(1)[0])           # k_glstm_fwd launches bracketed (k_glstm_fwd_dt is not)
out["gb_flops"] = float(model.engine.profile_read_kind(2)[2])      # algorithmic FLOP of the generator's BPTT launch (k_glstm_bwd_dt counts the discriminator half too)
model.engine.profile_read()
out["chain_launches"] = int(model.engine.profile_launches())
out["device_status"] = int(model.engine.device_status())
gv, dv = model.get_vars()
h = hashlib.sha256()
for k in sorted(gv): h.update(np.ascontiguousarray(gv[k]).tobytes())
for k in sorted(dv): h.update(np.ascontiguousarray(dv[k]).tobytes())
out["vars_sha"] = h.hexdigest()
out["g_norm"] = float(sum(float(np.square(v.astype(np.float64)).sum()) for v in gv.values()))
if side is not None:
    stop.set(); side.join(timeout=10)
print("RESULT " + json.dumps(out))
""" % ROOT


def _run(env):
    e = dict(os.environ)
    e.setdefault("RSRGAN_PAD_ROWS", "0")      # (these cases choose B to pick a path: no silent padding up to the persistent kernels' 32 rows)
    e["RSRGAN_DPIPE"] = "0"                   # (one switch at a time against the stream-ordered D-run; the pipelined D-run -- the Python layer's default, which an
                                              #  engine created earlier in this process has written into os.environ -- has its own cases, which set it)
    e.update(env)
    p = subprocess.run([sys.executable, "-c", WORKER], capture_output=True, text=True, env=e, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[7:])


@pytest.mark.parametrize("B,T", [(8, 7), (64, 100)])
def test_xcd_groups_do_not_change_a_bit(B, T):
    """(64, 100) is the benchmarked size: there the >= 12 %-share group rule and the split-K planner take the branches bench.py
    runs (three generator layers in XCD groups, phase B as split-K partials)."""
    size = {"RSRGAN_TEST_B": str(B), "RSRGAN_TEST_T": str(T)}
    a = _run(dict(size, RSRGAN_XCD_GROUPS="1"))
    b = _run(dict(size, RSRGAN_XCD_GROUPS="0"))
    assert a == b, (a, b)


def test_split_k_phase_b_groups_do_not_change_a_bit():
    """RSRGAN_BP_GROUPS=0 keeps every split-K phase-B job on all 8 XCD slots: same tiles, same summation order."""
    size = {"RSRGAN_TEST_B": "64", "RSRGAN_TEST_T": "12"}
    a = _run(dict(size, RSRGAN_BP_GROUPS="1"))
    b = _run(dict(size, RSRGAN_BP_GROUPS="0"))
    assert a == b, (a, b)


def test_folded_discriminator_forward_agrees():
    a = _run({"RSRGAN_DFOLD": "1"})
    b = _run({"RSRGAN_DFOLD": "0"})
    for k in ("d0", "g0", "d1", "g1"):
        assert np.allclose(a[k], b[k], rtol=2e-5, atol=1e-7), (k, a[k], b[k])
    assert abs(a["g_norm"] - b["g_norm"]) <= 1e-5 * b["g_norm"]


@pytest.mark.parametrize("B,T,mode", [(16, 9, 1), (16, 9, 2), (16, 9, 3), (64, 100, 3),
                                      (16, 1, 3), (32, 2, 3), (48, 3, 3)])      # one / two steps; 3 row tiles (clusters across XCDs)
def test_persistent_discriminator_recurrence_agrees(B, T, mode):
    """csrc/dpersist.hip: the discriminator's recurrences that run alone -- D(G(x)) of the G-run (mode bit 0) and the D-run's BPTT
    (bit 1) -- as ONE persistent launch each (partial products exchanged as generation-tagged granules) against the per-step
    launches.  Same products, the projection / state gradient summed per cell quarter: fp32 rounding apart.  The launch count proves
    which path ran: a chain of >= T per-step launches becomes one."""
    size = {"RSRGAN_TEST_B": str(B), "RSRGAN_TEST_T": str(T)}
    a = _run(dict(size, RSRGAN_DPERSIST=str(mode)))
    b = _run(dict(size, RSRGAN_DPERSIST="0"))
    if T > 2:
        assert b["chain_launches"] - a["chain_launches"] >= (T - 1) * (1 if mode < 3 else 2), (a["chain_launches"], b["chain_launches"])
    assert a["device_status"] == 0 and b["device_status"] == 0
    for k in ("d0", "g0", "d1", "g1"):
        assert np.allclose(a[k], b[k], rtol=2e-5, atol=1e-7), (k, a[k], b[k])
    assert abs(a["g_norm"] - b["g_norm"]) <= 1e-5 * b["g_norm"]
    c = _run(dict(size, RSRGAN_DPERSIST=str(mode)))
    assert a["vars_sha"] == c["vars_sha"]          # fixed summation order: reproducible bits


@pytest.mark.parametrize("B,T,mode", [(32, 9, 1), (64, 100, 1), (32, 9, 3), (64, 100, 3), (32, 1, 3), (32, 2, 3), (64, 7, 2)])
def test_persistent_generator_recurrence_agrees(B, T, mode):
    """csrc/gpersist.hip: the generator's forward recurrence (models/lstm.py:89-112; mode bit 0) and its BPTT (bit 1) as ONE persistent
    launch each -- weights resident, partial projections / partial state and input gradients reduce-scattered and the state (its
    gradient) all-gathered as sentinel-armed 16-byte pieces -- against the launch-per-phase wavefront.  Same products; the projection
    is summed per slice of 20 cells and the gates per k-block group: fp32 rounding apart.  Ragged lengths exercise dynamic_rnn's
    masking in the consumers; T = 1, 2, 7 the ring start-up, wrap-around (rings of 3 and 6 steps) and the re-arming epilogue (the
    second update of each net runs on re-armed slots).  The launch count proves which path ran."""
    size = {"RSRGAN_TEST_B": str(B), "RSRGAN_TEST_T": str(T)}
    a = _run(dict(size, RSRGAN_GPERSIST=str(mode)))
    b = _run(dict(size, RSRGAN_GPERSIST="0"))
    if T > 2:
        assert b["chain_launches"] - a["chain_launches"] >= (T - 1) * (1 if mode < 3 else 2), (a["chain_launches"], b["chain_launches"])
    assert a["device_status"] == 0 and b["device_status"] == 0
    for k in ("d0", "g0", "d1", "g1"):
        assert np.allclose(a[k], b[k], rtol=5e-5, atol=1e-7), (k, a[k], b[k])
    assert abs(a["g_norm"] - b["g_norm"]) <= 1e-5 * b["g_norm"]
    c = _run(dict(size, RSRGAN_GPERSIST=str(mode)))
    assert a["vars_sha"] == c["vars_sha"]          # fixed summation order: reproducible bits


@pytest.mark.parametrize("B,T,net", [(32, 9, "lstm"), (64, 100, "lstm"), (32, 1, "lstm"), (32, 2, "lstm"), (32, 9, "res_lstm_l"), (8, 7, "res_lstm_l")])
def test_trailing_discriminator_bptt_agrees(B, T, net):
    """Round 5: in the G-run (gan_rnn_placeholder.py:246-256: g_adv differentiated through D into G) the discriminator's BPTT runs in
    its trailing form (csrc/dpersist.hip k_dlstm_bwd_trail: two row tiles per workgroup, FC workgroups that turn layer 0's input
    gradient into dy(t) and d(outputs)(t) = dy(t) . W_out^T step by step) BESIDE the generator's BPTT, whose top layer polls that
    gradient -- against the two launches one after the other with the two GEMMs between them (RSRGAN_TRAIL=0).  Same products, the
    input gradient summed per cell quarter instead of per k-block of a GEMM: fp32 rounding apart; reproducible; no failed wait."""
    size = {"RSRGAN_TEST_B": str(B), "RSRGAN_TEST_T": str(T), "RSRGAN_TEST_NET": net, "RSRGAN_PAD_ROWS": "1"}
    a = _run(dict(size))
    b = _run(dict(size, RSRGAN_TRAIL="0"))
    assert a["device_status"] == 0 and b["device_status"] == 0
    assert a["gb_flops"] > b["gb_flops"] > 0, (a["gb_flops"], b["gb_flops"])      # the one launch carried the discriminator's products: that path ran
    for k in ("d0", "g0", "d1", "g1"):
        assert np.allclose(a[k], b[k], rtol=5e-5, atol=1e-7), (k, a[k], b[k])
    assert abs(a["g_norm"] - b["g_norm"]) <= 1e-5 * b["g_norm"]
    c = _run(dict(size))
    assert a["vars_sha"] == c["vars_sha"]


@pytest.mark.parametrize("B,T,net", [(32, 9, "lstm"), (64, 100, "lstm"), (32, 1, "lstm"), (32, 2, "lstm"), (32, 9, "res_lstm_l"), (8, 7, "res_lstm_l")])
def test_trailing_discriminator_forward_agrees(B, T, net):
    """Round 5: a G-run that recomputes the generator's forward (the second G-run of the shipped gen_updates = 2,
    run_gan_rnn_placeholder.sh:130) runs D(G(x)) INSIDE the generator's forward launch (csrc/gpersist.hip k_glstm_fwd_dt: FC workgroups turn
    the top layer's chunks into y(t) = m(t) . W_out + b step by step, the discriminator's recurrence -- two row tiles per workgroup,
    csrc/dpersist_dev.h dp_fwdt_body -- follows a few steps behind) against generator launch, output-FC GEMM, noise kernel and
    discriminator launch one after the other (RSRGAN_TRAIL_FWD=0).  The output FC is summed per quarter of its reduction: fp32 rounding
    apart; reproducible; no failed wait; the stand-alone generator launch is gone from the G-run."""
    size = {"RSRGAN_TEST_B": str(B), "RSRGAN_TEST_T": str(T), "RSRGAN_TEST_NET": net, "RSRGAN_PAD_ROWS": "1", "RSRGAN_TEST_REUSE": "0"}
    a = _run(dict(size))
    b = _run(dict(size, RSRGAN_TRAIL_FWD="0"))
    assert a["device_status"] == 0 and b["device_status"] == 0
    assert a["gp_n"] == b["gp_n"] - 1, (a["gp_n"], b["gp_n"])
    for k in ("d0", "g0", "d1", "g1"):
        assert np.allclose(a[k], b[k], rtol=5e-5, atol=1e-7), (k, a[k], b[k])
    assert abs(a["g_norm"] - b["g_norm"]) <= 1e-5 * b["g_norm"]
    c = _run(dict(size))
    assert a["vars_sha"] == c["vars_sha"]


@pytest.mark.parametrize("B,T,net", [(64, 100, "lstm"), (32, 9, "lstm"), (32, 1, "lstm"), (32, 9, "res_lstm_l")])
def test_pipelined_discriminator_run_agrees(B, T, net):
    """Round 5, RSRGAN_DPIPE=1 (the caller guarantees that labels and lengths of rsrgan_d_step are complete when the call is made): the
    D-run's D(real) (gan_rnn_placeholder.py:207: it depends on nothing of the generator) is staged and run on the side stream as soon as
    the previous run no longer needs the discriminator's stash -- beside the previous G-run's weight-gradient GEMMs when the host runs
    ahead -- over rows [0, B) of the stacked stash, and the D-run itself is k_glstm_fwd_dt with D(G(x)) trailing over rows [B, 2B), then the
    stacked BPTT.  24 steps enqueued without a host wait, three batches in turn: against RSRGAN_DPIPE=0 (fp32 rounding of the
    output FC apart), the same bits when the whole sequence runs again, no failed wait."""
    size = {"RSRGAN_TEST_B": str(B), "RSRGAN_TEST_T": str(T), "RSRGAN_TEST_NET": net, "RSRGAN_TEST_ASYNC": "24"}
    a = _run(dict(size, RSRGAN_DPIPE="1"))
    b = _run(dict(size, RSRGAN_DPIPE="0"))
    assert a["device_status"] == 0 and b["device_status"] == 0
    assert a["gp_n"] == b["gp_n"] - 1, (a["gp_n"], b["gp_n"])      # the D-run's stand-alone generator launch is gone: k_glstm_fwd_dt ran
    for k in ("d0", "g0", "d1", "g1", "async_last"):
        assert np.allclose(a[k], b[k], rtol=2e-4, atol=1e-6), (k, a[k], b[k])
    assert abs(a["g_norm"] - b["g_norm"]) <= 1e-5 * b["g_norm"]
    c = _run(dict(size, RSRGAN_DPIPE="1"))
    assert a["vars_sha"] == c["vars_sha"]


def test_pipelined_discriminator_run_converts_device_tensors_off_the_compute_stream():
    """Round 6 (advisor, engine_hip.upload_ready): under RSRGAN_DPIPE=1 the library reads labels and lengths on its side stream AHEAD of
    the caller's stream.  Device tensors that need a conversion (int64 lengths -- the common dtype --, a strided label) used to be
    converted by a kernel queued on the compute stream behind its backlog, so D(real) could read the conversion's output before it was
    written.  24 steps enqueued without a host wait over three ragged batches in turn, int64 lengths and strided labels: the same
    losses as with int32 / contiguous tensors (bit for bit: the conversion changes no value) and as RSRGAN_DPIPE=0."""
    size = {"RSRGAN_TEST_B": "32", "RSRGAN_TEST_T": "9", "RSRGAN_TEST_ASYNC": "24"}
    a = _run(dict(size, RSRGAN_DPIPE="1", RSRGAN_TEST_LEN64="1"))
    b = _run(dict(size, RSRGAN_DPIPE="1"))
    c = _run(dict(size, RSRGAN_DPIPE="0", RSRGAN_TEST_LEN64="1"))
    assert a["device_status"] == 0
    assert a["async_last"] == b["async_last"] and a["vars_sha"] == b["vars_sha"], (a["async_last"], b["async_last"])
    assert np.allclose(a["async_last"], c["async_last"], rtol=2e-4, atol=1e-6)


@pytest.mark.parametrize("B,T,net", [(64, 100, "lstm"), (32, 9, "lstm"), (32, 1, "lstm"), (32, 2, "lstm"), (8, 7, "res_lstm_l")])
def test_discriminator_weight_gradients_inside_the_bptt_launch_agree(B, T, net):
    """Round 6: the D-run's weight gradients (discriminator_lstm.py:70-104; gan_rnn_placeholder.py:144,177-183) are accumulated by
    workgroups that TRAIL the recurrence inside k_dlstm_bwd (csrc/dpersist.hip dp_dw_body: dz / dm read write-through behind progress
    words, one record of partial sums per 16-row tile, k_dw_reduce adds the tiles in a fixed order and leaves the clip's sums of
    squares) -- against the split-K GEMM / column-sum launches behind the recurrence (RSRGAN_DW_INKERNEL=0).  Same products summed in
    another order: fp32 rounding apart; T = 1, 2, 3 the drained-launch progress word and the two-step lag; reproducible; no failed wait."""
    size = {"RSRGAN_TEST_B": str(B), "RSRGAN_TEST_T": str(T), "RSRGAN_TEST_NET": net, "RSRGAN_PAD_ROWS": "1", "RSRGAN_TEST_TSEQ": "3,1,2" if T == 9 else ""}
    a = _run(dict(size))
    b = _run(dict(size, RSRGAN_DW_INKERNEL="0"))
    assert a["device_status"] == 0 and b["device_status"] == 0
    keys = ["d0", "g0", "d1", "g1"] + [k for k in a if k[:2] in ("sd", "sg")]
    for k in keys:
        assert np.allclose(a[k], b[k], rtol=5e-5, atol=1e-7), (k, a[k], b[k])
    assert abs(a["g_norm"] - b["g_norm"]) <= 1e-5 * b["g_norm"]
    c = _run(dict(size))
    assert a["vars_sha"] == c["vars_sha"]


@pytest.mark.parametrize("net", ["lstm", "res_lstm_l"])
def test_tagged_ring_slots_across_batches_of_different_lengths(net):
    """Round 5: the hop-1 / input-gradient ring slots of k_glstm_fwd / k_glstm_bwd carry the parity of the ring pass in every word's
    lowest mantissa bit (csrc/gpersist.hip gp_store_t) instead of being re-armed with sentinels; the ring position is a counter in the
    control block that runs on from launch to launch (T steps forward, T - 1 / T backward).  One handle, batches of 9, 4, 7, 1, 2, 5,
    9 frames in turn (every residue of the ring depths 3 and 6, a one-step launch that writes nothing to the state-gradient ring):
    against the sentinel form (RSRGAN_GP_TAGS=0; the values differ by the 23rd mantissa bit of the partial sums) and against the
    launch path, no failed wait, and the same bits when the whole sequence is run again in a fresh process."""
    size = {"RSRGAN_TEST_B": "32", "RSRGAN_TEST_T": "9", "RSRGAN_TEST_TSEQ": "4,7,1,2,5,9,3,9", "RSRGAN_TEST_NET": net}
    a = _run(dict(size))
    b = _run(dict(size, RSRGAN_GP_TAGS="0"))
    c = _run(dict(size, RSRGAN_GPERSIST="0"))
    assert a["device_status"] == 0 and b["device_status"] == 0 and c["device_status"] == 0
    assert c["chain_launches"] - a["chain_launches"] >= 2 * 8
    keys = ["d0", "g0", "d1", "g1"] + ["s%s%d" % (n, i) for i in range(8) for n in "dg"]
    for k in keys:
        assert np.allclose(a[k], b[k], rtol=5e-5, atol=1e-7), (k, a[k], b[k])
        assert np.allclose(a[k], c[k], rtol=5e-5, atol=1e-7), (k, a[k], c[k])
    assert abs(a["g_norm"] - c["g_norm"]) <= 1e-5 * c["g_norm"]
    d = _run(dict(size))
    assert a["vars_sha"] == d["vars_sha"]


@pytest.mark.parametrize("B,T,net,reuse", [(8, 100, "res_lstm_l", "0"), (8, 9, "res_lstm_l", "1"), (1, 30, "lstm", "1"), (16, 7, "lstm", "0"), (8, 1, "lstm", "1")])
def test_single_tile_lane_for_padding_rows_changes_no_bit(B, T, net, reuse):
    """Round 5: a padded model whose real rows fit ONE 16-row tile (the shipped batch_size = 8, run_gan_rnn_placeholder.sh:126; decode's
    single utterance) runs one tile lane of the persistent generator launches (GPersistArgs::nrt = 1: the second tile of the row group
    holds rows of length 0 -- dynamic_rnn's masking makes them inert, everything they own in the stashes is zero since the allocation
    -- and neither computes nor publishes nor gathers; the FC workgroups of k_glstm_fwd_dt take zeros for its chunks) -- against both
    lanes running (RSRGAN_GP_NRT=0).  The live tile's arithmetic and summation order are the same: the same bits, D-run, G-run (reusing
    and recomputing the forward: k_glstm_fwd_dt) and the updated variables; no failed wait."""
    size = {"RSRGAN_TEST_B": str(B), "RSRGAN_TEST_T": str(T), "RSRGAN_TEST_NET": net, "RSRGAN_PAD_ROWS": "1", "RSRGAN_TEST_REUSE": reuse}
    a = _run(dict(size))
    b = _run(dict(size, RSRGAN_GP_NRT="0"))
    c = _run(dict(size, RSRGAN_DP_NRT="0"))      # (the discriminator's halves of the fused launches drop the tile too: DPersistArgs::nrt)
    assert a["device_status"] == 0 and b["device_status"] == 0 and c["device_status"] == 0
    assert a["vars_sha"] == c["vars_sha"]
    assert a["gb_flops"] == b["gb_flops"] > 0, (a["gb_flops"], b["gb_flops"])      # the persistent launches ran
    for k in ("d0", "g0", "d1", "g1"):
        assert a[k] == b[k], (k, a[k], b[k])
    assert a["vars_sha"] == b["vars_sha"]


@pytest.mark.parametrize("B,T", [(32, 9), (32, 50), (32, 1), (32, 2), (8, 7), (64, 9), (64, 2)])
def test_persistent_residual_generator_agrees(B, T):
    """Round 5: res_lstm_l (the g_type run_gan_rnn_placeholder.sh:124 ships; models/res_lstm_l.py:101-194: four LSTMCell(760, num_proj=257)
    with inputs_{l+1} = outputs_l + inputs_l) on the persistent launches: the running sum and its gradient travel from reducer to
    reducer inside k_glstm_fwd / k_glstm_bwd (csrc/gpersist.hip RES) -- against the launch-per-phase wavefront (RSRGAN_GP_RES=0).  P = 257
    is no multiple of 4: the 16-byte pieces straddle the last column.  (8, 7): padded to one 32-row group.  (64, .): two row groups =
    304 workgroups, more than the device holds at once: ONE LAUNCH PER ROW GROUP (GPersistArgs::ngl), the discriminator's half over all
    rows riding the last forward / the first backward launch, the ring counters of the two groups advancing on their own."""
    size = {"RSRGAN_TEST_B": str(B), "RSRGAN_TEST_T": str(T), "RSRGAN_TEST_NET": "res_lstm_l", "RSRGAN_PAD_ROWS": "1"}
    a = _run(dict(size))
    b = _run(dict(size, RSRGAN_GP_RES="0", RSRGAN_PAD_ROWS="0"))
    if T > 2:
        assert b["chain_launches"] - a["chain_launches"] >= 2 * (T - 1), (a["chain_launches"], b["chain_launches"])
    assert a["device_status"] == 0 and b["device_status"] == 0
    for k in ("d0", "g0", "d1", "g1"):
        assert np.allclose(a[k], b[k], rtol=5e-5, atol=1e-7), (k, a[k], b[k])
    assert abs(a["g_norm"] - b["g_norm"]) <= 1e-5 * b["g_norm"]
    c = _run(dict(size))
    assert a["vars_sha"] == c["vars_sha"]          # fixed summation order: reproducible bits


def test_persistent_base_generator_agrees():
    """res_lstm_base (models/res_lstm_base.py: the same four projected cells without the residual sums, fed the input frames
    directly): the persistent launches' plain form with a 257-wide layer 0 input read from memory, against the launch path."""
    size = {"RSRGAN_TEST_B": "32", "RSRGAN_TEST_T": "9", "RSRGAN_TEST_NET": "res_lstm_base"}
    a = _run(dict(size))
    b = _run(dict(size, RSRGAN_GPERSIST="0"))
    assert b["chain_launches"] - a["chain_launches"] >= 2 * 8, (a["chain_launches"], b["chain_launches"])
    assert a["device_status"] == 0 and b["device_status"] == 0
    for k in ("d0", "g0", "d1", "g1"):
        assert np.allclose(a[k], b[k], rtol=5e-5, atol=1e-7), (k, a[k], b[k])
    assert abs(a["g_norm"] - b["g_norm"]) <= 1e-5 * b["g_norm"]


@pytest.mark.parametrize("B,T", [(32, 9), (64, 50), (32, 1), (32, 2)])
def test_persistent_unprojected_generator_forward_agrees(B, T):
    """Round 5: num_proj=None cells (BASELINE.json's "2-layer 512-unit LSTM generator": the state is h itself) -- the forward recurrence
    as ONE persistent launch with a single hand-off per step, the all-gather of h (csrc/gpersist.hip k_glstm_np_fwd), against the
    launch-per-phase wavefront (RSRGAN_GP_NOPROJ=0).  The backward pass is the launch path in both runs."""
    size = {"RSRGAN_TEST_B": str(B), "RSRGAN_TEST_T": str(T), "RSRGAN_TEST_NET": "noproj"}
    a = _run(dict(size))
    b = _run(dict(size, RSRGAN_GP_NOPROJ="0"))
    if T > 2:
        assert b["chain_launches"] - a["chain_launches"] >= T - 1, (a["chain_launches"], b["chain_launches"])
    assert a["device_status"] == 0 and b["device_status"] == 0
    for k in ("d0", "g0", "d1", "g1"):
        assert np.allclose(a[k], b[k], rtol=5e-5, atol=1e-7), (k, a[k], b[k])
    assert abs(a["g_norm"] - b["g_norm"]) <= 1e-5 * b["g_norm"]
    c = _run(dict(size))
    assert a["vars_sha"] == c["vars_sha"]


def test_batched_kernel_gradient_gemm_agrees():
    """Round 5: the three generator layers' kernel gradients [x | m]^T dZ (560 x 3040 x 6400 each) as ONE stream-K launch of k_gemm
    (csrc/gemm.hip launch_gemm_batch: the unit space runs over all problems' tiles) against one launch per layer: the same products,
    cut at other k positions -- fp32 rounding apart, and reproducible."""
    size = {"RSRGAN_TEST_B": "64", "RSRGAN_TEST_T": "100"}
    a = _run(dict(size))
    b = _run(dict(size, RSRGAN_GEMM_BATCH="0"))
    assert a["device_status"] == 0 and b["device_status"] == 0
    for k in ("d0", "g0", "d1", "g1"):
        assert np.allclose(a[k], b[k], rtol=5e-5, atol=1e-7), (k, a[k], b[k])
    assert abs(a["g_norm"] - b["g_norm"]) <= 1e-5 * b["g_norm"]
    c = _run(dict(size))
    assert a["vars_sha"] == c["vars_sha"]


@pytest.mark.parametrize("switch", ["RSRGAN_WGRAD_BATCH", "RSRGAN_DHEAD", "RSRGAN_FC_SIDE", "RSRGAN_WGRAD_STREAMS", "RSRGAN_LAZY_SWIZZLE",
                                    "RSRGAN_FUSED_SEG"])
def test_round4_launch_fusions_agree(switch):
    """Round 4 replaced launch sequences around the persistent recurrences by fewer launches: the weight gradients of same-shaped layers
    as one launch per kind (k_gemm16_b, k_lstm_colsums*_b), discriminator_lstm's head as one pass (k_dhead1/2), the FCs' parameter
    gradients and the layers' dWp / column sums on a side stream beside the dK GEMMs, weight copies rebuilt where they are read, the
    G-run as one graph segment.  Each has a switch that restores the launch sequence it replaced: same arithmetic, other summation
    orders at most -- three updates of each net must agree to fp32 rounding."""
    size = {"RSRGAN_TEST_B": "32", "RSRGAN_TEST_T": "9"}
    a = _run(dict(size))
    b = _run(dict(size, **{switch: "1" if switch == "RSRGAN_WGRAD_STREAMS" else "0"}))
    assert a["device_status"] == 0 and b["device_status"] == 0
    for k in ("d0", "g0", "d1", "g1"):
        assert np.allclose(a[k], b[k], rtol=5e-5, atol=1e-7), (k, a[k], b[k])
    assert abs(a["g_norm"] - b["g_norm"]) <= 1e-5 * b["g_norm"]


def test_persistent_launches_survive_a_busy_side_stream():
    """The persistent recurrences (csrc/gpersist.hip, dpersist.hip) wait for each other inside a launch, so every workgroup must become
    resident -- without a cooperative launch.  With foreign kernels on another stream holding CUs (a spin kernel, streaming adds:
    the N > 1 schedule's RCCL kernels on the communication stream, or another tenant) the workgroups arrive late but arrive: no
    bounded wait may expire (rsrgan_device_status == 0) and the step must produce the same bits as on an idle chip."""
    size = {"RSRGAN_TEST_B": "64", "RSRGAN_TEST_T": "40"}
    a = _run(dict(size, RSRGAN_TEST_SIDELOAD="1"))
    b = _run(dict(size, RSRGAN_TEST_SIDELOAD="0"))
    assert a["device_status"] == 0 and b["device_status"] == 0
    for k in ("d0", "g0", "d1", "g1"):
        assert a[k] == b[k], (k, a[k], b[k])
    assert a["vars_sha"] == b["vars_sha"]


def test_stream_k_gemm_step_is_reproducible():
    """Every time-batched product runs on the hand-written stream-K k_gemm (csrc/gemm.hip; no vendor library since round 3): the
    pieces of a cut tile are summed in k order by k_gemm_fixup, no float atomics, so two processes produce the same bits."""
    size = {"RSRGAN_TEST_B": "16", "RSRGAN_TEST_T": "16"}
    a = _run(dict(size))
    b = _run(dict(size))
    assert a == b, (a, b)


SEGAN_WORKER = r"""
import json, sys
import numpy as np
sys.path.insert(0, %r)
from tests.test_gpu_segan import _pair, _batch, NET_D, NET_G
cfg, m, o, rng = _pair(2, 300, 40, (16, 32, 32, 64, 64, 128), 20, 31, seed=11, l1=7.0)
x, lab, z, nz = _batch(cfg, 2, rng)
out = {"d": [float(v) for v in np.ravel(m.d_step(x, lab, z, nz, apply=False))]}
gd = m.get_grads(NET_D)
out["g"] = [float(v) for v in np.ravel(m.g_step(x, lab, z, (nz[0], nz[2]), apply=False))]
gg = m.get_grads(NET_G)
out["gd"] = {k: [float(np.linalg.norm(v)), float(np.sum(v.astype(np.float64)))] for k, v in gd.items()}
out["gg"] = {k: [float(np.linalg.norm(v)), float(np.sum(v.astype(np.float64)))] for k, v in gg.items()}
print("RESULT " + json.dumps(out))
""" % ROOT


def _run_worker(src, env):
    e = dict(os.environ); e.update(env)
    p = subprocess.run([sys.executable, "-c", src], capture_output=True, text=True, env=e, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])


def test_segan_column_reduction_forms_agree():
    """csrc/segan.hip launch_colred: the 16-byte / four-rows-in-flight form of each reduction mode (RSRGAN_COLRED_VEC bit m) against
    the scalar form: same sums to fp32 rounding, so the towers and every gradient's norm agree closely."""
    a = _run_worker(SEGAN_WORKER, {"RSRGAN_COLRED_VEC": "15"})
    b = _run_worker(SEGAN_WORKER, {"RSRGAN_COLRED_VEC": "0"})
    assert np.allclose(a["d"], b["d"], rtol=1e-5) and np.allclose(a["g"], b["g"], rtol=1e-5), (a["d"], b["d"], a["g"], b["g"])
    for net in ("gd", "gg"):
        for k in a[net]:
            na, nb = a[net][k][0], b[net][k][0]
            assert abs(na - nb) <= 1e-4 * nb + 1e-4, (net, k, na, nb)      # (biases in front of a VBN: zero gradient, fp32 noise)


RCED_WORKER = r"""
import json, sys
import numpy as np
sys.path.insert(0, %r)
from tests.test_gpu_trainers import rced_gradient_norms
print("RESULT " + json.dumps(rced_gradient_norms()))
""" % ROOT


def test_rced_weight_gradient_geometries_agree():
    """csrc/conv.hip k_conv_wgrad: six filter rows per workgroup with the rows inside the position loop (RSRGAN_WGRAD_DH=6, the
    default for multi-strip frames) against round 2's three rows: the same products in another summation order."""
    a = _run_worker(RCED_WORKER, {"RSRGAN_WGRAD_DH": "6"})
    b = _run_worker(RCED_WORKER, {"RSRGAN_WGRAD_DH": "3"})
    assert a.keys() == b.keys() and len(a) > 0
    for k in a:
        assert abs(a[k] - b[k]) <= 2e-5 * max(abs(b[k]), 1e-6), (k, a[k], b[k])


def test_rced_convolution_mfma_forms_agree():
    """csrc/conv.hip: the 4x4x1 forms (k_conv_fwd4, k_conv_wgrad4) on every layer (RSRGAN_CONV4=2), on the widths that are no multiple of 16 (the
    default) and nowhere (0): the same products in another summation order.  (Each form is also compared with the oracle:
    tests/test_gpu_trainers.py runs under the default, and its R-CED cases pass under RSRGAN_CONV4=2 as well.)"""
    a = _run_worker(RCED_WORKER, {"RSRGAN_CONV4": "2"})
    b = _run_worker(RCED_WORKER, {"RSRGAN_CONV4": "0"})
    c = _run_worker(RCED_WORKER, {})
    d = _run_worker(RCED_WORKER, {"RSRGAN_CONV4": "2", "RSRGAN_CONV4_KS": "4"})      # two group sets x k' quarters
    e = _run_worker(RCED_WORKER, {"RSRGAN_CONV4": "1", "RSRGAN_WGRAD4": "0"})          # the weight gradient alone on 16x16x4
    e1 = _run_worker(RCED_WORKER, {"RSRGAN_WGRAD4": "1"})                               # ... only its 12/20/24-channel layers on 4x4x1
    for k in b:
        assert abs(e1[k] - b[k]) <= 2e-5 * max(abs(b[k]), 1e-6), (k, e1[k], b[k])
    f = _run_worker(RCED_WORKER, {"RSRGAN_CONV_ROWS": "0"})                            # even strips (6 x 43) instead of 4 x 64 + 1 columns
    for k in b:
        assert abs(f[k] - b[k]) <= 2e-5 * max(abs(b[k]), 1e-6), (k, f[k], b[k])
    for k in b:
        assert abs(e[k] - b[k]) <= 2e-5 * max(abs(b[k]), 1e-6), (k, e[k], b[k])
    assert a.keys() == b.keys() == c.keys() == d.keys() and len(a) > 0
    for k in a:
        for other in (a, c, d):
            assert abs(other[k] - b[k]) <= 2e-5 * max(abs(b[k]), 1e-6), (k, other[k], b[k])


def test_wide_output_small_generator_with_reference_discriminator():
    """A generator narrower than its 40-dim output next to the reference's discriminator (2 x LSTMCell(256, num_proj=40)): dy
    [T*B][40] used to overflow the generator's gradient ping-pong buffers (sized by its layer widths), found by
    __graft_entry__.smoke().  Also the persistent discriminator recurrences against the ORACLE (not only against the per-step
    launches): losses of two D + G updates."""
    from oracle import rsrgan_oracle as O                 # noqa: F401
    from tests.helpers import build_hip_pair, rand_batch, small_cfg
    small = small_cfg("lstm")
    m0, _ = build_hip_pair(small, 4, 6, seed=1)           # (another model first: the overflow only faulted with a shifted heap)
    x0, l0, n0 = rand_batch(small, 4, 6, seed=2, ragged=True)
    m0.d_step(x0, l0, n0)
    cfg = small_cfg("lstm", output_dim=40, d_cells=256, d_proj=40)
    B, T = 16, 5
    model, oracle = build_hip_pair(cfg, B, T, seed=3, flags=3)
    x, lab, ln = rand_batch(cfg, B, T, seed=4, ragged=True)
    for _ in range(2):
        d_hip = np.ravel(model.d_step(x, lab, ln)); d_ref = np.ravel(oracle.d_step(x, lab, ln))
        g_hip = np.ravel(model.g_step(x, lab, ln, reuse_g_forward=True)); g_ref = np.ravel(oracle.g_step(x, lab, ln))
        assert np.allclose(d_hip, d_ref, rtol=1e-3), (d_hip, d_ref)
        assert np.allclose(g_hip, g_ref, rtol=1e-3), (g_hip, g_ref)
    assert model.engine.device_status() == 0


def test_rced_batch_norm_narrow_kernels_match_oracle():
    """csrc/bn.hip k_bn_part_narrow / k_bn_elem_narrow (the form the R-CED feature maps take from 4096 positions on): the oracle
    parity cases of tests/test_gpu_trainers.py::test_rced_batch_norm_matches_oracle are a few hundred positions, so they run here
    once more in a process where every eligible call takes the narrow form (RSRGAN_BN_NARROW = least row count)."""
    e = dict(os.environ); e["RSRGAN_BN_NARROW"] = "1"
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_trainers.py"), "-q", "-x", "-m", "gpu", "-k",
                        "rced_batch_norm_matches_oracle", "-p", "no:cacheprovider"], capture_output=True, text=True, env=e, cwd=ROOT, timeout=900)
    assert p.returncode == 0, (p.stdout[-3000:], p.stderr[-1000:])
    assert "4 passed" in p.stdout, p.stdout[-500:]
