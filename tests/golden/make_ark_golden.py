#!/usr/bin/env python
"""Generates tests/golden/ark_golden.npz: Kaldi archives (float, double, compressed) together with
the matrices the REFERENCE's own reader (io_funcs/kaldi_io.py:ArkReader, importable under Python 3)
decodes from them.  Runs only in the build container (needs /root/reference); the tests read only
the committed .npz.  The archive bytes are produced here (a float/double writer and a Kaldi
CompressedMatrix format-1 encoder), never copied from the reference."""
import os
import struct
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/io_funcs")
import kaldi_io as ref_kaldi_io                                            # noqa: E402  (the reference)


def mat_bytes(utt, m, kind):
    head = utt.encode()
    if kind == "F":
        body = struct.pack("<xcccc", b"B", b"F", b"M", b" ") + struct.pack("<bi", 4, m.shape[0]) + struct.pack("<bi", 4, m.shape[1]) + m.astype("<f4").tobytes()
    elif kind == "D":
        body = struct.pack("<xcccc", b"B", b"D", b"M", b" ") + struct.pack("<bi", 4, m.shape[0]) + struct.pack("<bi", 4, m.shape[1]) + m.astype("<f8").tobytes()
    else:   # Kaldi CompressedMatrix, kSpeechFeature (format 1)
        mn, mx = float(m.min()), float(m.max())
        rng = mx - mn
        u16 = lambda v: np.clip(np.round((v - mn) / rng * 65535.0), 0, 65535).astype("<u2")
        body = struct.pack("<xcccc", b"B", b"C", b"M", b" ") + struct.pack("<ffii", mn, rng, m.shape[0], m.shape[1])
        pcs, cols = [], []
        for c in range(m.shape[1]):
            col = np.sort(m[:, c])
            n = len(col)
            p = u16(np.array([col[0], col[n // 4], col[3 * n // 4], col[-1]]))
            p[1] = max(p[1], p[0] + 1); p[2] = max(p[2], p[1] + 1); p[3] = max(p[3], p[2] + 1)
            pcs.append(p)
            pf = mn + rng * 1.52590218966964e-05 * p.astype(np.float64)
            v = m[:, c]
            b = np.where(v < pf[1], np.round((v - pf[0]) / (pf[1] - pf[0]) * 64.0),
                         np.where(v < pf[2], np.round(64 + (v - pf[1]) / (pf[2] - pf[1]) * 128.0),
                                  np.round(192 + (v - pf[2]) / (pf[3] - pf[2]) * 63.0)))
            cols.append(np.clip(b, 0, 255).astype(np.uint8))
        body += b"".join(p.tobytes() for p in pcs) + b"".join(c.tobytes() for c in cols)
    return head, body


def main():
    rng = np.random.default_rng(7)
    mats = [("utt_f32", rng.standard_normal((13, 40)).astype(np.float32), "F"),
            ("utt_f64", rng.standard_normal((9, 257)), "D"),
            ("utt_cmp", (rng.standard_normal((37, 23)) * 3 + 1).astype(np.float32), "C"),
            ("utt_one", rng.standard_normal((1, 5)).astype(np.float32), "F")]
    with tempfile.TemporaryDirectory() as td:
        ark = os.path.join(td, "feats.ark"); scp = os.path.join(td, "feats.scp")
        blob, lines = b"", []
        for utt, m, kind in mats:
            head, body = mat_bytes(utt, m, kind)
            blob += head
            lines.append("%s %s:%d" % (utt, "feats.ark", len(blob)))
            blob += body
        open(ark, "wb").write(blob)
        open(scp, "w").write("".join(l.replace("feats.ark", ark) + "\n" for l in lines))
        reader = ref_kaldi_io.ArkReader()
        reader(scp)
        out = {"ark_bytes": np.frombuffer(blob, np.uint8), "scp_lines": np.array(lines)}
        for i, (utt, m, kind) in enumerate(mats):
            uid, data, looped = reader.read_next_utt()
            assert uid == utt and not looped
            out["ref/" + utt] = np.array(data)
            out["src/" + utt] = m
        assert reader.read_next_utt()[2] is True                  # loops back
    np.savez_compressed(os.path.join(HERE, "ark_golden.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
