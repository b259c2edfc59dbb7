"""TensorBoard event files without TensorFlow: what `tf.summary.FileWriter(save_dir/{train,eval})`, `scalar_summary`,
`histogram_summary`, `tf.summary.merge` and `writer.add_summary(summaries, step)` do in the reference (utils/ops.py:44-56,
models/gan_rnn_placeholder.py:81-86,219-223,270-298, models/gan.py:77-82,184-250; call sites
scripts/train_gan_rnn_placeholder.py:116-122,186-190 and scripts/train_gan_dnn.py:132-134,195-196).

The file format is TensorFlow's: a TFRecord stream (uint64 length, masked crc32c of the length, payload, masked crc32c of the
payload) of `Event` protocol buffers; the few message types needed (Event, Summary, Summary.Value, HistogramProto) are encoded by
hand.  Histograms use TensorFlow's default bucket limits (+-1e-12 * 1.1^k up to 1e20) and its run-length collapse of empty
buckets (core/lib/histogram/histogram.cc), so TensorBoard shows them as it shows the reference's."""
from __future__ import annotations

import os
import socket
import struct
import time
from typing import Dict, Iterable, Optional

import numpy as np

# ---- crc32c (Castagnoli), masked as TFRecord wants it ----
_POLY = 0x82F63B78
_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ _POLY if _c & 1 else _c >> 1
    _TABLE.append(_c)


def crc32c(data: bytes) -> int:
    c = 0xFFFFFFFF
    for b in data:
        c = _TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data: bytes) -> int:
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


# ---- protocol buffer wire format ----
def _varint(n: int) -> bytes:
    n &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _key(field: int, wire: int) -> bytes:
    return _varint((field << 3) | wire)


def _f_double(field, v):
    return _key(field, 1) + struct.pack("<d", float(v))


def _f_float(field, v):
    return _key(field, 5) + struct.pack("<f", float(v))


def _f_varint(field, v):
    return _key(field, 0) + _varint(int(v))


def _f_bytes(field, b: bytes):
    return _key(field, 2) + _varint(len(b)) + b


def _f_packed_doubles(field, vals):
    return _f_bytes(field, struct.pack("<%dd" % len(vals), *vals)) if len(vals) else b""


def _default_limits():
    pos, v = [], 1e-12
    while v < 1e20:
        pos.append(v)
        v *= 1.1
    pos.append(np.finfo(np.float64).max)
    return np.array([-x for x in reversed(pos)] + [0.0] + pos)


_LIMITS = _default_limits()


def scalar(tag: str, value) -> bytes:
    """one Summary.Value {tag, simple_value}"""
    return _f_bytes(1, _f_bytes(1, tag.encode()) + _f_float(2, value))


def histogram(tag: str, values) -> bytes:
    """one Summary.Value {tag, histo}: min, max, num, sum, sum_squares, bucket_limit[], bucket[] (histogram.cc EncodeToProto)"""
    x = np.asarray(values, np.float64).ravel()
    if x.size == 0:
        x = np.zeros(1)
    idx = np.searchsorted(_LIMITS, x, side="right")            # upper_bound: the first limit greater than the value
    counts = np.bincount(np.minimum(idx, len(_LIMITS) - 1), minlength=len(_LIMITS)).astype(np.float64)
    limits, buckets, i, n = [], [], 0, len(_LIMITS)
    while i < n:
        end, count = _LIMITS[i], counts[i]
        i += 1
        if count <= 0.0:
            while i < n and counts[i] <= 0.0:
                end, count = _LIMITS[i], counts[i]
                i += 1
        limits.append(end); buckets.append(count)
    h = (_f_double(1, x.min()) + _f_double(2, x.max()) + _f_double(3, x.size) + _f_double(4, x.sum()) + _f_double(5, np.square(x).sum()) +
         _f_packed_doubles(6, limits) + _f_packed_doubles(7, buckets))
    return _f_bytes(1, _f_bytes(1, tag.encode()) + _f_bytes(5, h))


def merge(values: Iterable[bytes]) -> bytes:
    """tf.summary.merge: a Summary is the concatenation of its values"""
    return b"".join(values)


class FileWriter:
    """tf.summary.FileWriter(logdir): events.out.tfevents.<time>.<host> with the version record first"""

    def __init__(self, logdir: str, filename_suffix: str = ""):
        os.makedirs(logdir, exist_ok=True)
        self.path = os.path.join(logdir, "events.out.tfevents.%010d.%s%s" % (int(time.time()), socket.gethostname(), filename_suffix))
        self._f = open(self.path, "wb")
        self._event(_f_double(1, time.time()) + _f_bytes(3, b"brain.Event:2"))
        self.flush()

    def _event(self, payload: bytes):
        head = struct.pack("<Q", len(payload))
        self._f.write(head + struct.pack("<I", masked_crc32c(head)) + payload + struct.pack("<I", masked_crc32c(payload)))

    def add_summary(self, summary: bytes, global_step: Optional[int] = None):
        ev = _f_double(1, time.time())
        if global_step is not None:
            ev += _f_varint(2, int(global_step))
        self._event(ev + _f_bytes(5, summary))
        self.flush()

    def flush(self):
        self._f.flush()

    def close(self):
        if not self._f.closed:
            self._f.close()


LOSS_TAGS = ("d_rl_loss", "d_fk_loss", "d_loss", "g_adv_loss", "g_mse_loss", "g_l2_loss", "g_loss")


def model_summaries(losses7, inputs=None, labels=None, g=None) -> bytes:
    """`model.summaries` of the reference graphs: the seven loss scalars and the histograms of the fed batch and of the
    generator's output -- with the reference's tags ('real_clean' is fed `inputs` and 'real_noise' `labels`,
    gan_rnn_placeholder.py:221-222).  The histograms of the discriminator logits ('d_real', 'd_fake') need tensors the C ABI does
    not hand out and are left out."""
    vals = [scalar(t, v) for t, v in zip(LOSS_TAGS, losses7)]
    for tag, a in (("real_clean", inputs), ("real_noise", labels), ("g_clean", g)):
        if a is not None:
            vals.append(histogram(tag, a))
    return merge(vals)


def read_events(path: str):
    """[(wall_time, step, {tag: simple_value | histogram dict}, file_version)] -- the inverse, for tests and quick looks"""
    def fields(buf):
        i, out = 0, []
        while i < len(buf):
            k, s = 0, 0
            while True:
                b = buf[i]; i += 1
                k |= (b & 0x7F) << s; s += 7
                if not b & 0x80:
                    break
            f, w = k >> 3, k & 7
            if w == 0:
                v, s = 0, 0
                while True:
                    b = buf[i]; i += 1
                    v |= (b & 0x7F) << s; s += 7
                    if not b & 0x80:
                        break
            elif w == 1:
                v = struct.unpack("<d", buf[i:i + 8])[0]; i += 8
            elif w == 5:
                v = struct.unpack("<f", buf[i:i + 4])[0]; i += 4
            elif w == 2:
                n, s = 0, 0
                while True:
                    b = buf[i]; i += 1
                    n |= (b & 0x7F) << s; s += 7
                    if not b & 0x80:
                        break
                v = buf[i:i + n]; i += n
            else:
                raise ValueError("wire type %d" % w)
            out.append((f, v))
        return out
    events = []
    with open(path, "rb") as f:
        data = f.read()
    i = 0
    while i < len(data):
        head = data[i:i + 8]
        (n,) = struct.unpack("<Q", head)
        if struct.unpack("<I", data[i + 8:i + 12])[0] != masked_crc32c(head):
            raise ValueError("length checksum")
        payload = data[i + 12:i + 12 + n]
        if struct.unpack("<I", data[i + 12 + n:i + 16 + n])[0] != masked_crc32c(payload):
            raise ValueError("payload checksum")
        i += 16 + n
        wall, step, vals, version = None, None, {}, None
        for f_, v in fields(payload):
            if f_ == 1:
                wall = v
            elif f_ == 2:
                step = v
            elif f_ == 3:
                version = v.decode()
            elif f_ == 5:
                for f2, v2 in fields(v):
                    if f2 != 1:
                        continue
                    tag, val = None, None
                    for f3, v3 in fields(v2):
                        if f3 == 1:
                            tag = v3.decode()
                        elif f3 == 2:
                            val = v3
                        elif f3 == 5:
                            h = dict(fields(v3))
                            val = dict(min=h[1], max=h[2], num=h[3], sum=h[4], sum_squares=h[5],
                                       bucket_limit=np.frombuffer(h.get(6, b""), "<f8"), bucket=np.frombuffer(h.get(7, b""), "<f8"))
                    vals[tag] = val
        events.append((wall, step, vals, version))
    return events
