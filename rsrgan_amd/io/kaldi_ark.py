"""Kaldi .ark/.scp IO with the same classes and methods as the reference's io_funcs/kaldi_io.py
(ArkReader :41-235, ArkWriter :238-282) and io_funcs/convert_cmvn_to_numpy.py (:19-81).

Binary archives only, like the reference: float ('BFM '), double ('BDM ') and Kaldi's compressed
speech-feature format ('BCM ', format 1: global header min/range/rows/cols, per-column uint16
percentiles, column-major uint8 payload; kaldi_io.py:100-161).  The compressed decode is
vectorised but evaluates exactly the reference's float64 expressions, so it is bit-identical to it
(pinned by tests/golden/ark_*.npz, which were decoded with the reference's own ArkReader)."""
from __future__ import annotations

import os
import random
import struct

import numpy as np


class ArkReader(object):
    """kaldi_io.py:41-235.  Call the instance with an .scp path, then read utterances."""

    def __init__(self, name="ArkReader"):
        self.name = name
        self.utt_ids, self.scp_data, self.scp_position = [], [], 0

    def __call__(self, scp_path):
        """Init utt_ids along with scp_data according to .scp file (:58-71)."""
        self.scp_position = 0
        self.utt_ids, self.scp_data = [], []
        with open(scp_path, "r") as fin:
            for line in fin:
                line = line.replace("\n", "")
                if line == "":
                    break
                utt_id, path_pos = line.split(" ")
                path, pos = path_pos.split(":")
                self.utt_ids.append(utt_id)
                self.scp_data.append((path, pos))

    def shuffle(self):
        """:73-78"""
        zipped = list(zip(self.utt_ids, self.scp_data))
        random.shuffle(zipped)
        self.utt_ids, self.scp_data = (list(t) for t in zip(*zipped)) if zipped else ([], [])
        self.scp_position = 0

    def read_ark(self, ark_file, ark_offset=0):
        """Read one matrix at `ark_offset` (:80-118).  Raises ValueError where the reference prints
        and sys.exit(1)s (not binary / empty / unsupported compressed format)."""
        with open(ark_file, "rb") as buf:
            buf.seek(int(ark_offset), 0)
            header = struct.unpack("<xcccc", buf.read(5))
            if header[0] != b"B":
                raise ValueError("%s: input .ark file is not binary" % ark_file)
            if header[1] == b"C":
                if header[2] == b"M" and header[3] != b"2":
                    min_value, rng, rows, cols = struct.unpack("<ffii", buf.read(16))
                    if cols == 0:
                        raise ValueError("Empty matrix.")
                    return self.read_compress(min_value, rng, rows, cols, buf)
                raise ValueError("Unsupport format. Maybe because of the matrices with 8 or fewer rows.")
            _, rows = struct.unpack("<bi", buf.read(5))
            _, cols = struct.unpack("<bi", buf.read(5))
            if header[1] == b"F":
                mat = np.frombuffer(buf.read(rows * cols * 4), dtype=np.float32)
            elif header[1] == b"D":
                mat = np.frombuffer(buf.read(rows * cols * 8), dtype=np.float64)
            else:
                raise ValueError("unknown matrix type %r" % (header[1],))
            return np.reshape(mat, (rows, cols))

    def read_shape(self, ark_file, ark_offset=0):
        """(rows, cols) of the matrix at `ark_offset` from its header alone (length bucketing without reading the data)."""
        with open(ark_file, "rb") as buf:
            buf.seek(int(ark_offset), 0)
            header = struct.unpack("<xcccc", buf.read(5))
            if header[0] != b"B":
                raise ValueError("%s: input .ark file is not binary" % ark_file)
            if header[1] == b"C":
                _, _, rows, cols = struct.unpack("<ffii", buf.read(16))
                return rows, cols
            _, rows = struct.unpack("<bi", buf.read(5))
            _, cols = struct.unpack("<bi", buf.read(5))
            return rows, cols

    def utt_shape_from_index(self, index):
        return self.read_shape(self.scp_data[index][0], self.scp_data[index][1])

    @staticmethod
    def uint16_to_float(min_value, rng, value):
        """:120-125 (the constant is 1/65535)."""
        return min_value + rng * 1.52590218966964e-05 * value

    @staticmethod
    def char_to_float(p0, p25, p75, p100, value):
        """:127-135, vectorised over `value` (uint8 array) with per-column percentiles."""
        value = value.astype(np.float64)
        lo = p0 + (p25 - p0) * value * (1 / 64.0)
        mid = p25 + (p75 - p25) * (value - 64) * (1 / 128.0)
        hi = p75 + (p100 - p75) * (value - 192) * (1 / 63.0)
        return np.where(value < 64, lo, np.where(value <= 192, mid, hi))

    @classmethod
    def read_compress(cls, min_value, rng, rows, cols, buf):
        """:137-161: per-column uint16 percentile headers, then column-major bytes."""
        pch = np.frombuffer(buf.read(8 * cols), dtype="<u2").reshape(cols, 4).astype(np.float64)
        p = cls.uint16_to_float(float(min_value), float(rng), pch)                 # [cols, 4]
        data = np.frombuffer(buf.read(rows * cols), dtype=np.uint8).reshape(cols, rows)
        out = cls.char_to_float(p[:, 0:1], p[:, 1:2], p[:, 2:3], p[:, 3:4], data)   # [cols, rows]
        return np.ascontiguousarray(out.T)

    def read_next_utt(self):
        """:163-185 -> (utt_id, matrix, looped)."""
        if len(self.scp_data) == 0:
            return None, None, True
        if self.scp_position >= len(self.scp_data):
            looped = True
            self.scp_position = 0
        else:
            looped = False
        self.scp_position += 1
        return self.utt_ids[self.scp_position - 1], self.read_utt_data_from_index(self.scp_position - 1), looped

    def read_next_scp(self):
        """:187-198"""
        if self.scp_position >= len(self.scp_data):
            self.scp_position = 0
        self.scp_position += 1
        return self.utt_ids[self.scp_position - 1]

    def read_previous_scp(self):
        """:200-211"""
        if self.scp_position < 0:
            self.scp_position = len(self.scp_data) - 1
        self.scp_position -= 1
        return self.utt_ids[self.scp_position + 1]

    def read_utt_data_from_id(self, utt_id):
        """:213-225"""
        return self.read_utt_data_from_index(self.utt_ids.index(utt_id))

    def read_utt_data_from_index(self, index):
        """:227-231"""
        return self.read_ark(self.scp_data[index][0], self.scp_data[index][1])

    def split(self):
        """:233-236 (the reference drops the last entry too: [pos:-1])."""
        self.scp_data = self.scp_data[self.scp_position:-1]
        self.utt_ids = self.utt_ids[self.scp_position:-1]


class ArkWriter(object):
    """kaldi_io.py:238-282: float32 'BFM ' matrices appended to an .ark, one scp line each.  Byte
    layout as the reference writes it: <utt_id> \\0 'B' 'F' 'M' ' ' \\4 <rows:i32> \\4 <cols:i32> <data>,
    the scp offset pointing at the \\0."""

    def __init__(self, scp_path):
        self.scp_path = scp_path
        self.scp_file_write = open(self.scp_path, "w")

    def write_next_utt(self, ark_path, utt_id, utt_mat):
        utt_mat = np.ascontiguousarray(np.asarray(utt_mat, dtype=np.float32))
        rows, cols = utt_mat.shape
        with open(ark_path, "ab") as f:
            f.write(utt_id.encode())
            pos = f.tell()
            f.write(struct.pack("<xcccc", b"B", b"F", b"M", b" "))
            f.write(struct.pack("<bi", 4, rows))
            f.write(struct.pack("<bi", 4, cols))
            f.write(utt_mat.tobytes())
        self.scp_file_write.write("%s %s:%s\n" % (utt_id, ark_path, pos))

    def close(self):
        self.scp_file_write.close()


def read_binary_file(filename, offset=0):
    """convert_cmvn_to_numpy.py:50-81: one uncompressed binary matrix."""
    with open(filename, "rb") as buf:
        buf.seek(int(offset), 0)
        header = struct.unpack("<xcccc", buf.read(5))
        if header[0] != b"B":
            raise ValueError("Input .ark file is not binary")
        if header[1] == b"C":
            raise ValueError("Input .ark file is compressed")
        _, rows = struct.unpack("<bi", buf.read(5))
        _, cols = struct.unpack("<bi", buf.read(5))
        if header[1] == b"F":
            mat = np.frombuffer(buf.read(rows * cols * 4), dtype=np.float32)
        else:
            mat = np.frombuffer(buf.read(rows * cols * 8), dtype=np.float64)
        return np.reshape(mat, (rows, cols))


def convert_cmvn_to_numpy(inputs_cmvn, labels_cmvn, save_dir=None):
    """convert_cmvn_to_numpy.py:19-47: Kaldi global CMVN stats ([2, D+1]: sums | sums of squares,
    frame count in the last column of row 0) -> mean / stddev; saved as train_cmvn.npz."""
    out = {}
    for tag, path in (("inputs", inputs_cmvn), ("labels", labels_cmvn)):
        stats = read_binary_file(path, 0)
        frames = stats[0][-1]
        cm = np.hsplit(stats, [stats.shape[1] - 1])[0]
        mean = cm[0] / frames
        out["mean_" + tag] = mean
        out["stddev_" + tag] = np.sqrt(cm[1] / frames - mean ** 2)
    if save_dir is not None:
        np.savez(os.path.join(save_dir, "train_cmvn.npz"), **out)
    return out
