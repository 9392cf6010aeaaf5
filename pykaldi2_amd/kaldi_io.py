"""Kaldi table I/O used by the path's tools: float-matrix archives (what `kaldi.util.table.MatrixWriter`
writes at reference bin/dump_loglikes.py:117-132, "ark:<file>", binary) and their reader.

Binary matrix entry: ``<key> <space> \\0B FM <space> \\4 <int32 rows> \\4 <int32 cols> <rows*cols float32>``
(DM with float64 is read too).  Text archives (``key  [ \\n rows ... ]``) are read as well.
"""
import struct

import numpy as np


class MatrixWriter:
    """with MatrixWriter("ark:out.ark") as w: w[utt_id] = matrix  (the PyKaldi idiom of the reference)."""

    def __init__(self, wspecifier):
        assert wspecifier.startswith("ark:"), "only 'ark:<file>' write specifiers are supported"
        self._f = open(wspecifier[4:], "wb")

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        if self._f:
            self._f.close()
            self._f = None

    def __setitem__(self, key, matrix):
        m = np.ascontiguousarray(matrix.detach().cpu().numpy() if hasattr(matrix, "detach") else matrix, dtype="<f4")
        assert m.ndim == 2 and " " not in key and key
        self._f.write(key.encode() + b" \0BFM " + b"\x04" + struct.pack("<i", m.shape[0]) + b"\x04" +
                      struct.pack("<i", m.shape[1]) + m.tobytes())


def read_matrix_ark(path):
    """Yields (key, float32 matrix) from a binary or text Kaldi matrix archive."""
    with open(path, "rb") as f:
        data = f.read()
    pos, n = 0, len(data)
    while pos < n:
        while pos < n and data[pos:pos + 1] in (b" ", b"\n", b"\t"):
            pos += 1
        if pos >= n:
            break
        sp = data.index(b" ", pos)
        key = data[pos:sp].decode()
        pos = sp + 1
        if data[pos:pos + 2] == b"\0B":
            tag = data[pos + 2:pos + 5]
            assert tag in (b"FM ", b"DM "), "unsupported Kaldi object %r for key %s" % (tag, key)
            dt = np.dtype("<f4") if tag == b"FM " else np.dtype("<f8")
            pos += 5
            assert data[pos] == 4
            rows = struct.unpack_from("<i", data, pos + 1)[0]
            assert data[pos + 5] == 4
            cols = struct.unpack_from("<i", data, pos + 6)[0]
            pos += 10
            m = np.frombuffer(data, dt, rows * cols, pos).reshape(rows, cols).astype(np.float32)
            pos += rows * cols * dt.itemsize
            yield key, m
        else:
            end = data.index(b"]", pos)
            body = data[pos:end].decode().replace("[", " ")
            rows = [[float(x) for x in line.split()] for line in body.strip().split("\n") if line.strip()]
            pos = end + 1
            yield key, np.asarray(rows, np.float32)
