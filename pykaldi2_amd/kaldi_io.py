"""Kaldi table I/O used by the path's tools: compact-lattice archives (`kaldi.util.table.CompactLatticeWriter`, reference
bin/latgen.py:156-181) and float-matrix archives (what `kaldi.util.table.MatrixWriter`
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


class CompactLatticeWriter:
    """with CompactLatticeWriter("ark:lat.ark") as w: w[utt_id] = lattice  (reference bin/latgen.py:156,181), `lattice` = the
    dict LatticeBatch.compact_lattice returns.  [upstream-knowledge: Kaldi's formats, no Kaldi in this environment]

    "ark:<file>" (binary, what the reference writes): ``<key> <space> \\0B`` followed by OpenFst's binary VectorFst with arc
    type "compactlattice44" -- header {int32 magic 2125659606, string "vector", string "compactlattice44", int32 version
    2, int32 flags 0, uint64 properties, int64 start, int64 num_states, int64 num_arcs}; per state its final weight
    {f32 graph, f32 acoustic, int32 n, n x int32 transition-ids} (+inf, +inf, 0 = not final) and int64 num_arcs, per arc
    {int32 ilabel, int32 olabel, weight as above, int32 nextstate}.
    "ark,t:<file>": text -- key line, arcs ``src dst word graph,acoustic,t1_t2``, finals ``state graph,acoustic,``, blank
    line.  Kaldi writes the start state first: states are renumbered so that the start state is 0.
    A determinised lattice (lattice.determinize_lattice / compact_lattice(determinize=True)) carries a transition-id STRING
    per arc and per final weight (`tid_off` / `tids`, `final_acoustic`, `final_tid_off` / `final_tids`) instead of one `tid`."""

    def __init__(self, wspecifier):
        opts, _, path = wspecifier.partition(":")
        assert opts.split(",")[0] == "ark" and path, "only 'ark:<file>' and 'ark,t:<file>' write specifiers are supported"
        self._text = "t" in opts.split(",")[1:]
        self._f = open(path, "wb")

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        if self._f:
            self._f.close()
            self._f = None

    def __setitem__(self, key, lat):
        assert key and " " not in key
        n = int(lat["num_states"])
        order = np.arange(n)
        start = int(lat["start"])
        if start != 0:
            order[0], order[start] = start, 0          # new id -> old id
        new_of = np.empty(n, np.int64)
        new_of[order] = np.arange(n)
        src, dst = new_of[lat["src"]], new_of[lat["dst"]]
        by_src = np.argsort(src, kind="stable")
        counts = np.bincount(src, minlength=n)
        fin = np.asarray(lat["final"], np.float32)[order]
        if "tid_off" in lat:          # determinised: strings
            arc_ids = lambda l: [int(t) for t in lat["tids"][lat["tid_off"][l]:lat["tid_off"][l + 1]]]
            fin_ac = np.asarray(lat["final_acoustic"], np.float32)[order]
            fin_ids = lambda s_: [int(t) for t in lat["final_tids"][lat["final_tid_off"][order[s_]]:lat["final_tid_off"][order[s_] + 1]]]
        else:
            arc_ids = lambda l: [int(lat["tid"][l])] if int(lat["tid"][l]) > 0 else []
            fin_ac = np.zeros(n, np.float32)
            fin_ids = lambda s_: []
        if self._text:
            lines = [key]
            for l in by_src:
                lines.append("%d\t%d\t%d\t%s,%s,%s" % (src[l], dst[l], lat["word"][l], repr(float(lat["graph"][l])),
                                                          repr(float(lat["acoustic"][l])), "_".join(str(t) for t in arc_ids(l))))
            for s_ in np.flatnonzero(np.isfinite(fin)):
                lines.append("%d\t%s,%s,%s" % (s_, repr(float(fin[s_])), repr(float(fin_ac[s_])) if fin_ac[s_] != 0 else "0",
                                                "_".join(str(t) for t in fin_ids(s_))))
            self._f.write(("\n".join(lines) + "\n\n").encode())
            return

        def fst_string(b):
            return struct.pack("<i", len(b)) + b
        out = [key.encode() + b" \0B", struct.pack("<i", 2125659606), fst_string(b"vector"), fst_string(b"compactlattice44"),
               struct.pack("<iiQqqq", 2, 0, 3, 0, n, src.shape[0])]
        pos = 0
        for s_ in range(n):
            if np.isfinite(fin[s_]):
                ids = fin_ids(s_)
                out.append(struct.pack("<ffi%di" % len(ids), float(fin[s_]), float(fin_ac[s_]), len(ids), *ids))
            else:
                out.append(struct.pack("<ffi", float("inf"), float("inf"), 0))
            out.append(struct.pack("<q", int(counts[s_])))
            for l in by_src[pos:pos + counts[s_]]:
                ids = arc_ids(l)
                w = int(lat["word"][l])
                out.append(struct.pack("<iiff", w, w, float(lat["graph"][l]), float(lat["acoustic"][l])))
                out.append(struct.pack("<i%di" % len(ids), len(ids), *ids))
                out.append(struct.pack("<i", int(dst[l])))
            pos += counts[s_]
        self._f.write(b"".join(out))


def read_compact_lattice_ark(path):
    """Yields (key, lattice dict) from a binary archive written by CompactLatticeWriter (round-trip tests / tooling)."""
    with open(path, "rb") as f:
        data = f.read()
    pos, n = 0, len(data)
    while pos < n:
        sp = data.index(b" ", pos)
        key = data[pos:sp].decode()
        pos = sp + 1
        assert data[pos:pos + 2] == b"\0B", "not a binary Kaldi object"
        pos += 2
        magic, = struct.unpack_from("<i", data, pos); pos += 4
        assert magic == 2125659606

        def rd_str(p):
            ln, = struct.unpack_from("<i", data, p)
            return data[p + 4:p + 4 + ln], p + 4 + ln
        ftype, pos = rd_str(pos)
        atype, pos = rd_str(pos)
        assert ftype == b"vector" and atype == b"compactlattice44", (ftype, atype)
        version, flags, props, start, ns, na = struct.unpack_from("<iiQqqq", data, pos); pos += 40

        def rd_weight(p):
            g, a, k = struct.unpack_from("<ffi", data, p)
            ids = list(struct.unpack_from("<%di" % k, data, p + 12)) if k else []
            return (g, a, ids), p + 12 + 4 * k
        finals, arcs = [], []
        for s_ in range(ns):
            w, pos = rd_weight(pos)
            finals.append(w)
            cnt, = struct.unpack_from("<q", data, pos); pos += 8
            for _ in range(cnt):
                il, ol = struct.unpack_from("<ii", data, pos); pos += 8
                w, pos = rd_weight(pos)
                nxt, = struct.unpack_from("<i", data, pos); pos += 4
                arcs.append((s_, nxt, il, ol, w))
        assert len(arcs) == na
        yield key, dict(start=start, num_states=ns, finals=finals, arcs=arcs)
