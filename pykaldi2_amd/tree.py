"""Kaldi's context-dependency tree (kaldi.tree.ContextDependency: reference bin/train_chain.py:178-180 reads
<chain_dir>/tree and hands it to proto_supervision_to_supervision, :272).

[upstream knowledge of the format -- ContextDependency::Write / EventMap::Write; no Kaldi file is available in
this environment, the readers are checked against files assembled by tests/test_supervision.py]

  ContextDependency <N> <P> ToPdf <EventMap> EndContextDependency
  EventMap :=  CE <answer>                                   constant
            |  TE <key> <size> ( <EventMap or NULL> x size )  table on the value of `key`
            |  SE <key> [ yes-values ] { <yes map> <no map> } split
  keys: -1 = pdf-class, 0..N-1 = phone at that position of the window.  Binary files start with "\\0B"; tokens
  are followed by a space, an int32 is the byte 4 plus four little-endian bytes (the table size, a uint32, has
  the byte -4), an integer vector is the byte 4, an int32 count and the raw values.

The tree is kept flattened (the arrays pk2_sup_model_create takes, include/pk2hip.h).
"""
import struct

import numpy as np

CE, TE, SE = 0, 1, 2


class _Stream:
    def __init__(self, raw, path):
        self.path = path
        self.binary = raw[:2] == b"\0B"
        if self.binary:
            self.raw, self.pos = raw, 2
        else:
            self.toks, self.pos = raw.decode(errors="replace").split(), 0

    def token(self, expect=None):
        if self.binary:
            end = self.raw.index(b" ", self.pos)
            t = self.raw[self.pos:end].decode()
            self.pos = end + 1
        else:
            t = self.toks[self.pos]
            self.pos += 1
        if expect is not None and t != expect:
            raise ValueError("%s: expected %s, found %s" % (self.path, expect, t))
        return t

    def peek(self):
        return chr(self.raw[self.pos]) if self.binary else self.toks[self.pos][0]

    def int(self):
        if not self.binary:
            return int(self.token())
        if self.raw[self.pos] not in (4, 0xFC):
            raise ValueError("%s: bad integer marker at byte %d" % (self.path, self.pos))
        v = struct.unpack_from("<i", self.raw, self.pos + 1)[0]
        self.pos += 5
        return v

    def ivec(self):
        if not self.binary:
            self.token("[")
            out = []
            while self.toks[self.pos] != "]":
                out.append(int(self.token()))
            self.token("]")
            return out
        if self.raw[self.pos] != 4:
            raise ValueError("%s: bad vector marker at byte %d" % (self.path, self.pos))
        n = struct.unpack_from("<i", self.raw, self.pos + 1)[0]
        out = list(struct.unpack_from("<%di" % n, self.raw, self.pos + 5))
        self.pos += 5 + 4 * n
        return out


class ContextDependency:
    """N = context width, P = central position, and the flattened ToPdf event map."""

    def __init__(self, N, P, kind, key, a, b, pool):
        self.N, self.P = int(N), int(P)
        self.kind = np.ascontiguousarray(kind, np.int32)
        self.key = np.ascontiguousarray(key, np.int32)
        self.a = np.ascontiguousarray(a, np.int32)
        self.b = np.ascontiguousarray(b, np.int32)
        self.pool = np.ascontiguousarray(pool, np.int32)

    def context_width(self):
        return self.N

    def central_position(self):
        return self.P

    def num_pdfs(self):
        return int(self.a[self.kind == CE].max()) + 1

    @classmethod
    def from_nested(cls, N, P, root):
        """root: ("CE", pdf) | ("TE", key, [child or None, ...]) | ("SE", key, [yes values], yes, no)."""
        kind, key, a, b, pool = [], [], [], [], []

        def add(node):
            if node is None:
                return -1
            me = len(kind)
            kind.append({"CE": CE, "TE": TE, "SE": SE}[node[0]]); key.append(0); a.append(0); b.append(0)
            if node[0] == "CE":
                a[me] = int(node[1])
            elif node[0] == "TE":
                key[me] = int(node[1])
                kids = [add(c) for c in node[2]]
                a[me], b[me] = len(pool), len(kids)
                pool.extend(kids)
            else:
                key[me] = int(node[1])
                yes = sorted(set(int(v) for v in node[2]))
                kids = [add(node[3]), add(node[4])]
                a[me], b[me] = len(pool), len(yes)
                pool.extend(yes + kids)
            return me

        add(root)
        return cls(N, P, kind, key, a, b, pool)

    @classmethod
    def read(cls, path):
        with open(path, "rb") as f:
            st = _Stream(f.read(), path)
        st.token("ContextDependency")
        N, P = st.int(), st.int()
        st.token("ToPdf")

        def event_map():
            c = st.peek()
            if c == "N":
                st.token("NULL")
                return None
            tag = st.token()
            if tag == "CE":
                return ("CE", st.int())
            if tag == "TE":
                k, size = st.int(), st.int()
                st.token("(")
                kids = [event_map() for _ in range(size)]
                st.token(")")
                return ("TE", k, kids)
            if tag == "SE":
                k, yes = st.int(), st.ivec()
                st.token("{")
                y, n = event_map(), event_map()
                st.token("}")
                return ("SE", k, yes, y, n)
            raise ValueError("%s: unknown event map %r" % (path, tag))

        import sys
        limit = sys.getrecursionlimit()
        sys.setrecursionlimit(max(limit, 20000))   # SE chains of real trees are deep
        try:
            root = event_map()
            st.token("EndContextDependency")
            return cls.from_nested(N, P, root)
        finally:
            sys.setrecursionlimit(limit)

    def write(self, path, binary=True):
        """Writes the tree in Kaldi's format (tests and synthetic recipes)."""
        out = bytearray(b"\0B" if binary else b"")

        def tok(t):
            out.extend((t + " ").encode())

        def i32(v, unsigned=False):
            out.extend((struct.pack("<b", -4 if unsigned else 4) + struct.pack("<i", v)) if binary else ("%d " % v).encode())

        def emit(n):
            if n < 0:
                return tok("NULL")
            p = self.pool[self.a[n]:]
            if self.kind[n] == CE:
                tok("CE"); i32(int(self.a[n]))
            elif self.kind[n] == TE:
                tok("TE"); i32(int(self.key[n])); i32(int(self.b[n]), unsigned=True); tok("(")
                for c in p[:self.b[n]]:
                    emit(int(c))
                tok(")")
            else:
                tok("SE"); i32(int(self.key[n]))
                yes = [int(v) for v in p[:self.b[n]]]
                if binary:
                    out.extend(struct.pack("<bi", 4, len(yes)) + struct.pack("<%di" % len(yes), *yes))
                else:
                    tok("[ " + " ".join(map(str, yes)) + " ]")
                tok("{"); emit(int(p[self.b[n]])); emit(int(p[self.b[n] + 1])); tok("}")

        tok("ContextDependency"); i32(self.N); i32(self.P); tok("ToPdf")
        emit(0)
        tok("EndContextDependency")
        with open(path, "wb") as f:
            f.write(bytes(out))

    def compute(self, window, pdf_class):
        """ContextDependency::Compute -> pdf-id or None.  (Host-side check of the flattened form; training goes
        through pk2_supervision_create.)"""
        node = 0
        while node >= 0:
            k = int(self.kind[node])
            if k == CE:
                return int(self.a[node])
            v = pdf_class if self.key[node] == -1 else window[self.key[node]]
            p = self.pool[self.a[node]:]
            if k == TE:
                if not 0 <= v < self.b[node]:
                    return None
                node = int(p[v])
            else:
                node = int(p[self.b[node]] if v in p[:self.b[node]] else p[self.b[node] + 1])
        return None
