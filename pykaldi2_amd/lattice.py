"""Host mirror of the PyKaldi objects behind the reference's lattice-based criteria (reference
ops/ops.py:41-75,119-156; bin/train_se.py:145-184):

  kaldi.hmm.TransitionModel                         -> TransitionModel (transition-id -> pdf / phone tables)
  kaldi.decoder.LatticeFasterDecoderOptions         -> LatticeFasterDecoderOptions
  kaldi.asr.MappedLatticeFasterRecognizer           -> MappedLatticeFasterRecognizer (.decode / .decode_batch)
  kaldi.lat.functions.lattice_forward_backward_mmi  -> LatticeBatch.mmi
  ... lattice_forward_backward_mpe_variants         -> LatticeBatch.mpe

Everything runs on the device through the C ABI (pk2_lattice_*): one workgroup per utterance decodes and
prunes the lattice, the lattice forward-backward reads it in place.  Nothing falls back to the CPU.
"""
import ctypes as C

import os
import numpy as np
import torch

from . import _lib


class LatticeFasterDecoderOptions:
    """Kaldi's LatticeFasterDecoderConfig fields the reference sets (bin/train_se.py:173-178) and the other
    pruning knobs at their Kaldi defaults."""

    def __init__(self, beam=16.0, lattice_beam=10.0, max_active=2 ** 31 - 1, min_active=200, beam_delta=0.5):
        self.beam, self.lattice_beam = beam, lattice_beam
        self.max_active, self.min_active, self.beam_delta = max_active, min_active, beam_delta
        self.determinize_lattice = False   # raw state-level lattices, as the reference asks for
        # pool sizing of the device lattices: tokens / links per frame on average (None = from max_active)
        self.tokens_per_frame = None
        self.links_per_frame = None


class TransitionModel:
    """Kaldi's TransitionModel as tables.  transition-id -> pdf-id / phone is all a lattice criterion needs;
    a model read from a file (or built with from_topology) also keeps the HMM topologies, the tuples and
    transition-id -> transition-state / self-loop / final flags, which splitting an alignment into phones and the
    chain supervision need.  transition-ids are 1-based; index 0 of every table is unused."""

    def __init__(self, tid2pdf, tid2phone):
        self.tid2pdf = np.ascontiguousarray(tid2pdf, dtype=np.int32)
        self.tid2phone = np.ascontiguousarray(tid2phone, dtype=np.int32)
        assert self.tid2pdf.shape == self.tid2phone.shape and self.tid2pdf.ndim == 1
        self._dev = {}
        self.phone2entry = self.entries = self.tuples = self.tid2tstate = self.tid_flags = None

    @classmethod
    def from_arrays(cls, d):
        return cls(d["tid2pdf"], d["tid2phone"])

    @classmethod
    def from_topology(cls, phone2entry, entries, tuples):
        """phone2entry: {phone: entry index}; entries[e] = HMM states [(forward pdf-class, self-loop pdf-class,
        [destination states])] with the non-emitting final state last (classes -1); tuples = (phone, hmm_state,
        forward pdf, self-loop pdf) in transition-state order.  Transition ids enumerate, for every tuple in
        order, the transitions of its HMM state in topology order (TransitionModel::ComputeDerived)."""
        tid2pdf, tid2phone, tid2tstate, flags = [-1], [0], [0], [0]
        for ts, (phone, hmm_state, fwd_pdf, loop_pdf) in enumerate(tuples, start=1):
            states = entries[phone2entry[phone]]
            for dst_state in states[hmm_state][2]:
                loop = dst_state == hmm_state
                tid2pdf.append(loop_pdf if loop else fwd_pdf)   # a self-loop may carry its own pdf (<Tuples>)
                tid2phone.append(phone)
                tid2tstate.append(ts)
                flags.append((1 if loop else 0) | (2 if dst_state == len(states) - 1 else 0))
        m = cls(tid2pdf, tid2phone)
        m.phone2entry = {int(k): int(v) for k, v in phone2entry.items()}
        m.entries = [[(int(f), int(l), [int(d) for d in dsts]) for f, l, dsts in st] for st in entries]
        m.tuples = np.asarray(tuples, dtype=np.int32).reshape(-1, 4)
        m.tid2tstate = np.asarray(tid2tstate, dtype=np.int32)
        m.tid_flags = np.asarray(flags, dtype=np.uint8)
        return m

    @classmethod
    def read(cls, path):
        """Kaldi transition model, text (`copy-transition-model --binary=false`) or binary (`final.mdl`) form
        [upstream knowledge of the formats: <TransitionModel> <Topology> ... </Topology> <Triples>|<Tuples> n ... ].
        Transition ids enumerate, for every tuple in file order, the transitions of its HMM state in topology order."""
        with open(path, "rb") as f:
            head = f.read(2)
            if head == b"\0B":
                return cls._read_binary(f.read(), path)
            toks = (head + f.read()).decode(errors="replace").split()
        pos = toks.index("<Topology>")
        phone_entry, entries = {}, []
        i = pos + 1
        while toks[i] != "</Topology>":
            if toks[i] == "<TopologyEntry>":
                i += 1
                assert toks[i] == "<ForPhones>"
                i += 1
                phones = []
                while toks[i] != "</ForPhones>":
                    phones.append(int(toks[i])); i += 1
                i += 1
                states = []
                while toks[i] != "</TopologyEntry>":
                    assert toks[i] == "<State>"
                    i += 2   # <State> index
                    dsts, fwd_class, loop_class = [], -1, -1
                    if toks[i] in ("<PdfClass>",):
                        fwd_class = loop_class = int(toks[i + 1])
                        i += 2
                    elif toks[i] == "<ForwardPdfClass>":
                        fwd_class, loop_class = int(toks[i + 1]), int(toks[i + 3])
                        i += 4   # <ForwardPdfClass> a <SelfLoopPdfClass> b
                    while toks[i] == "<Transition>":
                        dsts.append(int(toks[i + 1])); i += 3
                    assert toks[i] == "</State>"
                    i += 1
                    states.append((fwd_class, loop_class, dsts))
                i += 1
                for ph in phones:
                    phone_entry[ph] = len(entries)
                entries.append(states)
            else:
                i += 1
        i += 1
        tag = toks[i]
        assert tag in ("<Triples>", "<Tuples>"), tag
        n = int(toks[i + 1])
        i += 2
        width = 3 if tag == "<Triples>" else 4
        tuples = []
        for _ in range(n):
            vals = [int(x) for x in toks[i:i + width]]
            i += width
            tuples.append((vals[0], vals[1], vals[2], vals[-1]))
        return cls.from_topology(phone_entry, entries, tuples)

    @classmethod
    def _read_binary(cls, raw, path):
        """Kaldi binary form (`final.mdl`, `0.trans_mdl`; anything after </TransitionModel>, e.g. the GMMs, is
        ignored).  [upstream knowledge: TransitionModel::Read / HmmTopology::Read -- tokens are "<Name> ", an
        int32 / float is a size byte 4 followed by 4 little-endian bytes, an integer vector is a size byte 4, an
        int32 count and the raw values; the topology is dumped as phones, phone2idx, entries (a leading -1 marks
        the non-HMM form with separate self-loop pdf classes)]"""
        import struct
        pos = 0

        def token(expect=None):
            nonlocal pos
            end = raw.index(b" ", pos)
            t = raw[pos:end].decode()
            pos = end + 1
            if expect is not None and t != expect:
                raise ValueError("%s: expected %s, found %s" % (path, expect, t))
            return t

        def i32():
            nonlocal pos
            assert raw[pos] == 4, "%s: bad int marker at byte %d" % (path, pos)
            v = struct.unpack_from("<i", raw, pos + 1)[0]
            pos += 5
            return v

        def ivec():
            nonlocal pos
            assert raw[pos] == 4
            n = struct.unpack_from("<i", raw, pos + 1)[0]
            v = list(struct.unpack_from("<%di" % n, raw, pos + 5))
            pos += 5 + 4 * n
            return v

        token("<TransitionModel>")
        token("<Topology>")
        phones, phone2idx = ivec(), ivec()
        sz = i32()
        is_hmm = True
        if sz == -1:
            is_hmm = False
            sz = i32()
        entries = []
        for _ in range(sz):
            states = []
            for _ in range(i32()):
                fwd_class = loop_class = i32()
                if not is_hmm:
                    loop_class = i32()
                dsts = []
                for _ in range(i32()):
                    dsts.append(i32())
                    i32()                  # probability (float, same 5-byte encoding)
                states.append((fwd_class, loop_class, dsts))
            entries.append(states)
        token("</Topology>")
        tag = token()
        assert tag in ("<Triples>", "<Tuples>"), tag
        tuples = []
        for _ in range(i32()):
            phone, hmm_state, fwd_pdf = i32(), i32(), i32()
            tuples.append((phone, hmm_state, fwd_pdf, i32() if tag == "<Tuples>" else fwd_pdf))
        token("</Triples>" if tag == "<Triples>" else "</Tuples>")
        return cls.from_topology({ph: phone2idx[ph] for ph in phones}, entries, tuples)

    def num_transition_ids(self):
        return int(self.tid2pdf.shape[0] - 1)

    def num_pdfs(self):
        return int(self.tid2pdf.max() + 1)

    def transition_id_to_pdf(self, tid):
        return int(self.tid2pdf[tid])

    def transition_id_to_phone(self, tid):
        return int(self.tid2phone[tid])

    def device_tables(self, device):
        key = str(device)
        if key not in self._dev:
            self._dev[key] = (torch.from_numpy(self.tid2pdf).to(device), torch.from_numpy(self.tid2phone).to(device))
        return self._dev[key]

    def silence_mask(self, silence_phones, device):
        m = np.zeros(int(self.tid2phone.max()) + 2, np.uint8)
        for ph in silence_phones:
            if 0 <= int(ph) < m.shape[0]:
                m[int(ph)] = 1
        return torch.from_numpy(m).to(device)


class DecodeGraph:
    """HCLG (kaldi.fstext.StdVectorFst read from HCLG.fst, reference bin/train_se.py:145,180)."""

    def __init__(self, graph):
        L = _lib.lib()
        h = C.c_void_p()
        if isinstance(graph, (str, bytes)):
            _lib.check(L.pk2_decode_graph_from_openfst(graph.encode() if isinstance(graph, str) else graph, C.byref(h)))
        else:
            src = np.ascontiguousarray(graph["src"], np.int32); dst = np.ascontiguousarray(graph["dst"], np.int32)
            il = np.ascontiguousarray(graph["ilabel"], np.int32); w = np.ascontiguousarray(graph["weight"], np.float32)
            fin = np.ascontiguousarray(graph["final"], np.float32)
            assert fin.shape[0] == int(graph["num_states"])
            if graph.get("olabel") is not None:     # word ids: only the lattice-dumping tools need them
                ol = np.ascontiguousarray(graph["olabel"], np.int32)
                _lib.check(L.pk2_decode_graph_create_words(int(graph["num_states"]), int(graph["start"]), src.shape[0],
                                                           src.ctypes.data, dst.ctypes.data, il.ctypes.data, ol.ctypes.data,
                                                           w.ctypes.data, fin.ctypes.data, C.byref(h)))
            else:
                _lib.check(L.pk2_decode_graph_create(int(graph["num_states"]), int(graph["start"]), src.shape[0],
                                                     src.ctypes.data, dst.ctypes.data, il.ctypes.data, w.ctypes.data,
                                                     fin.ctypes.data, C.byref(h)))
        self._h = h
        ns, na, mi = C.c_int32(), C.c_int64(), C.c_int32()
        _lib.check(L.pk2_decode_graph_info(h, C.byref(ns), C.byref(na), C.byref(mi)))
        self.num_states, self.num_arcs, self.max_ilabel = ns.value, na.value, mi.value

    def link_words(self, src_state, dst_state, tid, graph_cost):
        """Word id (HCLG output label) of the arc behind each lattice link (host arrays of LatticeBatch.export)."""
        src_state = np.ascontiguousarray(src_state, np.int32); dst_state = np.ascontiguousarray(dst_state, np.int32)
        tid = np.ascontiguousarray(tid, np.int32); graph_cost = np.ascontiguousarray(graph_cost, np.float32)
        out = np.empty(src_state.shape[0], np.int32)
        _lib.check(_lib.lib().pk2_decode_graph_link_words(self._h, src_state.shape[0], src_state.ctypes.data, dst_state.ctypes.data,
                                                          tid.ctypes.data, graph_cost.ctypes.data, out.ctypes.data))
        return out

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.lib().pk2_decode_graph_destroy(self._h)
                self._h = None
        except Exception:
            pass


class LatticeBatch:
    """The lattices of one minibatch, resident in a device workspace."""

    def __init__(self, handle, workspace, lengths, device, trans_model, num_pdfs, graph=None):
        self._graph = graph   # the C batch object points into the graph handle: keep it alive
        self._h, self.workspace, self.lengths = handle, workspace, list(lengths)
        self.device, self.trans_model, self.num_pdfs = device, trans_model, num_pdfs
        self.status = self.num_tokens = self.num_links = self.best_cost = None
        self.time_major = False
        self._acoustic_scale = 1.0
        self._lattice_beam = 10.0

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.lib().pk2_lattice_batch_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def _ref(self, trans_ids):
        N, Tmax = len(self.lengths), max(self.lengths)
        ref = np.zeros((N, Tmax), np.int32)
        for n, ids in enumerate(trans_ids):
            ids = np.asarray(ids, np.int64).reshape(-1)
            assert ids.shape[0] >= self.lengths[n], "alignment of utterance %d is shorter than its %d frames" % (n, self.lengths[n])
            assert ids.min() >= 1 and ids.max() <= self.trans_model.num_transition_ids()
            ref[n, :self.lengths[n]] = ids[:self.lengths[n]]
        return torch.from_numpy(ref).to(self.device)

    def _zeros_post(self):
        # same memory layout as the log-likelihoods (time-major when the model produced them time-major)
        N, Tmax = len(self.lengths), max(self.lengths)
        if self.time_major:
            return torch.zeros(Tmax, N, self.num_pdfs, device=self.device).transpose(0, 1)
        return torch.zeros(N, Tmax, self.num_pdfs, device=self.device)

    def mmi(self, trans_ids, lm_scale=1.0, acoustic_scale=0.2, drop_frames=True):
        """-> (lat_like f64 [N], post f32 [N, Tmax, P] = numerator - denominator posteriors)."""
        N, Tmax = len(self.lengths), max(self.lengths)
        ref = self._ref(trans_ids)
        post = self._zeros_post()
        out = torch.empty(N, dtype=torch.float64, device=self.device)
        t2p, _ = self.trans_model.device_tables(self.device)
        _lib.check(_lib.lib().pk2_lattice_mmi(self._h, _lib.ptr(self.workspace), _lib.ptr(ref), ref.stride(0), _lib.ptr(t2p),
                                              float(lm_scale), float(acoustic_scale), int(bool(drop_frames)), _lib.ptr(post),
                                              post.stride(0), post.stride(1), _lib.ptr(out), _lib.stream_ptr(self.device)))
        return out, post

    def mpe(self, trans_ids, criterion, silence_phones, one_silence_class=True, lm_scale=1.0, acoustic_scale=1.0):
        """-> (expected frame accuracy f64 [N], post f32 [N, Tmax, P])."""
        assert criterion in ("smbr", "mpfe")
        N, Tmax = len(self.lengths), max(self.lengths)
        ref = self._ref(trans_ids)
        post = self._zeros_post()
        out = torch.empty(N, dtype=torch.float64, device=self.device)
        t2p, t2ph = self.trans_model.device_tables(self.device)
        sil = self.trans_model.silence_mask(silence_phones, self.device)
        _lib.check(_lib.lib().pk2_lattice_mpe(self._h, _lib.ptr(self.workspace), _lib.ptr(ref), ref.stride(0), _lib.ptr(t2p),
                                              _lib.ptr(t2ph), _lib.ptr(sil), 1 if criterion == "mpfe" else 0,
                                              int(bool(one_silence_class)), float(lm_scale), float(acoustic_scale),
                                              _lib.ptr(post), post.stride(0), post.stride(1), _lib.ptr(out),
                                              _lib.stream_ptr(self.device)))
        return out, post

    def export(self, n):
        """Pruned lattice of utterance n as host arrays (tests / tooling)."""
        L = _lib.lib()
        nt, nl = C.c_int32(), C.c_int32()
        sp = _lib.stream_ptr(self.device)
        _lib.check(L.pk2_lattice_export(self._h, _lib.ptr(self.workspace), n, C.byref(nt), C.byref(nl), None, None, None,
                                        None, None, None, None, None, None, sp))
        a = dict(tok_frame=np.empty(nt.value, np.int32), tok_state=np.empty(nt.value, np.int32),
                 tok_cost=np.empty(nt.value, np.float32), tok_final=np.empty(nt.value, np.float32),
                 link_src=np.empty(nl.value, np.int32), link_dst=np.empty(nl.value, np.int32),
                 link_tid=np.empty(nl.value, np.int32), link_graph=np.empty(nl.value, np.float32),
                 link_ac=np.empty(nl.value, np.float32))
        _lib.check(L.pk2_lattice_export(self._h, _lib.ptr(self.workspace), n, C.byref(nt), C.byref(nl),
                                        a["tok_frame"].ctypes.data, a["tok_state"].ctypes.data, a["tok_cost"].ctypes.data,
                                        a["tok_final"].ctypes.data, a["link_src"].ctypes.data, a["link_dst"].ctypes.data,
                                        a["link_tid"].ctypes.data, a["link_graph"].ctypes.data, a["link_ac"].ctypes.data, sp))
        return a


    def compact_lattice(self, n, acoustic_scale=None, determinize=False, beam=None, max_states=2000000):
        """`determinize=True`: the determinised form the reference's recogniser returns (determinize_lattice = True,
        bin/latgen.py:149) -- see determinize_lattice(); `beam` defaults to the decoder's lattice beam, costs are taken with
        the acoustic scale applied (as Kaldi prunes) and written with it removed."""
        raw = self._compact_lattice_raw(n, acoustic_scale)
        if not determinize:
            return raw
        sc = self._acoustic_scale if acoustic_scale is None else acoustic_scale
        det = determinize_lattice(raw, self._lattice_beam if beam is None else beam, max_states, acoustic_scale=sc)
        for k in ("best_words", "best_tids", "best_cost"):
            det[k] = raw[k]
        return det

    def _compact_lattice_raw(self, n, acoustic_scale=None):
        """Utterance n as a Kaldi CompactLattice (what `decoder_out["lattice"]` holds at reference bin/latgen.py:181): state =
        lattice token, arc = lattice link with ilabel = olabel = word id of the HCLG arc, weight (graph cost, acoustic
        cost with the acoustic scale removed) and the transition-id string ([tid], empty for epsilon links) -- Kaldi's
        ConvertLattice(Lattice -> CompactLattice) of the raw state-level lattice.  NOT determinised: the reference asks
        PyKaldi for `determinize_lattice = True` (DeterminizeLatticePhonePrunedWrapper); every Kaldi lattice tool reads
        this form too (lattice-determinize-pruned produces the reference's).  Returns a dict of host arrays for
        kaldi_io.CompactLatticeWriter, plus the best path: words, transition-ids and its total cost."""
        A = self.export(n)
        fr, st = A["tok_frame"], A["tok_state"]
        T = self.lengths[n]
        words = self._graph.link_words(st[A["link_src"]], st[A["link_dst"]], A["link_tid"], A["link_graph"]) \
            if self._graph is not None else np.zeros(A["link_src"].shape[0], np.int32)
        if (words < 0).any():
            raise _lib.Pk2Error("lattice link without a matching HCLG arc")
        last = fr == T
        fin = A["tok_final"].astype(np.float32).copy()
        if not np.isfinite(fin[last]).any():
            fin[last] = 0.0             # no final state reached: every surviving token is final (Kaldi's convention)
        fin[~last] = np.inf
        # token 0..: the start token is the frame-0 token with cost 0 reached by no link
        indeg = np.bincount(A["link_dst"], minlength=fr.shape[0])
        starts = np.flatnonzero((fr == 0) & (indeg == 0))
        start = int(starts[0]) if starts.size else 0
        # best path (for decoder_out["text"] / ["likelihood"]): tok_cost is the best forward cost of each token
        sc = self._acoustic_scale if acoustic_scale is None else acoustic_scale
        total = A["tok_cost"].astype(np.float64) + np.where(np.isfinite(fin), fin, np.inf)
        end = int(np.argmin(np.where(last, total, np.inf)))
        cost = A["link_graph"].astype(np.float64) + sc * A["link_ac"].astype(np.float64)
        order = np.argsort(A["link_dst"], kind="stable")
        lo = np.searchsorted(A["link_dst"][order], np.arange(fr.shape[0] + 1))
        path, tok, guard = [], end, 0
        while tok != start and guard < 4 * (T + 1) + 64:
            cand = order[lo[tok]:lo[tok + 1]]
            if cand.size == 0:
                break
            through = A["tok_cost"][A["link_src"][cand]].astype(np.float64) + cost[cand]
            l = int(cand[int(np.argmin(through))])
            path.append(l)
            tok = int(A["link_src"][l])
            guard += 1
        path.reverse()
        return dict(num_states=int(fr.shape[0]), start=start, src=A["link_src"], dst=A["link_dst"], word=words,
                    graph=A["link_graph"], acoustic=A["link_ac"], tid=A["link_tid"], final=fin,
                    best_words=[int(words[l]) for l in path if words[l] > 0],
                    best_tids=[int(A["link_tid"][l]) for l in path if A["link_tid"][l] > 0],
                    best_cost=float(total[end]))


def persistent_decoder_status():
    """(state, abort): state 1 = all frames of an utterance are decoded inside one persistent launch on this device,
    0 = disabled or it failed its host-verified first launch (the launch-per-frame decoder has taken over), -1 = no decode
    yet; abort != 0: a later launch timed out and its utterances were reported "not decoded"."""
    import ctypes
    state, flag = ctypes.c_int32(-1), ctypes.c_uint32(0)
    _lib.check(_lib.lib().pk2_lattice_persist_status(ctypes.byref(state), ctypes.byref(flag)))
    return int(state.value), int(flag.value)


class MappedLatticeFasterRecognizer:
    """On-the-fly lattice generator with the reference's construction signature
    (bin/train_se.py:180-183: `MappedLatticeFasterRecognizer.from_files(trans_model, HCLG, words_txt,
    acoustic_scale=..., decoder_opts=...)`)."""

    def __init__(self, trans_model, graph, acoustic_scale=0.1, decoder_opts=None):
        self.trans_model = trans_model
        self.graph = graph if isinstance(graph, DecodeGraph) else DecodeGraph(graph)
        self.acoustic_scale = float(acoustic_scale)
        self.decoder_opts = decoder_opts or LatticeFasterDecoderOptions()
        assert self.graph.max_ilabel <= trans_model.num_transition_ids(), "HCLG uses transition-ids the model lacks"
        self._grow = 1

    @classmethod
    def from_files(cls, trans_model_path, graph_path, words_txt=None, acoustic_scale=0.1, decoder_opts=None):
        return cls(TransitionModel.read(trans_model_path), DecodeGraph(graph_path), acoustic_scale, decoder_opts)

    def _opts(self, grow):
        o = self.decoder_opts
        max_active = int(min(o.max_active, 2 ** 31 - 1))
        # a frame holds the <= max_active expanded tokens' successors: 2x max_active covers the measured ~1.2x with room
        tpf = o.tokens_per_frame or min(2 * max_active, 40000)
        lpf = o.links_per_frame or 3 * tpf
        return _lib.DecoderOpts(float(o.beam), float(o.lattice_beam), float(o.beam_delta), self.acoustic_scale,
                                max_active, int(o.min_active), int(min(tpf * grow, 2 ** 30)), int(min(lpf * grow, 2 ** 30)))

    def _workspace(self, nbytes, dev):
        """The device workspace of a minibatch's lattices (15-20 GB at the lattice-MMI configuration: the token and link pools of
        every frame).  A fresh torch.empty per step sends blocks of that size through the caching allocator in ever different
        sizes; it splits them and now and then goes back to hipMalloc INSIDE a step (measured: one call in three timed steps,
        40 ms when it is lucky, 600 ms when cached blocks have to be released first: 305 instead of 147 ms per step on the bench
        line, twice in four runs).  The recognizer therefore keeps TWO workspaces (the LatticeBatch of the step before is usually
        still referenced -- by the loss tensor's graph -- when the next minibatch is decoded; the one before that is gone), each
        1.25 x the largest request it has seen (2 x 24 GB at the bench configuration), and hands out one nobody else holds; sizes
        settle within the first calls that see the largest minibatch.
        PK2_LAT_WS_CACHE=0: a fresh tensor per call."""
        import sys
        if os.environ.get("PK2_LAT_WS_CACHE", "1") == "0":
            return torch.empty(nbytes, dtype=torch.uint8, device=dev)
        slots = self.__dict__.setdefault("_ws_slots", [None, None])
        self._ws_max = max(getattr(self, "_ws_max", 0), int(nbytes))
        free = None
        for k in range(2):
            c = slots[k]
            if c is None:
                free = k if free is None else free
                continue
            idle = sys.getrefcount(c) <= 3          # (the list, `c`, the call's argument)
            if idle and c.device == dev and c.numel() >= self._ws_max:
                return c
            if idle and free is None:
                free = k
            del c
        if free is None:                            # (both handed out and alive: the caller keeps several LatticeBatches)
            return torch.empty(nbytes, dtype=torch.uint8, device=dev)
        slots[free] = None                          # (too small, or empty: back to the allocator before the larger one is requested)
        slots[free] = torch.empty(int(1.25 * self._ws_max), dtype=torch.uint8, device=dev)
        # the other slot follows at once (when nobody holds it): a loop that keeps the step's loss alive needs it in its next
        # step -- inside whatever is being timed -- while a loop that does not would never create it
        o = slots[1 - free]
        if o is None or (sys.getrefcount(o) <= 3 and o.numel() < self._ws_max):
            del o
            slots[1 - free] = None
            slots[1 - free] = torch.empty(int(1.25 * self._ws_max), dtype=torch.uint8, device=dev)
        return slots[free]

    def decode_batch(self, loglikes, lengths):
        """loglikes: CUDA f32 [N, Tmax, P] (unit pdf stride); lengths: frames per utterance.
        Returns a LatticeBatch.  Lattice pools that turn out too small are quadrupled and the decode repeated."""
        _lib.require_gpu()
        assert loglikes.dim() == 3 and loglikes.stride(2) == 1 and loglikes.dtype == torch.float32
        L = _lib.lib()
        dev = loglikes.device
        N, P = loglikes.shape[0], loglikes.shape[2]
        lens = np.ascontiguousarray(lengths, np.int32)
        assert lens.shape[0] == N and lens.max() == loglikes.shape[1], "pad the minibatch to its longest utterance"
        t2p, _ = self.trans_model.device_tables(dev)
        grow = self._grow
        for _ in range(8):
            h = C.c_void_p()
            opts = self._opts(grow)
            _lib.check(L.pk2_lattice_batch_create(self.graph._h, lens.ctypes.data, N, C.byref(opts), C.byref(h)))
            ws = self._workspace(int(L.pk2_lattice_batch_bytes(h)), dev)
            batch = LatticeBatch(h, ws, lens.tolist(), dev, self.trans_model, P, self.graph)
            batch._acoustic_scale = self.acoustic_scale
            batch._lattice_beam = float(self.decoder_opts.lattice_beam)
            _lib.check(L.pk2_lattice_decode(h, _lib.ptr(loglikes), loglikes.stride(0), loglikes.stride(1), P, _lib.ptr(t2p),
                                            self.trans_model.num_transition_ids(), _lib.ptr(ws), _lib.stream_ptr(dev)))
            status = np.zeros(N, np.int32); ntok = np.zeros(N, np.int32); nlink = np.zeros(N, np.int32)
            best = np.zeros(N, np.float32)
            rc = L.pk2_lattice_summary(h, _lib.ptr(ws), status.ctypes.data, ntok.ctypes.data, nlink.ctypes.data,
                                       best.ctypes.data, _lib.stream_ptr(dev))
            batch.status, batch.num_tokens, batch.num_links, batch.best_cost = status, ntok, nlink, best
            batch.time_major = N > 1 and loglikes.stride(0) < loglikes.stride(1)
            if rc == 0:
                self._grow = grow      # later minibatches start from pools that were large enough
                return batch
            if not np.all((status == 0) | (status == 1) | (status == 2)):
                _lib.check(rc)
            del batch, ws
            grow *= 4
        _lib.check(rc)

    def decode(self, loglikes):
        """One utterance [T, P] -> LatticeBatch of one (the reference's per-utterance call, ops/ops.py:55)."""
        return self.decode_batch(loglikes.unsqueeze(0), [loglikes.shape[0]])


def determinize_lattice(lat, beam, max_states=2000000, acoustic_scale=1.0):
    """Determinises a raw state-level lattice (the dict LatticeBatch.compact_lattice returns: arcs src / dst / word / tid /
    graph / acoustic, per-state final costs) on its word labels -- pk2_lattice_determinize, the host-side restatement of
    Kaldi's DeterminizeLatticePruned that replaces PyKaldi's `determinize_lattice = True` (reference bin/latgen.py:149).
    Every word sequence of the input survives once, with the costs and the transition-id alignment of its best path;
    what lies more than `beam` above the best path (total cost = graph + acoustic_scale * acoustic) is dropped.  When the
    construction exceeds `max_states` the beam is halved (at most 6 times), as Kaldi's wrapper retries with a tighter beam.
    Returns a dict in the writer's format: arcs carry transition-id STRINGS (`tid_off`, `tids`), finals a cost pair."""
    import ctypes as C
    L = _lib.lib()
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    src, dst, word, tid = i32(lat["src"]), i32(lat["dst"]), i32(lat["word"]), i32(lat["tid"])
    graph, ac = f32(lat["graph"]), f32(np.asarray(lat["acoustic"], np.float64) * acoustic_scale)
    fin = f32(lat["final"])
    h = C.c_void_p()
    b = float(beam)
    for attempt in range(7):
        rc = L.pk2_lattice_determinize(int(lat["num_states"]), int(lat["start"]), src.shape[0], _lib.ptr(src), _lib.ptr(dst),
                                       _lib.ptr(word), _lib.ptr(tid), _lib.ptr(graph), _lib.ptr(ac), _lib.ptr(fin), b,
                                       int(max_states), C.byref(h))
        if rc == 0:
            break
        if rc != -3 or attempt == 6:         # PK2_ERR_LIMIT = -3: too many states at this beam
            _lib.check(rc)
        b *= 0.5
    try:
        ns, st, na, nt, nf = C.c_int32(), C.c_int32(), C.c_int64(), C.c_int64(), C.c_int64()
        _lib.check(L.pk2_det_lattice_sizes(h, C.byref(ns), C.byref(st), C.byref(na), C.byref(nt), C.byref(nf)))
        ns, na, nt, nf = ns.value, na.value, nt.value, nf.value
        o = dict(src=np.empty(na, np.int32), dst=np.empty(na, np.int32), word=np.empty(na, np.int32),
                 graph=np.empty(na, np.float32), acoustic=np.empty(na, np.float32), tid_off=np.empty(na + 1, np.int64),
                 tids=np.empty(max(1, nt), np.int32), final=np.empty(ns, np.float32), final_acoustic=np.empty(ns, np.float32),
                 final_tid_off=np.empty(ns + 1, np.int64), final_tids=np.empty(max(1, nf), np.int32))
        _lib.check(L.pk2_det_lattice_export(h, *[_lib.ptr(o[k]) for k in ("src", "dst", "word", "graph", "acoustic", "tid_off",
                                                                          "tids", "final", "final_acoustic", "final_tid_off",
                                                                          "final_tids")]))
    finally:
        L.pk2_det_lattice_destroy(h)
    o["tids"], o["final_tids"] = o["tids"][:nt], o["final_tids"][:nf]
    if acoustic_scale != 1.0 and acoustic_scale != 0.0:       # stored with the acoustic scale removed, like the raw lattice
        o["acoustic"] = (o["acoustic"].astype(np.float64) / acoustic_scale).astype(np.float32)
        o["final_acoustic"] = np.where(np.isfinite(o["final_acoustic"]), o["final_acoustic"].astype(np.float64) / acoustic_scale,
                                       np.inf).astype(np.float32)
    o.update(num_states=ns, start=st.value, beam_used=b, determinized=True)
    return o
