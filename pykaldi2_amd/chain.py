"""Host-side mirror of the ``kaldi.chain`` objects the reference's LF-MMI path uses
(reference bin/train_chain.py:184-202, ops/ops.py:243-280), backed by libpk2hip.so.

  DenominatorGraph(den_fst, num_pdfs)       bin/train_chain.py:167,202
  ChainTrainingOptions                       bin/train_chain.py:191-193
  SupervisionOptions                         bin/train_chain.py:184-188
  Supervision                                bin/train_chain.py:271-272
  MappedAligner.to_phone_alignment           bin/train_chain.py:195-200,263
  alignment_to_proto_supervision             bin/train_chain.py:271
  proto_supervision_to_supervision           bin/train_chain.py:272
  compute_chain_objf_and_deriv(...)          ops/ops.py:265

``den_fst`` may be a path to an OpenFst binary ``den.fst`` or a dict of arc
arrays (``num_states, start, src, dst, pdf, prob`` as produced by
pykaldi2_amd.synth.den_graph_arcs).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


class ChainTrainingOptions:
    """kaldi.chain.ChainTrainingOptions (fields used by the reference)."""

    def __init__(self, leaky_hmm_coefficient=1.0e-05, xent_regularize=0.0, l2_regularize=0.0):
        self.leaky_hmm_coefficient = leaky_hmm_coefficient
        self.xent_regularize = xent_regularize
        self.l2_regularize = l2_regularize


class SupervisionOptions:
    """kaldi.chain.SupervisionOptions (reference bin/train_chain.py:184-188)."""

    def __init__(self):
        self.convert_to_pdfs = True
        self.frame_subsampling_factor = 3
        self.left_tolerance = 5
        self.right_tolerance = 5


class DenominatorGraph:
    def __init__(self, den_fst, num_pdfs):
        L = _lib.lib()
        h = C.c_void_p()
        if isinstance(den_fst, (str, bytes)):
            path = den_fst.encode() if isinstance(den_fst, str) else den_fst
            _lib.check(L.pk2_den_graph_from_openfst(path, int(num_pdfs), C.byref(h)))
        else:
            src = np.ascontiguousarray(den_fst["src"], dtype=np.int32)
            dst = np.ascontiguousarray(den_fst["dst"], dtype=np.int32)
            pdf = np.ascontiguousarray(den_fst["pdf"], dtype=np.int32)
            prob = np.ascontiguousarray(den_fst["prob"], dtype=np.float32)
            _lib.check(L.pk2_den_graph_create(int(den_fst["num_states"]), int(num_pdfs), src.shape[0],
                                              _lib.ptr(src), _lib.ptr(dst), _lib.ptr(pdf), _lib.ptr(prob),
                                              int(den_fst.get("start", 0)), C.byref(h)))
        self._h = h
        s, p, a = C.c_int32(), C.c_int32(), C.c_int64()
        _lib.check(L.pk2_den_graph_info(h, C.byref(s), C.byref(p), C.byref(a)))
        self._num_states, self._num_pdfs, self._num_arcs = s.value, p.value, a.value

    def num_states(self):
        return self._num_states

    def num_pdfs(self):
        return self._num_pdfs

    def num_arcs(self):
        return self._num_arcs

    def initial_probs(self):
        out = np.empty(self._num_states, dtype=np.float32)
        _lib.check(_lib.lib().pk2_den_graph_initial_probs(self._h, _lib.ptr(out)))
        return out

    def kernel_path(self, num_seqs):
        """0 = per-arc-pdf kernels, 1 = state-x kernels (a launch per frame), 2 = persistent recursion kernel."""
        return int(_lib.lib().pk2_den_graph_path(self._h, int(num_seqs)))

    def persist_form(self, num_seqs):
        """Form of the persistent kernel behind kernel_path() == 2: 1 = everything resident, 2 = chunked table + streamed
        overflow (csrc/chain_den_persist2.hip); 0 = none."""
        return int(_lib.lib().pk2_den_graph_persist_form(self._h, int(num_seqs)))

    def debug_persist2(self, which):
        """Second persistent layout of an ordering (test hook; csrc/chain_internal.h: HostPersist2); None if absent."""
        L = _lib.lib()
        info = np.zeros(32, dtype=np.int32)
        _lib.check(L.pk2_den_graph_debug_persist2(self._h, which, _lib.ptr(info), *([None] * 17)))
        if not info[0]:
            return None
        R, T, K, W, SP, SEG = (int(v) for v in info[26:32])
        MC = SEG - 2
        pieces, rows = int(info[7]), int(info[9])
        out = dict(prob=np.empty((R, K, T), np.float32), idx2=np.empty((R, K // 2, T), np.uint32),
                   ends=np.empty((R, 2, T), np.uint32), first_row=np.empty((R, 2, T), np.int32),
                   uncovered=np.empty((R, 2), np.int32), ncomp=np.empty((R, 2), np.int32), rmap=np.empty((2, rows), np.int16),
                   pbeg=np.empty((R, MC + 1), np.int32),
                   sprob=np.empty((pieces, T, SP), np.float32), sidx2=np.empty((pieces, T, SP // 2), np.uint32),
                   sends=np.empty((pieces, T), np.uint32), sfirst_row=np.empty((R, MC, T), np.int32),
                   wcrow=np.empty((R, SEG, W), np.int32), row_begin=np.empty(R + 1, np.int32),
                   grp_begin=np.empty(R + 1, np.int32), row_leak=np.empty(rows, np.float32), row_psum=np.empty(rows, np.float32))
        order = ("prob", "idx2", "ends", "first_row", "uncovered", "ncomp", "rmap", "pbeg", "sprob", "sidx2", "sends", "sfirst_row", "wcrow",
                 "row_begin", "grp_begin", "row_leak", "row_psum")
        _lib.check(L.pk2_den_graph_debug_persist2(self._h, which, _lib.ptr(info), *[_lib.ptr(out[k]) for k in order]))
        out.update(estep=int(info[1]), K=int(info[2]), R=int(info[3]), tfloats=int(info[4]), max_rows=int(info[5]),
                   max_groups=int(info[6]), pieces=pieces, cap=int(info[8]), cbeg=[int(v) for v in info[10:10 + MC + 1]],
                   lds_off=[int(v) for v in info[20:20 + MC]], SP=SP, W=W, T=T, slots=K, row_arrays=int(info[19]))
        return out

    def debug_persist(self, which):
        """Layout of an ordering for the persistent kernel (test hook): which = 0 forward, 1 backward; None when the
        graph does not fit it."""
        L = _lib.lib()
        info = np.zeros(9, dtype=np.int32)
        _lib.check(L.pk2_den_graph_debug_persist(self._h, which, _lib.ptr(info), None, None, None, None, None, None, None, None))
        if not info[0]:
            return None
        R, T, K, W, rows = (int(v) for v in info[3:8])
        out = dict(arcs=np.empty((R, K, T, 2), dtype=np.int32), ends=np.empty((R, T), dtype=np.uint64),
                   first_row=np.empty((R, T), dtype=np.int32), wcrow=np.empty((R, W), dtype=np.int32),
                   row_begin=np.empty(R + 1, dtype=np.int32), grp_begin=np.empty(R + 1, dtype=np.int32),
                   row_leak=np.empty(rows, dtype=np.float32), row_psum=np.empty(rows, dtype=np.float32))
        _lib.check(L.pk2_den_graph_debug_persist(self._h, which, _lib.ptr(info), *[_lib.ptr(out[k]) for k in
                   ("arcs", "ends", "first_row", "wcrow", "row_begin", "grp_begin", "row_leak", "row_psum")]))
        out.update(max_rows=int(info[1]), max_groups=int(info[2]), estep=int(info[8]))
        return out

    def debug_ordering(self, which):
        """Host-side work decomposition (test hook): which = 0 by dst, 1 by src, 2 by pdf (general kernels);
        3 = by virtual destination state, 4 = by src gathering virtual destination states (state-x kernels)."""
        L = _lib.lib()
        na, nc = C.c_int64(), C.c_int32()
        _lib.check(L.pk2_den_graph_debug_ordering(self._h, which, C.byref(na), C.byref(nc), None, None,
                                                  None, None, None, None))
        arcs = np.empty((na.value, 4), dtype=np.int32)
        k = int(L.pk2_den_graph_arcs_per_lane())
        meta = np.empty((na.value // k, 2), dtype=np.uint32)      # per lane {first chunk-local row, flush mask}
        wb_off = np.empty(nc.value + 1, dtype=np.int32)
        row0 = np.empty(nc.value, dtype=np.int32)
        nrows = np.empty(nc.value, dtype=np.int32)
        atomic = np.empty(nc.value, dtype=np.int32)
        _lib.check(L.pk2_den_graph_debug_ordering(self._h, which, None, None, _lib.ptr(arcs), _lib.ptr(meta),
                                                  _lib.ptr(wb_off), _lib.ptr(row0), _lib.ptr(nrows),
                                                  _lib.ptr(atomic)))
        out = dict(arcs=arcs, meta=meta, wb_off=wb_off, row0=row0, nrows=nrows, atomic=atomic, arcs_per_lane=k)
        if which >= 3:
            nv, nl, no = C.c_int32(), C.c_int64(), C.c_int32()
            _lib.check(L.pk2_den_graph_debug_virtual(self._h, which, C.byref(nv), None, None, None, None, None,
                                                     C.byref(nl), None, None, None, C.byref(no), None, None))
            S = self._num_states
            voff, ooff = np.empty(S + 1, dtype=np.int32), np.empty(S + 1, dtype=np.int32)
            vpdf, opdf = np.empty(nv.value, dtype=np.int32), np.empty(no.value, dtype=np.int32)
            real0, nreal, slot0 = (np.empty(nc.value, dtype=np.int32) for _ in range(3))
            leak = np.empty(nl.value, dtype=np.float32)
            loop_pdf, loop_prob = np.empty(S, dtype=np.int32), np.empty(S, dtype=np.float32)
            _lib.check(L.pk2_den_graph_debug_virtual(self._h, which, None, _lib.ptr(voff), _lib.ptr(vpdf),
                                                     _lib.ptr(real0), _lib.ptr(nreal), _lib.ptr(slot0), None,
                                                     _lib.ptr(leak), _lib.ptr(loop_pdf), _lib.ptr(loop_prob), None,
                                                     _lib.ptr(ooff), _lib.ptr(opdf)))
            out.update(voff=voff, vpdf=vpdf, real0=real0, nreal=nreal, slot0=slot0, row_leak=leak,
                       loop_pdf=loop_pdf, loop_prob=loop_prob, ooff=ooff, opdf=opdf)
        return out

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.lib().pk2_den_graph_destroy(self._h)
                self._h = None
        except Exception:
            pass


class Supervision:
    """kaldi.chain.Supervision for one utterance (num_sequences = 1): an acyclic FST in which
    every arc consumes one frame and carries a pdf label.  Arrays (numpy, host):
    src/dst int32 (state 0 initial), pdf int32, weight f32 (-log prob), arcs sorted by the
    frame of their source state with frame_offsets[t] = first arc of frame t."""

    def __init__(self, fst, weight=1.0, label_dim=None):
        self.weight = float(weight)
        self.num_sequences = 1
        self.frames_per_sequence = int(fst["frames"])
        self.label_dim = label_dim
        self.num_states = int(fst["num_states"])
        self.src = np.ascontiguousarray(fst["src"], dtype=np.int32)
        self.dst = np.ascontiguousarray(fst["dst"], dtype=np.int32)
        self.pdf = np.ascontiguousarray(fst["pdf"], dtype=np.int32)
        self.arc_weight = np.ascontiguousarray(fst["weight"], dtype=np.float32)
        self.frame_offsets = np.ascontiguousarray(fst["frame_offsets"], dtype=np.int32)
        self.final_states = np.ascontiguousarray(fst["final_states"], dtype=np.int32)
        self.final_weights = np.ascontiguousarray(fst["final_weights"], dtype=np.float32)
        self.state_time = fst.get("state_time")
        assert self.frame_offsets.shape[0] == self.frames_per_sequence + 1


class ProtoSupervision:
    """kaldi.chain.ProtoSupervision: the phone sequence with durations and the tolerance options; the
    allowed-phone sets are produced (with the supervision) by pk2_supervision_create."""

    def __init__(self, opts, phones, durations):
        self.phones = np.ascontiguousarray(phones, dtype=np.int32)
        self.durations = np.ascontiguousarray(durations, dtype=np.int32)
        if self.phones.ndim != 1 or self.phones.shape != self.durations.shape or self.phones.shape[0] == 0:
            raise ValueError("phones and durations must be non-empty lists of equal length")
        self.frame_subsampling_factor = int(opts.frame_subsampling_factor)
        self.left_tolerance, self.right_tolerance = int(opts.left_tolerance), int(opts.right_tolerance)


def alignment_to_proto_supervision(opts, phones, durations):
    """kaldi.chain.alignment_to_proto_supervision (reference bin/train_chain.py:271)."""
    return ProtoSupervision(opts, phones, durations)


class _SupModel:
    """pk2_sup_model of a (tree, transition model) pair."""

    def __init__(self, tree, trans_model):
        if trans_model.entries is None:
            raise ValueError("the transition model carries no topology (read it from 0.trans_mdl / final.mdl)")
        max_phone = max(trans_model.phone2entry)
        p2e = np.full(max_phone + 1, -1, np.int32)
        for ph, e in trans_model.phone2entry.items():
            p2e[ph] = e
        e_off, fwd, loop, t_off, t_dst = [0], [], [], [0], []
        for states in trans_model.entries:
            for f, l, dsts in states:
                fwd.append(f); loop.append(l); t_dst.extend(dsts); t_off.append(len(t_dst))
            e_off.append(len(fwd))
        arr = lambda v: np.ascontiguousarray(v, dtype=np.int32)   # noqa: E731
        e_off, fwd, loop, t_off, t_dst = arr(e_off), arr(fwd), arr(loop), arr(t_off), arr(t_dst)
        tuples = arr(trans_model.tuples)
        L = _lib.lib()
        self._h = L.pk2_sup_model_create(max_phone, _lib.ptr(p2e), len(trans_model.entries), _lib.ptr(e_off),
                                         _lib.ptr(fwd), _lib.ptr(loop), _lib.ptr(t_off), _lib.ptr(t_dst),
                                         tuples.shape[0], _lib.ptr(tuples), tree.N, tree.P, tree.kind.shape[0],
                                         _lib.ptr(tree.kind), _lib.ptr(tree.key), _lib.ptr(tree.a), _lib.ptr(tree.b),
                                         tree.pool.shape[0], _lib.ptr(tree.pool))
        if not self._h:
            raise _lib.Pk2Error(L.pk2_last_error().decode())
        self.label_dim = trans_model.num_pdfs()

    def pdf(self, window, pdf_class):
        w = np.ascontiguousarray(window, dtype=np.int32)
        out = C.c_int32()
        _lib.check(_lib.lib().pk2_sup_model_pdf(self._h, _lib.ptr(w), int(pdf_class), C.byref(out)))
        return out.value

    def __del__(self):
        try:
            if self._h:
                _lib.lib().pk2_sup_model_destroy(self._h)
                self._h = None
        except Exception:
            pass


_sup_models = {}


def supervision_model(tree, trans_model):
    key = (id(tree), id(trans_model))
    hit = _sup_models.get(key)
    if hit is None or hit[0] is not tree or hit[1] is not trans_model:
        hit = _sup_models[key] = (tree, trans_model, _SupModel(tree, trans_model))
    return hit[2]


def proto_supervision_to_supervision(tree, trans_model, proto, convert_to_pdfs=True, with_allowed=False):
    """kaldi.chain.proto_supervision_to_supervision (reference bin/train_chain.py:272).  Raises when no path
    satisfies the time constraints (Kaldi returns false with a warning)."""
    if not convert_to_pdfs:
        raise NotImplementedError("only convert_to_pdfs=True (reference bin/train_chain.py:185) is built")
    m = supervision_model(tree, trans_model)
    L = _lib.lib()
    h = L.pk2_supervision_create(m._h, _lib.ptr(proto.phones), _lib.ptr(proto.durations), proto.phones.shape[0],
                                 proto.frame_subsampling_factor, proto.left_tolerance, proto.right_tolerance)
    if not h:
        raise _lib.Pk2Error(L.pk2_last_error().decode())
    try:
        n = [C.c_int32() for _ in range(5)]
        L.pk2_supervision_sizes(h, *[C.byref(v) for v in n])
        frames, ns, na, nf, nal = (v.value for v in n)
        src, dst, pdf = (np.empty(na, np.int32) for _ in range(3))
        w, foff, stime = np.empty(na, np.float32), np.empty(frames + 1, np.int32), np.empty(ns, np.int32)
        fin, finw = np.empty(nf, np.int32), np.empty(nf, np.float32)
        aoff = np.empty(frames + 1, np.int32) if with_allowed else None
        aph = np.empty(nal, np.int32) if with_allowed else None
        _lib.check(L.pk2_supervision_copy(h, _lib.ptr(src), _lib.ptr(dst), _lib.ptr(pdf), _lib.ptr(w), _lib.ptr(foff),
                                          _lib.ptr(stime), _lib.ptr(fin), _lib.ptr(finw), _lib.ptr(aoff), _lib.ptr(aph)))
    finally:
        L.pk2_supervision_destroy(h)
    sup = Supervision(dict(num_states=ns, frames=frames, src=src, dst=dst, pdf=pdf, weight=w, frame_offsets=foff,
                           state_time=stime, final_states=fin, final_weights=finw), label_dim=m.label_dim)
    if with_allowed:
        sup.allowed_phones = [aph[aoff[t]:aoff[t + 1]].tolist() for t in range(frames)]
    return sup


def split_to_phones(trans_model, alignment):
    """Kaldi's SplitToPhones on a transition-id alignment -> (ok, [(phone, start, duration)])."""
    if trans_model.tid2tstate is None:
        raise ValueError("the transition model carries no topology (read it from final.mdl)")
    ali = np.ascontiguousarray(alignment, dtype=np.int32)
    T = ali.shape[0]
    phones, durs = np.empty(max(T, 1), np.int32), np.empty(max(T, 1), np.int32)
    n, ok = C.c_int32(), C.c_int32()
    _lib.check(_lib.lib().pk2_split_to_phones(_lib.ptr(trans_model.tid2tstate), _lib.ptr(trans_model.tid2phone),
                                              _lib.ptr(trans_model.tid_flags), trans_model.num_transition_ids(),
                                              _lib.ptr(ali), T, _lib.ptr(phones), _lib.ptr(durs), C.byref(n),
                                              C.byref(ok)))
    starts = np.concatenate([[0], np.cumsum(durs[:n.value])[:-1]]) if n.value else []
    return bool(ok.value), [(int(p), int(s), int(d)) for p, s, d in zip(phones[:n.value], starts, durs[:n.value])]


def supervision_from_alignment(aligner, tree, trans_model, opts, trans_ids):
    """The per-utterance block of reference bin/train_chain.py:262-272 as one call."""
    phone_ali = aligner.to_phone_alignment(trans_ids)
    proto = alignment_to_proto_supervision(opts, [item[0] for item in phone_ali], [item[2] for item in phone_ali])
    return proto_supervision_to_supervision(tree, trans_model, proto, opts.convert_to_pdfs)


class MappedAligner:
    """The one use the reference makes of kaldi.alignment.MappedAligner (bin/train_chain.py:195-200,263):
    to_phone_alignment on the transition-ids of the label files.  The decoding side of the aligner (tree, L.fst,
    beams) is not needed for that and is ignored."""

    def __init__(self, trans_model):
        self.transition_model = trans_model

    @classmethod
    def from_files(cls, model_rxfilename, tree_rxfilename=None, lexicon_rxfilename=None, symbols_filename=None,
                   disambig_rxfilename=None, **unused):
        from .lattice import TransitionModel
        return cls(TransitionModel.read(model_rxfilename))

    def to_phone_alignment(self, alignment, phones=None):
        """-> [(phone, start frame, duration)]."""
        return split_to_phones(self.transition_model, alignment)[1]


class _SupervisionBatch:
    """Concatenates per-utterance supervisions and ships them to the device in one
    pinned H2D copy (a few tens of KB)."""

    def __init__(self, sups, device):
        n = len(sups)
        arcs = np.cumsum([0] + [s.src.shape[0] for s in sups])
        self.total_arcs = int(arcs[-1])
        self.state_off = np.cumsum([0] + [s.num_states for s in sups]).astype(np.int32)
        self.final_off = np.cumsum([0] + [s.final_states.shape[0] for s in sups]).astype(np.int32)
        self.lengths = np.asarray([s.frames_per_sequence for s in sups], dtype=np.int32)
        frame_off = np.concatenate([s.frame_offsets + arcs[i] for i, s in enumerate(sups)]).astype(np.int32)
        ints = [np.concatenate([s.src for s in sups]), np.concatenate([s.dst for s in sups]),
                np.concatenate([s.pdf for s in sups]), frame_off,
                np.concatenate([s.final_states for s in sups])]
        flts = [np.concatenate([s.arc_weight for s in sups]), np.concatenate([s.final_weights for s in sups])]
        sizes = [a.shape[0] for a in ints + flts]
        offs = np.cumsum([0] + [(-(-s // 64)) * 64 for s in sizes])
        host = torch.empty(int(offs[-1]), dtype=torch.int32).pin_memory()
        hv = host.numpy()
        for a, o in zip(ints, offs[:5]):
            hv[o:o + a.shape[0]] = a
        for a, o in zip(flts, offs[5:7]):
            hv[o:o + a.shape[0]] = a.view(np.int32)
        self._host = host
        self.dev = host.to(device, non_blocking=True)
        base = self.dev.data_ptr()
        p = [C.c_void_p(base + int(o) * 4) for o in offs[:7]]
        self.struct = _lib.NumBatch(p[0], p[1], p[2], p[5], p[3], _lib.ptr(self.state_off), p[4], p[6],
                                    _lib.ptr(self.final_off), self.total_arcs, int(np.diff(arcs).max()) if n else 0)
        self.n = n


_workspace_cache = {}


def _workspace(device, nbytes):
    key = (device.index if device.index is not None else torch.cuda.current_device())
    ws = _workspace_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = None
        _workspace_cache[key] = None
        ws = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=device)
        _workspace_cache[key] = ws
    return ws


def compute_chain_objf_and_deriv(opts, den_graph, supervisions, nnet_output, lengths=None, operator_form=False):
    """Batched kaldi.chain.compute_chain_objf_and_deriv (reference ops/ops.py:265-267).

    nnet_output: f32 CUDA tensor [N, T, P] (or [T, P] with a single Supervision); row t of
    sequence n is frame t.  Returns (out, grad): out is a device tensor [3, N] =
    (objf, log p_num, log p_den) per sequence, grad is d objf / d nnet_output with
    xent_regularize * numerator posterior already added (zeros on padding frames).
    operator_form=True (ops.ChainObjtiveBatch): out is [3 N + 1] -- the same three rows followed by sum_n objf[n] -- and grad
    holds MINUS the derivative, what the reference operator's backward returns (ops/ops.py:276-280); both come out of the
    library call itself, no torch kernel runs around it.
    """
    _lib.require_gpu()
    if isinstance(supervisions, Supervision):
        supervisions = [supervisions]
    x = nnet_output
    if x.dim() == 2:
        x = x.unsqueeze(0)
    assert x.is_cuda and x.dtype == torch.float32 and x.stride(2) == 1
    N, T, P = x.shape
    assert N == len(supervisions) and P == den_graph.num_pdfs()
    sb = _SupervisionBatch(supervisions, x.device)
    assert int(sb.lengths.max()) <= T
    weight = supervisions[0].weight
    L = _lib.lib()
    Tmax = int(sb.lengths.max())
    bound = max(sb.total_arcs, int(sb.lengths.sum()) + N)
    nbytes = L.pk2_chain_workspace_bytes(den_graph._h, N, Tmax, bound)
    ws = _workspace(x.device, nbytes)
    grad = torch.empty_like(x)
    if Tmax < T:
        grad[:, Tmax:].zero_()
    if operator_form:
        out = torch.empty(3 * N + 1, dtype=torch.float32, device=x.device)
        _lib.check(L.pk2_chain_objf_and_deriv_op(den_graph._h, _lib.ptr(x), x.stride(0), x.stride(1),
                                                 _lib.ptr(sb.lengths), N, C.byref(sb.struct),
                                                 float(opts.leaky_hmm_coefficient), float(opts.xent_regularize),
                                                 float(opts.l2_regularize), float(weight), _lib.ptr(grad),
                                                 grad.stride(0), grad.stride(1), _lib.ptr(out), _lib.ptr(ws),
                                                 ws.numel(), -1.0, _lib.ptr(out[3 * N:]), _lib.stream_ptr(x.device)))
        out._pk2_keepalive = sb
        return out, grad
    out = torch.empty(3, N, dtype=torch.float32, device=x.device)
    _lib.check(L.pk2_chain_objf_and_deriv(den_graph._h, _lib.ptr(x), x.stride(0), x.stride(1),
                                          _lib.ptr(sb.lengths), N, C.byref(sb.struct),
                                          float(opts.leaky_hmm_coefficient), float(opts.xent_regularize),
                                          float(opts.l2_regularize), float(weight), _lib.ptr(grad),
                                          grad.stride(0), grad.stride(1), _lib.ptr(out), _lib.ptr(ws),
                                          ws.numel(), _lib.stream_ptr(x.device)))
    # keep the pinned staging buffer alive until the stream has consumed it
    out._pk2_keepalive = sb
    return out, grad


def den_forward_backward(den_graph, nnet_output, lengths, leaky):
    """Denominator only: (log p_den [N], occupancies [N,T,P]).  Test / profiling hook."""
    _lib.require_gpu()
    x = nnet_output
    N, T, P = x.shape
    lengths = np.ascontiguousarray(lengths, dtype=np.int32)
    L = _lib.lib()
    Tmax = int(lengths.max())
    nbytes = L.pk2_chain_workspace_bytes(den_graph._h, N, Tmax, 0)
    ws = _workspace(x.device, nbytes)
    gamma = torch.zeros_like(x)
    lp = torch.empty(N, dtype=torch.float32, device=x.device)
    _lib.check(L.pk2_chain_den_fwd_bwd(den_graph._h, _lib.ptr(x), x.stride(0), x.stride(1),
                                       _lib.ptr(lengths), N, float(leaky), _lib.ptr(lp), _lib.ptr(gamma),
                                       gamma.stride(0), gamma.stride(1), _lib.ptr(ws), ws.numel(),
                                       _lib.stream_ptr(x.device)))
    return lp, gamma
