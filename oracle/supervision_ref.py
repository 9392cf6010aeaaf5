"""CPU oracle for the chain supervision of one utterance -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
path (pykaldi2_amd/) never does.

What it restates
----------------
Per utterance the reference runs (bin/train_chain.py:262-272, options :184-188)

    phone_ali = aligner.to_phone_alignment(trans_ids)                 # Kaldi SplitToPhones
    proto     = kaldi_chain.alignment_to_proto_supervision(opts, phones, durations)
    sup       = kaldi_chain.proto_supervision_to_supervision(tree, trans_model, proto, True)

Kaldi / PyKaldi are third-party, un-vendored and un-pinned (reference docker/Dockerfile:57-64) and absent from
/root/reference, so this file restates Kaldi's published algorithms [upstream knowledge]: hmm/hmm-utils.cc
(SplitToPhones, IsReordered, GetHmmAsFsa, AddSelfLoops with reorder), hmm/transition-model.cc (ComputeDerived:
the numbering of transition-states and transition-ids), tree/context-dep.cc + tree/event-map.cc
(ContextDependency::Compute), chain/chain-supervision.cc (AlignmentToProtoSupervision,
ProtoSupervisionToSupervision, TimeEnforcerFst).

It works at the TRANSITION-ID level, as Kaldi does -- phone instance -> HMM without self-loops whose arcs carry
transition-ids, every arc followed by any number of the self-loop of its transition-state (reorder = true), a
label passes frame t when the phone of its transition-id is allowed there, labels mapped to pdf-ids at the end
-- whereas the product (csrc/chain_sup.hip) builds the frame-synchronous acceptor directly from the tree's
pdfs.  The two are compared as languages: the multiset of accepted pdf sequences (``label_sequences``, tiny
cases) and the number of accepted paths (``count_paths``, larger cases).

PARITY UNPINNED at the Kaldi boundary: the reference holds no golden vectors, tests or fixtures for this path
(SURVEY.md 8c).  One assumption is not checkable here: Kaldi's final FST is compared as a path multiset, so if
the tree ties the pdfs of two different HMM paths of one phone sequence Kaldi's result (whether or not it
determinizes) may count that pdf sequence once where this oracle and the product count it per HMM path.
"""
import bisect


class TransitionModelRef:
    """TransitionModel::ComputeDerived restated: tuples (phone, hmm_state, forward pdf, self-loop pdf) sorted,
    transition-state = 1 + index of the tuple, transition-ids number the transitions of each transition-state's
    HMM state consecutively (1-based)."""

    def __init__(self, phone2entry, entries, tuples):
        self.phone2entry = dict(phone2entry)
        self.entries = entries                       # entries[e][s] = (fwd_class, loop_class, [dst states])
        self.tuples = [tuple(int(v) for v in t) for t in tuples]
        assert self.tuples == sorted(self.tuples), "Kaldi keeps the tuples sorted"
        self.first_tid = [None]
        self.tid2ts = [0]
        for ts, (phone, hs, _, _) in enumerate(self.tuples, start=1):
            self.first_tid.append(len(self.tid2ts))
            self.tid2ts.extend([ts] * len(self.topo(phone)[hs][2]))

    def topo(self, phone):
        return self.entries[self.phone2entry[phone]]

    def tuple_to_tstate(self, phone, hs, fwd_pdf, loop_pdf):
        i = bisect.bisect_left(self.tuples, (phone, hs, fwd_pdf, loop_pdf))
        if i == len(self.tuples) or self.tuples[i] != (phone, hs, fwd_pdf, loop_pdf):
            raise KeyError("TupleToTransitionState: no tuple %r" % ((phone, hs, fwd_pdf, loop_pdf),))
        return i + 1

    def pair_to_tid(self, ts, index):
        return self.first_tid[ts] + index

    def _where(self, tid):
        ts = self.tid2ts[tid]
        phone, hs, fwd_pdf, loop_pdf = self.tuples[ts - 1]
        dst = self.topo(phone)[hs][2][tid - self.first_tid[ts]]
        return ts, phone, hs, fwd_pdf, loop_pdf, dst

    def tid_to_phone(self, tid):
        return self._where(tid)[1]

    def is_self_loop(self, tid):
        w = self._where(tid)
        return w[5] == w[2]

    def is_final(self, tid):
        w = self._where(tid)
        return w[5] == len(self.topo(w[1])) - 1

    def tid_to_pdf(self, tid):
        w = self._where(tid)
        return w[4] if w[5] == w[2] else w[3]

    def self_loop_of(self, ts):
        phone, hs, _, _ = self.tuples[ts - 1]
        dsts = self.topo(phone)[hs][2]
        return self.first_tid[ts] + dsts.index(hs) if hs in dsts else 0


def split_to_phones(tm, ali):
    """hmm-utils.cc SplitToPhones (+ IsReordered) -> (ok, [[tids of one phone], ...])."""
    def is_reordered():
        for i in range(len(ali) - 1):
            if tm.tid2ts[ali[i]] != tm.tid2ts[ali[i + 1]]:
                continue
            a, b = tm.is_self_loop(ali[i]), tm.is_self_loop(ali[i + 1])
            if a and not b:
                return False
            if not a and b:
                return True
        return True
    reordered = is_reordered()
    ok, ends = True, []
    i = 0
    while i < len(ali):
        if tm.is_final(ali[i]):
            if reordered:
                while i + 1 < len(ali) and tm.is_self_loop(ali[i + 1]) and tm.tid2ts[ali[i + 1]] == tm.tid2ts[ali[i]]:
                    i += 1
            ends.append(i + 1)
        elif i + 1 == len(ali):
            ok = False
            ends.append(i + 1)
        else:
            if tm.tid2ts[ali[i]] != tm.tid2ts[ali[i + 1]] and tm.tid_to_phone(ali[i]) != tm.tid_to_phone(ali[i + 1]):
                ok = False
                ends.append(i + 1)
        i += 1
    out, b = [], 0
    for e in ends:
        out.append(list(ali[b:e]))
        b = e
    return ok, out


def alignment_to_proto_supervision(phones, durations, factor=3, left_tolerance=5, right_tolerance=5):
    """chain-supervision.cc AlignmentToProtoSupervision -> allowed_phones[t] (sorted, unique) per subsampled frame."""
    T = sum(durations)
    Tp = (T + factor - 1) // factor
    allowed = [[] for _ in range(Tp)]
    cur = 0
    for ph, d in zip(phones, durations):
        t0 = max(0, cur - left_tolerance)
        t1 = min(T, cur + d + right_tolerance)
        for t in range((t0 + factor - 1) // factor, (t1 + factor - 1) // factor):
            allowed[t].append(ph)
        cur += d
    return [sorted(set(a)) for a in allowed]


def _hmm_arcs(tm, tree_compute, N, P, phones):
    """The transition-id FSA of the phone sequence before self-loops are added (context expansion + H without
    self-loops, composed with the linear phone acceptor).  FSA states: (instance, hmm_state), ("end",).
    Returns {state: [(tid, next state)]}."""
    arcs = {}
    n = len(phones)
    for i, ph in enumerate(phones):
        window = [phones[i - P + j] if 0 <= i - P + j < n else 0 for j in range(N)]
        topo = tm.topo(ph)
        for hs, (fwd_class, loop_class, dsts) in enumerate(topo[:-1]):
            fwd_pdf, loop_pdf = tree_compute(window, fwd_class), tree_compute(window, loop_class)
            ts = tm.tuple_to_tstate(ph, hs, fwd_pdf, loop_pdf)
            out = arcs.setdefault((i, hs), [])
            for idx, d in enumerate(dsts):
                if d == hs:
                    continue                                  # H is built without self-loops
                nxt = (i, d) if d < len(topo) - 1 else ((i + 1, 0) if i + 1 < n else ("end",))
                out.append((tm.pair_to_tid(ts, idx), nxt))
    return arcs


def label_sequences(tm, tree_compute, N, P, phones, allowed):
    """Every accepted pdf sequence (sorted list, with multiplicity) -- exponential, tiny cases only."""
    arcs = _hmm_arcs(tm, tree_compute, N, P, phones)
    Tp = len(allowed)
    out = []

    def ok(tid, t):
        return t < Tp and tm.tid_to_phone(tid) in allowed[t]

    def walk(state, t, labels):
        if state == ("end",):
            if t == Tp:
                out.append(tuple(labels))
            return
        for tid, nxt in arcs[state]:
            if not ok(tid, t):
                continue
            seq = labels + [tm.tid_to_pdf(tid)]
            walk(nxt, t + 1, seq)
            loop = tm.self_loop_of(tm.tid2ts[tid])           # reorder: the self-loops follow the transition
            tt = t + 1
            while loop and ok(loop, tt):
                seq = seq + [tm.tid_to_pdf(loop)]
                tt += 1
                walk(nxt, tt, seq)

    walk((0, 0), 0, [])
    return sorted(out)


def count_paths(tm, tree_compute, N, P, phones, allowed):
    """Number of accepted transition-id sequences by a frame-synchronous recursion over "the arc being held"."""
    arcs = _hmm_arcs(tm, tree_compute, N, P, phones)
    Tp = len(allowed)
    flat = [(s, tid, nxt) for s, lst in arcs.items() for tid, nxt in lst]
    leaving = {}
    for k, (s, _, _) in enumerate(flat):
        leaving.setdefault(s, []).append(k)
    cur = {}
    for k in leaving.get((0, 0), []):
        if tm.tid_to_phone(flat[k][1]) in allowed[0]:
            cur[k] = cur.get(k, 0) + 1
    for t in range(1, Tp):
        nxt = {}
        for k, c in cur.items():
            _, tid, dst = flat[k]
            loop = tm.self_loop_of(tm.tid2ts[tid])
            if loop and tm.tid_to_phone(loop) in allowed[t]:
                nxt[k] = nxt.get(k, 0) + c
            for k2 in leaving.get(dst, []):
                if tm.tid_to_phone(flat[k2][1]) in allowed[t]:
                    nxt[k2] = nxt.get(k2, 0) + c
        cur = nxt
    return sum(c for k, c in cur.items() if flat[k][2] == ("end",))


def fst_label_sequences(sup):
    """Accepted label sequences of a built supervision (arrays src/dst/pdf, final_states; state 0 initial)."""
    out_arcs = {}
    for s, d, p in zip(sup.src.tolist(), sup.dst.tolist(), sup.pdf.tolist()):
        out_arcs.setdefault(s, []).append((p, d))
    finals = set(sup.final_states.tolist())
    out = []

    def walk(state, labels):
        if state in finals and len(labels) == sup.frames_per_sequence:
            out.append(tuple(labels))
        for p, d in out_arcs.get(state, []):
            walk(d, labels + [p])

    walk(0, [])
    return sorted(out)


def fst_count_paths(sup):
    cnt = {0: 1}
    order = sorted(range(len(sup.src)), key=lambda k: sup.state_time[sup.src[k]])
    for k in order:     # arcs sorted by source time: a state's count is complete before its arcs are used
        s, d = int(sup.src[k]), int(sup.dst[k])
        cnt[d] = cnt.get(d, 0) + cnt.get(s, 0)
    return sum(cnt.get(int(f), 0) for f in sup.final_states)
