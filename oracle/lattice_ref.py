"""CPU oracle for the lattice-based sequence criteria (MMI / sMBR / MPFE) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product path (pykaldi2_amd/) never does.

What it restates
----------------
``MMIFunction`` / ``sMBRFunction`` (reference ops/ops.py:41-75, 119-156) call, per
utterance:

  1. ``asr_decoder.decode(loglikes)``                       on-the-fly lattice generation with Kaldi's
     LatticeFasterDecoder over HCLG (built at reference bin/train_se.py:172-183: beam, lattice_beam,
     max_active, acoustic_scale from the YAML, ``determinize_lattice = False`` -> raw state-level lattice;
     PyKaldi's recognizer then removes the acoustic scale from the lattice);
  2. MMI:   ``scale_lattice(lattice_scale(1.0, 0.2))`` then ``lattice_forward_backward_mmi(trans_model, lat,
            trans_ids, drop_frames=True, convert_to_pdf_ids=False, cancel=True)``  (ops/ops.py:57-60);
     sMBR/MPFE: ``lattice_forward_backward_mpe_variants(trans_model, silence_phones, lat, trans_ids,
            criterion, one_silence_class=True)`` on the unscaled lattice (ops/ops.py:136-141);
  3. ``Posterior.to_pdf_matrix(trans_model)`` -> dense [T, P] matrix; backward returns its negative.

Kaldi / PyKaldi are third-party, un-vendored and un-pinned (reference docker/Dockerfile:57-64) and absent
from /root/reference, so this file restates Kaldi's published algorithms: decoder/lattice-faster-decoder.cc
(token passing, GetCutoff, ProcessEmitting / ProcessNonemitting, PruneForwardLinks[Final]),
lat/lattice-functions.cc (LatticeForwardBackward, LatticeForwardBackwardMmi,
LatticeForwardBackwardMpeVariants), hmm/posterior.cc (AlignmentToPosterior, MergePosteriors).

PARITY UNPINNED at the Kaldi boundary: the reference holds no golden vectors, tests or fixtures for this
path.  Self-made pins (tests/test_oracle_lattice.py):
  (i)   brute-force path enumeration of tiny lattices for the MMI posteriors / total likelihood and for the
        expected frame accuracy and its derivative (``brute_force_lattice``);
  (ii)  the decoder against exhaustive Viterbi search with no pruning (best path cost) and the lattice-beam
        property (every arc kept lies on a path within lattice_beam of the best one; no such arc is missing
        when the search beam is wide open);
  (iii) invariants: per-frame denominator posteriors sum to 1; sMBR derivatives sum to 0 per frame.

One documented difference from Kaldi's serial decoder: ProcessEmitting tightens ``next_cutoff`` while it
walks the token list, so which arcs above the final cutoff are kept depends on hash-list order; here (and in
the HIP decoder) an arc is kept iff its cost is below the FINAL cutoff (best new cost + adaptive beam).
Such arcs lead to tokens above the frame's next_cutoff; they are not closed over epsilon arcs, and they are expanded in
the next frame only if that frame's own cutoff is looser (max_active binding on one frame and not on the next).
``decode(..., serial_order=...)`` emulates the serial rule for several walking orders; tests/test_oracle_lattice.py
measures what it changes in the pruned lattice and in the MMI posteriors.  Kaldi's per-frame cost offsets (numerical only) are not applied.
Word labels (olabels) are not carried: no criterion of the path uses them.
"""
import math

import numpy as np

INF = np.float32(np.inf)


# ----------------------------------------------------------------------------------------
# Decoding graph and transition model (plain arrays)
# ----------------------------------------------------------------------------------------
class DecodeGraphRef:
    """HCLG as arc arrays: arc a = src[a] -> dst[a], ilabel[a] (transition-id, 0 = epsilon), weight[a]
    (tropical graph cost).  final[s] = final cost (inf: not final)."""

    def __init__(self, num_states, start, src, dst, ilabel, weight, final):
        self.S, self.start = int(num_states), int(start)
        order = np.argsort(np.asarray(src), kind="stable")
        self.src = np.asarray(src, np.int32)[order]
        self.dst = np.asarray(dst, np.int32)[order]
        self.ilabel = np.asarray(ilabel, np.int32)[order]
        self.weight = np.asarray(weight, np.float32)[order]
        self.final = np.asarray(final, np.float32)
        self.off = np.zeros(self.S + 1, np.int64)
        np.add.at(self.off, self.src + 1, 1)
        self.off = np.cumsum(self.off)

    def arcs(self, s):
        return range(int(self.off[s]), int(self.off[s + 1]))


class DecoderOptionsRef:
    """LatticeFasterDecoderConfig defaults (lattice-faster-decoder.h) with the reference's YAML overrides
    (example/librispeech/configs/se.yaml decoder_config)."""

    def __init__(self, beam=13.0, lattice_beam=7.0, max_active=7000, min_active=200, beam_delta=0.5,
                 acoustic_scale=0.1):
        self.beam, self.lattice_beam = np.float32(beam), np.float32(lattice_beam)
        self.max_active, self.min_active = int(max_active), int(min_active)
        self.beam_delta, self.acoustic_scale = np.float32(beam_delta), np.float32(acoustic_scale)


def get_cutoff(costs, opts):
    """LatticeFasterDecoder::GetCutoff: returns (cutoff, adaptive_beam).  costs: float32 array of the
    frame's token costs."""
    best = costs.min()
    beam_cutoff = np.float32(best + opts.beam)
    n = costs.shape[0]
    max_active_cutoff = INF
    if n > opts.max_active:
        max_active_cutoff = np.partition(costs, opts.max_active)[opts.max_active]
    if max_active_cutoff < beam_cutoff:
        return max_active_cutoff, np.float32(np.float32(max_active_cutoff - best) + opts.beam_delta)
    min_active_cutoff = INF
    if n > opts.min_active:
        min_active_cutoff = best if opts.min_active == 0 else np.partition(costs, opts.min_active)[opts.min_active]
    if min_active_cutoff > beam_cutoff and min_active_cutoff != INF:
        return min_active_cutoff, np.float32(np.float32(min_active_cutoff - best) + opts.beam_delta)
    return beam_cutoff, opts.beam


class LatticeRef:
    """Raw state-level lattice, frame layered.  Token k = (frame tok_frame[k], HCLG state tok_state[k]).
    Link l: tok link_src[l] -> link_dst[l], transition-id link_tid[l] (0 = epsilon, stays in the frame),
    graph cost link_graph[l], acoustic cost link_ac[l] = -loglike (acoustic scale REMOVED, as PyKaldi's
    recognizer does before returning the lattice).  tok_final[k]: final cost (inf if none)."""

    def __init__(self):
        self.T = 0
        self.tok_frame, self.tok_state, self.tok_cost, self.tok_final = [], [], [], []
        self.link_src, self.link_dst, self.link_tid, self.link_graph, self.link_ac = [], [], [], [], []
        self.start_tok = 0

    def arrays(self):
        f = lambda x, t: np.asarray(x, t)
        return dict(tok_frame=f(self.tok_frame, np.int32), tok_state=f(self.tok_state, np.int32),
                    tok_cost=f(self.tok_cost, np.float32), tok_final=f(self.tok_final, np.float32),
                    link_src=f(self.link_src, np.int32), link_dst=f(self.link_dst, np.int32),
                    link_tid=f(self.link_tid, np.int32), link_graph=f(self.link_graph, np.float32),
                    link_ac=f(self.link_ac, np.float32))

    def link_set(self):
        """Canonical description of the links for comparisons: {(frame, src state, dst state, tid)}."""
        return set((self.tok_frame[s], self.tok_state[s], self.tok_state[d], t)
                   for s, d, t in zip(self.link_src, self.link_dst, self.link_tid))


def decode(graph, loglikes, tid2pdf, opts, serial_order=None):
    """LatticeFasterDecoder::Decode + GetRawLattice + final pruning, all arithmetic in float32 in the order
    (cur_cost + ac_cost) + graph_cost.  loglikes [T, P] float32; tid2pdf[tid] (index 0 unused).

    serial_order: None = an emitting arc is kept iff its cost is below the frame's FINAL next_cutoff (the rule of this
    build, see the module docstring).  Otherwise Kaldi's serial ProcessEmitting is emulated: the best token's arcs first
    bound next_cutoff, then the tokens are walked in the given order -- "ascending" / "descending" (by cost) or a
    numpy Generator (random order; Kaldi's own order is that of its hash list) -- and an arc is kept iff its cost is
    below the cutoff AS IT STANDS when the arc is reached, which tightens on the way."""
    loglikes = np.asarray(loglikes, np.float32)
    T = loglikes.shape[0]
    frames = []      # per frame: dict state -> cost
    links = []       # per frame t: list of (src_state, dst_state, tid, graph, ac_scaled); emitting t -> t+1
    eps_links = []   # per frame t: list of (src_state, dst_state, 0, graph, 0) inside frame t

    def nonemitting(toks, cutoff):
        # closure to the exact fixed point, then links from the final costs (equivalent to Kaldi's queue with
        # DeleteForwardLinks on every improvement)
        work = list(toks.keys())
        while work:
            nxt = set()
            for s in work:
                c = toks[s]
                if c >= cutoff:
                    continue
                for a in graph.arcs(s):
                    if graph.ilabel[a] != 0:
                        continue
                    tot = np.float32(c + graph.weight[a])
                    d = int(graph.dst[a])
                    if tot < cutoff and tot < toks.get(d, INF):
                        toks[d] = tot
                        nxt.add(d)
            work = list(nxt)
        out = []
        for s, c in toks.items():
            if c >= cutoff:
                continue
            for a in graph.arcs(s):
                if graph.ilabel[a] == 0:
                    tot = np.float32(c + graph.weight[a])
                    if tot < cutoff:
                        out.append((s, int(graph.dst[a]), 0, graph.weight[a], np.float32(0)))
        return out

    toks = {graph.start: np.float32(0)}
    eps_links.append(nonemitting(toks, opts.beam))          # InitDecoding: ProcessNonemitting(config_.beam)
    frames.append(toks)
    for t in range(T):
        cur = frames[t]
        costs = np.asarray(list(cur.values()), np.float32)
        cur_cutoff, adaptive_beam = get_cutoff(costs, opts)
        cand = []
        for s, c in cur.items():
            if c > cur_cutoff:
                continue
            for a in graph.arcs(s):
                tid = int(graph.ilabel[a])
                if tid == 0:
                    continue
                ac = np.float32(-(opts.acoustic_scale * loglikes[t, tid2pdf[tid]]))
                tot = np.float32(np.float32(c + ac) + graph.weight[a])
                cand.append((s, int(graph.dst[a]), tid, graph.weight[a], ac, tot))
        nxt = {}
        lk = []
        if cand and serial_order is not None:
            # LatticeFasterDecoder::ProcessEmitting as written: a running cutoff
            by_src = {}
            for x in cand:
                by_src.setdefault(x[0], []).append(x)
            srcs = list(by_src.keys())
            if isinstance(serial_order, str):
                srcs.sort(key=lambda q: (cur[q], q), reverse=(serial_order == "descending"))
            else:
                srcs = [srcs[i] for i in serial_order.permutation(len(srcs))]
            next_cutoff = INF
            best_src = min(cur, key=lambda q: (cur[q], q))
            for x in by_src.get(best_src, []):
                if np.float32(x[5] + adaptive_beam) < next_cutoff:
                    next_cutoff = np.float32(x[5] + adaptive_beam)
            for q in srcs:
                for s, d, tid, gw, ac, tot in by_src[q]:
                    if tot >= next_cutoff:
                        continue
                    if np.float32(tot + adaptive_beam) < next_cutoff:
                        next_cutoff = np.float32(tot + adaptive_beam)
                    lk.append((s, d, tid, gw, ac))
                    if tot < nxt.get(d, INF):
                        nxt[d] = tot
        elif cand:
            next_cutoff = np.float32(min(x[5] for x in cand) + adaptive_beam)
            for s, d, tid, gw, ac, tot in cand:
                if tot < next_cutoff:
                    lk.append((s, d, tid, gw, ac))
                    if tot < nxt.get(d, INF):
                        nxt[d] = tot
        else:
            next_cutoff = INF
        links.append(lk)
        if not nxt:
            raise RuntimeError("decoder: no surviving token at frame %d" % t)
        eps_links.append(nonemitting(nxt, next_cutoff))
        frames.append(nxt)

    # ---- final costs (ComputeFinalCosts) and backward pruning (PruneForwardLinksFinal / PruneForwardLinks) ----
    last = frames[T]
    fin = {s: graph.final[s] for s in last}
    if not any(v != INF for v in fin.values()):
        fin = {s: np.float32(0) for s in last}
    best_final = min(np.float32(last[s] + fin[s]) for s in last)
    extra = [dict() for _ in range(T + 1)]   # token extra costs (inf = pruned)
    keep_eps = [[] for _ in range(T + 1)]
    keep_em = [[] for _ in range(T)]
    for s in last:
        extra[T][s] = np.float32(np.float32(last[s] + fin[s]) - best_final) if fin[s] != INF else INF

    def prune_eps(t):
        cost = frames[t]
        ex = extra[t]
        changed = True
        while changed:
            changed = False
            for s, d, tid, gw, ac in eps_links[t]:
                e_d = ex.get(d, INF)
                le = np.float32(e_d + np.float32(np.float32(cost[s] + gw) - cost[d])) if e_d != INF else INF
                if le > opts.lattice_beam:
                    continue
                le = max(le, np.float32(0))
                if le < ex.get(s, INF):
                    ex[s] = le
                    changed = True
        for s, d, tid, gw, ac in eps_links[t]:
            e_d = ex.get(d, INF)
            if e_d == INF:
                continue
            le = np.float32(e_d + np.float32(np.float32(cost[s] + gw) - cost[d]))
            if le <= opts.lattice_beam:
                keep_eps[t].append((s, d, tid, gw, ac))

    prune_eps(T)
    for t in range(T - 1, -1, -1):
        cost, ncost = frames[t], frames[t + 1]
        for s, d, tid, gw, ac in links[t]:
            e_d = extra[t + 1].get(d, INF)
            if e_d == INF:
                continue
            le = np.float32(e_d + np.float32(np.float32(np.float32(cost[s] + ac) + gw) - ncost[d]))
            if le > opts.lattice_beam:
                continue
            le = max(le, np.float32(0))
            keep_em[t].append((s, d, tid, gw, ac))
            if le < extra[t].get(s, INF):
                extra[t][s] = le
        prune_eps(t)

    # ---- assemble (tokens that survive, frame by frame) ----
    lat = LatticeRef()
    lat.T = T
    index = [dict() for _ in range(T + 1)]
    for t in range(T + 1):
        for s in sorted(frames[t]):
            if extra[t].get(s, INF) != INF:
                index[t][s] = len(lat.tok_state)
                lat.tok_frame.append(t)
                lat.tok_state.append(s)
                lat.tok_cost.append(frames[t][s])
                lat.tok_final.append(fin[s] if t == T else INF)
    inv_scale = np.float32(1.0) / opts.acoustic_scale if opts.acoustic_scale != 0 else np.float32(1.0)
    for t in range(T + 1):
        if t > 0:
            for s, d, tid, gw, ac in keep_em[t - 1]:
                if s in index[t - 1] and d in index[t]:
                    lat.link_src.append(index[t - 1][s]); lat.link_dst.append(index[t][d])
                    lat.link_tid.append(tid); lat.link_graph.append(gw); lat.link_ac.append(np.float32(ac * inv_scale))
        for s, d, tid, gw, ac in keep_eps[t]:
            if s in index[t] and d in index[t]:
                lat.link_src.append(index[t][s]); lat.link_dst.append(index[t][d])
                lat.link_tid.append(0); lat.link_graph.append(gw); lat.link_ac.append(np.float32(0))
    lat.start_tok = index[0][graph.start]
    lat.best_cost = float(best_final)
    return lat


# ----------------------------------------------------------------------------------------
# Lattice forward-backward (float64, like Kaldi)
# ----------------------------------------------------------------------------------------
def _logadd(a, b):
    if a == -math.inf:
        return b
    if b == -math.inf:
        return a
    m = max(a, b)
    return m + math.log1p(math.exp(-abs(a - b)))


def _topo_links(lat):
    """Links in an order where every link's source token is complete: by frame; inside a frame the epsilon
    links in topological order of the tokens."""
    A = lat.arrays()
    nl = A["link_src"].shape[0]
    frame_of = A["tok_frame"]
    order = []
    em = [[] for _ in range(lat.T + 1)]
    ep = [[] for _ in range(lat.T + 1)]
    for l in range(nl):
        if A["link_tid"][l] != 0:
            em[frame_of[A["link_dst"][l]]].append(l)
        else:
            ep[frame_of[A["link_src"][l]]].append(l)
    for t in range(lat.T + 1):
        order.extend(em[t])
        # Kahn on the epsilon sub-graph of frame t
        indeg = {}
        outs = {}
        for l in ep[t]:
            indeg[A["link_dst"][l]] = indeg.get(A["link_dst"][l], 0) + 1
            indeg.setdefault(A["link_src"][l], 0)
            outs.setdefault(A["link_src"][l], []).append(l)
        ready = [k for k, v in indeg.items() if v == 0]
        done = 0
        while ready:
            k = ready.pop()
            for l in outs.get(k, []):
                order.append(l)
                done += 1
                indeg[A["link_dst"][l]] -= 1
                if indeg[A["link_dst"][l]] == 0:
                    ready.append(A["link_dst"][l])
        assert done == len(ep[t]), "epsilon cycle in the lattice"
    return A, order


def lattice_forward_backward(lat, lm_scale=1.0, ac_scale=1.0):
    """LatticeForwardBackward (lattice-functions.cc): returns (tot_like, alpha, beta, link_like, order, A)."""
    A, order = _topo_links(lat)
    nt = A["tok_state"].shape[0]
    # fst::ScaleLattice multiplies the float weights by the double scales and stores floats again
    like = -((lm_scale * A["link_graph"].astype(np.float64)).astype(np.float32).astype(np.float64) +
             (ac_scale * A["link_ac"].astype(np.float64)).astype(np.float32).astype(np.float64))
    alpha = np.full(nt, -math.inf)
    beta = np.full(nt, -math.inf)
    alpha[lat.start_tok] = 0.0
    for l in order:
        alpha[A["link_dst"][l]] = _logadd(alpha[A["link_dst"][l]], alpha[A["link_src"][l]] + like[l])
    tot = -math.inf
    for k in range(nt):
        if A["tok_final"][k] != INF:
            f = -float(np.float32(lm_scale * float(A["tok_final"][k])))
            beta[k] = f
            tot = _logadd(tot, alpha[k] + f)
    for l in reversed(order):
        beta[A["link_src"][l]] = _logadd(beta[A["link_src"][l]], beta[A["link_dst"][l]] + like[l])
    return tot, alpha, beta, like, order, A


def lattice_mmi(lat, trans_ids, tid2pdf, num_pdfs, lm_scale=1.0, ac_scale=0.2, drop_frames=True):
    """LatticeForwardBackwardMmi(..., drop_frames, convert_to_pdf_ids=False, cancel=True) followed by
    Posterior.to_pdf_matrix.  Returns (lat_like, post_mat[T, P]) with post = numerator - denominator."""
    tot, alpha, beta, like, order, A = lattice_forward_backward(lat, lm_scale, ac_scale)
    T = lat.T
    den = [dict() for _ in range(T)]
    for l in range(like.shape[0]):
        tid = int(A["link_tid"][l])
        if tid == 0:
            continue
        t = int(A["tok_frame"][A["link_src"][l]])
        p = math.exp(alpha[A["link_src"][l]] + like[l] + beta[A["link_dst"][l]] - tot)
        den[t][tid] = den[t].get(tid, 0.0) + p
    post = np.zeros((T, num_pdfs), np.float64)
    for t in range(T):
        ref = int(trans_ids[t])
        # MergePosteriors(num, -den, merge=True, drop_frames): a frame whose numerator transition-id does not
        # occur (with non-zero posterior) in the lattice is dropped
        if drop_frames and den[t].get(ref, 0.0) == 0.0:
            continue
        post[t, tid2pdf[ref]] += 1.0
        for tid, p in den[t].items():
            post[t, tid2pdf[tid]] -= p
    return tot, post


def lattice_mpe(lat, trans_ids, tid2pdf, tid2phone, silence_phones, num_pdfs, criterion="smbr",
                one_silence_class=True, lm_scale=1.0, ac_scale=1.0):
    """LatticeForwardBackwardMpeVariants + Posterior.to_pdf_matrix.  Returns (expected frame accuracy,
    post_mat[T, P]) where post = arc posterior * (accuracy through the arc - expected accuracy)."""
    tot, alpha, beta, like, order, A = lattice_forward_backward(lat, lm_scale, ac_scale)
    sil = set(int(x) for x in silence_phones)
    nl, nt, T = like.shape[0], alpha.shape[0], lat.T
    acc = np.zeros(nl)
    for l in range(nl):
        tid = int(A["link_tid"][l])
        if tid == 0:
            continue
        t = int(A["tok_frame"][A["link_src"][l]])
        ref = int(trans_ids[t])
        phone, ref_phone = int(tid2phone[tid]), int(tid2phone[ref])
        phone_sil, both_sil = phone in sil, (phone in sil and ref_phone in sil)
        if criterion == "mpfe":
            ok = (phone == ref_phone and not phone_sil) if not one_silence_class else (phone == ref_phone or both_sil)
        else:
            pdf, ref_pdf = int(tid2pdf[tid]), int(tid2pdf[ref])
            ok = (pdf == ref_pdf and not phone_sil) if not one_silence_class else (pdf == ref_pdf or both_sil)
        acc[l] = 1.0 if ok else 0.0
    a_s = np.zeros(nt)
    for l in order:
        s, d = A["link_src"][l], A["link_dst"][l]
        a_s[d] += math.exp(alpha[s] + like[l] - alpha[d]) * (a_s[s] + acc[l])
    tot_score = 0.0
    for k in range(nt):
        if A["tok_final"][k] != INF:
            tot_score += math.exp(alpha[k] - float(np.float32(lm_scale * float(A["tok_final"][k]))) - tot) * a_s[k]
    b_s = np.zeros(nt)
    for l in reversed(order):
        s, d = A["link_src"][l], A["link_dst"][l]
        if beta[s] != -math.inf and beta[d] != -math.inf:
            b_s[s] += math.exp(beta[d] + like[l] - beta[s]) * (b_s[d] + acc[l])
    post = np.zeros((T, num_pdfs), np.float64)
    for l in range(nl):
        tid = int(A["link_tid"][l])
        if tid == 0:
            continue
        s, d = A["link_src"][l], A["link_dst"][l]
        if beta[d] == -math.inf:
            continue
        p = math.exp(alpha[s] + like[l] + beta[d] - tot)
        post[int(A["tok_frame"][s]), tid2pdf[tid]] += p * (a_s[s] + acc[l] + b_s[d] - tot_score)
    return tot_score, post


# ----------------------------------------------------------------------------------------
# Self-made pins
# ----------------------------------------------------------------------------------------
def brute_force_lattice(lat, trans_ids, tid2pdf, tid2phone, silence_phones, num_pdfs, lm_scale, ac_scale,
                        criterion="smbr"):
    """Enumerates every start->final path of a tiny lattice.  Returns (tot_like, den_post[T,P],
    expected_accuracy, d(expected accuracy)/d(arc log-likelihood) summed per (t, pdf))."""
    A = lat.arrays()
    outs = {}
    for l in range(A["link_src"].shape[0]):
        outs.setdefault(int(A["link_src"][l]), []).append(l)
    sil = set(int(x) for x in silence_phones)
    paths = []

    def walk(k, ll, links):
        if A["tok_final"][k] != INF:
            paths.append((ll - float(np.float32(lm_scale * float(A["tok_final"][k]))), list(links)))
        for l in outs.get(k, []):
            links.append(l)
            walk(int(A["link_dst"][l]), ll - (float(np.float32(lm_scale * float(A["link_graph"][l]))) +
                                             float(np.float32(ac_scale * float(A["link_ac"][l])))), links)
            links.pop()

    walk(lat.start_tok, 0.0, [])
    lls = np.array([p[0] for p in paths])
    m = lls.max()
    tot = m + math.log(np.exp(lls - m).sum())
    w = np.exp(lls - tot)
    T = lat.T
    den = np.zeros((T, num_pdfs))
    accs = np.zeros(len(paths))
    for i, (_, links) in enumerate(paths):
        for l in links:
            tid = int(A["link_tid"][l])
            if tid == 0:
                continue
            t = int(A["tok_frame"][A["link_src"][l]])
            den[t, tid2pdf[tid]] += w[i]
            ref = int(trans_ids[t])
            both_sil = int(tid2phone[tid]) in sil and int(tid2phone[ref]) in sil
            if criterion == "mpfe":
                ok = int(tid2phone[tid]) == int(tid2phone[ref]) or both_sil
            else:
                ok = tid2pdf[tid] == tid2pdf[ref] or both_sil
            accs[i] += 1.0 if ok else 0.0
    exp_acc = float((w * accs).sum())
    grad = np.zeros((T, num_pdfs))
    for i, (_, links) in enumerate(paths):
        for l in links:
            tid = int(A["link_tid"][l])
            if tid != 0:
                grad[int(A["tok_frame"][A["link_src"][l]]), tid2pdf[tid]] += w[i] * (accs[i] - exp_acc)
    return tot, den, exp_acc, grad


def viterbi_best_cost(graph, loglikes, tid2pdf, acoustic_scale):
    """Exhaustive (unpruned) Viterbi over the decoding graph in float64: the best complete path cost."""
    T = loglikes.shape[0]

    def closure(c):
        changed = True
        while changed:
            changed = False
            for a in range(graph.src.shape[0]):
                if graph.ilabel[a] == 0 and c[graph.src[a]] + graph.weight[a] < c[graph.dst[a]] - 1e-12:
                    c[graph.dst[a]] = c[graph.src[a]] + graph.weight[a]
                    changed = True
        return c

    cost = np.full(graph.S, np.inf)
    cost[graph.start] = 0.0
    cost = closure(cost)
    for t in range(T):
        n = np.full(graph.S, np.inf)
        for a in range(graph.src.shape[0]):
            tid = graph.ilabel[a]
            if tid != 0 and cost[graph.src[a]] < np.inf:
                v = cost[graph.src[a]] - acoustic_scale * float(loglikes[t, tid2pdf[tid]]) + float(graph.weight[a])
                if v < n[graph.dst[a]]:
                    n[graph.dst[a]] = v
        cost = closure(n)
    fin = cost + graph.final.astype(np.float64)
    return float(fin.min()) if np.isfinite(fin).any() else float(cost.min())
