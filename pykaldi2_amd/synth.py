"""Seeded synthetic inputs shaped like the LibriSpeech recipe.

There is no dataset, Kaldi model directory or den.fst in this environment, so
bench.py and the tests draw LibriSpeech-shaped utterances, pdf alignments, a
denominator graph and per-utterance numerator FSTs from these generators.
Nothing here is on the timed path.
"""
import numpy as np


def utterance_durations(rng, n):
    """Seconds; clip(Gamma(k=4.2, theta=2.93), 1.3, 34.9), mean ~12.3 s."""
    return np.clip(rng.gamma(4.2, 2.93, size=n), 1.3, 34.9)


def waveform(rng, seconds, sr=16000):
    """0.05*N(0,1) low-passed by y[n] = x[n] + 0.9 y[n-1], peak-normalised to 0.5."""
    n = int(round(sr * seconds))
    x = (0.05 * rng.standard_normal(n)).astype(np.float64)
    try:
        from scipy.signal import lfilter
        y = lfilter([1.0], [1.0, -0.9], x)
    except Exception:  # pragma: no cover
        y = np.empty_like(x)
        acc = 0.0
        for i in range(n):
            acc = x[i] + 0.9 * acc
            y[i] = acc
    y *= 0.5 / max(1e-9, np.abs(y).max())
    return y.astype(np.float32)


def room_impulse_response(rng, sr=16000):
    """A synthetic room impulse response: direct path after 2-8 ms, then Gaussian noise under an exponential
    envelope with T60 in [0.1, 0.5] s (the range of the reference's reverb config, simulation/config.py:63-92),
    cut where the envelope has fallen by 60 dB."""
    t60 = float(rng.uniform(0.1, 0.5))
    delay = int(rng.uniform(0.002, 0.008) * sr)
    n_tail = int(t60 * sr)
    r = np.zeros(delay + 1 + n_tail, np.float32)
    r[delay] = 1.0
    env = 10.0 ** (-3.0 * np.arange(n_tail) / n_tail)        # -60 dB at t60
    r[delay + 1:] = (0.1 * rng.standard_normal(n_tail) * env).astype(np.float32)     # the direct path stays the peak
    return r


def num_fbank_frames(n_samples):
    """Frames produced by the reference extractor for an n_samples wav
    (reference simulation/freq_analysis.py:64-69 after data/sr_dataset.py:288
    dropped one sample): ceil((N-1-400)/160)+1."""
    m = n_samples - 1
    if m <= 400:
        return 1
    return -(-(m - 400) // 160) + 1


def pdf_alignment(rng, num_frames, num_pdfs):
    """Piecewise-constant pdf ids; segment length 3+Geometric(0.12) frames."""
    out = np.empty(num_frames, dtype=np.int64)
    t = 0
    prev = None
    while t < num_frames:
        seg = 3 + int(rng.geometric(0.12))
        pid = int(rng.integers(0, num_pdfs))
        if prev is not None and pid == prev:
            pid = (pid + 1) % num_pdfs
        if num_frames - t < 3 and prev is not None:
            pid = prev  # a <3-frame tail is merged into the previous segment
        out[t:t + seg] = pid
        prev = pid
        t += seg
    return out


def den_graph_arcs(num_states=30000, num_arcs=1000000, num_pdfs=6048, seed=0, loop_pdf_differs=False,
                   multi_entry_frac=0.0):
    """Phone-structured synthetic denominator graph.

    ``loop_pdf_differs=True`` gives the graph Kaldi's chain topology produces (one HMM state per phone with
    ForwardPdfClass != SelfLoopPdfClass, self-loops added after the forward transition): the arcs ENTERING a state carry
    its forward pdf, its self-loop carries a different (self-loop) pdf, so the pdf is no longer a function of the
    destination state -- every looping state has two entering pdfs.  ``multi_entry_frac`` additionally gives that
    fraction of the states a second forward pdf on half of their entering arcs (states merged by minimisation across
    phonetic contexts).

    States come in pairs (a 2-state left-to-right "phone"): the entry state has
    a self loop and a forward arc to the exit state; the exit state has a self
    loop and 1+Poisson fan-out arcs to entry states of other phones (a phone
    bigram).  Every state carries one pdf, emitted by arcs *entering* it.  Arc
    probabilities are Dirichlet-normalised per source state.  Returns a dict of
    numpy arrays (src, dst, pdf int32; prob float32), start state 0.
    """
    rng = np.random.default_rng(seed)
    S = int(num_states) // 2 * 2
    n_ph = S // 2
    state_pdf = rng.integers(0, num_pdfs, size=S)
    entry = np.arange(n_ph) * 2
    exit_ = entry + 1
    fixed = 3 * n_ph  # entry self, entry->exit, exit self
    fan_total = max(n_ph, int(num_arcs) - fixed)
    lam = max(0.0, fan_total / n_ph - 1.0)
    fan = 1 + rng.poisson(lam, size=n_ph)
    src = [entry, entry, exit_, np.repeat(exit_, fan)]
    # destinations: a skewed pool (some phones are far more frequent successors)
    pop = rng.gamma(0.7, 1.0, size=n_ph)
    pop /= pop.sum()
    fan_dst = rng.choice(n_ph, size=int(fan.sum()), p=pop) * 2
    dst = [entry, exit_, exit_, fan_dst]
    src = np.concatenate(src)
    dst = np.concatenate(dst)
    # make sure every entry state is reachable: phone i's exit -> phone (i+1)'s entry
    src = np.concatenate([src, exit_])
    dst = np.concatenate([dst, np.roll(entry, -1)])
    w = rng.gamma(1.0, 1.0, size=src.shape[0]) + 1e-3
    tot = np.bincount(src, weights=w, minlength=S)
    prob = w / tot[src]
    pdf = state_pdf[dst]
    if loop_pdf_differs:
        loop_pdf = (state_pdf + 1 + rng.integers(0, max(1, num_pdfs - 1), size=S)) % num_pdfs   # != state_pdf for P > 1
        is_loop = src == dst
        pdf = np.where(is_loop, loop_pdf[dst], pdf)
    if multi_entry_frac > 0:
        alt_pdf = rng.integers(0, num_pdfs, size=S)
        multi = rng.random(S) < multi_entry_frac
        flip = multi[dst] & (src != dst) & (rng.random(src.shape[0]) < 0.5)
        pdf = np.where(flip, alt_pdf[dst], pdf)
    order = np.lexsort((dst, src))
    return dict(num_states=S, start=0, num_pdfs=int(num_pdfs),
                src=src[order].astype(np.int32), dst=dst[order].astype(np.int32),
                pdf=pdf[order].astype(np.int32), prob=prob[order].astype(np.float32))


def numerator_fst_from_alignment(ali, subsample=3, tolerance=5):
    """Simplified stand-in for Kaldi's alignment -> Supervision pipeline
    (reference bin/train_chain.py:262-272 with SupervisionOptions :184-188;
    SURVEY Appendix A.3): T' = ceil(T/3); segment i occupying frames [b,e)
    may be emitted on subsampled frames [ceil(max(0,b-tol)/3), ceil(min(e+tol,T)/3));
    one pdf per segment, at least one frame per segment, unit weights.

    State (i, t) = "t frames consumed, last one from segment i".  Returns a
    dict of arrays for chain.Supervision: arcs sorted by source time, with
    ``frame_offsets[t]`` the first arc whose source state sits at frame t.
    """
    ali = np.asarray(ali)
    T = ali.shape[0]
    Tp = -(-T // subsample)
    change = np.flatnonzero(np.diff(ali)) + 1
    b = np.concatenate([[0], change])
    e = np.concatenate([change, [T]])
    seg_pdf = ali[b]
    n_seg = b.shape[0]
    lo = -(-np.maximum(0, b - tolerance) // subsample)
    hi = -(-np.minimum(e + tolerance, T) // subsample)
    hi = np.minimum(hi, Tp)
    # reachable (i, t): forward pass then backward pass over the trellis
    fwd = np.zeros((n_seg + 1, Tp + 1), dtype=bool)  # row 0 = start pseudo-segment -1
    fwd[0, 0] = True
    for t in range(Tp):
        prev = fwd[:, t]
        allowed = (lo <= t) & (t < hi)
        stay = prev[1:] & allowed
        adv = prev[:-1] & allowed
        fwd[1:, t + 1] = stay | adv
    bwd = np.zeros_like(fwd)
    bwd[n_seg, Tp] = fwd[n_seg, Tp]
    if not bwd[n_seg, Tp]:
        raise ValueError("alignment admits no path under the tolerance windows")
    for t in range(Tp - 1, -1, -1):
        allowed = (lo <= t) & (t < hi)
        nxt = bwd[1:, t + 1] & allowed          # arcs entering segment i at t+1 need frame t allowed for i
        bwd[1:, t] |= nxt                      # stay: (i,t)->(i,t+1)
        bwd[:-1, t] |= nxt                     # advance: (i-1,t)->(i,t+1)
        bwd[:, t] &= fwd[:, t]
    live = fwd & bwd
    sid = -np.ones(live.shape, dtype=np.int64)
    # number states in time-major order so the FST is top-sorted
    ii, tt = np.nonzero(live.T)[1], np.nonzero(live.T)[0]
    sid[ii, tt] = np.arange(ii.shape[0])
    state_time = tt.copy()
    src, dst, pdf = [], [], []
    offsets = [0]
    for t in range(Tp):
        allowed = (lo <= t) & (t < hi)
        for kind in (0, 1):  # 0 = stay, 1 = advance
            if kind == 0:
                ok = live[1:, t] & live[1:, t + 1] & allowed
                s_ = sid[1:, t][ok]
            else:
                ok = live[:-1, t] & live[1:, t + 1] & allowed
                s_ = sid[:-1, t][ok]
            d_ = sid[1:, t + 1][ok]
            src.append(s_); dst.append(d_); pdf.append(seg_pdf[ok])
        offsets.append(offsets[-1] + src[-1].shape[0] + src[-2].shape[0])
    src = np.concatenate(src); dst = np.concatenate(dst); pdf = np.concatenate(pdf)
    return dict(num_states=int(ii.shape[0]), frames=int(Tp),
                src=src.astype(np.int32), dst=dst.astype(np.int32), pdf=pdf.astype(np.int32),
                weight=np.zeros(src.shape[0], dtype=np.float32),
                frame_offsets=np.asarray(offsets, dtype=np.int32),
                state_time=state_time.astype(np.int32),
                final_states=np.asarray([sid[n_seg, Tp]], dtype=np.int32),
                final_weights=np.zeros(1, dtype=np.float32))


def minibatch(rng, batch, num_pdfs, sr=16000, ali_model=None, duration_rng=None):
    """One LibriSpeech-shaped minibatch: list of (wav f32[N], alignment i64[T]); the alignment holds pdf-ids,
    or transition-ids of `ali_model` (a TransitionModel: the label files of chain training).  `duration_rng`
    draws the utterance lengths from a separate generator (length-bucketed data parallelism: every rank passes
    the same one and gets utterances of the same lengths with different content)."""
    out = []
    for d in utterance_durations(duration_rng if duration_rng is not None else rng, batch):
        wav = waveform(rng, float(d), sr)
        T = num_fbank_frames(wav.shape[0])
        out.append((wav, phone_tid_alignment(rng, T, ali_model)[0] if ali_model is not None else pdf_alignment(rng, T, num_pdfs)))
    return out


# ----------------------------------------------------------------------------------------
# Lattice path (train_se): transition model, decoding graph, transition-id alignments
# ----------------------------------------------------------------------------------------
def transition_model_arrays(num_pdfs):
    """A context-independent 3-state left-to-right model over num_pdfs // 3 phones, in Kaldi's numbering:
    transition-ids are 1-based, two per HMM state (self-loop, then forward), pdf = 3*(phone-1) + hmm_state.
    Returns dict(tid2pdf, tid2phone [index 0 unused], num_phones, silence_phones)."""
    num_phones = num_pdfs // 3
    n = 6 * num_phones
    tid = np.arange(1, n + 1)
    tid2pdf = np.concatenate([[-1], (tid - 1) // 2]).astype(np.int32)
    tid2phone = np.concatenate([[0], (tid - 1) // 6 + 1]).astype(np.int32)
    return dict(tid2pdf=tid2pdf, tid2phone=tid2phone, num_phones=num_phones, silence_phones=[1])


def decoding_graph_arcs(num_words=2000, num_pdfs=5768, seed=0, max_phones=5):
    """Synthetic HCLG-shaped decoding graph: a word loop over `num_words` pronunciations of 2..max_phones
    phones, every phone 3 HMM states with a self-loop and a forward arc (ilabels = transition-ids of
    transition_model_arrays), word entry = epsilon arc carrying a Zipf unigram cost, word exit = epsilon arc
    back to the loop state (so tokens cross two epsilon arcs between words); the word-entry arc carries the word id
    (1-based) as its output label.  Returns dict(num_states, start, src, dst, ilabel, olabel, weight, final)."""
    rng = np.random.default_rng(seed)
    num_phones = num_pdfs // 3
    p = 1.0 / np.arange(1, num_words + 1)
    p /= p.sum()
    src, dst, ilab, w, olab = [], [], [], [], []
    loop = 0
    n_states = 1
    lp_self, lp_fwd = -np.log(0.6), -np.log(0.4)
    for wd in range(num_words):
        k = int(rng.integers(2, max_phones + 1))
        phones = rng.integers(1, num_phones + 1, size=k)
        if wd == 0:
            phones = np.array([1, 1])      # a silence "word"
        first = n_states
        src.append(loop); dst.append(first); ilab.append(0); w.append(-np.log(p[wd])); olab.append(wd + 1)   # the word label
        for ph in phones:
            for hs in range(3):
                s = n_states
                n_states += 1
                base = 1 + 2 * (3 * (int(ph) - 1) + hs)
                src.append(s); dst.append(s); ilab.append(base); w.append(lp_self); olab.append(0)
                src.append(s); dst.append(s + 1); ilab.append(base + 1); w.append(lp_fwd); olab.append(0)
        end = n_states        # word-end state reached by the last forward arc
        n_states += 1
        src.append(end); dst.append(loop); ilab.append(0); w.append(0.0); olab.append(0)
    final = np.full(n_states, np.inf, np.float32)
    final[loop] = 0.0
    return dict(num_states=n_states, start=loop, src=np.asarray(src, np.int32), dst=np.asarray(dst, np.int32),
                ilabel=np.asarray(ilab, np.int32), olabel=np.asarray(olab, np.int32), weight=np.asarray(w, np.float32),
                final=final)


def tid_alignment(rng, num_frames, num_pdfs):
    """A transition-id alignment (reference aux_label): phones drawn uniformly, each HMM state held for
    1 + Geometric(0.4) frames = self-loop ids followed by one forward id."""
    num_phones = num_pdfs // 3
    out = []
    while len(out) < num_frames:
        ph = int(rng.integers(1, num_phones + 1))
        for hs in range(3):
            base = 1 + 2 * (3 * (ph - 1) + hs)
            d = int(rng.geometric(0.4))
            out.extend([base] * (d - 1) + [base + 1])
    return np.asarray(out[:num_frames], np.int64)


# ----------------------------------------------------------------------------------------
# Chain model for the supervision builder: tree + transition model + transition-id alignments
# ----------------------------------------------------------------------------------------
CHAIN_TOPO = [(0, 1, [0, 1]), (-1, -1, [])]                         # Kaldi's chain topology: one emitting state
BAKIS_TOPO = [(0, 0, [0, 1]), (1, 1, [1, 2]), (2, 2, [2, 3]), (-1, -1, [])]
SKIP_TOPO = [(0, 1, [0, 1, 2]), (2, 2, [1, 2]), (-1, -1, [])]       # state 0 may skip state 1; state 1 loops second


def chain_model(num_pdfs, seed=0, mixed_topologies=False):
    """A left-biphone chain model in Kaldi's terms: returns (tree, trans_model).
    tree: ContextDependency with N = 2, P = 1 -- table on the central phone, split on the left phone (a random
    half of the phones, sometimes with 0 = utterance start), table on the pdf-class, constant leaves.
    Phones use the chain topology (forward pdf-class 0, self-loop pdf-class 1: 4 pdfs per phone); with
    mixed_topologies every third phone is a 3-state Bakis model and every fifth a 2-state model with a skip."""
    from .lattice import TransitionModel
    from .tree import ContextDependency
    rng = np.random.default_rng(seed)
    entries = [CHAIN_TOPO, BAKIS_TOPO, SKIP_TOPO]
    phone2entry, tuples, kids, next_pdf, ph = {}, [], [None], 0, 0
    while True:
        e = (1 if (ph + 1) % 3 == 0 else 2 if (ph + 1) % 5 == 0 else 0) if mixed_topologies else 0
        classes = sorted({c for f, l, _ in entries[e][:-1] for c in (f, l)})
        if next_pdf + 2 * len(classes) > num_pdfs:
            break
        ph += 1
        phone2entry[ph] = e
        branch = []
        for _ in range(2):
            leaf = [None] * (max(classes) + 1)
            for c in classes:
                leaf[c] = ("CE", next_pdf)
                next_pdf += 1
            branch.append(leaf)
        kids.append((branch[0], branch[1]))
    num_phones = ph
    root_kids = [None]
    for p in range(1, num_phones + 1):
        yes = rng.choice(num_phones + 1, size=max(1, (num_phones + 1) // 2), replace=False)
        b0, b1 = kids[p]
        root_kids.append(("SE", 0, [int(v) for v in yes], ("TE", -1, b0), ("TE", -1, b1)))
        for leaf in (b0, b1):
            for hs, (f, l, _) in enumerate(entries[phone2entry[p]][:-1]):
                tuples.append((p, hs, leaf[f][1], leaf[l][1]))
    tree = ContextDependency.from_nested(2, 1, ("TE", 1, root_kids))
    tm = TransitionModel.from_topology(phone2entry, entries, sorted(set(tuples)))
    return tree, tm


def phone_tid_alignment(rng, num_frames, trans_model, reorder=True):
    """A transition-id alignment of exactly `num_frames` frames from `trans_model` (the reference's label
    files): random phones, a random path through each phone's HMM, 3 + Geometric(0.12) frames per phone split
    over the HMM states of the path; with `reorder` the self-loops follow the forward transition (Kaldi's
    default graphs), otherwise they precede it.  Returns (tids i64[T], phones, durations)."""
    first_tid = {}
    for tid in range(1, trans_model.num_transition_ids() + 1):
        first_tid.setdefault(int(trans_model.tid2tstate[tid]), tid)
    by_state = {}
    for ts, (p, hs, _, _) in enumerate(trans_model.tuples.tolist(), start=1):
        by_state.setdefault((p, hs), []).append(ts)
    phone_ids = sorted(trans_model.phone2entry)
    tids, phones, durs = [], [], []
    last_loop = None      # (position of the last self-loop run, its transition-id): where a short tail is absorbed
    while len(tids) < num_frames:
        p = int(phone_ids[rng.integers(len(phone_ids))])
        states = trans_model.entries[trans_model.phone2entry[p]]
        path, hs = [], 0
        while hs != len(states) - 1:
            fwd = [k for k, d in enumerate(states[hs][2]) if d != hs]
            k = fwd[int(rng.integers(len(fwd)))]
            path.append((hs, k))
            hs = states[hs][2][k]
        remaining = num_frames - len(tids)
        if len(path) > remaining:
            if last_loop is None:
                raise ValueError("num_frames too small for one phone")
            at, loop = last_loop
            tids[at:at] = [loop] * remaining
            durs[-1] += remaining
            break
        total = max(len(path), min(int(rng.geometric(0.12)) + 3, remaining))
        piece = []
        for j, (hs, k) in enumerate(path):
            dsts = states[hs][2]
            ts = by_state[(p, hs)][int(rng.integers(len(by_state[(p, hs)])))]
            hold = total // len(path) + (1 if j < total % len(path) else 0) if hs in dsts else 1
            loops = [first_tid[ts] + dsts.index(hs)] * (hold - 1)
            if hs in dsts:
                last_loop = (len(tids) + len(piece) + (1 if reorder else 0), first_tid[ts] + dsts.index(hs))
            piece.extend([first_tid[ts] + k] + loops if reorder else loops + [first_tid[ts] + k])
        tids.extend(piece); phones.append(p); durs.append(len(piece))
    return np.asarray(tids, np.int64), phones, durs
