"""Chain supervision from an alignment -- host-side C ABI entries of libpk2hip.so against
oracle/supervision_ref.py, which restates Kaldi's pipeline at the transition-id level (no GPU needed: these
entries do integer work on the host).  Also the Kaldi `tree` reader, against byte strings assembled here from
the published format."""
import struct

import numpy as np
import pytest

from oracle import supervision_ref as ref
from pykaldi2_amd import _lib, chain, synth
from pykaldi2_amd.lattice import TransitionModel
from pykaldi2_amd.tree import ContextDependency


def _opts(factor=3, left=5, right=5):
    o = chain.SupervisionOptions()
    o.frame_subsampling_factor, o.left_tolerance, o.right_tolerance = factor, left, right
    return o


def _ref_model(tm):
    return ref.TransitionModelRef(tm.phone2entry, tm.entries, tm.tuples.tolist())


@pytest.fixture(scope="module")
def model():
    tree, tm = synth.chain_model(90, seed=3, mixed_topologies=True)
    return tree, tm, _ref_model(tm)


# ---------------------------------------------------------------------------------------------------
# tree
# ---------------------------------------------------------------------------------------------------
def test_tree_binary_bytes_assembled_by_hand(tmp_path):
    i32 = lambda v: b"\x04" + struct.pack("<i", v)
    u32 = lambda v: b"\xfc" + struct.pack("<I", v)
    raw = (b"\0BContextDependency " + i32(2) + i32(1) + b"ToPdf " +
           b"TE " + i32(1) + u32(3) + b"( " +
           b"NULL " +
           b"CE " + i32(5) +
           b"SE " + i32(0) + b"\x04" + struct.pack("<i", 2) + struct.pack("<2i", 0, 2) + b"{ " +
           b"TE " + i32(-1) + u32(2) + b"( CE " + i32(7) + b"CE " + i32(8) + b") " +
           b"CE " + i32(9) + b"} " +
           b") EndContextDependency ")
    p = tmp_path / "tree"
    p.write_bytes(raw)
    t = ContextDependency.read(str(p))
    assert (t.context_width(), t.central_position(), t.num_pdfs()) == (2, 1, 10)
    assert t.compute([0, 1], 0) == 5 and t.compute([3, 1], 1) == 5
    assert t.compute([0, 2], 0) == 7 and t.compute([2, 2], 1) == 8 and t.compute([1, 2], 0) == 9
    assert t.compute([0, 0], 0) is None and t.compute([0, 3], 0) is None and t.compute([0, 2], 2) is None


def test_tree_text_form_and_round_trip(tmp_path, model):
    txt = "ContextDependency 1 0 ToPdf TE 0 3 ( NULL SE -1 [ 0 ]\n{ CE 0 CE 1 }\nCE 2 )\nEndContextDependency\n"
    p = tmp_path / "tree.txt"
    p.write_text(txt)
    t = ContextDependency.read(str(p))
    assert [t.compute([1], 0), t.compute([1], 1), t.compute([2], 0), t.compute([0], 0)] == [0, 1, 2, None]
    tree = model[0]
    rng = np.random.default_rng(0)
    for binary in (True, False):
        q = tmp_path / ("t%d" % binary)
        tree.write(str(q), binary=binary)
        back = ContextDependency.read(str(q))
        for name in ("kind", "key", "a", "b", "pool"):
            assert np.array_equal(getattr(back, name), getattr(tree, name)), name
        for _ in range(50):
            w, c = rng.integers(0, 25, size=2).tolist(), int(rng.integers(0, 3))
            assert back.compute(w, c) == tree.compute(w, c)
    assert (back.N, back.P) == (tree.N, tree.P)


def test_tree_lookup_through_the_library_matches_host_walk(model):
    tree, tm, _ = model
    m = chain.supervision_model(tree, tm)
    rng = np.random.default_rng(1)
    nph = max(tm.phone2entry)
    hits = 0
    for _ in range(400):
        w = [int(rng.integers(0, nph + 2)), int(rng.integers(0, nph + 2))]
        c = int(rng.integers(-1, 4))
        want = tree.compute(w, c)
        if want is None:
            with pytest.raises(_lib.Pk2Error):
                m.pdf(w, c)
        else:
            hits += 1
            assert m.pdf(w, c) == want
    assert hits > 100


# ---------------------------------------------------------------------------------------------------
# transition model numbering, SplitToPhones
# ---------------------------------------------------------------------------------------------------
def test_transition_id_tables_match_compute_derived(model):
    _, tm, rm = model
    assert tm.num_transition_ids() == len(rm.tid2ts) - 1
    for tid in range(1, tm.num_transition_ids() + 1):
        assert tm.tid2tstate[tid] == rm.tid2ts[tid]
        assert tm.tid2pdf[tid] == rm.tid_to_pdf(tid) and tm.tid2phone[tid] == rm.tid_to_phone(tid)
        assert bool(tm.tid_flags[tid] & 1) == rm.is_self_loop(tid) and bool(tm.tid_flags[tid] & 2) == rm.is_final(tid)


@pytest.mark.parametrize("reorder", [True, False])
def test_split_to_phones(model, reorder):
    _, tm, rm = model
    rng = np.random.default_rng(5 + reorder)
    for T in (3, 17, 64, 301):
        ali, phones, durs = synth.phone_tid_alignment(rng, T, tm, reorder)
        assert ali.shape[0] == T and sum(durs) == T
        ok, pieces = chain.split_to_phones(tm, ali)
        rok, rpieces = ref.split_to_phones(rm, ali.tolist())
        assert ok and rok
        assert [(p, d) for p, _, d in pieces] == list(zip(phones, durs)) == [(rm.tid_to_phone(x[0]), len(x)) for x in rpieces]
        assert [s for _, s, _ in pieces] == np.concatenate([[0], np.cumsum(durs)[:-1]]).tolist()
        assert chain.MappedAligner(tm).to_phone_alignment(ali) == pieces


def test_split_to_phones_incomplete_alignments(model):
    _, tm, rm = model
    rng = np.random.default_rng(11)
    ali, _, _ = synth.phone_tid_alignment(rng, 40, tm, True)
    for bad in (ali[:-1] if not (tm.tid_flags[ali[-2]] & 2) else ali[:-2], np.concatenate([ali[:7], ali[19:]])):
        ok, pieces = chain.split_to_phones(tm, bad)
        rok, rpieces = ref.split_to_phones(rm, bad.tolist())
        assert ok == rok and [d for _, _, d in pieces] == [len(x) for x in rpieces]
    with pytest.raises(_lib.Pk2Error):
        chain.split_to_phones(tm, [1, tm.num_transition_ids() + 1])
    with pytest.raises(_lib.Pk2Error):
        chain.split_to_phones(tm, [0])


# ---------------------------------------------------------------------------------------------------
# supervision
# ---------------------------------------------------------------------------------------------------
def _check_structure(sup):
    T = sup.frames_per_sequence
    assert sup.frame_offsets[0] == 0 and sup.frame_offsets[-1] == sup.src.shape[0]
    st = sup.state_time
    assert st[0] == 0 and np.all(np.diff(st) >= 0) and np.all(st[1:] >= 1)
    for t in range(T):
        a, b = sup.frame_offsets[t], sup.frame_offsets[t + 1]
        assert b > a
        assert np.all(st[sup.src[a:b]] == t) and np.all(st[sup.dst[a:b]] == t + 1)
    assert np.all(st[sup.final_states] == T) and np.all(sup.arc_weight == 0) and np.all(sup.final_weights == 0)
    # connected: every state leaves an arc or is final, every state but 0 is entered
    has_out = np.zeros(sup.num_states, bool); has_out[sup.src] = True; has_out[sup.final_states] = True
    has_in = np.zeros(sup.num_states, bool); has_in[sup.dst] = True; has_in[0] = True
    assert has_out.all() and has_in.all()
    assert set(np.flatnonzero(st == T).tolist()) == set(sup.final_states.tolist())


def test_allowed_phones_and_language_tiny(model):
    tree, tm, rm = model
    rng = np.random.default_rng(21)
    n_checked = 0
    n_multi = n_empty = 0
    for trial in range(160):
        factor = int(rng.integers(1, 4))
        left, right = int(rng.integers(0, 5)), int(rng.integers(0, 5))
        if trial < 40:
            _, phones, durs = synth.phone_tid_alignment(rng, int(rng.integers(4, 15)), tm, True)
        else:   # short phones: several instances (all three topologies, repeated phones) inside a few frames
            n = int(rng.integers(2, 6))
            phones = [int(p) for p in rng.choice(sorted(tm.phone2entry)[:8], size=n)]
            durs = [int(d) for d in rng.integers(1, 2 + factor * 2, size=n)]
        if len(phones) > 5 or -(-sum(durs) // factor) > 11:
            continue
        proto = chain.alignment_to_proto_supervision(_opts(factor, left, right), phones, durs)
        want_allowed = ref.alignment_to_proto_supervision(phones, durs, factor, left, right)
        want = ref.label_sequences(rm, tree.compute, tree.N, tree.P, phones, want_allowed)
        if not want:
            n_empty += 1
            with pytest.raises(_lib.Pk2Error):
                chain.proto_supervision_to_supervision(tree, tm, proto, True)
            continue
        n_multi += len(phones) >= 3 and len(want) > 1
        sup = chain.proto_supervision_to_supervision(tree, tm, proto, True, with_allowed=True)
        assert sup.allowed_phones == want_allowed
        assert sup.frames_per_sequence == len(want_allowed) and sup.label_dim == tm.num_pdfs()
        _check_structure(sup)
        assert ref.fst_label_sequences(sup) == want
        assert ref.fst_count_paths(sup) == len(want) == ref.count_paths(rm, tree.compute, tree.N, tree.P, phones, want_allowed)
        n_checked += 1
    assert n_checked >= 80 and n_multi >= 20 and n_empty >= 5


@pytest.mark.parametrize("mixed", [False, True])
def test_path_counts_at_utterance_size(mixed):
    tree, tm = synth.chain_model(400, seed=7, mixed_topologies=mixed)
    rm = _ref_model(tm)
    rng = np.random.default_rng(33)
    for T in (60, 240, 900):
        ali, _, _ = synth.phone_tid_alignment(rng, T, tm, bool(T % 7))
        pieces = chain.MappedAligner(tm).to_phone_alignment(ali)
        phones, durs = [p for p, _, _ in pieces], [d for _, _, d in pieces]
        proto = chain.alignment_to_proto_supervision(_opts(), phones, durs)
        sup = chain.proto_supervision_to_supervision(tree, tm, proto, True, with_allowed=True)
        allowed = ref.alignment_to_proto_supervision(phones, durs)
        assert sup.allowed_phones == allowed and sup.frames_per_sequence == -(-T // 3)
        _check_structure(sup)
        assert ref.fst_count_paths(sup) == ref.count_paths(rm, tree.compute, tree.N, tree.P, phones, allowed) > 0


def test_supervision_failures(model):
    tree, tm, _ = model
    phones = sorted(tm.phone2entry)[:4]
    # four phones cannot share two subsampled frames
    with pytest.raises(_lib.Pk2Error, match="no path"):
        chain.proto_supervision_to_supervision(tree, tm, chain.alignment_to_proto_supervision(_opts(), phones, [1, 2, 1, 1]))
    with pytest.raises(_lib.Pk2Error, match="no topology"):
        chain.proto_supervision_to_supervision(tree, tm, chain.alignment_to_proto_supervision(_opts(), [max(tm.phone2entry) + 1], [9]))
    with pytest.raises(_lib.Pk2Error, match="duration"):
        chain.proto_supervision_to_supervision(tree, tm, chain.alignment_to_proto_supervision(_opts(), phones[:2], [4, 0]))
    with pytest.raises(ValueError):
        chain.alignment_to_proto_supervision(_opts(), [], [])
    with pytest.raises(NotImplementedError):
        chain.proto_supervision_to_supervision(tree, tm, chain.alignment_to_proto_supervision(_opts(), phones[:1], [6]), False)
    # a transition model whose tuples lack what the tree produces (TupleToTransitionState fails in Kaldi)
    fewer = TransitionModel.from_topology(tm.phone2entry, tm.entries, [tuple(t) for t in tm.tuples.tolist()[1:]])
    with pytest.raises(_lib.Pk2Error, match="tuple"):
        for left in [0] + phones:
            chain.proto_supervision_to_supervision(tree, fewer, chain.alignment_to_proto_supervision(
                _opts(), [left, tm.tuples[0][0]] if left else [int(tm.tuples[0][0])], [6, 6] if left else [6]))
    # a transition model read without its topology
    bare = TransitionModel(tm.tid2pdf, tm.tid2phone)
    with pytest.raises(ValueError):
        chain.split_to_phones(bare, [1, 2])
