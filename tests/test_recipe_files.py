"""The file-based recipe of tests/recipe.py read back through the library's own readers (no GPU): what the reference's
command lines load from <chain_dir>, <ali_dir>, <graph_dir> and the data yaml equals the arrays the files were written from."""
import numpy as np
import yaml

from pykaldi2_amd import chain, data, lattice, se, synth
from pykaldi2_amd.tree import ContextDependency

from recipe import chain_recipe, lattice_recipe


def test_chain_recipe_round_trips(tmp_path):
    r = chain_recipe(str(tmp_path))
    G_file = chain.DenominatorGraph(str(tmp_path / "chain" / "den.fst"), r["P"])
    G_arr = chain.DenominatorGraph(r["den"], r["P"])
    assert G_file.num_states() == G_arr.num_states() and G_file.num_arcs() == G_arr.num_arcs()
    assert np.abs(G_file.initial_probs() - G_arr.initial_probs()).max() < 1e-6
    tm = lattice.TransitionModel.read(str(tmp_path / "chain" / "0.trans_mdl"))
    assert np.array_equal(tm.tid2pdf, r["trans_model"].tid2pdf) and np.array_equal(tm.tid2phone, r["trans_model"].tid2phone)
    assert np.array_equal(tm.tid2tstate, r["trans_model"].tid2tstate) and np.array_equal(tm.tid_flags, r["trans_model"].tid_flags)
    tree = ContextDependency.read(str(tmp_path / "chain" / "tree"))
    assert tree.num_pdfs() == r["tree"].num_pdfs()
    for left in (0, 1, 3):
        for ph in (1, 2, 5):
            for cls in (0, 1):
                assert tree.compute([left, ph], cls) == r["tree"].compute([left, ph], cls)
    aligner = chain.MappedAligner.from_files(str(tmp_path / "ali" / "final.mdl"), str(tmp_path / "ali" / "tree"),
                                             str(tmp_path / "lang" / "L.fst"), None, str(tmp_path / "lang" / "phones" / "disambig.int"))
    cfg = yaml.safe_load(open(tmp_path / "data" / "data.yaml"))
    src = data.ZipWavSource([v for v in cfg["clean_source"].values()])
    assert len(src) == len(r["wavs"])
    for i in range(len(src)):
        wav, lab, aux, utt = src.get(i)
        assert np.array_equal(lab, r["alis"][utt]) and np.abs(wav - r["wavs"][utt][:wav.shape[0]]).max() < 1e-4
        # the supervision built from the file's alignment with the file's models is the one built from the arrays
        a = chain.supervision_from_alignment(aligner, tree, tm, chain.SupervisionOptions(), lab)
        b = chain.supervision_from_alignment(chain.MappedAligner(r["trans_model"]), r["tree"], r["trans_model"],
                                             chain.SupervisionOptions(), r["alis"][utt])
        assert a.frames_per_sequence == b.frames_per_sequence and np.array_equal(a.pdf, b.pdf) and np.array_equal(a.src, b.src)


def test_lattice_recipe_round_trips(tmp_path):
    r = lattice_recipe(str(tmp_path))
    graph = lattice.DecodeGraph(str(tmp_path / "graph" / "HCLG.fst"))
    assert (graph.num_states, graph.num_arcs) == (r["hclg"]["num_states"], len(r["hclg"]["src"]))
    tm = lattice.TransitionModel.read(str(tmp_path / "final.mdl"))
    assert np.array_equal(tm.tid2pdf, r["tm"]["tid2pdf"]) and np.array_equal(tm.tid2phone, r["tm"]["tid2phone"])
    assert np.allclose(se.read_kaldi_vector(str(tmp_path / "final.occs")), r["counts"])
    assert open(tmp_path / "graph" / "phones" / "silence.csl").read().strip() == "1"
    cfg = yaml.safe_load(open(tmp_path / "data" / "data.yaml"))
    src = data.ZipWavSource([v for v in cfg["clean_source"].values()])
    for i in range(len(src)):
        wav, lab, aux, utt = src.get(i)
        assert np.array_equal(lab, r["pdfs"][utt]) and np.array_equal(aux, r["tids"][utt])
