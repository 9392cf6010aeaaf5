"""The training / lattice command lines WITHOUT -synthetic: a tiny recipe written to disk in Kaldi's formats
(tests/recipe.py: den.fst, tree, 0.trans_mdl, final.mdl, HCLG.fst, words.txt, final.occs, a zip of wavs, label files) goes
through the same file branch a real recipe takes (reference bin/train_chain.py:162-202, bin/train_se.py:145-184,
bin/latgen.py:143-181), and the first-step objective equals what the library computes in-process from the arrays the files
were written from."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from pykaldi2_amd import chain, fbank, kaldi_io, lattice, lstm, ops, se

from recipe import chain_recipe, lattice_recipe, model_yaml

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _seed_model(path, P, hidden=64, layers=2):
    torch.manual_seed(7)
    m = lstm.LSTMAM(80, P, hidden, layers, 0.0, True)
    torch.save({"model": m.state_dict()}, path)
    return m


def _first_loss(stdout):
    m = re.search(r"Loss\s+([-+0-9.eE]+)", stdout)
    assert m, stdout[-1500:]
    return float(m.group(1))


def _quantised(wavs):
    """what the zip holds: 16-bit PCM"""
    return [np.clip(np.round(w * 32768.0), -32768, 32767).astype(np.float32) / 32768.0 for w in wavs]


def test_train_chain_from_files_matches_the_in_process_objective(tmp_path):
    r = chain_recipe(str(tmp_path))
    P = r["P"]
    seed = str(tmp_path / "seed.tar")
    model = _seed_model(seed, P)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "train_chain.py"), "-config", model_yaml(tmp_path / "mmi.yaml", P),
                          "-data", str(tmp_path / "data" / "data.yaml"), "-exp_dir", str(tmp_path / "exp"),
                          "-chain_dir", str(tmp_path / "chain"), "-ali_dir", str(tmp_path / "ali"), "-lang_dir", str(tmp_path / "lang"),
                          "-seed_model", seed, "-lr", "1e-3", "-batch_size", "2", "-sweep_size", "0.001", "-print_freq", "1",
                          "-xent_regularize", "0.1"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    got = _first_loss(out.stdout)
    # the same step in-process, from the arrays: both utterances in one minibatch (their order does not matter to the sum)
    utts = sorted(r["wavs"])
    wavs = _quantised([r["wavs"][u] for u in utts])
    dev = torch.device("cuda", 0)
    fb = fbank.FbankExtractor()
    feats, frames, row_off = fb(torch.from_numpy(np.concatenate(wavs)).to(dev), [w.shape[0] for w in wavs])
    x = fb.pad_roll_subsample(feats, row_off, frames, shift=0, subsample=3, time_major=True)
    aligner = chain.MappedAligner(r["trans_model"])
    sups = [chain.supervision_from_alignment(aligner, r["tree"], r["trans_model"], chain.SupervisionOptions(), r["alis"][u][:T])
            for u, T in zip(utts, frames)]
    den = chain.DenominatorGraph(r["den"], P)
    opts = chain.ChainTrainingOptions(leaky_hmm_coefficient=1e-4, xent_regularize=0.1)
    loss = ops.ChainObjtiveBatch.apply(model.to(dev).forward_time_major(x).transpose(0, 1), den, sups, opts)
    want = loss.item() / float(np.sum(frames))
    assert abs(got - want) <= 2e-4 * abs(want), (got, want)          # (printed with 5 significant digits)
    ck = torch.load(tmp_path / "exp" / "chain.model.0.tar", map_location="cpu", weights_only=False)
    assert set(ck) == {"model", "optimizer", "epoch"}


@pytest.mark.parametrize("criterion", ["mmi", "smbr"])
def test_train_se_from_files_matches_the_in_process_objective(tmp_path, criterion):
    r = lattice_recipe(str(tmp_path))
    P = r["P"]
    dc = dict(beam=10.0, lattice_beam=5.0, max_active=7000, acoustic_scale=0.2)
    seed = str(tmp_path / "seed.tar")
    model = _seed_model(seed, P)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "train_se.py"), "-config", model_yaml(tmp_path / "se.yaml", P, decoder=dc),
                          "-data", str(tmp_path / "data" / "data.yaml"), "-exp_dir", str(tmp_path / "exp"), "-criterion", criterion,
                          "-trans_model", str(tmp_path / "final.mdl"), "-prior_path", str(tmp_path / "final.occs"),
                          "-den_dir", str(tmp_path / "graph"), "-seed_model", seed, "-lr", "1e-5", "-batch_size", "2",
                          "-sweep_size", "0.001", "-print_freq", "1", "-ce_ratio", "0.1"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    got = _first_loss(out.stdout)
    utts = sorted(r["wavs"])
    wavs = _quantised([r["wavs"][u] for u in utts])
    dev = torch.device("cuda", 0)
    o = lattice.LatticeFasterDecoderOptions(beam=dc["beam"], lattice_beam=dc["lattice_beam"], max_active=dc["max_active"])
    o.determinize_lattice = False
    tm = lattice.TransitionModel.from_arrays(r["tm"])
    rec = lattice.MappedLatticeFasterRecognizer(tm, r["hclg"], acoustic_scale=dc["acoustic_scale"], decoder_opts=o)
    batch = dict(wav=torch.from_numpy(np.concatenate(wavs)).to(dev), lens=[w.shape[0] for w in wavs],
                 y=[r["pdfs"][u] for u in utts], aux=[r["tids"][u] for u in utts])
    loss, se_val, ce, frames = se.sequence_loss(model.to(dev), fbank.FbankExtractor(), batch, rec, tm,
                                                se.log_prior_from_counts(r["counts"]).to(dev), criterion, r["tm"]["silence_phones"],
                                                0.1, ops.CrossEntropyLoss(ignore_index=-100, reduction="sum"))
    want = loss.item() / float(np.sum(frames))       # what bin/train_se.py prints as Loss
    assert abs(got - want) <= 2e-4 * abs(want) + 1e-6, (got, want)


def test_latgen_from_files_writes_the_decoders_lattices(tmp_path):
    r = lattice_recipe(str(tmp_path))
    P = r["P"]
    dc = dict(beam=10.0, lattice_beam=5.0, max_active=7000, acoustic_scale=0.2)
    seed = str(tmp_path / "seed.tar")
    model = _seed_model(seed, P)
    out_file = str(tmp_path / "lat.ark")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "latgen.py"), "-config", model_yaml(tmp_path / "se.yaml", P, decoder=dc),
                          "-model_path", seed, "-data_path", str(tmp_path / "data" / "train.zip"), "-batch_size", "2",
                          "-prior_path", str(tmp_path / "final.occs"), "-out_file", out_file, "-trans_model", str(tmp_path / "final.mdl"),
                          "-graph_dir", str(tmp_path / "graph")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lats = dict(kaldi_io.read_compact_lattice_ark(out_file))
    utts = sorted(r["wavs"])
    assert sorted(lats) == utts
    # the same decode in-process from the arrays: same best path cost, words printed through words.txt
    wavs = _quantised([r["wavs"][u] for u in utts])
    dev = torch.device("cuda", 0)
    fb = fbank.FbankExtractor()
    feats, frames, row_off = fb(torch.from_numpy(np.concatenate(wavs)).to(dev), [w.shape[0] for w in wavs])
    x = fb.pad_roll_subsample(feats, row_off, frames, shift=0, subsample=1, time_major=True)
    o = lattice.LatticeFasterDecoderOptions(beam=dc["beam"], lattice_beam=dc["lattice_beam"], max_active=dc["max_active"])
    rec = lattice.MappedLatticeFasterRecognizer(lattice.TransitionModel.from_arrays(r["tm"]), r["hclg"], acoustic_scale=dc["acoustic_scale"],
                                                decoder_opts=o)
    with torch.no_grad():
        ll = model.to(dev).eval().forward_time_major(x).transpose(0, 1) - se.log_prior_from_counts(r["counts"]).to(dev)
        lat = rec.decode_batch(ll, [int(t) for t in frames])
    for j, u in enumerate(utts):
        cl = lat.compact_lattice(j)
        m = re.search(r"Log-like per-frame for utterance %s is ([-+0-9.eE]+)" % re.escape(u), out.stdout)
        assert m and abs(float(m.group(1)) - (-cl["best_cost"] / frames[j])) < 1e-4 * abs(cl["best_cost"] / frames[j]) + 1e-6
        assert (u + " " + " ".join("w%d" % w for w in cl["best_words"])).strip() in out.stdout
