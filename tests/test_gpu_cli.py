"""The training command lines run end to end on synthetic data (tiny model) and write the reference's
checkpoint format."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg(tmp_path, name, label_size, seq):
    cfg = dict(data_config=dict(frame_len=400, frame_shift=160, seg_len=80, seg_shift=80, sequence_mode=seq,
                                load_label=True, use_cmn=True, simulation_prob=0),
               model_config=dict(feat_dim=80, hidden_size=64, dropout=0.1, num_layers=2, label_size=label_size))
    p = tmp_path / name
    p.write_text(yaml.safe_dump(cfg))
    return str(p)


def test_train_chain_cli_synthetic(tmp_path):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "train_chain.py"), "-config",
                          _cfg(tmp_path, "mmi.yaml", 120, True), "-exp_dir", str(tmp_path / "exp"), "-lr", "1e-3",
                          "-batch_size", "2", "-sweep_size", "0.02", "-print_freq", "1", "-xent_regularize", "0.1",
                          "-synthetic", "-den_states", "400", "-den_arcs", "6000"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "Epoch: [0]" in out.stdout and "grad_norm" in out.stdout
    ck = torch.load(tmp_path / "exp" / "chain.model.0.tar", map_location="cpu", weights_only=False)
    assert set(ck) == {"model", "optimizer", "epoch"} and "lstm.weight_hh_l1_reverse" in ck["model"]


def test_simulated_batches_on_device():
    """sequence_batches with the dynamic simulation on: every utterance reverberated + noised on the device, peak
    normalised to 0.5, lengths unchanged; with simulation_prob = 0 the waveforms are the source's."""
    from pykaldi2_amd import data
    cfg = dict(data_config=dict(simulation_prob=1.0, use_dir_noise=True, use_reverb=True, gain_norm=True), synthetic=True)
    src = data.make_source(cfg, 120)
    np.random.seed(0)
    b = next(data.sequence_batches(src, 3, 1.0, torch.device("cuda")))
    off, clean = 0, data.make_source(dict(synthetic=True), 120)
    for n in b["lens"]:
        w = b["wav"][off:off + n]
        assert torch.isfinite(w).all() and abs(float(w.abs().max()) - 0.5) < 1e-5
        ref = clean.draw()[0]
        assert ref.shape[0] == n and not np.allclose(w.cpu().numpy(), ref, atol=1e-3)
        off += n
    assert off == b["wav"].numel()


def test_train_ce_cli_synthetic(tmp_path):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "train_ce.py"), "-train_config",
                          _cfg(tmp_path, "ce.yaml", 120, False), "-exp_dir", str(tmp_path / "exp"), "-lr", "1e-3",
                          "-batch_size", "16", "-sweep_size", "0.05", "-print_freq", "1", "-synthetic",
                          "-global_mvn", "1", "-mvn_utterances", "6"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "Epoch: [0]" in out.stdout and "Loss" in out.stdout
    ck = torch.load(tmp_path / "exp" / "model.0.tar", map_location="cpu", weights_only=False)
    assert ck["epoch"] == 0 and "output_layer.bias" in ck["model"]
    # -global_mvn wrote the transform; the sequence CLIs take it back through -transform
    from pykaldi2_amd import fbank
    t = fbank.GlobalMeanVarianceNormalization.load(str(tmp_path / "exp" / "transform.pkl"))
    assert t.mean_vec.shape == (1, 80) and np.isfinite(t.mean_vec).all() and (t.std_vec >= 1e-2).all()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "train_chain.py"), "-config",
                          _cfg(tmp_path, "mmi.yaml", 120, True), "-exp_dir", str(tmp_path / "exp2"), "-lr", "1e-3",
                          "-batch_size", "2", "-sweep_size", "0.01", "-print_freq", "1", "-synthetic", "-den_states", "400",
                          "-den_arcs", "6000", "-transform", str(tmp_path / "exp" / "transform.pkl")],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "Epoch: [0]" in out.stdout


@pytest.mark.parametrize("criterion", ["mmi", "smbr"])
def test_train_se_cli_synthetic(tmp_path, criterion):
    cfg = yaml.safe_load(open(_cfg(tmp_path, "se.yaml", 120, True)))
    cfg["decoder_config"] = dict(beam=9.0, lattice_beam=4.0, max_active=400, acoustic_scale=0.3, align_beam=10)
    (tmp_path / "se.yaml").write_text(yaml.safe_dump(cfg))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "train_se.py"), "-config", str(tmp_path / "se.yaml"),
                          "-exp_dir", str(tmp_path / "exp"), "-lr", "1e-4", "-momentum", "0.9", "-criterion", criterion,
                          "-batch_size", "2", "-sweep_size", "0.02", "-print_freq", "1", "-synthetic", "-graph_words", "60"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "Epoch: [0]" in out.stdout and "grad_norm" in out.stdout
    ck = torch.load(tmp_path / "exp" / "model.se.0.tar", map_location="cpu", weights_only=False)
    assert set(ck) == {"model", "optimizer", "epoch"} and "lstm.weight_hh_l1_reverse" in ck["model"]


def test_train_transformer_ce_cli_synthetic(tmp_path):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "train_transformer_ce.py"), "-train_config",
                          _cfg(tmp_path, "ce.yaml", 120, True), "-exp_dir", str(tmp_path / "exp"), "-lr", "1e-3",
                          "-batch_size", "3", "-sweep_size", "0.02", "-print_freq", "1", "-synthetic", "-nlayers", "2",
                          "-dim_model", "64", "-nheads", "4", "-ff_size", "128", "-look_ahead", "6", "-warmup_step", "10"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "Epoch: [0]" in out.stdout and "grad_norm" in out.stdout
    ck = torch.load(tmp_path / "exp" / "model.0.tar", map_location="cpu", weights_only=False)
    assert set(ck) == {"model", "optimizer", "epoch"}
    assert "transformer.layers.1.encoder_layer.self_attn.in_proj_weight" in ck["model"] and "output_layer.weight" in ck["model"]


def test_train_transformer_se_cli_synthetic(tmp_path):
    cfg = yaml.safe_load(open(_cfg(tmp_path, "se.yaml", 120, True)))
    cfg["decoder_config"] = dict(beam=9.0, lattice_beam=4.0, max_active=400, acoustic_scale=0.3, align_beam=10)
    (tmp_path / "se.yaml").write_text(yaml.safe_dump(cfg))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "train_transformer_se.py"), "-config", str(tmp_path / "se.yaml"),
                          "-exp_dir", str(tmp_path / "exp"), "-lr", "1e-4", "-momentum", "0.9", "-criterion", "mmi",
                          "-batch_size", "2", "-sweep_size", "0.02", "-print_freq", "1", "-synthetic", "-graph_words", "60",
                          "-nlayers", "2", "-dim_model", "64", "-nheads", "4", "-ff_size", "128", "-dataPath", ""],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "Epoch: [0]" in out.stdout and "grad_norm" in out.stdout
    ck = torch.load(tmp_path / "exp" / "model.se.0.tar", map_location="cpu", weights_only=False)
    assert set(ck) == {"model", "optimizer", "epoch"} and "transformer.layers.0.conv1d.weight" in ck["model"]


def test_dump_loglikes_cli_synthetic(tmp_path):
    from pykaldi2_amd import kaldi_io
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "dump_loglikes.py"), "-config",
                          _cfg(tmp_path, "ce.yaml", 120, True), "-out_file", str(tmp_path / "ll.ark"), "-batch_size", "2",
                          "-synthetic", "3"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    mats = list(kaldi_io.read_matrix_ark(str(tmp_path / "ll.ark")))
    assert len(mats) == 3 and all(m.shape[1] == 120 and m.shape[0] > 100 and np.isfinite(m).all() for _, m in mats)


def test_latgen_cli_synthetic_writes_compact_lattices(tmp_path):
    """bin/latgen.py (reference bin/latgen.py:143-181): model forward -> on-device lattice generation -> Kaldi compact-lattice
    archive, best word sequence and per-frame log-likelihood printed per utterance."""
    from pykaldi2_amd import kaldi_io
    import yaml as _yaml
    cfg = _yaml.safe_load(open(_cfg(tmp_path, "se.yaml", 120, True)))
    cfg["decoder_config"] = dict(beam=13, lattice_beam=7, max_active=7000, acoustic_scale=0.1)
    (tmp_path / "se.yaml").write_text(_yaml.safe_dump(cfg))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "latgen.py"), "-config", str(tmp_path / "se.yaml"),
                          "-out_file", str(tmp_path / "lat.ark"), "-synthetic", "3", "-synthetic_words", "300", "-batch_size", "2"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lats = list(kaldi_io.read_compact_lattice_ark(str(tmp_path / "lat.ark")))
    assert [k for k, _ in lats] == ["synth-1", "synth-2", "synth-3"]
    for key, lat in lats:
        # determinised on the word labels, like the reference's (decoder_opts.determinize_lattice = True): every arc carries
        # a word and a transition-id string, no two arcs of a state carry the same word
        assert lat["start"] == 0 and len(lat["arcs"]) >= 1 and any(np.isfinite(f[0]) for f in lat["finals"])
        assert all(0 <= a[1] < lat["num_states"] and a[2] == a[3] and a[2] > 0 for a in lat["arcs"])
        assert len(set((a[0], a[2]) for a in lat["arcs"])) == len(lat["arcs"])
        assert sum(len(a[4][2]) for a in lat["arcs"]) > 0
        assert "Log-like per-frame for utterance %s is " % key in out.stdout
    # -no_determinize: the raw lattice-beam-pruned state-level lattice (arc = decoder link, at most one transition-id)
    raw = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "latgen.py"), "-config", str(tmp_path / "se.yaml"), "-no_determinize",
                          "-out_file", str(tmp_path / "raw.ark"), "-synthetic", "3", "-synthetic_words", "300", "-batch_size", "2"],
                         capture_output=True, text=True, timeout=600)
    assert raw.returncode == 0, raw.stdout[-2000:] + raw.stderr[-2000:]
    for (key, lat), (_, det) in zip(kaldi_io.read_compact_lattice_ark(str(tmp_path / "raw.ark")), lats):
        assert len(lat["arcs"]) > 500 and len(lat["arcs"]) > len(det["arcs"]) and all(len(a[4][2]) <= 1 for a in lat["arcs"])
        assert any(a[2] > 0 for a in lat["arcs"])
    # the best word sequence of every utterance is printed (ids: the synthetic graph has no words.txt)
    texts = [l for l in out.stdout.splitlines() if l.startswith("synth-")]
    assert len(texts) == 3 and all(len(t.split()) >= 2 and all(w.isdigit() for w in t.split()[1:]) for t in texts)


def test_best_path_of_the_compact_lattice_is_the_decoders():
    """LatticeBatch.compact_lattice: the back-traced best path costs what the decoder reported, consumes one transition-id
    per frame, and its words are the output labels of the HCLG arcs it crossed."""
    import torch
    from pykaldi2_amd import lattice, synth
    P, T = 90, 57
    g = synth.decoding_graph_arcs(80, P, seed=5)
    tm = synth.transition_model_arrays(P)
    rec = lattice.MappedLatticeFasterRecognizer(lattice.TransitionModel.from_arrays(tm), g, 0.3,
                                                lattice.LatticeFasterDecoderOptions(beam=12.0, lattice_beam=5.0))
    ll = torch.from_numpy((2.0 * np.random.default_rng(2).standard_normal((T, P))).astype(np.float32)).cuda()
    lat = rec.decode(ll)
    cl = lat.compact_lattice(0)
    assert abs(cl["best_cost"] - float(lat.best_cost[0])) < 1e-3 * max(1.0, abs(float(lat.best_cost[0])))
    assert len(cl["best_tids"]) == T and len(cl["best_words"]) >= 1
    assert set(cl["best_words"]) <= set(g["olabel"][g["olabel"] > 0].tolist())
    # every word arc of the lattice carries the label of a word-entry arc of the graph; emitting links carry none
    assert (cl["word"][cl["tid"] > 0] == 0).all() and (cl["word"] >= 0).all()
    # determinised form (reference bin/latgen.py:149): deterministic on words, and its best path costs what the decoder's does
    det = lat.compact_lattice(0, determinize=True)
    assert det["determinized"] and (det["word"] > 0).all() and det["src"].shape[0] < cl["src"].shape[0]
    assert len(set(zip(det["src"].tolist(), det["word"].tolist()))) == det["src"].shape[0]
    n = det["num_states"]
    d = np.full(n, np.inf); d[det["start"]] = 0.0
    cost = det["graph"].astype(np.float64) + 0.3 * det["acoustic"].astype(np.float64)
    for _ in range(n):                                   # Bellman-Ford on the small acyclic result
        nd = d.copy()
        np.minimum.at(nd, det["dst"], d[det["src"]] + cost)
        if np.array_equal(nd, d):
            break
        d = nd
    fin = det["final"].astype(np.float64) + 0.3 * np.where(np.isfinite(det["final_acoustic"]), det["final_acoustic"], 0.0)
    assert abs(float(np.min(d + fin)) - cl["best_cost"]) < 2e-3 * max(1.0, abs(cl["best_cost"]))
    assert det["best_words"] == cl["best_words"]


def test_decode_py_dumps_subsampled_loglikes(tmp_path):
    """decode.py (reference decode.py:102-122): log-likelihood matrices, every third frame with -frame_subsampling_factor 3."""
    from pykaldi2_amd import kaldi_io
    out = subprocess.run([sys.executable, os.path.join(ROOT, "decode.py"), "-config", _cfg(tmp_path, "ce.yaml", 120, True),
                          "-out_file", str(tmp_path / "ll.ark"), "-synthetic", "2", "-frame_subsampling_factor", "3"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    full = subprocess.run([sys.executable, os.path.join(ROOT, "decode.py"), "-config", _cfg(tmp_path, "ce.yaml", 120, True),
                           "-out_file", str(tmp_path / "ll1.ark"), "-synthetic", "2"], capture_output=True, text=True, timeout=600)
    assert full.returncode == 0
    a = dict(kaldi_io.read_matrix_ark(str(tmp_path / "ll.ark")))
    b = dict(kaldi_io.read_matrix_ark(str(tmp_path / "ll1.ark")))
    assert set(a) == set(b) and len(a) == 2
    for k in a:
        assert a[k].shape == (b[k].shape[0] // 3, b[k].shape[1]) and np.isfinite(a[k]).all()


def test_bench_rccl_path_single_rank_group():
    """The driver's N>1 launch line with one rank: process group over RCCL, bucketed gradient all-reduce on the
    side stream while the step graphs replay, barrier-bracketed timing.  With one rank the exchange is the
    identity, so the objective after the same steps must equal the run without a process group."""
    import json
    base = [os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    env = dict(os.environ, PK2_HVD_SINGLE_RANK_GROUP="1", PK2_HVD_OVERLAP="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                           "--master-addr", "127.0.0.1", "--master-port", "29541"] + base,
                          capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert dist.returncode == 0, dist.stdout[-2000:] + dist.stderr[-3000:]
    solo = subprocess.run([sys.executable] + base, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert solo.returncode == 0, solo.stdout[-2000:] + solo.stderr[-3000:]
    a = json.loads(dist.stdout.strip().splitlines()[-1])
    b = json.loads(solo.stdout.strip().splitlines()[-1])
    assert a["n_gpus"] == 1 and a["config"]["parallelism"] == "dp1"
    # the gradients went through the library's own RCCL communicator (pk2_comm_init / pk2_allreduce_bucket), bucketed
    assert a["exchange"]["api"] == "pk2_allreduce_bucket" and "rccl" in a["exchange"]["library"]
    assert a["exchange"]["schedule"] == "overlap" and b["exchange"]["api"] == "torch.distributed.all_reduce"
    # the communicator's own rank count (pk2_comm_info), the per-rank block of one rank, no bucketed second window
    assert a["exchange"]["ranks"] == 1 and b["exchange"]["ranks"] == 1 and len(a["per_rank"]) == 1 and a["irtf_bucketed"] is None
    assert a["exchange"]["exchange_ms"] > 0 and a["exchange"]["straggler_wait_ms"] == 0 and b["exchange"]["exchange_ms"] == 0
    assert a["per_rank"][0]["gpu_ms_max"] >= a["per_rank"][0]["gpu_ms_mean"] > 0 and a["ideal_weak_irtf"] > 0
    # split-K GEMMs accumulate with float atomics: equal up to summation order
    assert abs(a["last_objf_per_frame"] - b["last_objf_per_frame"]) <= 2e-3 * abs(b["last_objf_per_frame"]) + 1e-4


def _check_scaling_fields(d, world):
    """an N > 1 line must explain its own scaling -- ranks of the communicator, per-rank GPU time of the
    timed steps, exchange / straggler split, the weak-scaling ideal, and the length-bucketed protocol's number from a
    second short window (bench.py::scaling_report; DESIGN section 6 says how to read them)."""
    ex = d["exchange"]
    assert ex["ranks"] == world
    for k in ("exchange_ms", "exchange_net_ms", "straggler_wait_ms", "straggler_wait_ms_worst_rank", "bytes_per_step"):
        assert k in ex and ex[k] >= 0, k
    assert ex["exchange_ms"] > 0 and 80e6 < ex["bytes_per_step"] < 90e6          # the flat gradient buffer: 21 M floats
    assert ex["exchange_net_ms"] <= ex["exchange_ms"] + 1e-6
    pr = d["per_rank"]
    assert [p["rank"] for p in pr] == list(range(world))
    for p in pr:
        assert 0 < p["gpu_ms_mean"] <= p["gpu_ms_max"] and 0 < p["pre_exchange_ms_mean"] <= p["gpu_ms_mean"]
        assert p["audio_s"] > 0 and p["own_irtf"] > 0 and p["straggler_wait_ms_mean"] >= 0
    # somebody is the slowest rank of every step, nobody waits a negative time
    assert min(p["straggler_wait_ms_mean"] for p in pr) <= ex["straggler_wait_ms"] <= max(p["straggler_wait_ms_mean"] for p in pr) + 1e-6
    assert abs(d["ideal_weak_irtf"] - sum(p["own_irtf"] for p in pr)) < 0.05 * world
    assert d["irtf_bucketed"]["value"] > 0 and d["irtf_bucketed"]["steps"] >= 2


def test_bench_two_ranks_on_one_gpu_over_gloo():
    """Two bench.py ranks (the driver's torch.distributed.run line with N = 2) sharing this GPU over gloo -- RCCL refuses
    two ranks on one device, the collective pattern is the same: parameter broadcast, bucketed gradient all-reduce on the
    side stream, barrier-bracketed timing, MAX / SUM of the timings, and every rank reaching the same collectives (a
    rank-0-only extra step would hang here).  Rank 0 prints the one JSON line with n_gpus = 2."""
    import json
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29547", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
           "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT,
                         env=dict(os.environ, PK2_HVD_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 8
    assert d["scaling"] == "weak" and d["value"] > 0 and d["cpu_baseline"] is None
    # with more than one rank the exchange schedule is measured at start-up (2 + 2 x 6 steps, the two schedules in turn)
    # and every rank switches to the same one; bench.py runs that calibration to its end BEFORE its warm-up steps
    assert "[hvd] gradient exchange schedule: " in out.stderr and d["exchange"]["schedule"] in ("single", "overlap")
    assert d["exchange"]["calibration_steps"] == 14 and d["steps"] == 4
    _check_scaling_fields(d, 2)


def test_overlap_schedule_50_steps_keeps_the_persistent_kernels(tmp_path):
    """the bucketed schedule (RCCL all-reduces of finished buckets on the side stream, here on a one-rank
    communicator) pinned for 50 LF-MMI steps: no persistent kernel may time out or fall back while RCCL's kernels share the
    chip with them (lstm_persist abort flag 0, denominator still on the persistent path), and the objective equals the
    single-schedule run of the same steps."""
    import json
    base = [os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "50", "--warmup", "2", "--no-cpu-baseline"]
    env = dict(os.environ, PK2_HVD_SINGLE_RANK_GROUP="1", PK2_HVD_OVERLAP="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                           "--master-addr", "127.0.0.1", "--master-port", "29551"] + base,
                          capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert dist.returncode == 0, dist.stdout[-2000:] + dist.stderr[-3000:]
    solo = subprocess.run([sys.executable] + base, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert solo.returncode == 0, solo.stdout[-2000:] + solo.stderr[-3000:]
    a = json.loads(dist.stdout.strip().splitlines()[-1])
    b = json.loads(solo.stdout.strip().splitlines()[-1])
    assert a["exchange"]["schedule"] == "overlap" and a["exchange"]["api"] == "pk2_allreduce_bucket"
    for d in (a, b):
        assert d["persistent_health"] == dict(lstm_persist_abort=0, den_kernel_path=2, den_persist_form=2, guard_raised=False), d["persistent_health"]
        assert np.isfinite(d["last_objf_per_frame"])
    # 52 optimiser steps apart the two runs still agree (split-K float atomics: equal up to summation order)
    assert abs(a["last_objf_per_frame"] - b["last_objf_per_frame"]) <= 5e-3 * abs(b["last_objf_per_frame"]) + 1e-3


def test_overlap_schedule_survives_a_cu_hogging_peer():
    """what the all-reduce kernel of an 8-rank job does to the chip, on this one GPU -- in front of every bucket's
    all-reduce, on the side stream, 32 workgroups stream a reduce-copy over the bucket 12 times (PK2_HVD_FAKE_PEER=32,12:
    ~2.5 ms of co-resident memory traffic per step, under the backward recurrences).  The persistent kernels must neither time
    out nor fall back, the gradients keep their values (the stand-in adds zeros), and the step pays a fraction of what the
    same exchange costs behind backward on the compute stream (profiles/r05_peer_coresidency.txt)."""
    import json
    base = [os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "40", "--warmup", "3", "--no-cpu-baseline"]
    out = {}
    for tag, extra in (("plain", {}), ("peer", dict(PK2_HVD_FAKE_PEER="32,12"))):
        env = dict(os.environ, PK2_HVD_SINGLE_RANK_GROUP="1", PK2_HVD_OVERLAP="1", HSA_ENABLE_IPC_MODE_LEGACY="0", **extra)
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                            "--master-addr", "127.0.0.1", "--master-port", "29557"] + base,
                           capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        out[tag] = json.loads(r.stdout.strip().splitlines()[-1])
    a, b = out["peer"], out["plain"]
    for d in (a, b):
        assert d["exchange"]["schedule"] == "overlap"
        assert d["persistent_health"] == dict(lstm_persist_abort=0, den_kernel_path=2, den_persist_form=2, guard_raised=False), d["persistent_health"]
    assert a["exchange"]["exchange_ms"] > 1.0 > b["exchange"]["exchange_ms"]          # the stand-in really ran: > 1 ms per step
    assert abs(a["last_objf_per_frame"] - b["last_objf_per_frame"]) <= 5e-3 * abs(b["last_objf_per_frame"]) + 1e-3
    # hidden under backward: the step grows by far less than the exchange takes
    assert a["ms_per_step"] - b["ms_per_step"] < 0.5 * a["exchange"]["exchange_ms"], (a["ms_per_step"], b["ms_per_step"], a["exchange"])


def test_bench_eight_ranks_on_one_gpu_over_gloo():
    """The driver's N = 8 line (`torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8`) with all ranks on this GPU
    over gloo: rendezvous, parameter broadcast, the start-up trial of the two exchange schedules, the same number of
    collectives on every rank, one JSON line from rank 0 with the whole-job aggregate."""
    import json
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", "29553", os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=ROOT,
                         env=dict(os.environ, PK2_HVD_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2"))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["config"]["parallelism"] == "dp8" and d["config"]["global_batch"] == 32
    assert d["scaling"] == "weak" and d["value"] > 0 and d["exchange"]["calibration_steps"] == 14
    assert d["persistent_health"]["lstm_persist_abort"] == 0
    _check_scaling_fields(d, 8)
