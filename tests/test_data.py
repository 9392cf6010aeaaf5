"""Host-side data layer: the reference's on-disk formats (zip of wavs + label text files)."""
import io
import wave
import zipfile

import os

import numpy as np

from pykaldi2_amd import data, synth


def _wav_bytes(x):
    buf = io.BytesIO()
    with wave.open(buf, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
        w.writeframes((x * 32768.0).astype("<i2").tobytes())
    return buf.getvalue()


def test_zip_wav_source_joins_wavs_and_labels(tmp_path):
    rng = np.random.default_rng(0)
    wavs = {"spk1-utt1": rng.uniform(-0.5, 0.5, 16000), "spk1-utt2": rng.uniform(-0.5, 0.5, 8000),
            "unlabelled": rng.uniform(-0.5, 0.5, 4000)}
    zpath = tmp_path / "train.zip"
    with zipfile.ZipFile(zpath, "w") as z:
        for k, v in wavs.items():
            z.writestr("LibriSpeech/%s.wav" % k, _wav_bytes(v))
    T1, T2 = synth.num_fbank_frames(16000), synth.num_fbank_frames(8000)
    (tmp_path / "pdf.txt").write_text("spk1-utt1 " + " ".join(["7"] * T1) + "\nspk1-utt2 " + " ".join(["3"] * (T2 - 5)) + "\n")
    (tmp_path / "tid.txt").write_text("spk1-utt1 " + " ".join(["70"] * T1) + "\nspk1-utt2 " + " ".join(["30"] * (T2 - 5)) + "\n")
    src = data.ZipWavSource([dict(type="Librispeech", wav=str(zpath), label=str(tmp_path / "pdf.txt"),
                                  aux_label=str(tmp_path / "tid.txt"))])
    assert len(src) == 2 and [it[2] for it in src.items] == ["spk1-utt1", "spk1-utt2"]
    seen = {}
    for _ in range(20):
        wav, lab, aux, utt = src.draw()
        seen[utt] = (wav, lab, aux)
    w1, l1, a1 = seen["spk1-utt1"]
    assert w1.dtype == np.float32 and np.abs(w1 - np.round(wavs["spk1-utt1"] * 32768) / 32768).max() < 1e-4
    assert l1.shape[0] == T1 and (a1 == 70).all()
    # labels shorter than the features: both are cut to the label length (reference data/sr_dataset.py:349-363)
    w2, l2, _ = seen["spk1-utt2"]
    assert l2.shape[0] == T2 - 5 and synth.num_fbank_frames(w2.shape[0]) == T2 - 5


def test_synthetic_source_shapes():
    src = data.SyntheticSource(6048, seed=1)
    wav, lab, aux, utt = src.draw()
    assert wav.dtype == np.float32 and 1.3 * 16000 <= wav.shape[0] <= 34.9 * 16000 + 1
    assert lab.shape[0] == synth.num_fbank_frames(wav.shape[0]) and lab.max() < 6048 and aux is None


def test_simulation_pool_from_config_switches():
    from pykaldi2_amd import data
    off = dict(data_config=dict(simulation_prob=0, use_dir_noise=True, use_reverb=True), synthetic=True)
    assert data.SimulationPool.from_config(off) is None
    assert data.SimulationPool.from_config(dict(data_config=dict(simulation_prob=0.5, use_dir_noise=False, use_reverb=False))) is None
    assert data.make_source(off, 120).simulation is None
    on = dict(data_config=dict(simulation_prob=0.7, use_dir_noise=True, use_reverb=True, gain_norm=True, snr_range=[5, 15]),
              synthetic=True)
    pool = data.SimulationPool.from_config(on, seed=1)
    assert pool.prob == 0.7 and pool.gain_norm and len(pool.noises) == 8 and len(pool.rirs) == 16
    assert all(r.dtype == np.float32 and np.argmax(r) >= 32 and r[np.argmax(r)] == 1.0 for r in pool.rirs)
    only_rir = data.SimulationPool.from_config(dict(data_config=dict(simulation_prob=1, use_dir_noise=False, use_reverb=True), synthetic=True))
    assert not only_rir.noises and only_rir.rirs and only_rir.sim.use_rir and not only_rir.sim.use_noise


def _zip_source(tmp_path, n_utts, rank, world, seed=3):
    rng = np.random.default_rng(seed)
    zpath = tmp_path / "many.zip"
    if not zpath.exists():
        with zipfile.ZipFile(zpath, "w") as z:
            for k in range(n_utts):
                z.writestr("u%03d.wav" % k, _wav_bytes(rng.uniform(-0.1, 0.1, int(rng.integers(2000, 30000)))))
    return data.ZipWavSource([dict(wav=str(zpath))], rank=rank, world=world)


def test_every_rank_runs_the_same_number_of_steps(tmp_path):
    """each step ends in a gradient all-reduce, so the number of minibatches of an epoch must not
    depend on the utterance lengths a rank happens to draw.  Finite source = the reference's DistributedSampler
    (data/dataloader.py:83-84): every utterance once per epoch across the ranks, equal counts, padding by wrap-around;
    synthetic source: a count computed from the sweep size alone."""
    import torch
    world, batch = 3, 2
    plans = [data.epoch_plan(_zip_source(tmp_path, 23, r, world), batch, 0, r, world, epoch=1) for r in range(world)]
    assert len({len(p) for p in plans}) == 1 and len(plans[0]) == 4          # ceil(23 / 6) steps on every rank
    flat = [i for p in plans for step in p for i in step]
    assert len(flat) == 24 and set(flat) == set(range(23))                   # one wrapped duplicate
    assert plans != [data.epoch_plan(_zip_source(tmp_path, 23, r, world), batch, 0, r, world, epoch=2) for r in range(world)]
    # the generator itself: same number of minibatches on every rank, utterances really read
    counts = []
    for r in range(world):
        bs = list(data.sequence_batches(_zip_source(tmp_path, 23, r, world), batch, 0, torch.device("cpu"), epoch=0))
        counts.append(len(bs))
        assert all(len(b["lens"]) == batch and b["wav"].shape[0] == sum(b["lens"]) for b in bs)
    assert counts == [4, 4, 4]
    # sweep_size caps the epoch identically on all ranks
    short = [len(data.epoch_plan(_zip_source(tmp_path, 23, r, world), batch, 1e-4, r, world)) for r in range(world)]
    assert short == [1, 1, 1]
    # synthetic generator: ranks draw different lengths, yet the same number of steps
    n = [len(list(data.sequence_batches(data.SyntheticSource(50, rank=r, world=4), 2, 0.02, torch.device("cpu"))))
         for r in range(4)]
    assert n == [3, 3, 3, 3]       # ceil(72 s / (12.3 s * 2))


def test_length_bucketed_plan_gives_the_ranks_of_a_step_similar_lengths(tmp_path):
    world, batch = 4, 2
    srcs = [_zip_source(tmp_path, 200, r, world, seed=5) for r in range(world)]
    dur = srcs[0].durations()
    assert abs(dur[0] - srcs[0].get(0)[0].shape[0] / 16000.0) < 1e-6        # from the archive directory, no decoding
    def spread(bucketed):
        plans = [data.epoch_plan(srcs[r], batch, 0, r, world, epoch=0, length_bucketed=bucketed) for r in range(world)]
        assert len({len(p) for p in plans}) == 1
        flat = sorted(i for p in plans for st in p for i in st)
        assert flat == list(range(200))
        longest = np.array([[max(dur[i] for i in plans[r][k]) for r in range(world)] for k in range(len(plans[0]))])
        return float(np.mean(longest.max(1) - longest.min(1)))
    assert spread(True) < 0.35 * spread(False)
    # synthetic: durations shared by the ranks, content per rank
    a = data.epoch_plan(data.SyntheticSource(50, rank=0, world=2), 2, 0.05, 0, 2, length_bucketed=True)
    b = data.epoch_plan(data.SyntheticSource(50, rank=1, world=2), 2, 0.05, 1, 2, length_bucketed=True)
    assert a == b and all(isinstance(x, float) for st in a for x in st)


def test_wav_durations_come_from_the_headers():
    """durations() reads the fmt / data chunks (any rate, channel count, extra chunks before `data`); an unreadable header
    falls back to 16 kHz 16-bit mono with a 44-byte header."""
    import io
    import struct
    import wave
    from pykaldi2_amd.data import _wav_seconds

    def mk(rate, ch, n):
        b = io.BytesIO()
        with wave.open(b, "wb") as w:
            w.setnchannels(ch); w.setsampwidth(2); w.setframerate(rate); w.writeframes(b"\0" * (2 * ch * n))
        return b.getvalue()
    x = mk(16000, 1, 32000)
    assert _wav_seconds(x[:4096], len(x)) == 2.0
    x = mk(8000, 2, 8000)
    assert _wav_seconds(x[:4096], len(x)) == 1.0
    x = mk(16000, 1, 16000)
    y = x[:36] + b"LIST" + struct.pack("<I", 10) + b"0123456789" + x[36:]
    assert _wav_seconds(y[:4096], len(y)) == 1.0
    assert _wav_seconds(b"junk", 32044) is None            # (the caller estimates and says so)
    # a `data` chunk beyond the first 4 KB (a large LIST chunk): not an estimate any more -- the caller reads further
    z = x[:36] + b"LIST" + struct.pack("<I", 6000) + b"\0" * 6000 + x[36:]
    assert _wav_seconds(z[:4096], len(z)) is None and _wav_seconds(z[:8192], len(z)) == 1.0


def test_durations_read_past_large_header_chunks_and_are_cached(tmp_path, capsys):
    """ZipWavSource.durations() reads on when the `data` chunk is not in the first 4 KB, names the members it had
    to estimate, and keeps the result next to the archive (keyed by the archive's size and mtime) so that the next
    process -- every rank of every job -- does not open every member again."""
    import io
    import json
    import struct
    import wave
    import zipfile

    def mk(n):
        b = io.BytesIO()
        with wave.open(b, "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(b"\0" * (2 * n))
        return b.getvalue()
    x = mk(16000)
    big_list = x[:36] + b"LIST" + struct.pack("<I", 20000) + b"\0" * 20000 + x[36:]
    zpath = str(tmp_path / "a.zip")
    with zipfile.ZipFile(zpath, "w") as z:
        z.writestr("u1.wav", mk(8000)); z.writestr("u2.wav", big_list); z.writestr("u3.wav", b"not a wav file at all" * 100)
    src = data.ZipWavSource([dict(wav=zpath)])
    dur = dict(zip([it[2] for it in src.items], src.durations()))
    assert dur["u1"] == 0.5 and dur["u2"] == 1.0 and abs(dur["u3"] - (2100 - 44) / 32000.0) < 1e-9
    err = capsys.readouterr().err
    assert "1 wav header(s) could not be parsed" in err and "u3.wav" in err and "u2.wav" not in err
    blob = json.load(open(zpath + ".durations.json"))
    assert blob["seconds"]["u2.wav"] == 1.0 and blob["stamp"][0] == os.path.getsize(zpath)
    # an estimate is NOT cached (a later run -- or a fixed parser -- looks at the member again and warns again)
    assert "u3.wav" not in blob["seconds"] and blob["stamp"][2] == data._DURATIONS_CACHE_VERSION
    # a second source answers from the cache: no parsed member is opened again (the archive's member reader records it)
    src2 = data.ZipWavSource([dict(wav=zpath)])
    orig, opened = zipfile.ZipFile.open, []
    try:
        def spy(self, name, *a, **k):
            opened.append(getattr(name, "filename", name))
            return orig(self, name, *a, **k)
        zipfile.ZipFile.open = spy
        assert list(src2.durations()) == list(src.durations())
    finally:
        zipfile.ZipFile.open = orig
    assert opened == ["u3.wav"]
    assert "u3.wav" in capsys.readouterr().err
    # a changed archive invalidates the cache
    with zipfile.ZipFile(zpath, "a") as z:
        z.writestr("u4.wav", mk(4000))
    os.utime(zpath, (1, 1))
    src3 = data.ZipWavSource([dict(wav=zpath)])
    assert dict(zip([it[2] for it in src3.items], src3.durations()))["u4"] == 0.25
