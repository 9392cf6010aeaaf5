"""Host-side data layer: the reference's on-disk formats (zip of wavs + label text files)."""
import io
import wave
import zipfile

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
