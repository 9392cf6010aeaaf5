"""oracle/frontend_ref.py against the goldens produced by the reference's own code."""
import numpy as np

from oracle import frontend_ref as F
from pykaldi2_amd import fbank as fb_mod, synth, utils


def test_mel_generator_reproduces_reference_table(golden):
    g = golden("fbank")
    assert np.array_equal(fb_mod.mel_filterbank(), g["mel"])


def test_oracle_fbank_and_cmn_match_reference(golden):
    g = golden("fbank")
    for i, n in enumerate(g["lens"]):
        got = F.logfbank(g["wav%d" % i], g["mel"])
        want = g["fbank%d" % i]
        assert got.shape == want.shape == (synth.num_fbank_frames(int(n)), 80)
        assert np.abs(got - want).max() <= 2e-5, (i, np.abs(got - want).max())
        assert np.abs(F.cmn(got) - g["cmn%d" % i]).max() <= 3e-5
    assert not g["fbank4"].any()  # all-zero waveform -> exactly 0


def test_pad_and_subsample_match_reference(golden):
    g = golden("misc")
    lens = g["collate_lens"]
    feats = np.split(g["collate_feats"], np.cumsum(lens)[:-1])
    assert np.array_equal(F.pad_roll_subsample(feats), g["collate_x"])
    for e in range(3):
        assert np.array_equal(F.pad_roll_subsample(feats, shift=-(e % 3), subsample=3), g["subsample_e%d" % e])
    y = g["collate_y"]
    assert y.dtype == np.int64 and (y[1, 5:] == -100).all() and list(g["collate_num_frs"]) == list(lens)


def test_noam_decay_matches_reference(golden):
    g = golden("misc")
    for s, lr in zip(g["noam_steps"], g["noam_lr"]):
        assert utils.noam_decay(int(s), 4000, 1e-3) == lr


def test_chunking_and_mvn_match_reference(golden):
    g = golden("chunk_mvn")
    for i in range(5):
        segs = F.utt2seg(g["feats%d" % i], 80, 80)
        want = g["seg_x%d" % i]
        assert len(segs) == want.shape[0]
        if segs:
            assert np.array_equal(np.stack(segs), want)
            assert np.array_equal(np.stack(F.utt2seg(g["labels%d" % i], 80, 80)), g["seg_y%d" % i])
    assert np.array_equal(F.mvn_apply(g["mvn_in"], g["mvn_mean"], g["mvn_std"]), g["mvn_out"])
    # the std floor of learn_mean_and_variance_from_stats (reader/preprocess.py:141-149)
    m = fb_mod.GlobalMeanVarianceNormalization.from_stats(np.full(4, 20.0), np.full(4, 40.0), 10)
    assert np.array_equal(m.mean_vec, np.full((1, 4), 2.0, np.float32)) and np.array_equal(m.std_vec, np.full((1, 4), 1e-2, np.float32))
