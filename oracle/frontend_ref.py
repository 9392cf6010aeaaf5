"""CPU oracle for the feature front end -- TEST INFRASTRUCTURE ONLY (see oracle/chain_ref.py header).

numpy restatement of
  DataGeneratorTrain._logfbank_extractor   reference data/sr_dataset.py:279-296
  stft / _enframe (end='pad')              reference simulation/freq_analysis.py:113-150, 41-110
  cmn(axis=0)                              reference reader/preprocess.py:34-41
  SeqDataloader padding                    reference data/dataloader.py:96-103
  roll + unfold subsampling                reference bin/train_chain.py:251-255
Pinned against tests/golden/fbank.npz and misc.npz, which tools/gen_golden.py produced by running
the reference's own code (tests/test_oracle_frontend.py).
"""
import numpy as np


def normalised_mel(mel):
    """sr_dataset.py:283-286: divide each FFT-bin column by its sum over filters (0 -> -1)."""
    t1 = np.sum(mel, 0)
    t1[t1 == 0] = -1
    return mel.dot(np.diag(1 / t1)).T  # [257, 80] float32


def enframe_pad(y, shift=160, length=400):
    """freq_analysis.py:64-69: zero-pad the tail so that (N + shift - length) % shift == 0."""
    n = y.shape[0]
    rem = (n + shift - length) % shift
    if rem != 0:
        y = np.concatenate([y, np.zeros(shift - rem, dtype=y.dtype)])
    nfr = (y.shape[0] + shift - length) // shift
    idx = np.arange(length)[None, :] + shift * np.arange(nfr)[:, None]
    return y[idx]


def logfbank(wav, mel):
    wav = np.asarray(wav, dtype=np.float32)
    y = wav[1:] - np.float32(0.96) * wav[:-1]                  # float32 (sr_dataset.py:288)
    frames = enframe_pad(y)
    spec = np.fft.fft(np.hamming(400)[None, :] * frames, n=512, axis=1)[:, :257].astype(np.complex64)
    power = np.abs(spec) ** 2                                  # float32
    return np.log(power.dot(normalised_mel(mel) * 32768 ** 2) + 1)


def cmn(feats):
    return feats - np.mean(feats, axis=0, keepdims=True)


def pad_roll_subsample(feat_list, shift=0, subsample=1):
    """x[n, j] = roll(pad(feat_n), shift)[j*subsample]; shape [N, (Tmax-1)//subsample+1, D]."""
    tmax = max(f.shape[0] for f in feat_list)
    x = np.zeros((len(feat_list), tmax, feat_list[0].shape[1]), dtype=np.float32)
    for i, f in enumerate(feat_list):
        x[i, :f.shape[0]] = f
    x = np.roll(x, shift, axis=1)
    return x[:, ::subsample]


def utt2seg(data, seg_len, seg_shift):
    """_utt2seg on a [T, D] matrix (reference data/sr_dataset.py:40-52 works on the transpose):
    n_seg = floor((T - seg_len) / seg_shift) + 1 non-overlapping-by-shift windows, tail dropped."""
    T = data.shape[0]
    n_seg = int(np.floor((T - seg_len) / seg_shift)) + 1
    return [data[i * seg_shift:i * seg_shift + seg_len] for i in range(max(0, n_seg))]


def mvn_apply(x, mean_vec, std_vec):
    """GlobalMeanVarianceNormalization.apply_on_ndarray (reference reader/preprocess.py:211-229)."""
    return (x - mean_vec) / std_vec
