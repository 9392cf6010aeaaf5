"""On-device 80-dim log-mel filterbank front end (libpk2hip.so).

Replaces DataGeneratorTrain._logfbank_extractor + cmn (reference data/sr_dataset.py:279-296,
365-366) and the roll/unfold frame subsampling of bin/train_chain.py:251-255 for waveforms that
are already in HBM.  The mel matrix of the reference is the table data/mel80_window.txt, produced
(per the comment at data/sr_dataset.py:270-273) by librosa.filters.mel(16000, 512, n_mels=80,
fmax=7690, htk=True); ``mel_filterbank()`` below regenerates it bit for bit (pinned by
tests/golden/fbank.npz), and ``load_mel(path)`` reads a user-supplied table in the same format.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib

def mel_filterbank(sr=16000, n_fft=512, n_mels=80, fmin=0.0, fmax=7690.0):
    """HTK-scale triangular mel filters with Slaney area normalisation, [n_mels, n_fft/2+1] f32."""
    def hz2mel(f):
        return 2595.0 * np.log10(1.0 + f / 700.0)

    def mel2hz(m):
        return 700.0 * (10.0 ** (m / 2595.0) - 1.0)
    fftfreqs = np.linspace(0, sr / 2, n_fft // 2 + 1)
    mel_f = mel2hz(np.linspace(hz2mel(fmin), hz2mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    w = np.zeros((n_mels, n_fft // 2 + 1))
    for i in range(n_mels):
        w[i] = np.maximum(0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w.astype(np.float32)


def load_mel(path):
    with open(path) as f:
        rows = [np.asarray([np.float32(v) for v in line.rstrip("\n").split(",")]) for line in f if line.strip()]
    mel = np.vstack(rows).astype(np.float32)
    assert mel.shape == (80, 257), mel.shape
    return mel


def num_frames(num_samples):
    return int(_lib.lib().pk2_fbank_num_frames(int(num_samples)))


class FbankExtractor:
    def __init__(self, mel=None):
        mel = mel_filterbank() if mel is None else np.ascontiguousarray(mel, dtype=np.float32)
        h = C.c_void_p()
        _lib.check(_lib.lib().pk2_fbank_create(_lib.ptr(mel), C.byref(h)))
        self._h = h

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.lib().pk2_fbank_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def __call__(self, wav, wav_lengths, apply_cmn=True):
        """wav: CUDA f32 1-D tensor holding the utterances back to back; wav_lengths: list of sample
        counts.  Returns (feats [sum_T, 80] CUDA f32, frames list, row_off CUDA int64 [N+1])."""
        _lib.require_gpu()
        assert wav.is_cuda and wav.dtype == torch.float32 and wav.is_contiguous()
        n = len(wav_lengths)
        wav_off = np.zeros(n + 1, dtype=np.int64)
        wav_off[1:] = np.cumsum(wav_lengths)
        frames = [num_frames(l) for l in wav_lengths]
        row_off_h = np.zeros(n + 1, dtype=np.int64)
        row_off_h[1:] = np.cumsum(frames)
        # pinned + non_blocking: a pageable copy would block the host until the stream has drained the previous
        # training step (the host then never runs ahead of the GPU)
        row_off = torch.from_numpy(row_off_h).pin_memory().to(wav.device, non_blocking=True)
        feats = torch.empty(int(row_off_h[-1]), 80, dtype=torch.float32, device=wav.device)
        _lib.check(_lib.lib().pk2_fbank_compute(self._h, _lib.ptr(wav), _lib.ptr(wav_off), n, _lib.ptr(feats),
                                                _lib.ptr(row_off), 1 if apply_cmn else 0,
                                                _lib.stream_ptr(wav.device)))
        return feats, frames, row_off

    @staticmethod
    def pad_roll_subsample(feats, row_off, frames, shift=0, subsample=1, time_major=False):
        """Zero-pad to max(frames), torch.roll by `shift` along time, keep every `subsample`-th frame
        (reference bin/train_chain.py:251-255; with subsample=1, shift=0 it is SeqDataloader's padding,
        data/dataloader.py:96-103).  Returns [N, T', 80] or [T', N, 80]."""
        n = len(frames)
        max_t = int(max(frames))
        out_t = (max_t - 1) // subsample + 1
        shape = (out_t, n, 80) if time_major else (n, out_t, 80)
        x = torch.empty(*shape, dtype=torch.float32, device=feats.device)
        _lib.check(_lib.lib().pk2_pad_roll_subsample(_lib.ptr(feats), _lib.ptr(row_off), n, max_t, int(shift),
                                                     int(subsample), _lib.ptr(x), out_t, 1 if time_major else 0,
                                                     _lib.stream_ptr(feats.device)))
        return x


def utt2seg(feats, seg_len, seg_shift):
    """Cuts one utterance [T, D] into segments [n_seg, seg_len, D], n_seg = floor((T-seg_len)/seg_shift)+1,
    the tail is dropped (reference data/sr_dataset.py:40-52, used for CE chunks at :374-382).  Pure
    re-indexing: a strided view of the feature tensor, materialised once."""
    T = feats.shape[0]
    n_seg = (T - seg_len) // seg_shift + 1 if T >= seg_len else 0
    if n_seg <= 0:
        return feats.new_zeros((0, seg_len) + tuple(feats.shape[1:]))
    inner = int(np.prod(feats.shape[1:])) if feats.dim() > 1 else 1
    f = feats.contiguous()
    view = f.as_strided((n_seg, seg_len) + tuple(f.shape[1:]), (seg_shift * inner, inner) + tuple(f.stride()[1:]))
    return view.contiguous()


class _MvnState:
    """What GlobalMeanVarianceNormalization.save pickles (attribute names of the reference's class)."""

    def __init__(self, mean_vec, std_vec):
        self.mean_vec, self.std_vec, self.mean_norm, self.var_norm = mean_vec, std_vec, True, True


class GlobalMeanVarianceNormalization:
    """Apply side of reader/preprocess.py:89-229: (x - mean_vec) / std_vec with the pickled [1, D] vectors
    (std floored at 1e-2 when it was estimated, :141-149).  Estimation stays a host-side, one-off step."""

    def __init__(self, mean_vec, std_vec):
        self.mean_vec = np.ascontiguousarray(mean_vec, dtype=np.float32).reshape(1, -1)
        self.std_vec = np.ascontiguousarray(std_vec, dtype=np.float32).reshape(1, -1)
        self._dev = None

    @classmethod
    def from_stats(cls, mean_stats, var_stats, n_frame):
        """learn_mean_and_variance_from_stats (reader/preprocess.py:141-151)."""
        mean = np.asarray(mean_stats, dtype=np.float32).reshape(-1, 1) / n_frame
        std = np.sqrt(np.asarray(var_stats, dtype=np.float32).reshape(-1, 1) / n_frame - mean ** 2)
        std = np.maximum(std, 1e-2)
        std[np.isnan(std)] = 1.0
        std[np.isinf(std)] = 1.0
        return cls(mean.T, std.T)

    @classmethod
    def load(cls, path):
        """The `-transform` file of the reference CLIs: a pickle of reader.preprocess.GlobalMeanVarianceNormalization
        (bin/train_ce.py:98-100).  Only its mean_vec / std_vec (and the mean_norm / var_norm switches) are needed,
        so the object is rebuilt as a plain attribute bag instead of importing the reference's class."""
        import pickle

        class _Bag:
            def __setstate__(self, state):
                self.__dict__.update(state)

        class _Unpickler(pickle.Unpickler):
            def find_class(self, module, name):
                if module.startswith("numpy") or module in ("builtins", "collections", "_codecs"):
                    return super().find_class(module, name)
                return _Bag

        with open(path, "rb") as f:
            o = _Unpickler(f).load()
        mean = np.asarray(o.mean_vec, np.float32).reshape(1, -1)
        std = np.asarray(o.std_vec, np.float32).reshape(1, -1)
        if not getattr(o, "mean_norm", True):
            mean = np.zeros_like(mean)
        if not getattr(o, "var_norm", True):
            std = np.ones_like(std)
        return cls(mean, std)

    def save(self, path):
        """Pickles the transform with the attribute names of the reference's class (mean_vec, std_vec [1, D],
        mean_norm, var_norm: reader/preprocess.py:89-112), so that `load` reads files from either side."""
        import pickle
        with open(path, "wb") as f:
            pickle.dump(_MvnState(self.mean_vec, self.std_vec), f, pickle.HIGHEST_PROTOCOL)

    @classmethod
    def estimate(cls, source, extractor, device, n_sample_to_use=2000, apply_cmn=True, batch=8):
        """learn_mean_and_variance_from_train_loader (reference bin/train_ce.py:112-118, reader/preprocess.py:114-151):
        sum and sum of squares of the features of `n_sample_to_use` utterances drawn from the source (accumulated on
        the device in float64), then the floored standard deviation."""
        s1 = torch.zeros(80, dtype=torch.float64, device=device)
        s2 = torch.zeros(80, dtype=torch.float64, device=device)
        n_frame, seen = 0, 0
        while seen < n_sample_to_use:
            utts = [source.draw() for _ in range(min(batch, n_sample_to_use - seen))]
            seen += len(utts)
            lens = [u[0].shape[0] for u in utts]
            wav = torch.from_numpy(np.concatenate([u[0] for u in utts])).to(device)
            feats, frames, _ = extractor(wav, lens, apply_cmn=apply_cmn)
            f64 = feats.double()
            s1 += f64.sum(0)
            s2 += (f64 * f64).sum(0)
            n_frame += int(sum(frames))
        return cls.from_stats(s1.cpu().numpy(), s2.cpu().numpy(), n_frame)

    def __call__(self, x):
        return self.apply_on_tensor(x)

    def apply_on_tensor(self, x):
        _lib.require_gpu()
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.shape[-1] == self.mean_vec.shape[1]
        if self._dev is None or self._dev[0].device != x.device:
            self._dev = (torch.from_numpy(self.mean_vec).to(x.device), torch.from_numpy(self.std_vec).to(x.device))
        y = torch.empty_like(x)
        _lib.check(_lib.lib().pk2_mvn_apply(_lib.ptr(x), _lib.ptr(self._dev[0]), _lib.ptr(self._dev[1]),
                                            x.numel() // x.shape[-1], x.shape[-1], _lib.ptr(y),
                                            _lib.stream_ptr(x.device)))
        return y
