"""Dynamic data simulation on the device: reverberation by a room impulse response and additive noise at a
sampled SNR, the single-channel / single-source path of the reference's simulator (reference
simulation/simulation.py:181-234 `SimpleSimulator`, simulation/_distorter.py:84-154 `Distorter`), which the
reference runs with numpy inside DataLoader workers (data/sr_dataset.py:321-345).  Waveforms, impulse responses
and noises are CUDA float32 tensors; the power / peak statistics stay in device memory between the kernels of an
utterance (no host round trip), only the random draws are made on the host -- with numpy's global generator and
in the reference's order, so `np.random.seed` reproduces the reference's choices.

Reference behaviours kept on purpose (oracle/simulation_ref.py): the SNR of a directional noise comes from
uniform[0, 20] dB whatever `snr_range` is; the power used to scale a second noise already includes the first.
"""
import numpy as np
import torch

from . import _lib


def _check(x):
    _lib.require_gpu()
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 1 and x.is_contiguous(), "expected a 1-D CUDA float32 tensor"
    return x


def _power(x):
    """device f64[2] = (sum of squares, max |x|)"""
    stats = torch.zeros(2, dtype=torch.float64, device=x.device)
    _lib.check(_lib.lib().pk2_sim_power(_lib.ptr(x), x.numel(), _lib.ptr(stats), _lib.stream_ptr(x.device)))
    return stats


class Distorter:
    """Distorter.apply_rir / Distorter.add_noise of the reference on CUDA tensors."""

    @staticmethod
    def apply_rir(wav, rir, delay=None):
        """Reverberant signal, sample-synchronised with the input (sync=True): (rir * wav)[delay-1 : delay+n-1] with
        delay = argmax(rir).  `delay` may be passed when it is known (it costs a host read otherwise)."""
        wav, rir = _check(wav), _check(rir)
        if delay is None:
            delay = int(torch.argmax(rir).item())
        out = torch.empty_like(wav)
        _lib.check(_lib.lib().pk2_sim_apply_rir(_lib.ptr(wav), wav.numel(), _lib.ptr(rir), rir.numel(), int(delay),
                                                _lib.ptr(out), _lib.stream_ptr(wav.device)))
        return out

    @staticmethod
    def add_noise(signal, noise, snr, start=None):
        """signal + noise scaled to `snr` dB and placed by the 'sample_noise' scheme.  `start` = the sampled position
        (drawn here like the reference does when None).  Returns (distorted, start); `signal` is not modified."""
        signal, noise = _check(signal), _check(noise)
        n, m = signal.numel(), noise.numel()
        if start is None:
            n_extra = abs(n - m)
            start = int(np.random.randint(0, high=n_extra, size=1)[0]) if n_extra > 0 else 0     # _sampling.get_sample 'uniform_int'
        out = signal.clone()
        ps, pn = _power(signal), _power(noise)       # both alive until the kernel that reads them is enqueued
        _lib.check(_lib.lib().pk2_sim_add_noise(_lib.ptr(out), n, _lib.ptr(noise), m, int(start), float(snr),
                                                _lib.ptr(ps), _lib.ptr(pn), _lib.stream_ptr(signal.device)))
        return out, start


class SimpleSimulator:
    """Single speech source simulator (reference simulation/simulation.py:181-234):
    ``SimpleSimulator(use_rir, use_noise, snr_range)(source_wav, dir_noise_wavs, source_rir, dir_noise_rirs,
    normalize_gain=...)`` -> (simulated waveform, sentence config)."""

    def __init__(self, array_geometry=None, use_rir=True, use_noise=True, snr_range=(0, 30)):
        assert array_geometry is None, "single-channel simulation only"
        self.use_rir, self.use_noise = use_rir, use_noise
        self.snr_range = tuple(snr_range)      # sets `global_snr` in the reference, which its simulate() never reads

    def __call__(self, source_wav, dir_noise_wavs=None, source_rir=None, dir_noise_rirs=None, normalize_gain=True,
                 rir_delays=None):
        noises = list(dir_noise_wavs) if dir_noise_wavs is not None else []
        use_rir = source_rir is not None
        if use_rir and noises:
            assert dir_noise_rirs is not None and len(dir_noise_rirs) == len(noises), \
                "number of dir_noise_rir does not equal to number of directional noise sources"
        delays = list(rir_delays) if rir_delays is not None else [None] * (1 + len(noises))
        mixed = Distorter.apply_rir(source_wav, source_rir, delays[0]) if use_rir else _check(source_wav).clone()
        cfg = {}
        if noises:
            cfg["dir_snr"] = np.random.uniform(low=0.0, high=20.0, size=len(noises))      # config.py:39-40
            cfg["dir_start"] = []
            stream = _lib.stream_ptr(mixed.device)
            for i, nz in enumerate(noises):
                nz = Distorter.apply_rir(nz, dir_noise_rirs[i], delays[1 + i]) if use_rir else _check(nz)
                n, m = mixed.numel(), nz.numel()
                n_extra = abs(n - m)
                start = int(np.random.randint(0, high=n_extra, size=1)[0]) if n_extra > 0 else 0
                cfg["dir_start"].append(start)
                ps, pn = _power(mixed), _power(nz)
                _lib.check(_lib.lib().pk2_sim_add_noise(_lib.ptr(mixed), n, _lib.ptr(nz), m, start, float(cfg["dir_snr"][i]),
                                                        _lib.ptr(ps), _lib.ptr(pn), stream))
        if normalize_gain:
            peak = _power(mixed)
            _lib.check(_lib.lib().pk2_sim_gain_norm(_lib.ptr(mixed), mixed.numel(), _lib.ptr(peak),
                                                    _lib.stream_ptr(mixed.device)))
        return mixed, cfg
