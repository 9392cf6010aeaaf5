"""CPU oracle for the dynamic data simulation (reverberation + additive noise) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
path (pykaldi2_amd/) never does.

Restates, in numpy float64, what the reference runs per training utterance when `simulation_prob` fires
(reference data/sr_dataset.py:321-345 -> simulation/simulation.py:55-178 `_Simulator.simulate` for ONE speech
source, single channel):

  Distorter.apply_rir(wav, rir, sync=True)      simulation/_distorter.py:118-154 (FFT convolution :8-26, the
                                                result cut to reverb[delay-1 : delay+n-1], delay = argmax(rir))
  Distorter.add_noise(mixed, noise, snr, 'sample_noise')   :86-116 with _comp_noise_scale_given_snr :28-32 and
                                                _NoiseSampler.sample_noise :36-58
  gain normalisation 0.5 / max|x|               simulation/simulation.py:170-172

PINNED: tests/golden/simulation.npz holds inputs and outputs of the reference's own Distorter / SimpleSimulator,
produced by tools/gen_golden_sim.py (which imports the reference); tests/test_simulation.py checks this file
against them.  Reference behaviours kept on purpose: the SNR of a directional noise is drawn from
uniform[0, 20] dB whatever `snr_range` says (config.py:39-40 vs :79-80: `snr_range` only sets `global_snr`, which
`simulate` never reads); `mixed_noisy` aliases `mixed`, so the power that scales the second and later noises
already contains the earlier ones (simulation.py:138-147).
"""
import numpy as np


def fftconvolve1d(a, b):
    """_fftconvolve1d for 1-D inputs: full linear convolution through rfft of a fast length."""
    from scipy.fft import next_fast_len
    rlen = a.shape[0] + b.shape[0] - 1
    nfft = next_fast_len(int(rlen))
    return np.fft.irfft(np.fft.rfft(a, nfft) * np.fft.rfft(b, nfft), nfft)[:rlen]


def apply_rir(wav, rir):
    """sync=True: reverb[delay-1 : delay+n-1]."""
    wav = np.asarray(wav, np.float64)
    rir = np.asarray(rir, np.float64)
    n, delay = wav.shape[0], int(np.argmax(rir))
    rv = fftconvolve1d(rir, wav)
    return rv[delay - 1:delay + n - 1]


def noise_scale(signal, noise, snr):
    return np.sqrt(np.mean(np.asarray(signal, np.float64) ** 2) / np.mean(np.asarray(noise, np.float64) ** 2) * 10 ** ((-snr) / 10))


def place_noise(noise, n, start):
    """_NoiseSampler.sample_noise with the start already drawn."""
    m = noise.shape[0]
    if m <= n:
        out = np.zeros(n)
        out[start:start + m] = noise
        return out
    return noise[start:start + n]


def simulate(source, noises=(), source_rir=None, noise_rirs=(), snrs=(), starts=(), normalize_gain=True):
    """One speech source, any number of directional noises; snrs / starts = the host's random draws."""
    mixed = apply_rir(source, source_rir) if source_rir is not None else np.asarray(source, np.float64).copy()
    for i, nz in enumerate(noises):
        nz = apply_rir(nz, noise_rirs[i]) if source_rir is not None else np.asarray(nz, np.float64)
        mixed = mixed + place_noise(nz * noise_scale(mixed, nz, snrs[i]), mixed.shape[0], starts[i])
    if normalize_gain:
        mixed = mixed * (0.5 / np.max(np.abs(mixed)))
    return mixed
