"""Generates tests/golden/simulation.npz by running the REFERENCE's own simulation code
(/root/reference/simulation: Distorter, SimpleSimulator) on seeded inputs.  Run in the build container only;
the reference never travels, the fixture does.

    python tools/gen_golden_sim.py
"""
import os
import sys

import numpy as np

np.int = int      # the reference uses the removed alias
sys.path.insert(0, "/root/reference")
from simulation import SimpleSimulator            # noqa: E402
from simulation._distorter import Distorter       # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rng = np.random.default_rng(20260927)
out = {}


def make_rir(k, delay, t60_samples):
    r = np.zeros(k)
    r[delay] = 1.0
    tail = np.arange(k - delay - 1)
    r[delay + 1:] = 0.4 * rng.standard_normal(k - delay - 1) * np.exp(-tail / t60_samples)
    return r.astype(np.float32)


def make_wav(n, amp):
    x = rng.standard_normal(n)
    y = np.zeros(n)
    for i in range(1, n):
        y[i] = x[i] + 0.9 * y[i - 1]
    return (amp * y / np.abs(y).max()).astype(np.float32)


# ---- Distorter.apply_rir ----
for name, n, k, delay in (("rir_a", 4000, 800, 40), ("rir_b", 1500, 3000, 7), ("rir_c", 10000, 257, 1)):
    wav, rir = make_wav(n, 0.3), make_rir(k, delay, k / 6.0)
    rv, _ = Distorter.apply_rir(wav.astype(np.float64), rir.astype(np.float64)[:, None])
    out[name + "_wav"], out[name + "_rir"], out[name + "_out"] = wav, rir, rv[:, 0]

# ---- Distorter.add_noise, both placements ----
for name, n, m, snr, seed in (("noise_short", 5000, 1800, 7.5, 11), ("noise_long", 3000, 9000, 15.0, 12), ("noise_equal", 2048, 2048, 0.0, 13)):
    sig, nz = make_wav(n, 0.4), make_wav(m, 0.1)
    np.random.seed(seed)
    d, _ = Distorter.add_noise(sig.astype(np.float64)[:, None], nz.astype(np.float64)[:, None], snr, "sample_noise")
    rs = np.random.RandomState(seed)
    n_extra = abs(n - m)
    start = int(rs.randint(0, high=n_extra, size=1)[0]) if n_extra > 0 else 0
    out[name + "_sig"], out[name + "_noise"], out[name + "_out"] = sig, nz, d[:, 0]
    out[name + "_snr"], out[name + "_start"], out[name + "_seed"] = np.float64(snr), np.int64(start), np.int64(seed)

# ---- SimpleSimulator: the call of data/sr_dataset.py:340-344 ----
for name, n, m, use_rir, use_noise, norm, seed in (("sim_full", 6000, 2500, True, True, True, 21),
                                                   ("sim_long_noise", 3000, 8000, True, True, False, 22),
                                                   ("sim_noise_only", 4000, 1200, False, True, True, 23),
                                                   ("sim_rir_only", 5000, 0, True, False, True, 24)):
    src = make_wav(n, 0.3)
    nz = make_wav(m, 0.2) if use_noise else None
    rir_s = make_rir(900, 33, 150.0) if use_rir else None
    rir_n = make_rir(700, 51, 120.0) if use_rir and use_noise else None
    np.random.seed(seed)
    sim = SimpleSimulator(use_rir=use_rir, use_noise=use_noise, snr_range=(0, 30))
    y, _, _, cfg = sim(src.astype(np.float64).copy(),
                       dir_noise_wavs=[nz.astype(np.float64).copy()] if use_noise else None,
                       source_rir=rir_s.astype(np.float64).copy() if use_rir else None,
                       dir_noise_rirs=[rir_n.astype(np.float64).copy()] if rir_n is not None else None,
                       gen_mask=False, normalize_gain=norm)
    out[name + "_src"], out[name + "_out"], out[name + "_seed"] = src, y[:, 0], np.int64(seed)
    out[name + "_norm"] = np.int64(norm)
    if use_noise:
        rs = np.random.RandomState(seed)
        snr = rs.uniform(low=0.0, high=20.0, size=1)
        assert np.array_equal(snr, cfg["dir_snr"]), (snr, cfg)
        n_extra = abs(n - m)
        start = int(rs.randint(0, high=n_extra, size=1)[0]) if n_extra > 0 else 0
        out[name + "_noise"], out[name + "_snr"], out[name + "_start"] = nz, np.float64(snr[0]), np.int64(start)
    if use_rir:
        out[name + "_rir_s"] = rir_s
        if rir_n is not None:
            out[name + "_rir_n"] = rir_n

np.savez_compressed(os.path.join(ROOT, "tests", "golden", "simulation.npz"), **out)
print("wrote", len(out), "arrays")
