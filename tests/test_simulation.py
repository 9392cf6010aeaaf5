"""Dynamic data simulation (reverberation + noise): the oracle against the reference's own outputs
(tests/golden/simulation.npz, written by tools/gen_golden_sim.py from the reference's Distorter / SimpleSimulator),
and -- on the GPU -- the device kernels against the same vectors."""
import numpy as np
import pytest

from oracle import simulation_ref as R


@pytest.fixture(scope="module")
def G(golden):
    return golden("simulation")


def test_oracle_apply_rir_matches_reference(G):
    for name in ("rir_a", "rir_b", "rir_c"):
        got = R.apply_rir(G[name + "_wav"], G[name + "_rir"])
        assert np.abs(got - G[name + "_out"]).max() < 1e-12


def test_oracle_add_noise_matches_reference(G):
    for name in ("noise_short", "noise_long", "noise_equal"):
        sig, nz = G[name + "_sig"].astype(np.float64), G[name + "_noise"].astype(np.float64)
        got = sig + R.place_noise(nz * R.noise_scale(sig, nz, float(G[name + "_snr"])), sig.shape[0], int(G[name + "_start"]))
        assert np.abs(got - G[name + "_out"]).max() < 1e-12


def _sim_case(G, name):
    src = G[name + "_src"]
    noises = [G[name + "_noise"]] if name + "_noise" in G.files else []
    rir_s = G[name + "_rir_s"] if name + "_rir_s" in G.files else None
    rir_n = [G[name + "_rir_n"]] if name + "_rir_n" in G.files else []
    snrs = [float(G[name + "_snr"])] if noises else []
    starts = [int(G[name + "_start"])] if noises else []
    return src, noises, rir_s, rir_n, snrs, starts, bool(G[name + "_norm"]), int(G[name + "_seed"])


SIM_CASES = ["sim_full", "sim_long_noise", "sim_noise_only", "sim_rir_only"]


@pytest.mark.parametrize("name", SIM_CASES)
def test_oracle_simulate_matches_reference(G, name):
    src, noises, rir_s, rir_n, snrs, starts, norm, _ = _sim_case(G, name)
    got = R.simulate(src, noises, rir_s, rir_n, snrs, starts, norm)
    assert np.abs(got - G[name + "_out"]).max() < 1e-12


# ---------------------------------------------------------------------------------------------------
# device kernels (float32 direct-form convolution against the reference's float64 FFT convolution)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_gpu_apply_rir_matches_reference(G):
    import torch
    from pykaldi2_amd import simulation
    for name in ("rir_a", "rir_b", "rir_c"):
        wav, rir = torch.from_numpy(G[name + "_wav"]).cuda(), torch.from_numpy(G[name + "_rir"]).cuda()
        got = simulation.Distorter.apply_rir(wav, rir).cpu().numpy()
        want = G[name + "_out"]
        assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max() + 1e-7
        assert np.array_equal(got, simulation.Distorter.apply_rir(wav, rir, int(np.argmax(G[name + "_rir"]))).cpu().numpy())


@pytest.mark.gpu
def test_gpu_add_noise_matches_reference(G):
    import torch
    from pykaldi2_amd import simulation
    for name in ("noise_short", "noise_long", "noise_equal"):
        sig, nz = torch.from_numpy(G[name + "_sig"]).cuda(), torch.from_numpy(G[name + "_noise"]).cuda()
        np.random.seed(int(G[name + "_seed"]))           # the reference's own draw of the noise position
        got, start = simulation.Distorter.add_noise(sig, nz, float(G[name + "_snr"]))
        assert start == int(G[name + "_start"])
        want = G[name + "_out"]
        assert np.abs(got.cpu().numpy() - want).max() <= 2e-6 * np.abs(want).max()
        assert np.array_equal(sig.cpu().numpy(), G[name + "_sig"])       # the input is left alone


@pytest.mark.gpu
@pytest.mark.parametrize("name", SIM_CASES)
def test_gpu_simulator_matches_reference(G, name):
    """Same seed, same draws (SNR from uniform[0, 20], noise position), same waveform as the reference's
    SimpleSimulator to float32 accuracy."""
    import torch
    from pykaldi2_amd import simulation
    src, noises, rir_s, rir_n, snrs, starts, norm, seed = _sim_case(G, name)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()      # noqa: E731
    sim = simulation.SimpleSimulator(use_rir=rir_s is not None, use_noise=bool(noises), snr_range=(0, 30))
    np.random.seed(seed)
    got, cfg = sim(dev(src), [dev(n) for n in noises] or None, dev(rir_s) if rir_s is not None else None,
                   [dev(r) for r in rir_n] or None, normalize_gain=norm)
    if noises:
        assert abs(cfg["dir_snr"][0] - snrs[0]) < 1e-12 and cfg["dir_start"] == starts
    want = G[name + "_out"]
    assert np.abs(got.cpu().numpy() - want).max() <= 5e-6 * np.abs(want).max()
    if norm:
        assert abs(float(got.abs().max()) - 0.5) < 1e-6


@pytest.mark.gpu
def test_gpu_simulation_argument_errors():
    import torch
    from pykaldi2_amd import _lib, simulation
    x = torch.randn(1000, device="cuda")
    with pytest.raises(_lib.Pk2Error):
        simulation.Distorter.add_noise(x, torch.randn(400, device="cuda"), 5.0, start=700)     # 700 + 400 > 1000
    with pytest.raises(_lib.Pk2Error):
        simulation.Distorter.apply_rir(x, torch.randn(64, device="cuda"), delay=64)
    with pytest.raises(AssertionError):
        simulation.Distorter.apply_rir(x.double(), torch.randn(64, device="cuda"))
