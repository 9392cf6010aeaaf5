"""Minibatch sources for the training CLIs: raw waveforms go to the GPU, features are computed there.

The reference's data layer (reference data/sr_dataset.py, data/dataloader.py, reader/*) reads 16 kHz
wavs out of zip archives, joins them with per-utterance label files by utterance id and computes
fbank+CMN in DataLoader workers.  Here the host only decodes PCM and ships it; fbank / CMN / chunking /
padding run on the device (pykaldi2_amd.fbank).  Two sources share one interface:

* ``ZipWavSource``  -- the reference's on-disk formats: a zip of wav files
  (``clean_source/{n}/wav``), ``label`` = text file ``utt_id pdf pdf ...`` (CE targets, 100 fps),
  ``aux_label`` = same format with transition ids.  Utterances are the sorted intersection of the
  archive members and the label keys (reference reader/stream.py:563-588).
* ``SyntheticSource`` -- the seeded LibriSpeech-shaped generator of SURVEY.md 8(d) (no dataset here).

Batch dict (cf. reference data/dataloader.py:128-134): ``wav`` (1-D CUDA f32, utterances back to
back), ``lens`` (samples), ``y`` (list of int64 pdf alignments, one per utterance), ``aux`` (list),
``utt_ids``, ``seconds``.
"""
import io
import os
import sys
import wave
import zipfile

import numpy as np
import torch

from . import _lib, synth


def read_label_file(path):
    """``utt_id id id id ...`` per line (reference example/librispeech/README.md:40-44)."""
    out = {}
    with open(path) as f:
        for line in f:
            parts = line.split()
            if len(parts) >= 2:
                out[parts[0]] = np.asarray(parts[1:], dtype=np.int64)
    return out


def decode_wav(data):
    """RIFF PCM16 mono/multi-channel -> float32 in [-1, 1) (first channel), like reader/zip_io.py:145."""
    with wave.open(io.BytesIO(data)) as w:
        assert w.getsampwidth() == 2, "only 16-bit PCM wav is supported"
        n, ch = w.getnframes(), w.getnchannels()
        pcm = np.frombuffer(w.readframes(n), dtype="<i2").reshape(-1, ch)[:, 0]
        sr = w.getframerate()
    return (pcm.astype(np.float32) / 32768.0), sr


class ZipWavSource:
    def __init__(self, sources, data_path="", seed=0, rank=0, world=1):
        self.items = []   # (zip path, member, utt_id, labels, aux)
        for src in sources:
            zpath = os.path.join(data_path, src["wav"]) if data_path else src["wav"]
            labels = read_label_file(os.path.join(data_path, src["label"]) if data_path else src["label"]) \
                if src.get("label") else {}
            aux = read_label_file(os.path.join(data_path, src["aux_label"]) if data_path else src["aux_label"]) \
                if src.get("aux_label") else {}
            with zipfile.ZipFile(zpath) as z:
                members = sorted(m for m in z.namelist() if m.lower().endswith(".wav"))
            for m in members:
                utt = os.path.splitext(os.path.basename(m))[0]
                if labels and utt not in labels:
                    continue
                self.items.append((zpath, m, utt, labels.get(utt), aux.get(utt)))
        assert self.items, "no utterance found"
        self.rng = np.random.default_rng(seed + rank)
        self._zips = {}
        self.rank, self.world = rank, world

    def __len__(self):
        return len(self.items)

    def _read(self, zpath, member):
        z = self._zips.get(zpath)
        if z is None:
            z = self._zips[zpath] = zipfile.ZipFile(zpath)
        wav, sr = decode_wav(z.read(member))
        assert sr == 16000, "%s: expected 16 kHz audio" % member
        return wav

    def draw(self):
        return self.get(int(self.rng.integers(len(self.items))))

    def durations(self):
        """Seconds per utterance from the WAV headers in the archives (fmt chunk: channels, rate, bits; data chunk: payload
        size), without decoding the audio.  The result is cached next to each archive (`<archive>.durations.json`, keyed by
        the archive's size and mtime; a directory that cannot be written is simply not cached) -- a 960 h corpus is ~280 k
        member opens, which every rank of every job would otherwise repeat at start-up.  A header whose `data`
        chunk lies beyond the first 4 KB (large LIST / bext chunks) is read further, up to 1 MB; a member whose header
        cannot be parsed at all counts as 16 kHz 16-bit mono and is named in a warning."""
        if not hasattr(self, "_dur"):
            import json
            want = {}
            for it in self.items:
                want.setdefault(it[0], set()).add(it[1])
            secs, fell_back = {}, []
            for zpath in sorted(want):
                st = os.stat(zpath)
                # (stamp: archive size, mtime, cache format version -- a fixed header parser invalidates old caches)
                cache_path, stamp = zpath + ".durations.json", [int(st.st_size), int(st.st_mtime), _DURATIONS_CACHE_VERSION]
                cached = {}
                try:
                    with open(cache_path) as f:
                        blob = json.load(f)
                    if blob.get("stamp") == stamp:
                        cached = blob.get("seconds", {})
                except (OSError, ValueError):
                    pass
                missing = [m for m in want[zpath] if m not in cached]
                estimated = {}        # fall-back estimates are NOT cached: the next run parses (and warns) again
                if missing:
                    miss = set(missing)
                    with zipfile.ZipFile(zpath) as z:
                        for info in z.infolist():
                            if info.filename not in miss:
                                continue
                            with z.open(info) as f:
                                head = f.read(4096)
                                sec = _wav_seconds(head, info.file_size)
                                # (a member that is no RIFF/WAVE file at all is not read any further)
                                while sec is None and head[:4] == b"RIFF" and len(head) < min(info.file_size, 1 << 20):
                                    more = f.read(len(head))           # the data chunk was not in what has been read: double it
                                    if not more:
                                        break
                                    head += more
                                    sec = _wav_seconds(head, info.file_size)
                            if sec is None:
                                estimated[info.filename] = max(0, info.file_size - 44) / 2.0 / 16000.0
                                fell_back.append("%s@/%s" % (zpath, info.filename))
                            else:
                                cached[info.filename] = sec
                    try:
                        tmp = "%s.%d.tmp" % (cache_path, os.getpid())
                        with open(tmp, "w") as f:
                            json.dump(dict(stamp=stamp, seconds=cached), f)
                        os.replace(tmp, cache_path)       # (atomic: several ranks may write the same file)
                    except OSError:
                        pass
                for m in want[zpath]:
                    secs[(zpath, m)] = cached[m] if m in cached else estimated[m]
            if fell_back:
                sys.stderr.write("[data] %d wav header(s) could not be parsed, durations estimated as 16 kHz / 16-bit / mono: %s%s\n"
                                 % (len(fell_back), ", ".join(fell_back[:5]), " ..." if len(fell_back) > 5 else ""))
            self._dur = np.array([secs[(it[0], it[1])] for it in self.items])
        return self._dur

    def get(self, index):
        zpath, member, utt, lab, aux = self.items[index]
        wav = self._read(zpath, member)
        T = synth.num_fbank_frames(wav.shape[0])
        if lab is not None:   # truncate to min(n_label, n_fbank) like data/sr_dataset.py:349-363
            n = min(T, lab.shape[0])
            lab = lab[:n]
            wav = wav[:401 + 160 * (n - 1)] if n < T else wav
        return wav, lab, aux, utt


_DURATIONS_CACHE_VERSION = 2


def _wav_seconds(head, file_size):
    """Duration from the first bytes of a RIFF/WAVE file: walks the chunks up to `data`.  None when the `data` chunk (or the
    `fmt ` chunk before it) is not inside `head` or the file is no RIFF/WAVE at all -- the caller reads further or estimates."""
    import struct
    if len(head) < 12 or head[:4] != b"RIFF" or head[8:12] != b"WAVE":
        return None
    pos, rate, block = 12, None, None
    while pos + 8 <= len(head):
        tag, size = head[pos:pos + 4], struct.unpack("<I", head[pos + 4:pos + 8])[0]
        if tag == b"fmt " and pos + 24 <= len(head):
            _, ch, rate, _, block, bits = struct.unpack("<HHIIHH", head[pos + 8:pos + 24])
            block = block or max(1, ch * bits // 8)
        elif tag == b"data":
            if not rate or not block:
                return None
            payload = min(size, file_size - (pos + 8)) if size not in (0, 0xFFFFFFFF) else file_size - (pos + 8)
            return max(0, payload) / float(block) / float(rate)
        pos += 8 + size + (size & 1)
    return None


class SyntheticSource:
    def __init__(self, num_pdfs, seed=0, rank=0, world=1, with_tids=False, ali_model=None):
        self.num_pdfs = num_pdfs
        self.with_tids = with_tids     # also draw a transition-id alignment (aux_label) for the lattice criteria
        self.ali_model = ali_model     # chain training: `label` holds transition-ids of this TransitionModel
        self.rng = np.random.default_rng(1234 + seed + rank)
        self.count = 0
        self.rank, self.world = rank, world

    def __len__(self):
        return 1 << 30

    MEAN_SECONDS = 12.3     # of synth.utterance_durations

    def draw(self, seconds=None):
        d = float(synth.utterance_durations(self.rng, 1)[0]) if seconds is None else float(seconds)
        wav = synth.waveform(self.rng, d)
        T = synth.num_fbank_frames(wav.shape[0])
        self.count += 1
        if self.ali_model is not None:
            return wav, synth.phone_tid_alignment(self.rng, T, self.ali_model)[0], None, "synth-%d" % self.count
        if self.with_tids:
            tids = synth.tid_alignment(self.rng, T, self.num_pdfs)
            return wav, (tids - 1) // 2, tids, "synth-%d" % self.count      # pdf of a transition-id (synth.transition_model_arrays)
        return wav, synth.pdf_alignment(self.rng, T, self.num_pdfs), None, "synth-%d" % self.count


def make_source(config, num_pdfs, rank=0, world=1, with_tids=False, ali_model=None):
    if config.get("synthetic") or not config.get("source_paths"):
        source = SyntheticSource(num_pdfs, rank=rank, world=world, with_tids=with_tids, ali_model=ali_model)
    else:
        source = ZipWavSource(config["source_paths"], config.get("data_path", ""), rank=rank, world=world)
    source.simulation = SimulationPool.from_config(config, seed=rank)     # None unless data_config switches it on
    return source


class SimulationPool:
    """Noises and room impulse responses for the dynamic simulation (reference data/sr_dataset.py:94-126,321-345):
    with probability `simulation_prob` an utterance is reverberated by a sampled RIR and mixed with one sampled
    directional noise (reverberated by a second RIR of the pool) at an SNR drawn by pykaldi2_amd.simulation --
    on the device.  Sources: zips of 16 kHz wav files (`dir_noise` / `rir` entries of the data yaml, first channel,
    one impulse response per file), or the synthetic generator."""

    def __init__(self, noises, rirs, use_reverb=True, use_noise=True, simulation_prob=0.5, gain_norm=False, snr_range=(0, 30)):
        from . import simulation
        self.noises = noises if use_noise else []
        self.rirs = rirs if use_reverb else []
        self.prob, self.gain_norm = float(simulation_prob), bool(gain_norm)
        self.sim = simulation.SimpleSimulator(use_rir=bool(self.rirs), use_noise=bool(self.noises), snr_range=snr_range)
        self._dev = {}

    @classmethod
    def from_config(cls, config, seed=0):
        """data_config keys of the reference YAMLs (use_dir_noise, use_reverb, snr_range, simulation_prob, gain_norm)
        plus `dir_noise_paths` / `rir_paths` (bin/train_ce.py:73-76); None when the simulation is off."""
        dc = config.get("data_config", {})
        prob = dc.get("simulation_prob", 0)
        if not prob or not (dc.get("use_dir_noise") or dc.get("use_reverb")):
            return None
        if config.get("synthetic") or not (config.get("dir_noise_paths") or config.get("rir_paths")):
            rng = np.random.default_rng(4321 + seed)
            noises = [synth.waveform(rng, float(d)) for d in rng.uniform(2.0, 12.0, size=8)]
            rirs = [synth.room_impulse_response(rng) for _ in range(16)]
        else:
            noises = cls._read_zips(config.get("dir_noise_paths") or [], config.get("data_path", ""))
            rirs = cls._read_zips(config.get("rir_paths") or [], config.get("data_path", ""))
        return cls(noises, rirs, dc.get("use_reverb", False), dc.get("use_dir_noise", False), prob, dc.get("gain_norm", False),
                   dc.get("snr_range", (0, 30)))

    @staticmethod
    def _read_zips(sources, data_path):
        out = []
        for src in sources:
            zpath = os.path.join(data_path, src["wav"]) if data_path else src["wav"]
            with zipfile.ZipFile(zpath) as z:
                for m in sorted(m for m in z.namelist() if m.lower().endswith(".wav")):
                    out.append(decode_wav(z.read(m))[0])
        return out

    def _on_device(self, kind, idx, device):
        key = (kind, idx, str(device))
        if key not in self._dev:
            arr = (self.noises if kind == "n" else self.rirs)[idx]
            self._dev[key] = (torch.from_numpy(np.ascontiguousarray(arr, np.float32)).to(device), int(np.argmax(arr)))
        return self._dev[key]

    def maybe_simulate(self, wav, device):
        """wav: host float32 -> device tensor, simulated with probability simulation_prob (the draws use numpy's
        global generator like the reference: data/sr_dataset.py:321-336)."""
        x = _lib.h2d(wav, device)
        if np.random.random() > self.prob:
            return x
        noise = noise_rir = src_rir = None
        delays = []
        if self.noises:
            noise = self._on_device("n", int(np.random.choice(len(self.noises))), device)[0]
        if self.rirs:
            picks = np.random.choice(len(self.rirs), 2 if noise is not None else 1, replace=len(self.rirs) < 2)
            src_rir, d0 = self._on_device("r", int(picks[0]), device)
            delays.append(d0)
            if noise is not None:
                noise_rir, d1 = self._on_device("r", int(picks[1]), device)
                delays.append(d1)
        y, _ = self.sim(x, [noise] if noise is not None else None, src_rir, [noise_rir] if noise_rir is not None else None,
                        normalize_gain=self.gain_norm, rir_delays=delays or None)
        return y


def epoch_plan(source, batch_size, hours, rank=0, world=1, epoch=0, length_bucketed=False, seed=0):
    """What every rank does in one epoch, computed from rank-independent quantities only, so that ALL ranks run the same
    number of steps (each step ends in a gradient all-reduce: a rank that stops early would leave the others
    blocked in the collective).  Returns a list of steps; a step is a list of `batch_size` items, an item being an
    utterance index (finite sources) or a duration in seconds / None (the synthetic generator).

    * Finite source (ZipWavSource): the reference's DistributedSampler (data/dataloader.py:83-84): a permutation of the
      utterance list seeded by (seed, epoch) identically on every rank, padded by wrapping to a multiple of
      world * batch_size, rank r taking every world-th group -- each utterance once per epoch, same count everywhere.
      `hours` (of audio PER RANK, as for the synthetic source and as the reference's per-process sweep_size) caps the
      epoch through the mean duration of the list.
    * Synthetic source: n = ceil(hours * 3600 / (12.3 s * batch_size)) steps on every rank.
    * length_bucketed: the ranks of one step get utterances of similar length (a step waits at the all-reduce for its
      longest minibatch): super-blocks of 16 steps are sorted by duration and dealt group by group to the ranks, the
      steps of a super-block are then shuffled.  Synthetic: the durations come from a generator shared by the ranks."""
    finite = hasattr(source, "items")
    rng = np.random.default_rng([int(seed), int(epoch), 7919])          # the same stream on every rank
    if not finite:
        n_steps = max(1, int(np.ceil(hours * 3600.0 / (source.MEAN_SECONDS * batch_size))))
        if not length_bucketed:
            return [[None] * batch_size for _ in range(n_steps)]
        durs = synth.utterance_durations(rng, n_steps * batch_size).reshape(n_steps, batch_size)
        return [list(map(float, row)) for row in durs]
    n = len(source.items)
    if n == 0:
        return []
    group = world * batch_size
    perm = rng.permutation(n)
    total = -(-n // group) * group
    perm = np.concatenate([perm, perm[:total - n]]) if total > n else perm     # wrap-around padding (DistributedSampler)
    if total > perm.shape[0]:
        perm = np.resize(perm, total)
    n_steps = total // group
    if hours and hours > 0:
        mean = float(np.mean(source.durations())) if n else 1.0
        n_steps = max(1, min(n_steps, int(np.ceil(hours * 3600.0 / (max(mean, 1e-3) * batch_size)))))      # hours PER RANK
    perm = perm[:n_steps * group]
    if length_bucketed:
        dur = source.durations()
        out = []
        for b0 in range(0, n_steps, 16):
            blk = perm[b0 * group:(b0 + 16) * group]
            blk = blk[np.argsort(dur[blk], kind="stable")]
            steps = blk.reshape(-1, world, batch_size)            # consecutive (similar-length) groups -> the ranks of a step
            steps = steps[rng.permutation(steps.shape[0])]
            out.extend(steps[:, rank].tolist())
        return out
    return perm.reshape(n_steps, world, batch_size)[:, rank].tolist()


def sequence_batches(source, batch_size, hours, device, simulation=None, rank=None, world=None, epoch=0,
                     length_bucketed=False):
    """Whole-utterance minibatches of one epoch: `hours` of audio per rank (the reference's sweep_size,
    data/sr_dataset.py:226) on the synthetic generator, one sharded pass over the utterance list on a finite source
    (see epoch_plan: the number of steps is the same on every rank).  `simulation`: a SimulationPool, or None."""
    rank = getattr(source, "rank", 0) if rank is None else rank
    world = getattr(source, "world", 1) if world is None else world
    if simulation is None:
        simulation = getattr(source, "simulation", None)
    finite = hasattr(source, "items")
    for step in epoch_plan(source, batch_size, hours, rank, world, epoch, length_bucketed):
        utts = [source.get(int(it)) if finite else source.draw(it) for it in step]
        lens = [u[0].shape[0] for u in utts]
        if simulation is not None:
            wav = torch.cat([simulation.maybe_simulate(u[0], device) for u in utts])
        else:
            wav = _lib.h2d(np.concatenate([u[0] for u in utts]), device)      # (pinned: the copy does not wait for the step in flight)
        yield dict(wav=wav, lens=lens, y=[u[1] for u in utts], aux=[u[2] for u in utts],
                   utt_ids=[u[3] for u in utts], seconds=sum(lens) / 16000.0)
