"""iRTF benchmark of the LF-MMI training step (BASELINE.json: 3x512 BLSTM LF-MMI, LibriSpeech-shaped
utterances, batch 4 x var-len per GPU, train_chain.py semantics) on N MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one minibatch of raw waveforms already resident in HBM:
fbank+CMN+subsample -> 3x512 BLSTM forward -> LF-MMI (numerator + denominator forward-backward)
-> BLSTM backward -> gradient all-reduce (N>1) -> Noam lr, global-norm clip, Adam(amsgrad).
Weak scaling: every rank processes its own 4-utterance minibatch per step.
The per-utterance supervisions are built inside the step from the transition-id alignments (SplitToPhones ->
proto-supervision -> supervision: host-side entries of libpk2hip.so, as the reference does per utterance at
bin/train_chain.py:262-272) and uploaded to the GPU.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from pykaldi2_amd import chain, fbank, hvd, lstm, ops, optim, synth, utils  # noqa: E402

P = 6048            # configs/mmi.yaml label_size
S_DEN, A_DEN = 30000, 1000000
HBM_PEAK_GBS = 8000.0
# Denominator graph topology: "chain" = what Kaldi's chain topology gives a real den.fst (the arcs entering a state carry
# its forward pdf, its self-loop a different pdf: reference bin/train_chain.py:167,202); "unique" = round 1's graph
# whose self-loops emit the entering pdf (kept for A/B only).
DEN_TOPOLOGY = os.environ.get("PK2_BENCH_DEN_TOPOLOGY", "chain")


def den_graph_arrays():
    return synth.den_graph_arcs(S_DEN, A_DEN, P, seed=0, loop_pdf_differs=(DEN_TOPOLOGY == "chain"))


_chain_model = None


def chain_model():
    """Synthetic tree + transition model of the chain system (also the alignment model of the label files)."""
    global _chain_model
    if _chain_model is None:
        tree, tm = synth.chain_model(P, seed=0)
        _chain_model = (chain.MappedAligner(tm), tree, tm, chain.SupervisionOptions())
    return _chain_model


def build_supervisions(alis):
    """reference bin/train_chain.py:262-272, per utterance: alignment -> phones -> proto-supervision -> supervision
    (host-side C ABI entries of libpk2hip.so)."""
    aligner, tree, tm, sopts = chain_model()
    return [chain.supervision_from_alignment(aligner, tree, tm, sopts, a) for a in alis]


def make_batches(rng, n_unique, batch, device, duration_rng=None):
    out = []
    for _ in range(n_unique):
        mb = synth.minibatch(rng, batch, P, ali_model=chain_model()[2], duration_rng=duration_rng)
        lens = [w.shape[0] for w, _ in mb]
        wav = torch.from_numpy(np.concatenate([w for w, _ in mb])).to(device)
        out.append(dict(wav=wav, lens=lens, alis=[a for _, a in mb], seconds=sum(lens) / 16000.0, host=mb))
    return out


class Trainer:
    """One training step per call.  The per-utterance supervisions (host work: SplitToPhones -> proto-supervision ->
    supervision, ~0.3 ms per utterance in libpk2hip.so, the GIL released) are built by ONE worker thread while the main thread
    enqueues the step's kernels -- `prefetch(mb)` starts the minibatch the next call will use, as a DataLoader worker would;
    without a prefetch the step builds them inline.  Every step still builds its own supervisions inside the timed loop; on a
    box whose host cores are slow or shared this keeps the loop GPU-bound (one gpurun box in ten ran the inline form at 20 ms
    per step with an unchanged 13.5 ms of GPU work)."""

    def __init__(self, device, den, base_lr=1e-3, xent=0.1, arch="blstm"):
        from concurrent.futures import ThreadPoolExecutor
        self._pool = ThreadPoolExecutor(max_workers=1)
        self._pending = {}
        torch.manual_seed(0)
        self.arch = arch
        if arch == "transformer":   # configs[4]: 12-layer TransformerAM (dim 512, 8 heads, FFN 2048), secondary workload
            from pykaldi2_amd import transformer
            self.model = transformer.TransformerAM(80, 512, 8, 2048, 12, 0.0, P).to(device)
        else:
            self.model = lstm.LSTMAM(80, P, 512, 3, 0.0, True).to(device)
        self.base_opt = optim.Adam(self.model, lr=base_lr, amsgrad=True)
        hvd.broadcast_parameters(self.model.state_dict(), root_rank=0)
        self.opt = hvd.DistributedOptimizer(self.base_opt, named_parameters=self.model.named_parameters())
        self.fb = fbank.FbankExtractor()
        self.den = den
        self.opts = chain.ChainTrainingOptions(leaky_hmm_coefficient=1e-4, xent_regularize=xent)
        self.base_lr = base_lr
        self.step_no = 0
        self.last = {}

    def prefetch(self, mb):
        if id(mb) not in self._pending:
            self._pending[id(mb)] = self._pool.submit(build_supervisions, mb["alis"])

    def step(self, mb, epoch=0, events=None):
        def mark(name):
            if events is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                events.append((name, e))
        mark("start")
        feats, frames, row_off = self.fb(mb["wav"], mb["lens"])
        x = self.fb.pad_roll_subsample(feats, row_off, frames, shift=-(epoch % 3), subsample=3, time_major=True)
        mark("fbank")
        if self.arch == "transformer":
            Tp = x.shape[0]
            kpm = torch.ones(len(frames), Tp, dtype=torch.bool)
            for n_, f_ in enumerate(frames):
                kpm[n_, :-(-f_ // 3)] = False
            logits = self.model(x, None, kpm)       # (host mask: pinned, non-blocking inside forward)
        else:
            logits = self.model.forward_time_major(x)
        mark("model_fwd" if self.arch == "transformer" else "lstm_fwd")
        fut = self._pending.pop(id(mb), None)    # host work: prefetched by the worker thread, or built here while the
        sups = fut.result() if fut is not None else build_supervisions(mb["alis"])     # device is still in the forward pass
        loss = ops.ChainObjtiveBatch.apply(logits.transpose(0, 1), self.den, sups, self.opts)
        mark("chain")
        self.opt.zero_grad()
        loss.backward()
        mark("model_bwd" if self.arch == "transformer" else "lstm_bwd")
        self.step_no += 1
        lr = utils.noam_decay(self.step_no, 4000, self.base_lr)
        for grp in self.opt.param_groups:
            grp["lr"] = lr
        norm = optim.clip_grad_norm_(self.opt, 5.0)
        self.opt.step()
        mark("optim")
        self.last = dict(loss=loss, norm=norm, logits=logits, frames=frames, sups=sups)
        return loss


def scaling_report(step_events, exchange_timing, audio, dev, rank, world):
    """What an N > 1 line needs to explain its own scaling, from device events of the timed steps:
    per rank the GPU time of a step (mean / max), the time until its last all-reduce was issued (`pre_exchange`: forward,
    loss and backward -- in the bucketed schedule the earlier buckets overlap the rest of backward), the duration of its
    all-reduce calls; gathered over the ranks, per step:  straggler wait of rank r = max over ranks of pre_exchange -
    pre_exchange of r (a collective finishes when its slowest rank arrives), exchange net of it = duration of the last
    all-reduce call - that wait; `ideal_weak_irtf` = sum over ranks of audio / (GPU time - straggler wait), i.e. what weak
    scaling would give if no rank ever waited for another and the exchange cost what it costs net."""
    K = len(step_events)
    gpu = np.array([a.elapsed_time(b) for a, b in step_events], dtype=np.float64)
    pre = gpu.copy()
    last = np.zeros(K); total = np.zeros(K); nbytes = 0
    for i, calls in enumerate(exchange_timing or []):
        if i >= K or not calls:
            continue
        pre[i] = step_events[i][0].elapsed_time(calls[-1][0])
        last[i] = calls[-1][0].elapsed_time(calls[-1][1])
        total[i] = sum(a.elapsed_time(b) for a, b, _ in calls)
        nbytes = sum(n for _, _, n in calls)
    mine = np.stack([gpu, pre, last, total])                       # [4, K]
    if world > 1:
        t = torch.from_numpy(mine).to(dev if torch.distributed.get_backend() == "nccl" else "cpu")
        allv = [torch.zeros_like(t) for _ in range(world)]
        torch.distributed.all_gather(allv, t)
        allv = np.stack([a.cpu().numpy() for a in allv])           # [world, 4, K]
        a_t = torch.tensor([audio], dtype=torch.float64, device=t.device)
        audios = [torch.zeros_like(a_t) for _ in range(world)]
        torch.distributed.all_gather(audios, a_t)
        audios = [float(a.item()) for a in audios]
    else:
        allv, audios = mine[None], [audio]
    wait = allv[:, 1, :].max(axis=0, keepdims=True) - allv[:, 1, :]            # [world, K]
    net = np.maximum(allv[:, 2, :] - wait, 0.0)
    own = np.maximum(allv[:, 0, :] - wait, 1e-6)
    per_rank = []
    for r in range(allv.shape[0]):
        per_rank.append({"rank": r, "gpu_ms_mean": round(float(allv[r, 0].mean()), 3), "gpu_ms_max": round(float(allv[r, 0].max()), 3),
                         "pre_exchange_ms_mean": round(float(allv[r, 1].mean()), 3),
                         "exchange_ms_mean": round(float(allv[r, 3].mean()), 3),
                         "straggler_wait_ms_mean": round(float(wait[r].mean()), 3),
                         "audio_s": round(audios[r], 2), "own_irtf": round(audios[r] / (own[r].sum() * 1e-3), 2),
                         "gpu_ms_steps": [round(float(v), 2) for v in allv[r, 0]]})       # (device events around every timed step)
    return {"per_rank": per_rank, "ideal_weak_irtf": round(sum(p["own_irtf"] for p in per_rank), 2),
            "exchange": {"exchange_ms": round(float(allv[:, 3, :].mean()), 3), "exchange_net_ms": round(float(net.mean()), 3),
                         "straggler_wait_ms": round(float(wait.mean()), 3), "straggler_wait_ms_worst_rank": round(float(wait.mean(axis=1).max()), 3),
                         "bytes_per_step": int(nbytes),
                         "how_to_read": "gpu_ms = pre_exchange + last all-reduce (incl. the wait for the slowest rank) + optimiser; "
                         "value falls short of ideal_weak_irtf by straggler_wait (utterance lengths differ per rank: see "
                         "irtf_bucketed) plus exchange_net (RCCL over xGMI)"}}


def dump_chain_parity_inputs(tr, mb, path):
    """The logits of the breakdown minibatch, its alignments, and what the device computes for them (objective per
    sequence, d objf / d logits): the `parity` leg of the cpu-baseline child checks them against the C port.  TransformerAM
    (configs[4]): also the model input, the padding mask and the weights, so that the child can run the torch CPU modules
    on the same minibatch and compare the logits."""
    logits = tr.last["logits"].detach().transpose(0, 1).contiguous()          # [N, T', P]
    out, grad = chain.compute_chain_objf_and_deriv(tr.opts, tr.den, tr.last["sups"], logits)
    torch.cuda.synchronize()
    arrs = {"ali%d" % n: np.asarray(a) for n, a in enumerate(mb["alis"])}
    np.savez(path, n=len(mb["alis"]), xent=tr.opts.xent_regularize, logits=logits.cpu().numpy(),
             gpu_out=out.cpu().numpy(), gpu_grad=grad.cpu().numpy(), **arrs)
    if tr.arch == "transformer":
        with torch.no_grad():      # (the logits above are those of the step BEFORE its update: these are of the saved weights)
            feats, frames, row_off = tr.fb(mb["wav"], mb["lens"])
            x = tr.fb.pad_roll_subsample(feats, row_off, frames, shift=0, subsample=3, time_major=True)
            kpm = torch.ones(len(frames), x.shape[0], dtype=torch.bool)
            for n_, f_ in enumerate(frames):
                kpm[n_, :-(-int(f_) // 3)] = False
            now = tr.model(x, None, kpm.to(x.device)).transpose(0, 1).contiguous()
        torch.save(dict(state_dict={k: v.cpu() for k, v in tr.model.state_dict().items()}, x=x.cpu(), frames=list(frames),
                        logits=now.cpu()), path + ".model.pt")


def transformer_reference_forward(m, x, kpm):
    """The torch CPU computation of the reference's TransformerAM (models/transformer.py:52-94) on the modules
    pykaldi2_amd.transformer.TransformerAM keeps as parameter containers, layer by layer (the fast-path probe
    of torch >= 2 rejects the custom layer)."""
    import torch.nn.functional as F
    h = m.input_layer(x)
    for lp in m.transformer.layers:
        h = lp.encoder_layer(h, None, kpm)
        h = F.relu(lp.conv1d(h.permute(1, 2, 0))).permute(2, 0, 1)
    return m.output_layer(m.transformer.norm(h))


DEN_ROOF_LENS = [589, 410, 377, 502]     # the fixed denominator workload of `--den-only`, the PMC passes and `roofline`


def den_roofline(den, dev, reps=5):
    """Times the denominator forward-backward (one C-ABI call = Tmax frame launches + the parallel passes around them)
    with events on the stream it is launched on, on the FIXED workload the committed PMC passes were taken on
    (4 sequences of DEN_ROOF_LENS frames, random logits): `algorithmic_bytes`, `traffic` and the timing all describe the
    same call.  Algorithmic bytes: SURVEY.md 8(d)."""
    lens = DEN_ROOF_LENS
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    x = torch.randn(len(lens), max(lens), P, device=dev, generator=gen)
    for _ in range(2):
        chain.den_forward_backward(den, x, lens, 1e-4)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        chain.den_forward_backward(den, x, lens, 1e-4)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    Tsum, Tmax = int(sum(lens)), int(max(lens))
    S, A = den.num_states(), den.num_arcs()
    rows = Tsum * 4 * (3 * P + 2 * (S + 1))
    byts = rows + 24 * A * Tmax                  # both arc lists streamed once per frame
    resident = rows + 24 * A                     # its lower bound: arc lists read once per call (cache-resident)
    ach = byts / (ms * 1e-3) / 1e9
    # HBM bytes per call from the committed PMC passes (tools/gpu_profile.sh) of this same workload: counters cannot be
    # read live.  Only reported when the profile was taken on the same lengths and graph shape.
    traffic, traffic_note = None, "no PMC profile of this workload committed"
    for name in ("r06_den_traffic.json", "r05_den_traffic.json", "r04_den_traffic.json"):         # the newest committed PMC passes of this workload
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                prof = json.load(f)
            if prof.get("lengths") == lens and prof.get("topology") == DEN_TOPOLOGY and prof.get("arcs") == A:
                traffic = int(prof["traffic_bytes_raw"])
                traffic_note = ("profiles/%s: rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE (separate passes), raw; with FETCH_SIZE doubled "
                                "(gfx950 wide-read correction, an upper bound here): %d" % (name, int(prof["traffic_bytes_fetch_x2"])))
                break
        except Exception:
            pass
    # what the memory system really moved per second (counter bytes of the committed profile over this run's timing): the
    # formula counts the per-frame arc re-stream the persistent kernel no longer performs
    counter = None if traffic is None else round(traffic / (ms * 1e-3) / 1e9, 1)
    return dict(bound="hbm", achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4),
                traffic=traffic, traffic_note=traffic_note, achieved_counter=counter,
                frac_counter=None if counter is None else round(counter / HBM_PEAK_GBS, 4),
                persist_form=den.persist_form(len(lens)),
                kernel={2: ("pk2::den_persist2_kernel" if den.persist_form(len(lens)) == 2 else "pk2::den_persist_kernel") +
                           " (one launch per denominator call: the alpha and beta recursions of the 4 "
                           "sequences, one per XCD, arcs in registers, state vector in LDS) + the parallel exp / occupancy passes",
                        1: "pk2::den_step_sx<4> x Tmax (one denominator forward-backward call; forward frame t and "
                           "backward frame Tmax-1-t share a launch)"}.get(den.kernel_path(len(lens)), "pk2::den_fwd_step / den_bwd_step x Tmax"),
                workload="4 sequences of %s frames, %s den graph (%d states, %d arcs, %d pdfs)" % (lens, DEN_TOPOLOGY, S, A, P),
                ms_per_launch=round(ms, 3), algorithmic_bytes=byts,
                us_per_frame=round(1e3 * ms / Tmax, 2),
                frac_cache_resident=round(resident / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                cache_resident_bytes=resident)


def lstm_flops_per_frame(feat=80, hidden=512, layers=3, pdfs=P):
    """Forward FLOPs of the BLSTM + output layer per network frame: input and recurrent products of both
    directions of every layer plus the affine output; forward + backward = 3x."""
    f, d_in = 0, feat
    for _ in range(layers):
        f += 2 * (2 * d_in * 4 * hidden + 2 * hidden * 4 * hidden)
        d_in = 2 * hidden
    return f + 2 * d_in * pdfs


def transformer_flops_per_frame(T, feat=80, dim=512, ff=2048, layers=12, pdfs=P):
    """Forward FLOPs of TransformerAM (models/transformer.py:52-94) per frame of a T-frame utterance: input Linear, per layer
    the QKV / output projections, the two T x T attention products over all heads, the FFN, Conv1d(k = 3), and the output Linear."""
    per_layer = 2 * dim * 3 * dim + 2 * dim * dim + 2 * 2 * T * dim + 2 * 2 * dim * ff + 2 * 3 * dim * dim
    return 2 * feat * dim + layers * per_layer + 2 * dim * pdfs


def gemm_arith_name():
    from pykaldi2_amd import _lib
    return "bf16x3" if _lib.lib().pk2_gemm_get_arith() == 1 else "f32"


def dtype_string():
    """The arithmetic the path computes in: f32 storage and accumulation everywhere; the GEMMs multiply either on the f32 MFMA or,
    by default, on the bf16 MFMA through an exact three-way bf16 split of both operands (csrc/gemm_bf16x3.h)."""
    return "f32 (GEMMs: bf16x3 split, f32 accumulate)" if gemm_arith_name() == "bf16x3" else "f32"


def gemm_mfma_roofline(dev, rows, reps=10):
    """MFMA utilisation of the BLSTM GEMMs (north star): the layer-1/2 input projection of this minibatch,
    [rows, 1024] x [1024, 4096] (both directions' gates), timed with events on the launch stream under BOTH arithmetic paths of
    pk2_gemm_f32.  `achieved` is the path the run uses, in f32-equivalent TFLOP/s (2 M N K per product); for bf16x3 the executed
    bf16 MFMA work is six times that and is priced against the 2.5 PFLOP/s dense bf16 peak."""
    from pykaldi2_amd import _lib
    from pykaldi2_amd.lstm import _gemm, _p
    L = _lib.lib()
    M, N, K = int(rows), 4096, 1024
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev)
    active = L.pk2_gemm_get_arith()
    res = {}
    for arith in (0, 1):
        _lib.check(L.pk2_gemm_set_arith(arith))
        for _ in range(3):
            _gemm(0, 1, M, N, K, _p(A), K, _p(W), K, _p(C), N)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            _gemm(0, 1, M, N, K, _p(A), K, _p(W), K, _p(C), N)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        res[arith] = (ms, 2.0 * M * N * K / (ms * 1e-3) / 1e12)
    _lib.check(L.pk2_gemm_set_arith(active))
    ms, tf = res[active]
    out = dict(bound="mfma", achieved=round(tf, 1), unit="TFLOP/s", ms_per_launch=round(ms, 4),
               arith="bf16x3" if active == 1 else "f32",
               f32_path=dict(achieved=round(res[0][1], 1), peak=157.3, frac=round(res[0][1] / 157.3, 4), ms_per_launch=round(res[0][0], 4),
                             kernel="v_mfma_f32_32x32x2_f32"),
               bf16x3_path=dict(achieved_f32_equivalent=round(res[1][1], 1), executed_bf16_tflops=round(6 * res[1][1], 1), peak_bf16=2500.0,
                                frac=round(6 * res[1][1] / 2500.0, 4), ms_per_launch=round(res[1][0], 4),
                                kernel="v_mfma_f32_32x32x16_bf16, six part products per k-step, f32 accumulate"))
    if active == 1:
        out.update(peak=round(2500.0 / 6, 1), frac=round(6 * tf / 2500.0, 4),
                   kernel="pk2::gemm_f32_kernel<.., bf16x3> (v_mfma_f32_32x32x16_bf16 x 6): BLSTM input projection %d x %d x %d; peak = "
                          "2.5 PFLOP/s dense bf16 / 6 executed products per f32-equivalent product" % (M, N, K))
    else:
        out.update(peak=157.3, frac=round(tf / 157.3, 4),
                   kernel="pk2::gemm_f32_kernel (v_mfma_f32_32x32x2_f32): BLSTM input projection %d x %d x %d" % (M, N, K))
    return out


def persistent_health(den, batch):
    """After the run: did any persistent kernel give up (a poll timed out -> abort flag, NaN-poisoned output) or fall back
    to its launch-per-step / per-frame form?  (with RCCL's kernels on the side stream the persistent kernels
    share the CUs; a timeout would show here, not as a hang.)"""
    import ctypes
    from pykaldi2_amd import _lib
    flag = ctypes.c_uint32(7)
    rc = _lib.lib().pk2_lstm_persist_status(ctypes.byref(flag))
    return dict(lstm_persist_abort=int(flag.value) if rc == 0 else -1, den_kernel_path=den.kernel_path(batch),
                den_persist_form=den.persist_form(batch), guard_raised=bool(_lib.persist_guard_raised()))


def usable_cores():
    """Cores this process may really use (affinity mask and cgroup quota), capped at 32."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, min(n, 32))


def chain_parity(g, pi, path):
    """In-job parity gate (north star "LF-MMI objective within 1e-3 rel"): the objective and the logit
    gradient the device computed for one minibatch of the run (dumped by the parent: logits, alignments, device results)
    against the C restatement of Kaldi's chain computation (oracle/chain_oracle.c) on the same logits."""
    from oracle import chain_c
    d = np.load(path)
    sups = build_supervisions([d["ali%d" % n] for n in range(int(d["n"]))])
    want_out, want_grad = chain_c.chain_batch(g, pi, d["logits"], sups, 1e-4, float(d["xent"]), double=True)
    got = d["gpu_out"].astype(np.float64)
    rel = float(abs(got[0].sum() - want_out[0].sum()) / abs(want_out[0].sum()))
    rel_seq = float((np.abs(got[0] - want_out[0]) / np.abs(want_out[0])).max())
    gerr = float(np.abs(d["gpu_grad"] - want_grad).max())
    return dict(objective_device=round(float(got[0].sum()), 4), objective_cpu_port=round(float(want_out[0].sum()), 4),
                objective_rel_err=float("%.3g" % rel), objective_rel_err_worst_sequence=float("%.3g" % rel_seq),
                den_logprob_rel_err=float("%.3g" % float((np.abs(got[2] - want_out[2]) / np.abs(want_out[2])).max())),
                grad_max_abs_err=float("%.3g" % gerr), tolerance={"objective_rel": 1e-3, "grad_abs": 1e-4},
                ok=bool(rel <= 1e-3 and rel_seq <= 1e-3 and gerr <= 1e-4),
                frames=[int(s.frames_per_sequence) for s in sups],
                against="oracle/chain_oracle.c, float64 build (restatement of Kaldi's DenominatorComputation / NumeratorComputation, "
                        "SURVEY App. A; unpinned at the Kaldi boundary) on the logits of one minibatch of this run")


def cpu_baseline_worker(seed, threads, parity_path=None):
    """Runs in a child process (no HIP context, bounded by a timeout): the same step on the host.
    numpy front end (oracle), the reference's torch CPU model path (nn.LSTM + nn.Linear,
    models/lstm.py:45-54), the C port of the chain objective, Adam(amsgrad) + clip."""
    os.environ["OMP_NUM_THREADS"] = str(threads)
    from oracle import chain_c, chain_ref, frontend_ref
    torch.set_num_threads(threads)
    g = den_graph_arrays()
    pi = chain_ref.initial_probs_ref(g["num_states"], g["src"].astype(np.int64), g["dst"].astype(np.int64),
                                     g["prob"].astype(np.float64), 0)
    parity = None
    if parity_path:
        try:
            parity = chain_parity(g, pi, parity_path)
        except Exception as e:       # the baseline is still worth reporting
            parity = dict(ok=False, error=repr(e)[:300])
    rng = np.random.default_rng(seed)
    ali_model = chain_model()[2]
    torch.manual_seed(0)
    arch = os.environ.get("PK2_BENCH_ARCH", "blstm")
    if arch == "transformer":        # configs[4]: the reference's torch modules of TransformerAM on the host
        from pykaldi2_amd import transformer
        tf = transformer.TransformerAM(80, 512, 8, 2048, 12, 0.0, P)
        params = list(tf.parameters())
        if parity_path and parity is not None and os.path.exists(parity_path + ".model.pt"):
            try:                     # logits of the run's own minibatch and weights: device vs torch CPU
                blob = torch.load(parity_path + ".model.pt")
                tf.load_state_dict(blob["state_dict"])
                Tp = blob["x"].shape[0]
                kpm = torch.ones(len(blob["frames"]), Tp, dtype=torch.bool)
                for n_, f_ in enumerate(blob["frames"]):
                    kpm[n_, :-(-int(f_) // 3)] = False
                with torch.no_grad():
                    want = transformer_reference_forward(tf.eval(), blob["x"], kpm).transpose(0, 1)     # [N, T', P]
                got = blob["logits"]
                valid = ~kpm
                err = float((got - want)[valid].abs().max())
                scale = float(want[valid].abs().max())
                parity["logits_max_abs_err"] = float("%.3g" % err)
                parity["logits_rel_to_max"] = float("%.3g" % (err / max(scale, 1e-30)))
                parity["logits_against"] = "torch CPU forward of the reference's TransformerAM modules with the run's weights"
                parity["ok"] = bool(parity.get("ok") and err <= 2e-4 * max(1.0, scale))
                tf.train()
            except Exception as e:
                parity["logits_error"] = repr(e)[:200]
    else:
        rnn = torch.nn.LSTM(80, 512, 3, batch_first=True, bidirectional=True)
        lin = torch.nn.Linear(1024, P)
        params = list(rnn.parameters()) + list(lin.parameters())
    opt = torch.optim.Adam(params, lr=1e-3, amsgrad=True)
    mel = fbank.mel_filterbank()
    # a bounded sample of the same workload: whole training steps until ~12 s of host work are done (at most 8 minibatches)
    seconds, dt, steps = 0.0, 0.0, 0
    while steps < 8 and dt < 12.0:
        host = synth.minibatch(rng, 4, P, ali_model=ali_model)
        t0 = time.time()
        feats = [frontend_ref.cmn(frontend_ref.logfbank(w, mel)).astype(np.float32) for w, _ in host]
        x = torch.from_numpy(frontend_ref.pad_roll_subsample(feats, 0, 3).copy())
        if arch == "transformer":
            kpm = torch.ones(x.shape[0], x.shape[1], dtype=torch.bool)
            for n_, f_ in enumerate(feats):
                kpm[n_, :-(-f_.shape[0] // 3)] = False
            logits = transformer_reference_forward(tf, x.transpose(0, 1), kpm).transpose(0, 1)
        else:
            logits = lin(rnn(x)[0])
        sups = build_supervisions([a for _, a in host])
        out, grad = chain_c.chain_batch(g, pi, logits.detach().numpy(), sups, 1e-4, 0.1)
        opt.zero_grad()
        logits.backward(torch.from_numpy(-grad))
        torch.nn.utils.clip_grad_norm_(params, 5.0)
        opt.step()
        dt += time.time() - t0
        seconds += sum(w.shape[0] for w, _ in host) / 16000.0
        steps += 1
    print(json.dumps(dict(value=round(seconds / dt, 2), unit="hours of audio per wall-clock hour", cores=threads,
                          kind="port", parity=parity,
                          sample="%d minibatches of 4 utterances (%.1f s audio): numpy fbank oracle + torch CPU %s fwd/bwd + "
                                 "C port of the LF-MMI objective (OpenMP over sequences) + Adam; %.1f s wall on %d threads"
                                 % (steps, seconds, "12-layer TransformerAM" if arch == "transformer" else "3x512 BLSTM", dt, threads))),
          flush=True)


def ce_parity(path):
    """north star "CE frame posteriors within 1e-4 rel": the device's posteriors for chunks of one 256 x 80 minibatch of the
    run (model in eval mode) against the reference's torch CPU modules (nn.LSTM + nn.Linear, models/lstm.py:45-54)
    loaded with the same state_dict."""
    d = torch.load(path)
    PC = d["logits"].shape[-1]
    rnn = torch.nn.LSTM(80, 512, 3, batch_first=True, dropout=0.2, bidirectional=True).eval()
    lin = torch.nn.Linear(1024, PC).eval()
    rnn.load_state_dict({k[5:]: v for k, v in d["state_dict"].items() if k.startswith("lstm.")})
    lin.load_state_dict({k[13:]: v for k, v in d["state_dict"].items() if k.startswith("output_layer.")})
    with torch.no_grad():
        want = torch.softmax(lin(rnn(d["x"])[0]).double(), -1)
    got = torch.softmax(d["logits"].double(), -1)
    rel = float(((got - want).abs() / want.clamp_min(1e-30)).max())
    return dict(posterior_max_rel_err=float("%.3g" % rel), tolerance={"posterior_rel": 1e-4}, ok=bool(rel <= 1e-4),
                chunks=int(d["x"].shape[0]), frames_per_chunk=int(d["x"].shape[1]),
                against="torch CPU nn.LSTM(80, 512, 3, bidirectional) + nn.Linear with the run's state_dict "
                        "(the reference's LSTMAM modules, models/lstm.py:45-54)")


def cpu_ce_worker(threads, parity_path=None):
    """SURVEY 8(d) config 1 in a child process: the reference's torch CPU CE path (nn.LSTM + nn.Linear = models/lstm.py:45-54,
    nn.CrossEntropyLoss, clip 5, Adam(amsgrad, lr 1e-4)) on the GPU leg's own minibatch shape x[256,80,80], P=5768, dropout 0.2,
    on the node's host cores (the CPU leg used to run 64 chunks against the GPU's 256)."""
    torch.set_num_threads(threads)
    parity = None
    if parity_path:
        try:
            parity = ce_parity(parity_path)
        except Exception as e:
            parity = dict(ok=False, error=repr(e)[:300])
    torch.manual_seed(0)
    B, T, PC = 256, 80, 5768
    rnn = torch.nn.LSTM(80, 512, 3, batch_first=True, dropout=0.2, bidirectional=True)
    lin = torch.nn.Linear(1024, PC)
    params = list(rnn.parameters()) + list(lin.parameters())
    opt = torch.optim.Adam(params, lr=1e-4, amsgrad=True)
    crit = torch.nn.CrossEntropyLoss(ignore_index=-100)
    x = torch.randn(B, T, 80)
    y = torch.randint(0, PC, (B, T))

    def step():
        opt.zero_grad()
        loss = crit(lin(rnn(x)[0]).reshape(-1, PC), y.reshape(-1))
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 5.0)
        opt.step()
    step()
    n, t0 = 0, time.time()
    while n < 2 or (time.time() - t0 < 15.0 and n < 12):
        step()
        n += 1
    dt = (time.time() - t0) / n
    print(json.dumps(dict(value=round(B * T * 0.01 / dt, 2), unit="hours of audio per wall-clock hour", cores=threads,
                          kind="reference", parity=parity,
                          sample="%d steps of the reference's torch CPU CE path (nn.LSTM 3x512 bidirectional + Linear, "
                                 "CrossEntropyLoss, clip 5, Adam amsgrad) on x[256,80,80], P=5768: %.2f s per step on %d threads"
                                 % (n, dt, threads))), flush=True)


def cpu_baseline(seed, timeout=240, worker="--cpu-baseline-worker", parity_path=None):
    import subprocess
    threads = usable_cores()
    cmd = [sys.executable, os.path.abspath(__file__), worker, str(seed), str(threads)] + ([parity_path] if parity_path else [])
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=dict(os.environ, HIP_VISIBLE_DEVICES=""))
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode == 0 and line:
            return json.loads(line[-1])
        return dict(value=None, cores=threads, kind="port", sample="cpu baseline failed: " + r.stderr[-300:])
    except subprocess.TimeoutExpired:
        return dict(value=None, cores=threads, kind="port", sample="cpu baseline exceeded %d s" % timeout)


def ce_workload(args, dev, rank, world):
    """configs[1]: CE on 256 chunks of 80 frames per step from raw waveforms in HBM: fbank + CMN per utterance,
    chunking, 3x512 BLSTM (dropout 0.2, configs/ce.yaml), fused softmax-CE, clip 5, Adam(amsgrad, lr 1e-4)."""
    PC, CH, BATCH = 5768, 80, 256
    rng = np.random.default_rng(99 + rank)
    torch.manual_seed(0)
    model = lstm.LSTMAM(80, PC, 512, 3, 0.2, True).to(dev).train()
    opt = hvd.DistributedOptimizer(optim.Adam(model, lr=1e-4, amsgrad=True), named_parameters=model.named_parameters())
    crit = ops.CrossEntropyLoss(ignore_index=-100)
    fb = fbank.FbankExtractor()
    batches = []
    for _ in range(3):
        utts, chunks = [], 0
        while chunks < BATCH:
            w = synth.waveform(rng, float(synth.utterance_durations(rng, 1)[0]))
            T = synth.num_fbank_frames(w.shape[0])
            utts.append((w, synth.pdf_alignment(rng, T, PC)))
            chunks += T // CH
        lens = [w.shape[0] for w, _ in utts]
        batches.append(dict(wav=torch.from_numpy(np.concatenate([w for w, _ in utts])).to(dev), lens=lens,
                            labels=[torch.from_numpy(a).to(dev).unsqueeze(1) for _, a in utts]))

    def step(mb):
        feats, frames, row_off = fb(mb["wav"], mb["lens"])
        off = np.concatenate([[0], np.cumsum(frames)])
        xs = [fbank.utt2seg(feats[off[n]:off[n + 1]], CH, CH) for n in range(len(frames))]
        ys = [fbank.utt2seg(l, CH, CH) for l in mb["labels"]]
        x, y = torch.cat(xs)[:BATCH], torch.cat(ys)[:BATCH]
        logits = model.forward_time_major(x.transpose(0, 1).contiguous())          # [T=80, B=256, P]
        loss = crit(logits.transpose(0, 1), y.squeeze(2))
        opt.zero_grad()
        loss.backward()
        optim.clip_grad_norm_(opt, 5.0)
        opt.step()
        return loss
    for i in range(args.warmup):
        step(batches[i % 3])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(batches[i % 3])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    audio = args.steps * BATCH * CH * 0.01
    if rank == 0:
        base = parity = None
        # whole-step MFMA utilisation (125.5 MFLOP per network frame forward + backward at P = 5768, all of it
        # MFMA-eligible) and the dominant GEMM of the step measured alone, events on the launch stream
        ms_step = 1e3 * dt / args.steps
        flops = 3.0 * lstm_flops_per_frame(pdfs=PC) * BATCH * CH
        tf = flops / (ms_step * 1e-3) / 1e12
        roof = dict(bound="mfma", achieved=round(tf, 1), peak=157.3, unit="TFLOP/s", frac=round(tf / 157.3, 4), traffic=None,
                    gemm_arith=gemm_arith_name(), frac_of_bf16x3_peak=round(6.0 * tf / 2500.0, 4),
                    kernel="whole CE step (3x512 BLSTM + output layer forward + backward: GEMMs pk2::gemm_f32_kernel in the arithmetic "
                    "`gemm_arith` + the persistent recurrences pk2::lstm_*_big_persist on the f32 MFMA) over the step's wall time; "
                    "FLOPs are f32-equivalent; peak / frac = the f32 MFMA peak (what the step is priced against since round 1), "
                    "frac_of_bf16x3_peak = the same FLOPs against 2.5 PFLOP/s dense bf16 / 6 executed products",
                    flops_per_step=flops, ms=round(ms_step, 3),
                    input_projection_gemm=gemm_mfma_roofline(dev, BATCH * CH))
        if world == 1 and not args.no_cpu_baseline:
            import tempfile
            with tempfile.TemporaryDirectory() as td:
                # posteriors of the first 32 chunks of one minibatch, model in eval mode (dropout off), for the child
                path = os.path.join(td, "ce_parity.pt")
                mb = batches[0]
                model.eval()
                with torch.no_grad():
                    feats, frames, row_off = fb(mb["wav"], mb["lens"])
                    off = np.concatenate([[0], np.cumsum(frames)])
                    x = torch.cat([fbank.utt2seg(feats[off[n]:off[n + 1]], CH, CH) for n in range(len(frames))])[:BATCH]
                    lg = model.forward_time_major(x.transpose(0, 1).contiguous()).transpose(0, 1)[:32]
                model.train()
                torch.save(dict(state_dict={k: v.cpu() for k, v in model.state_dict().items()}, x=x[:32].cpu(),
                                logits=lg.contiguous().cpu()), path)
                base = cpu_baseline(0, worker="--cpu-ce-worker", parity_path=path)
            parity = base.pop("parity", None)
        print(json.dumps({"metric": "iRTF (hrs audio/hr) 3x512 BLSTM CE, 256x80 chunks (secondary workload, configs[1])",
                          "value": round(audio / dt * world, 2), "n_gpus": world, "steps": args.steps,
                          "unit": "hours of audio per wall-clock hour", "higher_is_better": True,
                          "config": {"workload": "SECONDARY configs[1]: 3x512 BLSTM CE, batch 256 x 80 x 80 fbank from raw waveforms in "
                                     "HBM, P=5768, dropout 0.2, Adam(amsgrad)+clip 5"},
                          "ms_per_step": round(1e3 * dt / args.steps, 3), "dtype": dtype_string(), "loss": round(float(loss.item()), 4),
                          "roofline": roof, "cpu_baseline": base, "parity": parity,
                          "hbm_peak_allocated_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
                          "reference_published": "README.md:43-45: 190 iRTF (64x80, 1 V100), 520 iRTF (256x80, 4 V100)"}),
              flush=True)
    hvd.shutdown()


def se_workload(args, dev, rank, world):
    """configs[3]: lattice MMI (train_se.py) with on-the-fly lattices: `--batch` var-len utterances per GPU at
    100 fps, 3x512 BLSTM, P=5768, word-loop HCLG, decoder_config of the reference's se.yaml (beam 13,
    lattice_beam 7, max_active 7000, acoustic_scale 0.1), CE regulariser 0.1, clip 5, SGD."""
    from pykaldi2_amd import data, lattice, se
    PS = 5768
    batch = args.batch if args.batch != 4 else 8
    torch.manual_seed(0)
    model = lstm.LSTMAM(80, PS, 512, 3, 0.0, True).to(dev).train()
    opt = hvd.DistributedOptimizer(optim.SGD(model, lr=1e-5, momentum=0.9), named_parameters=model.named_parameters())
    crit = ops.CrossEntropyLoss(ignore_index=-100, reduction="sum")
    fb = fbank.FbankExtractor()
    tm = synth.transition_model_arrays(PS)
    trans_model = lattice.TransitionModel.from_arrays(tm)
    o = lattice.LatticeFasterDecoderOptions(beam=13.0, lattice_beam=7.0, max_active=7000, min_active=200)
    words = int(os.environ.get("PK2_SE_WORDS", "20000"))
    log("building the synthetic HCLG (%d words)" % words)
    g = synth.decoding_graph_arcs(words, PS, seed=0)
    rec = lattice.MappedLatticeFasterRecognizer(trans_model, g, acoustic_scale=0.1, decoder_opts=o)
    log("HCLG: %d states, %d arcs" % (rec.graph.num_states, rec.graph.num_arcs))
    log_prior = se.log_prior_from_counts(np.ones(PS)).to(dev)
    src = data.SyntheticSource(PS, seed=7, rank=rank, world=world, with_tids=True)
    batches = []
    for _ in range(3):
        utts = [src.draw() for _ in range(batch)]
        lens = [u[0].shape[0] for u in utts]
        batches.append(dict(wav=torch.from_numpy(np.concatenate([u[0] for u in utts])).to(dev), lens=lens,
                            y=[u[1] for u in utts], aux=[u[2] for u in utts], seconds=sum(lens) / 16000.0))
    crit_name = os.environ.get("PK2_SE_CRITERION", "mmi")

    def step(mb):
        loss, se_val, ce, frames = se.sequence_loss(model, fb, mb, rec, trans_model, log_prior, crit_name,
                                                    tm["silence_phones"], 0.1, crit)
        opt.zero_grad()
        loss.backward()
        optim.clip_grad_norm_(opt, 5.0)
        opt.step()
        return loss, frames
    for i in range(max(1, args.warmup)):
        step(batches[i % 3])
    torch.cuda.synchronize()
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
    allocs0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
    t0 = time.perf_counter()
    audio = 0.0
    step_ms = []
    for i in range(args.steps):
        ts = time.perf_counter()
        loss, frames = step(batches[i % 3])
        audio += float(np.sum(frames)) * 0.01
        step_ms.append(round(1e3 * (time.perf_counter() - ts), 1))      # (host time of the step's enqueueing; the decode's summary synchronises)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    mstat = torch.cuda.memory_stats(dev)
    device_allocs = int(mstat.get("num_device_alloc", 0) - allocs0)     # hipMalloc calls inside the timed region
    free_b, total_b = torch.cuda.mem_get_info(dev)
    mem_gb = dict(device_total=round(total_b / 2 ** 30, 1), device_free=round(free_b / 2 ** 30, 1),
                  reserved_peak=round(mstat.get("reserved_bytes.all.peak", 0) / 2 ** 30, 1),
                  allocated_peak=round(mstat.get("allocated_bytes.all.peak", 0) / 2 ** 30, 1), alloc_retries=int(mstat.get("num_alloc_retries", 0)))
    # breakdown of the lattice part on the last minibatch (outside the timed region)
    mb = batches[(args.steps - 1) % 3]
    with torch.no_grad():
        feats, frames, row_off = fb(mb["wav"], mb["lens"])
        x = fb.pad_roll_subsample(feats, row_off, frames, shift=0, subsample=1, time_major=True)
        ll = model.forward_time_major(x).transpose(0, 1) - log_prior
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        lat = rec.decode_batch(ll, [int(t) for t in frames])
        ev[1].record()
        if crit_name == "mmi":
            lat.mmi(mb["aux"], 1.0, 0.2, True)
        else:
            lat.mpe(mb["aux"], crit_name, tm["silence_phones"], True)
        ev[2].record()
        torch.cuda.synchronize()
    if rank == 0:
        # Dominant kernel: the persistent lattice decoder (integer / pointer work on HBM-resident graphs; no MFMA).  Algorithmic
        # bytes per decode call: per link created 56 B (16 B HCLG arc record + 4 B log-likelihood read, 16 B of
        # the per-state {cost, token} table read and written, 16 B link record + 4 B acoustic cost written), per token 48 B
        # (20 B token fields written, read again by the next frame, 8 B arc-range lookup) -- over the kept links / tokens
        # the call reports, so a lower bound of what the search touched.
        dec_ms = ev[0].elapsed_time(ev[1])
        toks, links = float(lat.num_tokens.sum()), float(lat.num_links.sum())
        byts = 56.0 * links + 48.0 * toks
        ach = byts / (dec_ms * 1e-3) / 1e9
        roof = dict(bound="hbm", achieved=round(ach, 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 5), traffic=None,
                    kernel="pk2::lat_frames_persist + the pruning kernels behind it (one decode-and-prune call of the minibatch: "
                    "a team of workgroups per utterance, all frames in one launch); latency-bound token passing, see DESIGN.md 4.4",
                    ms_per_launch=round(dec_ms, 2), algorithmic_bytes=int(byts),
                    us_per_frame=round(1e3 * dec_ms / float(np.max(frames)), 2))
        base = parity = None
        if world == 1 and not args.no_cpu_baseline and crit_name == "mmi":
            import tempfile
            F = min(120, int(frames[0]))       # bounded sample: the first 1.2 s of utterance 0 of this minibatch
            with torch.no_grad():
                ll0 = ll[0, :F].contiguous()
                lat0 = rec.decode(ll0)
                ali0 = np.asarray(mb["aux"][0])[:F]
                like0, post0 = lat0.mmi([ali0], 1.0, 0.2, True)
                torch.cuda.synchronize()
                exp0 = lat0.export(0)          # the kept (pruned) lattice as arrays
            with tempfile.TemporaryDirectory() as td:
                path = os.path.join(td, "se_parity.npz")
                np.savez(path, ll=ll0.cpu().numpy(), ali=ali0, words=words, wav=mb["wav"][:401 + 160 * (F - 1)].cpu().numpy(),
                         best_cost=np.float32(lat0.best_cost[0]), num_links=int(len(exp0["link_src"])), num_tokens=int(lat0.num_tokens[0]),
                         like=float(like0.item()), post=post0[0].cpu().numpy(),
                         # the whole minibatch for the timed CPU leg (waveforms, their lengths, transition-id alignments)
                         mb_wav=mb["wav"].cpu().numpy(), mb_lens=np.asarray(mb["lens"], np.int64),
                         mb_ali=np.concatenate([np.asarray(a, np.int32) for a in mb["aux"]]),
                         mb_ali_lens=np.asarray([len(a) for a in mb["aux"]], np.int64))
                base = cpu_baseline(0, timeout=420, worker="--cpu-se-worker", parity_path=path)
            parity = base.pop("parity", None)
        print(json.dumps({"metric": "iRTF (hrs audio/hr) 3x512 BLSTM lattice-%s, on-the-fly lattices (secondary workload, configs[3])" % crit_name.upper(),
                          "value": round(audio / dt * world, 2), "unit": "hours of audio per wall-clock hour", "higher_is_better": True,
                          "config": {"workload": "SECONDARY configs[3]: 3x512 BLSTM lattice-%s (train_se.py), on-the-fly lattices, batch %d x "
                                     "var-len per GPU at 100 fps, P=5768, %d-word HCLG, beam 13 / lattice_beam 7 / max_active 7000, "
                                     "CE regulariser 0.1, SGD+clip 5" % (crit_name.upper(), batch, words)},
                          "n_gpus": world, "steps": args.steps,
                          "ms_per_step": round(1e3 * dt / args.steps, 3), "dtype": "f32", "batch_per_gpu": batch,
                          "step_host_ms": step_ms, "device_allocs_in_timed_region": device_allocs, "memory_gb": mem_gb,
                          "hclg": {"states": rec.graph.num_states, "arcs": rec.graph.num_arcs},
                          "lattice_ms": {"decode_and_prune": round(ev[0].elapsed_time(ev[1]), 2),
                                         "forward_backward": round(ev[1].elapsed_time(ev[2]), 2)},
                          "lattice_tokens_per_frame": round(float(lat.num_tokens.sum()) / float(np.sum(frames)), 1),
                          "lattice_links_per_frame": round(float(lat.num_links.sum()) / float(np.sum(frames)), 1),
                          "loss": round(float(loss.item()), 2),
                          "roofline": roof, "cpu_baseline": base, "parity": parity,
                          "hbm_peak_allocated_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
                          "reference_published": "README.md:47-49: lattice MMI 16.7 iRTF (1 V100, 4 utterances), 34.5 iRTF (4 V100)"}),
              flush=True)
    hvd.shutdown()


def cpu_se_worker(threads, parity_path):
    """configs[3] on the host, in a child process.  Two legs:
    * parity (unchanged): the first 1.2 s of one utterance of the run -- the numpy restatement of Kaldi's LatticeFasterDecoder
      and of the lattice MMI forward-backward (oracle/lattice_ref.py) on the DEVICE's log-likelihoods of that sample against
      the device's lattice (best cost, kept links, MMI log-likelihood, posteriors); the C port of the same oracle
      (oracle/lattice_oracle.c) is checked against the numpy oracle on that sample as well;
    * cpu_baseline (round 6, VERDICT r5 #4: "1.2 s through a single-threaded numpy decoder is a number, not a baseline"): ONE
      WHOLE minibatch of the run (8 utterances, ~96 s of audio): numpy fbank oracle, the reference's torch CPU modules
      (nn.LSTM 3x512 bidirectional + Linear) forward, the C decoder + lattice MMI per utterance on a pool of host threads,
      backward through the model with the posteriors, clip 5, SGD."""
    os.environ["OMP_NUM_THREADS"] = str(threads)
    torch.set_num_threads(threads)
    from concurrent.futures import ThreadPoolExecutor
    from oracle import frontend_ref, lattice_c, lattice_ref as LR
    d = np.load(parity_path)
    PS, F = d["ll"].shape[1], d["ll"].shape[0]
    g = synth.decoding_graph_arcs(int(d["words"]), PS, seed=0)
    tm = synth.transition_model_arrays(PS)
    G = LR.DecodeGraphRef(g["num_states"], g["start"], g["src"], g["dst"], g["ilabel"], g["weight"], g["final"])
    o = LR.DecoderOptionsRef(beam=13.0, lattice_beam=7.0, max_active=7000, min_active=200, beam_delta=0.5, acoustic_scale=0.1)
    # ---- parity leg
    want = LR.decode(G, d["ll"], tm["tid2pdf"], o)
    wl, wp = LR.lattice_mmi(want, d["ali"], tm["tid2pdf"], PS, 1.0, 0.2, True)
    cport = lattice_c.decode_mmi(G, d["ll"], tm["tid2pdf"], o, d["ali"], PS, 1.0, 0.2, True)
    perr = float(np.abs(d["post"] - wp).max())
    lrel = float(abs(float(d["like"]) - wl) / max(1.0, abs(wl)))
    c_ok = bool(np.float32(cport["best_cost"]) == np.float32(want.best_cost) and cport["links"] == len(want.link_src)
                and abs(cport["like"] - wl) <= 1e-9 * max(1.0, abs(wl)) and float(np.abs(cport["post"] - wp).max()) <= 1e-9)
    parity = dict(best_cost_device=float(d["best_cost"]), best_cost_oracle=float(np.float32(want.best_cost)),
                  kept_links_device=int(d["num_links"]), kept_links_oracle=int(len(want.link_src)),
                  mmi_loglike_rel_err=float("%.3g" % lrel), posterior_max_abs_err=float("%.3g" % perr),
                  tolerance={"best_cost": "bit-equal (float32)", "kept_links": "equal", "loglike_rel": 1e-9, "posterior_abs": 2e-6},
                  c_port_equals_numpy_oracle=c_ok,
                  ok=bool(np.float32(d["best_cost"]) == np.float32(want.best_cost) and int(d["num_links"]) == len(want.link_src)
                          and lrel <= 1e-9 and perr <= 2e-6 and c_ok), frames=int(F),
                  against="oracle/lattice_ref.py (numpy restatement of Kaldi's LatticeFasterDecoder + lattice MMI forward-backward; "
                          "unpinned at the Kaldi boundary) on the device's log-likelihoods of the sample")
    # ---- timed leg: one whole minibatch
    torch.manual_seed(0)
    rnn = torch.nn.LSTM(80, 512, 3, batch_first=True, bidirectional=True)
    lin = torch.nn.Linear(1024, PS)
    params = list(rnn.parameters()) + list(lin.parameters())
    opt = torch.optim.SGD(params, lr=1e-5, momentum=0.9)
    mel = fbank.mel_filterbank()
    lens = [int(v) for v in d["mb_lens"]]
    woff = np.concatenate([[0], np.cumsum(lens)])
    aoff = np.concatenate([[0], np.cumsum(d["mb_ali_lens"])])
    log_prior = float(-np.log(PS))
    t0 = time.time()
    with ThreadPoolExecutor(max_workers=threads) as pool:
        feats = list(pool.map(lambda n: frontend_ref.cmn(frontend_ref.logfbank(d["mb_wav"][woff[n]:woff[n + 1]], mel)).astype(np.float32),
                              range(len(lens))))
        t_feat = time.time()
        frames = [f.shape[0] for f in feats]
        x = torch.zeros(len(lens), max(frames), 80)
        for n, f in enumerate(feats):
            x[n, :f.shape[0]] = torch.from_numpy(f)
        logits = lin(rnn(x)[0])
        ll = (torch.log_softmax(logits.detach(), -1) - log_prior).numpy()
        t_fwd = time.time()
        res = list(pool.map(lambda n: lattice_c.decode_mmi(G, ll[n, :frames[n]], tm["tid2pdf"], o, d["mb_ali"][aoff[n]:aoff[n + 1]], PS,
                                                           1.0, 0.2, True), range(len(lens))))
        t_lat = time.time()
    grad = torch.zeros_like(logits)
    for n, r in enumerate(res):
        grad[n, :frames[n]] = torch.from_numpy(-r["post"].astype(np.float32))
    opt.zero_grad()
    torch.log_softmax(logits, -1).backward(grad)
    torch.nn.utils.clip_grad_norm_(params, 5.0)
    opt.step()
    dt = time.time() - t0
    audio = sum(frames) * 0.01
    print(json.dumps(dict(value=round(audio / dt, 2), unit="hours of audio per wall-clock hour", cores=threads, kind="port",
                          parity=parity,
                          sample="one whole minibatch of the run (%d utterances, %.1f s of audio, %d frames): numpy fbank oracle %.1f s + torch "
                                 "CPU 3x512 BLSTM forward %.1f s + C port of the lattice decoder / MMI oracle (oracle/lattice_oracle.c, one "
                                 "utterance per host thread; %d links kept) %.1f s + backward, clip, SGD %.1f s = %.1f s wall on %d threads"
                                 % (len(lens), audio, sum(frames), t_feat - t0, t_fwd - t_feat, sum(r["links"] for r in res), t_lat - t_fwd,
                                    time.time() - t_lat, dt, threads))), flush=True)


def run_secondaries():
    """BASELINE configs[1], [3], [4] next to the headline: each runs as a short child job of its own
    (`bench.py --ce | --se | --transformer`: own process, own HIP context, this GPU; the parent is idle meanwhile) with its own
    timed window, roofline of its dominant kernel, CPU baseline on a bounded sample and parity check; the parent keeps the
    fields a reader needs.  Nothing of this runs inside the headline's timed region."""
    import subprocess
    out = {}
    for name, flags in (("ce", ["--ce", "--steps", "8", "--warmup", "3"]), ("se", ["--se", "--steps", "3", "--warmup", "3"]),
                        ("transformer", ["--transformer", "--steps", "8", "--warmup", "3"])):
        t0 = time.time()
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__)] + flags, capture_output=True, text=True, timeout=420)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not line:
                out[name] = dict(error=("exit %d: " % r.returncode) + r.stderr[-300:])
                continue
            d = json.loads(line[-1])
            roof = d.get("roofline_model") if name == "transformer" else d.get("roofline")
            keep = dict(metric=d["metric"], workload=d.get("config", {}).get("workload"), value=d["value"], unit=d.get("unit"),
                        ms_per_step=d["ms_per_step"], steps=d["steps"], dtype=d.get("dtype"),
                        roofline=None if roof is None else {k: roof.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel")},
                        cpu_baseline=d.get("cpu_baseline"), parity=d.get("parity"), wall_s=round(time.time() - t0, 1))
            if name == "transformer":
                keep["roofline_denominator"] = {k: d["roofline"].get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "us_per_frame")}
                keep["breakdown_ms"] = d.get("breakdown_ms")
            if name == "se":
                keep["lattice_ms"] = d.get("lattice_ms")
                keep["step_host_ms"] = d.get("step_host_ms")
                keep["device_allocs_in_timed_region"] = d.get("device_allocs_in_timed_region")
            out[name] = keep
        except subprocess.TimeoutExpired:
            out[name] = dict(error="exceeded 420 s")
        log("secondary %s done (%.0f s)" % (name, time.time() - t0))
    return out


def log(msg):
    sys.stderr.write("[bench %.1fs] %s\n" % (time.time() - T_START, msg))
    sys.stderr.flush()


T_START = time.time()


def main():
    if len(sys.argv) >= 4 and sys.argv[1] == "--cpu-baseline-worker":
        return cpu_baseline_worker(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] if len(sys.argv) > 4 else None)
    if len(sys.argv) >= 4 and sys.argv[1] == "--cpu-ce-worker":
        return cpu_ce_worker(int(sys.argv[3]), sys.argv[4] if len(sys.argv) > 4 else None)
    if len(sys.argv) >= 5 and sys.argv[1] == "--cpu-se-worker":
        return cpu_se_worker(int(sys.argv[3]), sys.argv[4])
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short child runs of configs[1], [3], [4] that the default "
                    "one-GPU run appends to its line as `secondary`")
    ap.add_argument("--reserve-gb", type=int, default=None, help="HBM handed to the caching allocator in one piece at start-up "
                    "(0: let it grow lazily; default 12, --se 48: their runs peak at 8 / 33 GB)")
    ap.add_argument("--length-bucketed", action="store_true", help="N > 1: every rank draws the same utterance lengths "
                    "(different audio / alignments), i.e. length-bucketed data parallelism without stragglers")
    ap.add_argument("--den-only", action="store_true", help="time only the denominator forward-backward")
    ap.add_argument("--den-states", type=int, default=None, help="states of the synthetic den graph (default 30000); with "
                    "--den-only: the graph-size sweep of profiles/r03_den_sweep.txt")
    ap.add_argument("--den-arcs", type=int, default=None, help="arcs of the synthetic den graph (default 1000000)")
    ap.add_argument("--den-topology", choices=["chain", "unique"], default=None,
                    help="synthetic den.fst shape: chain (default; self-loop pdf != entering pdf, as Kaldi's chain topology "
                    "gives) or unique (round 1: one pdf per destination state)")
    ap.add_argument("--lstm-only", action="store_true", help="time only one LSTM layer forward")
    ap.add_argument("--gemm-only", action="store_true", help="time the f32 MFMA GEMM on the model's shapes")
    ap.add_argument("--gemm-shapes", default="", help="with --gemm-only: 'ta,tb,M,N,K;...' instead of the built-in shapes")
    ap.add_argument("--transformer", action="store_true", help="secondary workload configs[4]: 12-layer TransformerAM "
                    "LF-MMI instead of the BLSTM")
    ap.add_argument("--se", action="store_true", help="secondary workload configs[3]: lattice MMI with on-the-fly lattices "
                    "(PK2_SE_CRITERION=smbr|mpfe for the other criteria)")
    ap.add_argument("--ce", action="store_true", help="secondary workload configs[1]: 3x512 BLSTM CE, 256 x 80-frame "
                    "chunks per step (not the headline metric)")
    args = ap.parse_args()
    global DEN_TOPOLOGY, S_DEN, A_DEN
    if args.den_states:
        S_DEN = args.den_states
    if args.den_arcs:
        A_DEN = args.den_arcs
    if args.den_topology:
        DEN_TOPOLOGY = args.den_topology
        os.environ["PK2_BENCH_DEN_TOPOLOGY"] = DEN_TOPOLOGY     # the cpu-baseline child builds the same graph

    hvd.init()
    rank, world = hvd.rank(), hvd.size()
    assert world == args.gpus or world == 1, "launch with torch.distributed.run for --gpus > 1"
    dev = torch.device("cuda", hvd.local_device())
    torch.cuda.set_device(dev)
    # Start-up, like building the graphs: the caching allocator gets the run's memory in ONE piece (288 GB of HBM: a few GB are
    # nothing), so that it never has to go back to the driver inside the timed region.  Its growth is lazy -- the longest
    # minibatch of the run may first appear among the timed steps (W < 8 unique minibatches) -- and a hipMalloc of a few GB costs
    # ~1 ms on most boxes of the pool but ~100 ms on some (memory cleared on allocation): two of ~25 bench runs of rounds 2 / 3
    # read 20.2 / 20.3 ms per step for 14.6 with an unchanged per-step GPU time (tools/step_diag.py).
    if args.reserve_gb is None:
        args.reserve_gb = 48 if args.se else 12
    if args.reserve_gb > 0:
        reserve = torch.empty(args.reserve_gb << 30, dtype=torch.uint8, device=dev)
        del reserve

    if args.ce:
        return ce_workload(args, dev, rank, world)
    if args.se:
        return se_workload(args, dev, rank, world)
    if args.gemm_only:
        from pykaldi2_amd.lstm import _gemm, _p
        shapes = [(0, 1, 2356, 4096, 1024), (0, 1, 2356, 6048, 1024), (1, 0, 6048, 1024, 2356), (0, 0, 2356, 1024, 6048),
                  (1, 0, 4096, 1024, 2356), (1, 0, 2048, 512, 2352), (0, 0, 2356, 1024, 4096), (0, 1, 2356, 4096, 80),
                  (0, 1, 20480, 4096, 1024), (0, 1, 4096, 4096, 4096),
                  (1, 0, 4096, 1024, 20480), (1, 0, 2048, 512, 20480), (1, 0, 5768, 1024, 20480), (0, 0, 20480, 1024, 4096)]
        if args.gemm_shapes:      # "ta,tb,M,N,K;..." instead of the built-in list
            shapes = [tuple(int(v) for v in sh.split(",")) for sh in args.gemm_shapes.split(";") if sh]
        for ta, tb, M, N, K in shapes:
            A = torch.randn((K, M) if ta else (M, K), device=dev)
            Bm = torch.randn((N, K) if tb else (K, N), device=dev)
            C = torch.empty(M, N, device=dev)
            for _ in range(3):
                _gemm(ta, tb, M, N, K, _p(A), A.shape[1], _p(Bm), Bm.shape[1], _p(C), N)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                _gemm(ta, tb, M, N, K, _p(A), A.shape[1], _p(Bm), Bm.shape[1], _p(C), N)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            print("gemm ta=%d tb=%d M=%5d N=%5d K=%5d  %7.1f us  %6.1f TFLOP/s" % (ta, tb, M, N, K, 1e3 * ms, 2.0 * M * N * K / ms / 1e9), flush=True)
        return
    if args.lstm_only:
        # one layer's recurrences alone (forward and backward pass), us per time step: the BASELINE minibatch shape
        from pykaldi2_amd import _lib
        L = _lib.lib()
        for (T, B) in ((589, 4), (1166, 4), (200, 4)):
            H, D = 512, 2
            gx = torch.randn(T, B, D * 4 * H, device=dev) * 0.1
            whh = torch.randn(D, 4 * H, H, device=dev) * 0.04
            y = torch.empty(T, B, D * H, device=dev); gates = torch.empty(D, T, B, 4 * H, device=dev)
            cells = torch.empty(D, T, B, H, device=dev)
            dy = torch.randn(T, B, D * H, device=dev) * 0.1
            dgx = torch.empty(T, B, D * 4 * H, device=dev)
            scratch = torch.empty(max(1, L.pk2_lstm_bwd_scratch_floats(B, H, D)), device=dev)
            def fwd():
                _lib.check(L.pk2_lstm_layer_fwd(_lib.ptr(gx), _lib.ptr(whh), None, B, T, H, D, _lib.ptr(y), _lib.ptr(gates),
                                                _lib.ptr(cells), None, _lib.stream_ptr()))
            def bwd():
                _lib.check(L.pk2_lstm_layer_bwd(_lib.ptr(dy), _lib.ptr(whh), _lib.ptr(gates), _lib.ptr(cells), B, T, H, D,
                                                _lib.ptr(dgx), _lib.ptr(scratch), _lib.stream_ptr()))
            res = {}
            for name, fn in (("fwd", fwd), ("bwd", bwd)):
                for _ in range(2):
                    fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    fn()
                e1.record(); torch.cuda.synchronize()
                res[name] = 1e3 * e0.elapsed_time(e1) / 5 / T
            print("lstm recurrences T=%d B=%d: fwd %.3f us/step, bwd %.3f us/step  (y %.4f dgx %.4f)" % (
                T, B, res["fwd"], res["bwd"], y.abs().mean().item(), dgx.abs().mean().item()), flush=True)
        return
    log("rank %d/%d: building synthetic den graph" % (rank, world))
    g = den_graph_arrays()
    den = chain.DenominatorGraph(g, P)
    log("den graph ready")
    if args.den_only:
        print(json.dumps(den_roofline(den, dev)), flush=True)
        return
    rng = np.random.default_rng(1234 + rank)
    #  the number of distinct minibatches divides --steps and the timed step i uses minibatch i % n_unique:
    # the timed window holds every minibatch steps / n_unique times, whatever --warmup is (8 minibatches over 20 steps read
    # 4245 iRTF at W = 5 and 4330 at W = 3 for the same code)
    n_unique = max(d for d in range(1, 11) if args.steps % d == 0)
    # default: SURVEY 8(d) protocol, everything seeded per rank (ranks draw different utterance lengths, so a step
    # waits for the rank with the longest minibatch); --length-bucketed gives every rank the same lengths
    batches = make_batches(rng, n_unique, args.batch, dev,
                           np.random.default_rng(1234) if args.length_bucketed else None)
    log("%d minibatches ready" % n_unique)
    tr = Trainer(dev, den, arch="transformer" if args.transformer else "blstm")
    if args.transformer:
        os.environ["PK2_BENCH_ARCH"] = "transformer"      # the cpu-baseline child runs the same architecture

    # world > 1: hvd times its two gradient-exchange schedules against each other during the first steps of a job (with a
    # host synchronisation per step).  That calibration is start-up work, like building the graphs: it is run to its end
    # here, before the W warm-up steps, so that neither it nor its host synchronisations fall into the timed region.
    calibration_steps = 0
    while getattr(tr.opt, "_trial", None) is not None and calibration_steps < 64:
        tr.step(batches[calibration_steps % n_unique])
        calibration_steps += 1
    # One priming step, start-up work as well (hygiene 8e): the first launch of a persistent kernel in a process
    # verifies the device and takes ~15 ms; it must not depend on W >= 1 to stay out of the timed region.
    tr.step(batches[n_unique - 1])
    for i in range(args.warmup):
        tr.step(batches[(n_unique - 1 - i) % n_unique])
    torch.cuda.synchronize()
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
    # device events around every timed step and around every all-reduce call of it (two event records per step and per
    # call; read after the timed region): what the N > 1 line reports per rank
    tr.opt.timing = []
    step_events = []
    allocs0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
    t0 = time.perf_counter()
    audio = 0.0
    for i in range(args.steps):
        mb = batches[i % n_unique]
        if i + 1 < args.steps:          # the next step's supervisions: built by the worker thread during this step
            tr.prefetch(batches[(i + 1) % n_unique])
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
        tr.step(mb)
        e1 = torch.cuda.Event(enable_timing=True); e1.record()
        step_events.append((e0, e1))
        audio += mb["seconds"]
    torch.cuda.synchronize()
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    device_allocs = int(torch.cuda.memory_stats(dev).get("num_device_alloc", 0) - allocs0)      # hipMalloc calls inside the timed region
    log("timed region done: %.3f s (%d device allocations inside it)" % (dt, device_allocs))
    exchange_timing, tr.opt.timing = tr.opt.timing, None
    scaling = scaling_report(step_events, exchange_timing, audio, dev, rank, world)
    irtf_bucketed = None
    if world > 1 and not args.length_bucketed:
        # the same job on minibatches whose utterance lengths are bucketed across the ranks (bin/train_chain.py
        # -length_bucketed): a second, short timed window, so that one line shows what the stragglers of the default
        # protocol (everything seeded per rank) cost
        bb = make_batches(np.random.default_rng(4321 + rank), n_unique, args.batch, dev, np.random.default_rng(4321))
        k2 = max(2, min(args.steps, 10))
        for i in range(2):
            tr.step(bb[i % n_unique])
        torch.cuda.synchronize(); torch.distributed.barrier()
        tb = time.perf_counter()
        audio_b = 0.0
        for i in range(k2):
            tr.step(bb[(2 + i) % n_unique])
            audio_b += bb[(2 + i) % n_unique]["seconds"]
        torch.cuda.synchronize(); torch.distributed.barrier()
        sb = torch.tensor([time.perf_counter() - tb, audio_b], dtype=torch.float64, device=dev)
        tb_max = sb[0:1].clone(); torch.distributed.all_reduce(tb_max, op=torch.distributed.ReduceOp.MAX)
        ab_sum = sb[1:2].clone(); torch.distributed.all_reduce(ab_sum, op=torch.distributed.ReduceOp.SUM)
        irtf_bucketed = {"value": round(float(ab_sum.item()) / float(tb_max.item()), 2), "steps": k2,
                         "protocol": "utterance lengths of a step drawn from one stream shared by the ranks"}
        del bb
    stats = torch.tensor([dt, audio], dtype=torch.float64, device=dev)
    if torch.distributed.is_initialized():
        tmax = stats[0:1].clone()
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        asum = stats[1:2].clone()
        torch.distributed.all_reduce(asum, op=torch.distributed.ReduceOp.SUM)
        dt, audio = float(tmax.item()), float(asum.item())
    loss_val = float(tr.last["loss"].item())

    # The same timed window once more with the GEMMs on the f32 MFMA ("carries the pure-f32 number beside it"):
    # same minibatches in the same order, same process; not part of `value`.
    f32_window = None
    if world == 1 and gemm_arith_name() == "bf16x3" and not args.den_states and not args.den_arcs:
        from pykaldi2_amd import _lib
        _lib.check(_lib.lib().pk2_gemm_set_arith(0))
        try:
            tr.step(batches[n_unique - 1])
            torch.cuda.synchronize()
            tf0 = time.perf_counter()
            for i in range(args.steps):
                if i + 1 < args.steps:
                    tr.prefetch(batches[(i + 1) % n_unique])
                tr.step(batches[i % n_unique])
            torch.cuda.synchronize()
            dtf = time.perf_counter() - tf0
            f32_window = dict(arith="f32 (v_mfma_f32_32x32x2_f32)", ms_per_step=round(1e3 * dtf / args.steps, 3), value=round(audio / dtf, 2),
                              note="the timed window repeated in the same process with PK2_GEMM_ARITH=f32 semantics (pk2_gemm_set_arith(0))")
        finally:
            _lib.check(_lib.lib().pk2_gemm_set_arith(1))

    # per-phase breakdown of one extra (untimed) step -- on EVERY rank: the step contains the gradient all-reduce
    # The step whose phases are read follows two untimed ones WITHOUT a synchronisation in between, so that its launches are
    # queued behind work the device still has, as in the timed loop: right behind a synchronisation the phases would
    # hold the host's enqueue latency as well (400 launches of the TransformerAM step take the host 11.6 ms against the
    # 12.7 ms the device needs for them; the `fbank` phase of either model read 0.4-3.9 ms for 0.1 ms of kernels).
    events = []
    mb = batches[0]
    for _ in range(2):
        tr.step(mb)
    tr.step(mb, events=events)
    torch.cuda.synchronize()
    if rank != 0:
        hvd.shutdown()
        return
    breakdown = {events[i][0]: round(events[i - 1][1].elapsed_time(events[i][1]), 3) for i in range(1, len(events))}
    lens = [s.frames_per_sequence for s in tr.last["sups"]]
    roof = den_roofline(den, dev)
    log("roofline done")
    # whole-model MFMA utilisation of the breakdown step: all BLSTM + output-layer FLOPs (forward + backward) of the real
    # (unpadded) frames over the time of the two model phases, against the f32 MFMA peak
    lstm_ms = (breakdown["model_fwd"] + breakdown["model_bwd"]) if args.transformer else (breakdown["lstm_fwd"] + breakdown["lstm_bwd"])
    if args.transformer:      # the model phases hold the 12-layer TransformerAM: its own FLOP count and label
        flops = 3.0 * sum(transformer_flops_per_frame(T) * T for T in lens)
        model_tf = flops / (lstm_ms * 1e-3) / 1e12
        roof_lstm = dict(bound="mfma", achieved=round(model_tf, 2), peak=157.3, unit="TFLOP/s", frac=round(model_tf / 157.3, 4),
                         kernel="12-layer TransformerAM (dim 512, 8 heads, FFN 2048, conv k=3) + output layer, forward + backward "
                         "of the breakdown minibatch (GEMMs in the arithmetic `gemm_arith`, fused f32-MFMA attention, row kernels); "
                         "FLOPs are f32-equivalent, peak = the f32 MFMA peak", gemm_arith=gemm_arith_name(), frames=int(sum(lens)),
                         padded_rows=len(lens) * max(lens), ms=round(lstm_ms, 3), flops_fwd_bwd=flops)
    else:
        lstm_tf = 3.0 * lstm_flops_per_frame() * sum(lens) / (lstm_ms * 1e-3) / 1e12
        roof_lstm = dict(bound="mfma", achieved=round(lstm_tf, 2), peak=157.3, unit="TFLOP/s", frac=round(lstm_tf / 157.3, 4),
                         kernel="3x512 BLSTM + output layer, forward + backward of the breakdown minibatch (recurrence "
                         "step kernels on the f32 MFMA + GEMMs in the arithmetic `gemm_arith`); FLOPs are f32-equivalent (2 per "
                         "multiply-add of the model), peak = the f32 MFMA peak", gemm_arith=gemm_arith_name(),
                         frames=int(sum(lens)), padded_rows=len(lens) * max(lens),
                         ms=round(lstm_ms, 3), flops_per_frame_fwd_bwd=3 * lstm_flops_per_frame(),
                         input_projection_gemm=gemm_mfma_roofline(dev, len(lens) * max(lens)))
    result = {
        "metric": ("iRTF (hrs audio/hr) 12-layer TransformerAM LF-MMI (secondary workload, configs[4])" if args.transformer else
                   "iRTF (hrs audio/hr) 3x512 BLSTM LF-MMI, LibriSpeech-shaped synthetic utterances"),
        "value": round(audio / dt, 2), "unit": "hours of audio per wall-clock hour",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": dtype_string(), "data": "synthetic (seeded LibriSpeech-shaped waveforms, "
        "transition-id alignments over a synthetic left-biphone tree, 30k-state/1M-arc denominator graph (%s); "
        "random-init 3x512 BLSTM; supervisions built from the alignments inside the step%s)"
        % ("Kaldi chain topology: self-loop pdf != entering pdf" if DEN_TOPOLOGY == "chain" else "one pdf per destination state",
           "; utterance lengths bucketed across ranks" if args.length_bucketed else ""),
        "config": {"workload": (("SECONDARY configs[4]: 12-layer TransformerAM (dim 512, 8 heads, FFN 2048) LF-MMI "
                                 "(train_transformer_se.py's model under train_chain.py's loss)" if args.transformer else
                                 "configs[2]: 3x512 BLSTM LF-MMI (train_chain.py)") +
                                ", batch %d x var-len per GPU, P=6048, subsample 3, leaky 1e-4, xent_regularize 0.1, "
                                "Adam(amsgrad)+Noam+clip 5" % args.batch),
                   "global_batch": args.batch * world, "parallelism": "dp%d" % world,
                   "den_graph": {"states": S_DEN, "arcs": int(den.num_arcs()), "pdfs": P, "topology": DEN_TOPOLOGY},
                   "allocator_reserve_gb": args.reserve_gb,
                   "hbm_peak_reserved_gb": round(torch.cuda.max_memory_reserved() / 2 ** 30, 2)},
        "exchange": dict({"library": hvd.comm_library() or ("torch.distributed/" + (torch.distributed.get_backend()
                          if torch.distributed.is_initialized() else "none")),
                          "ranks": hvd.comm_ranks(),
                          "schedule": getattr(tr.opt, "_mode", "single"), "calibration_steps": calibration_steps,
                          "api": "pk2_allreduce_bucket" if hvd.comm_library() else "torch.distributed.all_reduce"},
                         **scaling["exchange"]),
        "per_rank": scaling["per_rank"], "ideal_weak_irtf": scaling["ideal_weak_irtf"], "irtf_bucketed": irtf_bucketed,
        "roofline": roof, ("roofline_model" if args.transformer else "roofline_lstm"): roof_lstm, "last_objf_per_frame": round(loss_val / sum(lens), 4),
        "persistent_health": persistent_health(den, args.batch),
    }
    if world == 1 and not args.no_cpu_baseline:
        log("cpu baseline + parity (child process, %d cores)" % usable_cores())
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "parity.npz")
            dump_chain_parity_inputs(tr, mb, path)
            result["cpu_baseline"] = cpu_baseline(1234, parity_path=path)
        result["parity"] = result["cpu_baseline"].pop("parity", None)
    else:
        result["cpu_baseline"] = None
        result["parity"] = None
    # the other BASELINE configurations, as short child jobs behind the headline (one GPU, default workload only)
    if world == 1 and not (args.no_secondary or args.no_cpu_baseline or args.transformer or args.den_states or args.den_arcs
                           or args.den_topology or args.batch != 4 or "RANK" in os.environ):
        del tr, batches
        torch.cuda.empty_cache()
        result["secondary"] = run_secondaries()
    # the tail of the line (the driver keeps the last 2000 characters): the numbers a reader needs first
    result["n_unique_minibatches"] = n_unique
    result["device_allocs_in_timed_region"] = device_allocs
    result["gemm_arith"] = gemm_arith_name()
    result["f32_gemm_window"] = f32_window
    result[("roofline_model" if args.transformer else "roofline_lstm") + "_frac"] = roof_lstm["frac"]
    result["roofline_frac"] = roof.get("frac")
    result["breakdown_ms"] = breakdown
    if "secondary" in result:
        def short(d):
            if not d or "error" in d:
                return None
            cb, par = d.get("cpu_baseline") or {}, d.get("parity") or {}
            return [d.get("ms_per_step"), d.get("value"), (d.get("roofline") or {}).get("frac"), cb.get("value"), par.get("ok")]
        result["secondary_summary"] = dict({k: short(v) for k, v in result["secondary"].items()},
                                           fields="[ms_per_step, iRTF, roofline_frac, cpu_iRTF, parity_ok]")
    print(json.dumps(result), flush=True)
    hvd.shutdown()


if __name__ == "__main__":
    main()
