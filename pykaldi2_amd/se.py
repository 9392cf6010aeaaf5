"""One sequence-discriminative training step (lattice MMI / sMBR / MPFE + CE regulariser) for a whole
minibatch on the device: the body of the reference's bin/train_se.py:226-262 without its per-utterance Python
loop and host round trips.

    prediction = model(x)                                           bin/train_se.py:231
    ce_loss = CrossEntropyLoss(ignore_index=-100, reduction='sum')  :214,235
    se_loss = sum_j criterion(prediction[j, :num_frs[j]] - log_prior, asr_decoder, trans_model, trans_id_j)  :237-249
    loss = se_loss + ce_ratio * ce_loss                             :251
"""
import numpy as np
import torch

from . import ops


def read_kaldi_vector(path):
    """Kaldi vector / single-row matrix in text (" [ 1 2 3 ]") or binary ("\\0B" + FV/DV/FM/DM) form -- the
    `final.occs` priors file (reference bin/train_se.py:186-187: read_matrix(prior_path)[0])."""
    with open(path, "rb") as f:
        raw = f.read()
    if raw[:2] == b"\0B":
        tag = raw[2:5]
        pos = 5
        dt = np.dtype("<f4") if tag[:1] == b"F" else np.dtype("<f8")

        def rd_int(p):
            assert raw[p] == 4
            return int(np.frombuffer(raw, "<i4", 1, p + 1)[0]), p + 5
        if tag in (b"FV ", b"DV "):
            n, pos = rd_int(pos)
            return np.frombuffer(raw, dt, n, pos).astype(np.float64)
        if tag in (b"FM ", b"DM "):
            r, pos = rd_int(pos)
            c, pos = rd_int(pos)
            return np.frombuffer(raw, dt, r * c, pos).astype(np.float64).reshape(r, c)[0]
        raise ValueError("%s: unsupported Kaldi object %r" % (path, tag))
    toks = raw.decode().replace("[", " ").replace("]", " ").split()
    return np.asarray([float(t) for t in toks], np.float64)


def log_prior_from_counts(counts):
    """log(prior / sum(prior)) as float32 (reference bin/train_se.py:187)."""
    c = np.asarray(counts, np.float64)
    return torch.tensor(np.log(c / c.sum()), dtype=torch.float32)


def sequence_loss(model, fb, batch, asr_decoder, trans_model, log_prior, criterion, silence_ids, ce_ratio, ce_criterion,
                  forward=None, transform=None):
    """Forward of one minibatch: batch = dict(wav, lens, y = pdf alignments, aux = transition-id alignments)
    from pykaldi2_amd.data.  `forward(model, x[Tmax, N, 80], frames) -> [N, Tmax, P]` replaces the BLSTM call
    (TransformerAM with its masks: bin/train_transformer_se.py).  Returns (loss, se_value, ce_loss, frames)."""
    feats, frames, row_off = fb(batch["wav"], batch["lens"])
    if transform is not None:      # the `-transform` MVN statistics (reference bin/train_se.py:104-108)
        feats = transform(feats)
    x = fb.pad_roll_subsample(feats, row_off, frames, shift=0, subsample=1, time_major=True)      # [Tmax, N, 80]
    if forward is not None:
        prediction = forward(model, x, frames)
    else:
        prediction = model.forward_time_major(x).transpose(0, 1)                                   # [N, Tmax, P] view
    N, Tmax = prediction.shape[0], prediction.shape[1]
    y = np.full((N, Tmax), -100, np.int64)
    for n, lab in enumerate(batch["y"]):
        y[n, :frames[n]] = np.asarray(lab)[:frames[n]]
    ce_loss = ce_criterion(prediction, torch.from_numpy(y).to(prediction.device))
    loglikes = prediction - log_prior
    se = ops.LatticeBatchFunction.apply(loglikes, [int(t) for t in frames], asr_decoder, trans_model, batch["aux"],
                                        criterion, silence_ids)
    return se + ce_ratio * ce_loss, se, ce_loss, frames
