#!/usr/bin/env python
"""Sequence-discriminative training (lattice MMI / sMBR / MPFE with on-the-fly lattices) on MI355X -- command
line of the reference's bin/train_se.py (same flags, same YAML schema, same checkpoint format
model.se.{epoch}.tar = {'model','optimizer','epoch'}), running on libpk2hip.so.

  python -m torch.distributed.run --nproc-per-node 8 bin/train_se.py -config configs/se.yaml -data configs/data.yaml \
      -exp_dir exp/se -criterion mmi -seed_model exp/ce/model.ce.0.tar -trans_model exp/tri/final.mdl.txt \
      -prior_path exp/tri/final.occs -den_dir exp/tri/graph -lr 1e-5 -batch_size 8

Lattices are generated, pruned and consumed on the device for the whole minibatch (pykaldi2_amd.lattice); the
reference decodes each utterance on the CPU with Kaldi (ops/ops.py:55).  Differences by necessity (no Kaldi
here): `-trans_model` is read in Kaldi's binary or text form by the library's own reader;
`words.txt` is not needed (no criterion uses word labels); -synthetic trains on the seeded generators
(LibriSpeech-shaped utterances, a word-loop HCLG, a 3-state monophone transition model, uniform priors).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch as th
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pykaldi2_amd import data, fbank, hvd, lattice, lstm, ops, optim, se, synth, transformer, utils  # noqa: E402


def parse_config(argv=None, arch="blstm"):
    """arch = "transformer": the command line of the reference's bin/train_transformer_se.py (TransformerAM built
    from -dim_model / -nheads / -ff_size / -nlayers / -dropout, -look_ahead attention mask, -dataPath)."""
    parser = argparse.ArgumentParser()
    parser.add_argument("-config")
    parser.add_argument("-data", help="data yaml file")
    if arch == "transformer":
        parser.add_argument("-dataPath", dest="data_path", default='', type=str, help="path of data files")
        parser.add_argument("-dropout", default=0, type=float, help="set the dropout ratio")
        parser.add_argument("-nheads", default=4, type=int, help="the number of attention heads")
        parser.add_argument("-dim_model", default=512, type=int, help="the model dimension")
        parser.add_argument("-ff_size", default=2048, type=int, help="the size of feed-forward layer")
        parser.add_argument("-nlayers", default=6, type=int, help="the number of layers")
        parser.add_argument("-look_ahead", default=-1, type=int, help="the number of frames to look ahead")
    else:
        parser.add_argument("-data_path", default='', type=str, help="path of data files")
    parser.add_argument("-seed_model", default='', help="the seed nerual network model")
    parser.add_argument("-exp_dir", help="the directory to save the outputs")
    parser.add_argument("-transform", help="feature transformation matrix or mvn statistics")
    parser.add_argument("-criterion", type=str, choices=["mmi", "mpfe", "smbr"], default="mmi",
                        help="set the sequence training crtierion")
    parser.add_argument("-trans_model", help="the HMM transistion model, used for lattice generation")
    parser.add_argument("-prior_path", help="the prior for decoder, usually named as final.occs in kaldi setup")
    parser.add_argument("-den_dir", help="the decoding graph directory to find HCLG and words.txt files")
    parser.add_argument("-lr", type=float, default=1e-5, help="set the learning rate")
    parser.add_argument("-ce_ratio", default=0.1, type=float, help="the ratio for ce regularization")
    parser.add_argument("-momentum", default=0, type=float, help="set the momentum")
    parser.add_argument("-batch_size", default=32, type=int, help="Override the batch size in the config")
    parser.add_argument("-data_loader_threads", default=0, type=int, help="number of workers for data loading")
    parser.add_argument("-max_grad_norm", default=5, type=float, help="max_grad_norm for gradient clipping")
    parser.add_argument("-sweep_size", default=100, type=float, help="process n hours of data per sweep (default:60)")
    parser.add_argument("-length_bucketed", action="store_true", help="data parallel: the ranks of one step get utterances "
                        "of similar length (a step waits at the gradient all-reduce for its longest minibatch)")
    parser.add_argument("-num_epochs", default=1, type=int, help="number of training epochs (default:1)")
    parser.add_argument('-print_freq', default=10, type=int, metavar='N', help='print frequency (default: 10)')
    parser.add_argument('-save_freq', default=1000, type=int, metavar='N', help='save model frequency (default: 1000)')
    parser.add_argument('-synthetic', action='store_true', help='seeded synthetic utterances, HCLG and transition model')
    parser.add_argument('-graph_words', default=2000, type=int, help='(synthetic) vocabulary of the word-loop HCLG')
    args = parser.parse_args(argv)

    with open(args.config) as f:
        config = yaml.safe_load(f)
    config['data_path'] = args.data_path
    config["sweep_size"] = args.sweep_size
    if args.data and not args.synthetic:
        with open(args.data) as f:
            d = yaml.safe_load(f)
            config["source_paths"] = [j for i, j in d['clean_source'].items()]
            if 'dir_noise' in d:      # reference bin/train_ce.py:73-76
                config["dir_noise_paths"] = [j for i, j in d['dir_noise'].items()]
            if 'rir' in d:
                config["rir_paths"] = [j for i, j in d['rir'].items()]
    config["synthetic"] = args.synthetic
    print("pytorch version:{}".format(th.__version__))
    print("Experiment starts with config {}".format(json.dumps(config, sort_keys=True, indent=4)))
    return args, config


def main(arch="blstm"):
    args, config = parse_config(None, arch)

    hvd.init()
    th.cuda.set_device(hvd.local_rank())
    dev = th.device("cuda", hvd.local_rank())
    print("Run experiments with world size {}".format(hvd.size()))
    if args.exp_dir and not os.path.isdir(args.exp_dir):
        os.makedirs(args.exp_dir, exist_ok=True)

    mc = config["model_config"]
    P = mc["label_size"]
    forward = None
    if arch == "transformer":     # reference bin/train_transformer_se.py:128
        model = transformer.TransformerAM(mc["feat_dim"], args.dim_model, args.nheads, args.ff_size, args.nlayers,
                                          args.dropout, P).to(dev)
        forward = lambda m, x, frames: transformer.padded_forward(m, x, frames, args.look_ahead)   # noqa: E731
    else:
        model = lstm.LSTMAM(mc["feat_dim"], P, mc["hidden_size"], mc["num_layers"], mc["dropout"], True).to(dev)
    if args.seed_model:
        if not os.path.isfile(args.seed_model):
            sys.stderr.write('ERROR: The model file %s does not exist!\n' % (args.seed_model))
            sys.exit(0)
        sd = th.load(args.seed_model, map_location="cpu")["model"]
        model.load_state_dict({(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()})
        print("=> loaded checkpoint '{}' ".format(args.seed_model))
    elif not args.synthetic:
        sys.stderr.write('ERROR: The model file %s does not exist!\n' % (args.seed_model))
        sys.exit(0)
    optimizer = optim.SGD(model, lr=args.lr, momentum=args.momentum)
    hvd.broadcast_parameters(model.state_dict(), root_rank=0)
    hvd.broadcast_optimizer_state(optimizer, root_rank=0)
    optimizer = hvd.DistributedOptimizer(optimizer, named_parameters=model.named_parameters())

    dc = config["decoder_config"]
    decoder_opts = lattice.LatticeFasterDecoderOptions()
    decoder_opts.beam = dc["beam"]
    decoder_opts.lattice_beam = dc["lattice_beam"]
    decoder_opts.max_active = dc["max_active"]
    decoder_opts.determinize_lattice = False
    if args.synthetic:
        tm = synth.transition_model_arrays(P)
        trans_model = lattice.TransitionModel.from_arrays(tm)
        silence_ids = tm["silence_phones"]
        asr_decoder = lattice.MappedLatticeFasterRecognizer(trans_model, synth.decoding_graph_arcs(args.graph_words, P, seed=0),
                                                            acoustic_scale=dc["acoustic_scale"], decoder_opts=decoder_opts)
        log_prior = se.log_prior_from_counts(np.ones(P))
    else:
        HCLG = (args.den_dir or "") + "/HCLG.fst"
        silence_phones = (args.den_dir or "") + "/phones/silence.csl"
        for what, path in (("HCLG", HCLG), ("silence phone", silence_phones), ("trans_model", args.trans_model or ""),
                           ("prior", args.prior_path or "")):
            if not os.path.isfile(path):
                sys.stderr.write('ERROR: The %s file %s does not exist!\n' % (what, path))
                sys.exit(0)
        with open(silence_phones) as f:
            silence_ids = [int(i) for i in f.readline().strip().split(':')]
        asr_decoder = lattice.MappedLatticeFasterRecognizer.from_files(args.trans_model, HCLG, None,
                                                                       acoustic_scale=dc["acoustic_scale"],
                                                                       decoder_opts=decoder_opts)
        trans_model = asr_decoder.trans_model
        log_prior = se.log_prior_from_counts(se.read_kaldi_vector(args.prior_path))
    source = data.make_source(config, P, hvd.rank(), hvd.size(), with_tids=True)
    transform = None
    if args.transform is not None and os.path.isfile(args.transform):
        transform = fbank.GlobalMeanVarianceNormalization.load(args.transform)
    fb = fbank.FbankExtractor()

    model.train()
    for epoch in range(args.num_epochs):
        run_train_epoch(model, optimizer, log_prior.to(dev), source, fb, epoch, asr_decoder, trans_model, silence_ids, args, dev,
                        forward, transform)
        hvd.finish()      # collective: a persistent-kernel time-out of the last steps stops every rank before the checkpoint
        if hvd.rank() == 0 and args.exp_dir:
            th.save({'model': model.state_dict(), 'optimizer': optimizer.state_dict(), 'epoch': epoch},
                    args.exp_dir + '/model.se.' + str(epoch) + '.tar')
    hvd.shutdown()


def run_train_epoch(model, optimizer, log_prior, source, fb, epoch, asr_decoder, trans_model, silence_ids, args, dev,
                    forward=None, transform=None):
    batch_time = utils.AverageMeter('Time', ':6.3f')
    losses = utils.AverageMeter('Loss', ':.4e')
    grad_norm = utils.AverageMeter('grad_norm', ':.4e')
    n_batches = max(1, int(args.sweep_size * 3600 / (12.3 * args.batch_size)))
    progress = utils.ProgressMeter(n_batches, batch_time, losses, grad_norm, prefix="Epoch: [{}]".format(epoch))
    ce_criterion = ops.CrossEntropyLoss(ignore_index=-100, reduction='sum')
    end = time.time()
    for i, batch in enumerate(data.sequence_batches(source, args.batch_size, args.sweep_size, dev, epoch=epoch,
                                                         length_bucketed=args.length_bucketed)):
        loss, se_val, ce_loss, frames = se.sequence_loss(model, fb, batch, asr_decoder, trans_model, log_prior, args.criterion,
                                                         silence_ids, args.ce_ratio, ce_criterion, forward, transform)
        optimizer.zero_grad()
        loss.backward()
        norm = optim.clip_grad_norm_(optimizer, args.max_grad_norm)
        optimizer.step()
        if i % args.print_freq == 0:
            grad_norm.update(norm.item())
            losses.update(loss.item() / float(np.sum(frames)))
            batch_time.update(time.time() - end)
            if hvd.rank() == 0:
                progress.print(i)
        end = time.time()
        if hvd.rank() == 0 and args.exp_dir and i > 0 and i % args.save_freq == 0:
            th.save({'model': model.state_dict(), 'optimizer': optimizer.state_dict()},
                    args.exp_dir + '/model.se.' + str(i) + '.tar')


if __name__ == '__main__':
    main()
