#!/usr/bin/env python
"""LF-MMI ("chain") training on MI355X -- command line of the reference's bin/train_chain.py
(same flags, same YAML schema, same checkpoint format chain.model.{epoch}.tar = {'model','optimizer',
'epoch'}), running on libpk2hip.so.

  python -m torch.distributed.run --nproc-per-node 8 bin/train_chain.py -config configs/mmi.yaml \
      -data configs/data.yaml -exp_dir exp/chain -chain_dir exp/chain_tree -lr 1e-3 -batch_size 4

No Kaldi here: <chain_dir>/den.fst, <chain_dir>/0.trans_mdl, <chain_dir>/tree and
<ali_dir>/final.mdl are read by the library's own readers, and the per-utterance supervision is built from the
transition-id alignment (`label`) by the library's SplitToPhones / AlignmentToProtoSupervision /
ProtoSupervisionToSupervision (pykaldi2_amd.chain, csrc/chain_sup.hip), with the reference's options
(frame_subsampling_factor 3, tolerances 5, convert_to_pdfs).  -synthetic trains on the seeded
LibriSpeech-shaped generator with a synthetic denominator graph, tree and transition model.
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
from pykaldi2_amd import chain, data, fbank, hvd, lstm, ops, optim, synth, utils  # noqa: E402
from pykaldi2_amd.lattice import TransitionModel  # noqa: E402
from pykaldi2_amd.tree import ContextDependency  # noqa: E402


def parse_config(argv=None):
    """Command line + YAML merge of the reference script (same flags, same keys; keys this framework does not use pass through
    untouched): returns (args, config).  tests/test_host_logic.py pushes the reference's own configs/*.yaml through it."""
    parser = argparse.ArgumentParser()
    parser.add_argument("-config")
    parser.add_argument("-data", help="data yaml file")
    parser.add_argument("-dataPath", default='', type=str, help="path of data files")
    parser.add_argument("-seed_model", default='', help="the seed nerual network model")
    parser.add_argument("-exp_dir", help="the directory to save the outputs")
    parser.add_argument("-transform", help="feature transformation matrix or mvn statistics")
    parser.add_argument("-ali_dir", help="the directory to load trans_model and tree used for alignments")
    parser.add_argument("-lang_dir", help="the lexicon directory to load L.fst")
    parser.add_argument("-chain_dir", help="the directory to load trans_model, tree and den.fst for chain model")
    parser.add_argument("-lr", type=float, default=1e-3, help="set the base learning rate")
    parser.add_argument("-warmup_steps", default=4000, type=int, help="the number of warmup steps to adjust the learning rate")
    parser.add_argument("-xent_regularize", default=0, type=float, help="cross-entropy regularization weight")
    parser.add_argument("-momentum", default=0, type=float, help="set the momentum")
    parser.add_argument("-weight_decay", default=1e-4, type=float, help="set the L2 regularization weight")
    parser.add_argument("-batch_size", default=32, type=int, help="Override the batch size in the config")
    parser.add_argument("-data_loader_threads", default=0, type=int, help="number of workers for data loading")
    parser.add_argument("-max_grad_norm", default=5, type=float, help="max_grad_norm for gradient clipping")
    parser.add_argument("-sweep_size", default=100, type=float, help="process n hours of data per sweep (default:100)")
    parser.add_argument("-length_bucketed", action="store_true", help="data parallel: the ranks of one step get utterances "
                        "of similar length (a step waits at the gradient all-reduce for its longest minibatch)")
    parser.add_argument("-num_epochs", default=1, type=int, help="number of training epochs (default:1)")
    parser.add_argument("-anneal_lr_epoch", default=2, type=int, help="start to anneal the learning rate from this epoch")
    parser.add_argument("-anneal_lr_ratio", default=0.5, type=float, help="the ratio to anneal the learning rate ratio")
    parser.add_argument('-print_freq', default=10, type=int, metavar='N', help='print frequency (default: 10)')
    parser.add_argument('-save_freq', default=1000, type=int, metavar='N', help='save model frequency (default: 1000)')
    parser.add_argument('-synthetic', action='store_true', help='seeded synthetic utterances and denominator graph')
    parser.add_argument('-den_states', default=30000, type=int, help='(synthetic) denominator graph states')
    parser.add_argument('-den_arcs', default=1000000, type=int, help='(synthetic) denominator graph arcs')
    args = parser.parse_args(argv)

    with open(args.config) as f:
        config = yaml.safe_load(f)
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
    config["data_path"] = args.dataPath
    print("pytorch version:{}".format(th.__version__))
    print("Experiment starts with config {}".format(json.dumps(config, sort_keys=True, indent=4)))
    return args, config


def main():
    args, config = parse_config()

    hvd.init()
    th.cuda.set_device(hvd.local_rank())
    dev = th.device("cuda", hvd.local_rank())
    print("Run experiments with world size {}".format(hvd.size()))
    if args.exp_dir and not os.path.isdir(args.exp_dir):
        os.makedirs(args.exp_dir, exist_ok=True)

    mc = config["model_config"]
    P = mc["label_size"]
    model = lstm.LSTMAM(mc["feat_dim"], P, mc["hidden_size"], mc["num_layers"], mc["dropout"], True).to(dev)
    if args.seed_model:   # weights only; a "module." prefix is stripped (reference bin/train_chain.py:147-160)
        sd = th.load(args.seed_model, map_location="cpu")["model"]
        model.load_state_dict({(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()})
    optimizer = optim.Adam(model, lr=args.lr, amsgrad=True)
    hvd.broadcast_parameters(model.state_dict(), root_rank=0)
    hvd.broadcast_optimizer_state(optimizer, root_rank=0)
    optimizer = hvd.DistributedOptimizer(optimizer, named_parameters=model.named_parameters())

    if args.synthetic:
        den = chain.DenominatorGraph(synth.den_graph_arcs(args.den_states, args.den_arcs, P, seed=0, loop_pdf_differs=True), P)
        chain_tree, chain_trans_model = synth.chain_model(P, seed=0)
        aligner = chain.MappedAligner(chain_trans_model)
    else:
        den_path = os.path.join(args.chain_dir or "", "den.fst")
        chain_model_path = os.path.join(args.chain_dir or "", "0.trans_mdl")
        for what, path in (("chain denominator graph", den_path), ("trans_model", chain_model_path),
                           ("chain tree", os.path.join(args.chain_dir or "", "tree")),
                           ("alignment model", os.path.join(args.ali_dir or "", "final.mdl"))):
            if not os.path.isfile(path):     # reference bin/train_chain.py:171-177: message, exit status 0
                sys.stderr.write('ERROR: The {} {} does not exist!\n'.format(what, path))
                sys.exit(0)
        den = chain.DenominatorGraph(den_path, P)
        chain_trans_model = TransitionModel.read(chain_model_path)
        chain_tree = ContextDependency.read(os.path.join(args.chain_dir, "tree"))
        aligner = chain.MappedAligner.from_files(os.path.join(args.ali_dir, "final.mdl"), os.path.join(args.ali_dir, "tree"),
                                                 os.path.join(args.lang_dir or "", "L.fst"), None,
                                                 os.path.join(args.lang_dir or "", "phones/disambig.int"))
    supervision_opts = chain.SupervisionOptions()
    chain_opts = chain.ChainTrainingOptions(leaky_hmm_coefficient=1e-4, xent_regularize=args.xent_regularize)
    source = data.make_source(config, P, hvd.rank(), hvd.size(), ali_model=aligner.transition_model)
    transform = None
    if args.transform is not None and os.path.isfile(args.transform):
        transform = fbank.GlobalMeanVarianceNormalization.load(args.transform)
    sup_model = (aligner, chain_tree, chain_trans_model)
    fb = fbank.FbankExtractor()

    model.train()
    for epoch in range(args.num_epochs):
        run_train_epoch(model, optimizer, source, fb, epoch, supervision_opts, den, chain_opts, args, dev, sup_model, transform)
        hvd.finish()      # collective: a persistent-kernel time-out of the last steps stops every rank before the checkpoint
        if hvd.rank() == 0 and args.exp_dir:
            th.save({'model': model.state_dict(), 'optimizer': optimizer.state_dict(), 'epoch': epoch},
                    args.exp_dir + '/chain.model.' + str(epoch) + '.tar')
    hvd.shutdown()


def run_train_epoch(model, optimizer, source, fb, epoch, supervision_opts, den, chain_opts, args, dev, sup_model, transform):
    batch_time = utils.AverageMeter('Time', ':6.3f')
    losses = utils.AverageMeter('Loss', ':.4e')
    grad_norm = utils.AverageMeter('grad_norm', ':.4e')
    n_batches = max(1, int(args.sweep_size * 3600 / (12.3 * args.batch_size)))
    progress = utils.ProgressMeter(n_batches, batch_time, losses, grad_norm, prefix="Epoch: [{}]".format(epoch))
    sub = supervision_opts.frame_subsampling_factor
    end = time.time()
    for i, batch in enumerate(data.sequence_batches(source, args.batch_size, args.sweep_size, dev, epoch=epoch,
                                                         length_bucketed=args.length_bucketed)):
        frame_shift = (epoch % sub) * -1
        feats, frames, row_off = fb(batch["wav"], batch["lens"])
        if transform is not None:     # dataset.transform of the reference (bin/train_chain.py:118-122)
            feats = transform(feats)
        x = fb.pad_roll_subsample(feats, row_off, frames, shift=frame_shift, subsample=sub, time_major=True)
        aligner, tree, trans_model = sup_model
        sups = [chain.supervision_from_alignment(aligner, tree, trans_model, supervision_opts, np.asarray(y)[:T])
                for y, T in zip(batch["y"], frames)]    # reference bin/train_chain.py:262-272
        prediction = model.forward_time_major(x).transpose(0, 1)
        loss = ops.ChainObjtiveBatch.apply(prediction, den, sups, chain_opts)
        optimizer.zero_grad()
        loss.backward()
        step = n_batches * epoch + i + 1
        lr = utils.noam_decay(step, args.warmup_steps, args.lr)
        for param_group in optimizer.param_groups:
            param_group['lr'] = lr
        norm = optim.clip_grad_norm_(optimizer, args.max_grad_norm)
        optimizer.step()
        if i % args.print_freq == 0:   # .item() synchronises: only when printing
            grad_norm.update(norm.item())
            losses.update(loss.item() / float(np.sum(frames)))
            batch_time.update(time.time() - end)
            if hvd.rank() == 0:
                progress.print(i)
        end = time.time()
        if hvd.rank() == 0 and args.exp_dir and i > 0 and i % args.save_freq == 0:
            th.save({'model': model.state_dict(), 'optimizer': optimizer.state_dict()},
                    args.exp_dir + '/chain.model.' + str(i) + '.tar')


if __name__ == '__main__':
    main()
