#!/usr/bin/env python
"""Cross-entropy training of the TransformerAM on whole utterances on MI355X -- command line of the reference's
bin/train_transformer_ce.py (same flags, YAML schema, checkpoints model.{epoch}.tar = {'model','optimizer',
'epoch'}): SeqDataloader-style zero-padded minibatches, key-padding mask and optional look-ahead mask
(:188-201), CrossEntropyLoss(ignore_index=-100), Adam(amsgrad) with Noam decay (-warmup_step), clip.

  python -m torch.distributed.run --nproc-per-node 8 bin/train_transformer_ce.py -train_config configs/ce.yaml \
      -data_config configs/data.yaml -exp_dir exp/tce -lr 1e-3 -batch_size 16 -nlayers 12 -nheads 8

-synthetic trains on the seeded LibriSpeech-shaped generator.
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
from pykaldi2_amd import _lib, data, fbank, hvd, ops, optim, transformer, utils  # noqa: E402


def parse_config(argv=None):
    """Command line + YAML merge of the reference script (same flags, same keys; keys this framework does not use pass through
    untouched): returns (args, config).  tests/test_host_logic.py pushes the reference's own configs/*.yaml through it."""
    parser = argparse.ArgumentParser()
    parser.add_argument("-train_config")
    parser.add_argument("-data_config")
    parser.add_argument("-dataPath", default='', type=str, help="path of data files")
    parser.add_argument("-exp_dir")
    parser.add_argument("-lr", default=0.0001, type=float, help="Override the LR in the config")
    parser.add_argument("-batch_size", default=32, type=int, help="Override the batch size in the config")
    parser.add_argument("-data_loader_threads", default=0, type=int, help="number of workers for data loading")
    parser.add_argument("-max_grad_norm", default=5, type=float, help="max_grad_norm for gradient clipping")
    parser.add_argument("-sweep_size", default=200, type=float, help="process n hours of data per sweep (default:200)")
    parser.add_argument("-num_epochs", default=1, type=int, help="number of training epochs (default:1)")
    parser.add_argument("-global_mvn", default=False, type=bool, help="if apply global mean and variance normalization")
    parser.add_argument("-mvn_utterances", default=2000, type=int, help="utterances used to estimate the global mean and variance")
    parser.add_argument("-resume_from_model", type=str, help="the model from which you want to resume training")
    parser.add_argument("-dropout", default=0, type=float, help="set the dropout ratio")
    parser.add_argument("-warmup_step", default=4000, type=int, help="the number of warmup steps to adjust the learning rate")
    parser.add_argument("-nheads", default=4, type=int, help="the number of attention heads")
    parser.add_argument("-dim_model", default=512, type=int, help="the model dimension")
    parser.add_argument("-ff_size", default=2048, type=int, help="the size of feed-forward layer")
    parser.add_argument("-nlayers", default=6, type=int, help="the number of layers")
    parser.add_argument("-look_ahead", default=-1, type=int, help="the number of frames to look ahead")
    parser.add_argument('-print_freq', default=100, type=int, metavar='N', help='print frequency (default: 100)')
    parser.add_argument('-hvd', default=True, type=bool, help="whether to use horovod for training")
    parser.add_argument('-synthetic', action='store_true', help='seeded synthetic utterances')
    args = parser.parse_args(argv)

    with open(args.train_config) as f:
        config = yaml.safe_load(f)
    config["sweep_size"] = args.sweep_size
    if args.data_config and not args.synthetic:
        with open(args.data_config) as f:
            d = yaml.safe_load(f)
            config["source_paths"] = [j for i, j in d['clean_source'].items()]
            if 'dir_noise' in d:      # reference bin/train_ce.py:73-76
                config["dir_noise_paths"] = [j for i, j in d['dir_noise'].items()]
            if 'rir' in d:
                config["rir_paths"] = [j for i, j in d['rir'].items()]
    config["synthetic"] = args.synthetic
    config['data_path'] = args.dataPath
    print("Experiment starts with config {}".format(json.dumps(config, sort_keys=True, indent=4)))
    return args, config


def main():
    args, config = parse_config()

    if args.hvd:
        hvd.init()
        print("Run experiments with world size {}".format(hvd.size()))
    dev = th.device("cuda", hvd.local_rank())
    th.cuda.set_device(dev)
    if args.exp_dir and not os.path.isdir(args.exp_dir):
        os.makedirs(args.exp_dir, exist_ok=True)

    mc = config["model_config"]
    model = transformer.TransformerAM(mc["feat_dim"], args.dim_model, args.nheads, args.ff_size, args.nlayers,
                                      args.dropout, mc["label_size"]).to(dev)
    optimizer = optim.Adam(model, lr=args.lr, amsgrad=True)
    start_epoch = 0
    if args.resume_from_model:
        assert os.path.isfile(args.resume_from_model), "ERROR: model file {} does not exit!".format(args.resume_from_model)
        checkpoint = th.load(args.resume_from_model, map_location=dev)
        model.load_state_dict(checkpoint['model'])
        optimizer.load_state_dict(checkpoint['optimizer'])
        start_epoch = checkpoint['epoch']
        print("=> loaded checkpoint '{}' ".format(args.resume_from_model))
    if args.hvd:
        hvd.broadcast_parameters(model.state_dict(), root_rank=0)
        hvd.broadcast_optimizer_state(optimizer, root_rank=0)
        optimizer = hvd.DistributedOptimizer(optimizer, named_parameters=model.named_parameters())
    criterion = ops.CrossEntropyLoss(ignore_index=-100)
    source = data.make_source(config, mc["label_size"], hvd.rank(), hvd.size())
    fb = fbank.FbankExtractor()
    transform = None
    if args.global_mvn:      # reference bin/train_ce.py:110-121
        print("Estimating global mean and variance of feature vectors...")
        transform = fbank.GlobalMeanVarianceNormalization.estimate(source, fb, dev, n_sample_to_use=args.mvn_utterances,
                                                                   apply_cmn=config["data_config"].get("use_cmn", True))
        print("Global mean and variance transform trained successfully!")
        if args.exp_dir and (not args.hvd or hvd.rank() == 0):
            transform.save(args.exp_dir + "/transform.pkl")

    model.train()
    for epoch in range(start_epoch, args.num_epochs):
        run_train_epoch(model, optimizer, criterion, source, fb, epoch, config, args, dev, transform)
        hvd.finish()      # collective: a persistent-kernel time-out of the last steps stops every rank before the checkpoint
        if (not args.hvd or hvd.rank() == 0) and args.exp_dir:
            th.save({'model': model.state_dict(), 'optimizer': optimizer.state_dict(), 'epoch': epoch},
                    args.exp_dir + '/model.' + str(epoch) + '.tar')
    hvd.shutdown()


def run_train_epoch(model, optimizer, criterion, source, fb, epoch, config, args, dev, transform):
    batch_time = utils.AverageMeter('Time', ':6.3f')
    losses = utils.AverageMeter('Loss', ':.4e')
    grad_norm = utils.AverageMeter('grad_norm', ':.4e')
    n_batches = max(1, int(args.sweep_size * 3600 / (12.3 * args.batch_size)))
    progress = utils.ProgressMeter(n_batches, batch_time, losses, grad_norm, prefix="Epoch: [{}]".format(epoch))
    use_cmn = config["data_config"].get("use_cmn", True)
    end = time.time()
    for i, batch in enumerate(data.sequence_batches(source, args.batch_size, args.sweep_size, dev, epoch=epoch)):
        feats, frames, row_off = fb(batch["wav"], batch["lens"], apply_cmn=use_cmn)
        if transform is not None:
            feats = transform(feats)
        x = fb.pad_roll_subsample(feats, row_off, frames, shift=0, subsample=1, time_major=True)     # [Tmax, N, 80]
        prediction = transformer.padded_forward(model, x, frames, args.look_ahead)                    # [N, Tmax, P]
        N, Tmax = prediction.shape[0], prediction.shape[1]
        y = np.full((N, Tmax), -100, np.int64)          # SeqDataloader pads the labels with -100 (data/dataloader.py:94-136)
        for n, lab in enumerate(batch["y"]):
            y[n, :frames[n]] = np.asarray(lab)[:frames[n]]
        loss = criterion(prediction, _lib.h2d(y, dev))
        optimizer.zero_grad()
        loss.backward()
        norm = optim.clip_grad_norm_(optimizer, args.max_grad_norm)
        step = n_batches * epoch + i + 1
        lr = utils.noam_decay(step, args.warmup_step, args.lr)
        for param_group in optimizer.param_groups:
            param_group['lr'] = lr
        optimizer.step()
        if i % args.print_freq == 0:   # .item() synchronises: only when printing
            grad_norm.update(norm.item()); losses.update(loss.item(), N); batch_time.update(time.time() - end)
            if not args.hvd or hvd.rank() == 0:
                progress.print(i)
        end = time.time()


if __name__ == '__main__':
    main()
