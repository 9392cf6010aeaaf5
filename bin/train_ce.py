#!/usr/bin/env python
"""Cross-entropy training on MI355X -- command line of the reference's bin/train_ce.py (same flags, YAML
schema and checkpoint format model.{epoch}.tar), running on libpk2hip.so: on-device fbank + CMN,
80-frame chunks through a shuffle pool (reference DataBuffer, data/sr_dataset.py:55-84), 3x512 BLSTM,
fused softmax-CE, clip + Adam(amsgrad).

  python bin/train_ce.py -train_config configs/ce.yaml -data_config configs/data.yaml -exp_dir exp/ce \
      -lr 1e-4 -batch_size 64 [-hvd True] [-synthetic]
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
from pykaldi2_amd import _lib, data, fbank, hvd, lstm, ops, optim, utils  # noqa: E402


class ChunkPool:
    """Device-side shuffle buffer: whole utterances are cut into seg_len-frame chunks (tail dropped),
    pooled, and minibatches are drawn at random once the pool holds `capacity` chunks."""

    def __init__(self, capacity, seed):
        self.x, self.y, self.n = [], [], 0
        self.capacity = capacity
        self.gen = th.Generator().manual_seed(seed)

    def add(self, x, y):
        if x.shape[0]:
            self.x.append(x); self.y.append(y); self.n += x.shape[0]

    def batches(self, batch_size, flush=False):
        while self.n >= (batch_size if flush else self.capacity):
            X, Y = th.cat(self.x), th.cat(self.y)
            perm = th.randperm(X.shape[0], generator=self.gen).to(X.device)
            take = perm[:batch_size * max(1, (X.shape[0] // 2) // batch_size)]
            keep = perm[take.shape[0]:]
            for b in range(0, take.shape[0], batch_size):
                idx = take[b:b + batch_size]
                yield X[idx], Y[idx]
            self.x, self.y, self.n = [X[keep]], [Y[keep]], int(keep.shape[0])


def parse_config(argv=None):
    """Command line + YAML merge of the reference script (same flags, same keys; keys this framework does not use pass through
    untouched): returns (args, config).  tests/test_host_logic.py pushes the reference's own configs/*.yaml through it."""
    parser = argparse.ArgumentParser()
    parser.add_argument("-exp_dir")
    parser.add_argument("-dataPath", default='', type=str, help="path of data files")
    parser.add_argument("-train_config")
    parser.add_argument("-data_config")
    parser.add_argument("-lr", default=0.0001, type=float, help="Override the LR in the config")
    parser.add_argument("-batch_size", default=32, type=int, help="Override the batch size in the config")
    parser.add_argument("-data_loader_threads", default=0, type=int, help="number of workers for data loading")
    parser.add_argument("-max_grad_norm", default=5, type=float, help="max_grad_norm for gradient clipping")
    parser.add_argument("-sweep_size", default=200, type=float, help="process n hours of data per sweep (default:200)")
    parser.add_argument("-num_epochs", default=1, type=int, help="number of training epochs (default:1)")
    parser.add_argument("-global_mvn", default=False, type=bool, help="if apply global mean and variance normalization")
    parser.add_argument("-mvn_utterances", default=2000, type=int, help="utterances used to estimate the global mean and variance")
    parser.add_argument("-resume_from_model", type=str, help="the model from which you want to resume training")
    parser.add_argument("-dropout", type=float, help="set the dropout ratio")
    parser.add_argument("-anneal_lr_epoch", default=2, type=int, help="start to anneal the learning rate from this epoch")
    parser.add_argument("-anneal_lr_ratio", default=0.5, type=float, help="the ratio to anneal the learning rate")
    parser.add_argument('-print_freq', default=100, type=int, metavar='N', help='print frequency (default: 100)')
    parser.add_argument('-hvd', default=False, type=bool, help="whether to use horovod for training")
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
    dropout = args.dropout if args.dropout is not None else mc["dropout"]
    model = lstm.LSTMAM(mc["feat_dim"], mc["label_size"], mc["hidden_size"], mc["num_layers"], dropout, True).to(dev)
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
        if epoch > args.anneal_lr_epoch:
            for param_group in optimizer.param_groups:
                param_group['lr'] *= args.anneal_lr_ratio
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
    dc = config["data_config"]
    seg_len, seg_shift = dc["seg_len"], dc["seg_shift"]
    n_batches = max(1, int(args.sweep_size * 3600 / (seg_len * 0.01 * args.batch_size)))
    progress = utils.ProgressMeter(n_batches, batch_time, losses, grad_norm, prefix="Epoch: [{}]".format(epoch))
    pool = ChunkPool(capacity=max(4 * args.batch_size, 2048), seed=epoch + 17 * hvd.rank())
    end = time.time()
    i = 0

    def train_on(x, y):
        nonlocal i, end
        prediction = model(x)
        loss = criterion(prediction.view(-1, prediction.shape[2]), y.reshape(-1))
        optimizer.zero_grad()
        loss.backward()
        norm = optim.clip_grad_norm_(optimizer, args.max_grad_norm)
        optimizer.step()
        if i % args.print_freq == 0:
            grad_norm.update(norm.item()); losses.update(loss.item(), x.size(0)); batch_time.update(time.time() - end)
            if not args.hvd or hvd.rank() == 0:
                progress.print(i)
        end = time.time()
        i += 1

    # Every rank runs exactly n_batches steps (one gradient all-reduce each): utterances are drawn until the chunk pool
    # has produced them, whatever lengths this rank happened to get (the reference's chunk mode has a fixed number of
    # minibatches per sweep too, data/sr_dataset.py:226-232).
    def utterance_stream():
        k = 0
        while True:
            yield from data.sequence_batches(source, 8, args.sweep_size, dev, epoch=epoch * 1000 + k)
            k += 1
    stream = utterance_stream()
    while i < n_batches:
        batch = next(stream)
        feats, frames, row_off = fb(batch["wav"], batch["lens"], apply_cmn=dc.get("use_cmn", True))
        if transform is not None:
            feats = transform(feats)
        off = np.concatenate([[0], np.cumsum(frames)])
        for n, y in enumerate(batch["y"]):
            f = feats[off[n]:off[n + 1]]
            lab = _lib.h2d(np.ascontiguousarray(np.asarray(y)[:frames[n]]), dev).unsqueeze(1)
            pool.add(fbank.utt2seg(f, seg_len, seg_shift), fbank.utt2seg(lab, seg_len, seg_shift))
        for x, y in pool.batches(args.batch_size):
            if i >= n_batches:
                break
            train_on(x, y)


if __name__ == '__main__':
    main()
