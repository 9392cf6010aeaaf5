#!/usr/bin/env python
"""Decodes a set of utterances with the acoustic model and writes their lattices to a Kaldi compact-lattice archive --
command line of the reference's bin/latgen.py (same flags), running on libpk2hip.so: fbank + CMN + BLSTM forward,
log-likelihoods minus log priors, token-passing lattice generation over HCLG and lattice-beam pruning all on the device;
only the finished lattices come back to the host to be written.

  python bin/latgen.py -config configs/se.yaml -model_path exp/se/model.se.0.tar -data_path dev.zip \
      -prior_path exp/tri/final.occs -trans_model exp/tri/final.mdl -graph_dir exp/tri/graph -out_file exp/se/lat.ark

Per utterance it prints what the reference prints (bin/latgen.py:173-178): the best word sequence (`words.txt` symbols
when -graph_dir has one, ids otherwise) and the per-frame log-likelihood of the best path.  The lattices are Kaldi
CompactLattices, DETERMINISED on their word labels like the reference's (`decoder_opts.determinize_lattice = True`,
reference bin/latgen.py:149): every word sequence once, with the costs and the alignment of its best path, pruned with
the lattice beam (pykaldi2_amd.lattice.determinize_lattice: host-side restatement of Kaldi's DeterminizeLatticePruned);
-no_determinize writes the raw lattice-beam-pruned state-level lattice instead (arc = decoder link).  -out_file ending
in .txt writes a text archive ("ark,t:").  -synthetic N decodes N seeded utterances against a synthetic word-loop HCLG.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch as th
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pykaldi2_amd import data, fbank, kaldi_io, lattice, lstm, se, synth  # noqa: E402


def read_words_txt(path):
    table = {}
    with open(path) as f:
        for line in f:
            parts = line.split()
            if len(parts) == 2:
                table[int(parts[1])] = parts[0]
    return table


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("-config")
    parser.add_argument("-model_path")
    parser.add_argument("-data_path")
    parser.add_argument("-batch_size", default=32, type=int, help="Override the batch size in the config")
    parser.add_argument("-prior_path", help="the path to load the final.occs file")
    parser.add_argument("-transform", help="feature transformation matrix or mvn statistics")
    parser.add_argument("-out_file", help="write out the lattice to this file")
    parser.add_argument("-trans_model", help="the HMM transistion model, used for lattice generation")
    parser.add_argument("-graph_dir", help="the decoding graph directory")
    parser.add_argument("-sweep_size", default=200, type=float, help="process n hours of data per sweep (default:60)")
    parser.add_argument("-data_loader_threads", default=4, type=int, help="number of workers for data loading")
    parser.add_argument("-no_determinize", action="store_true", help="write the raw state-level lattices (the reference "
                        "determinises: decoder_opts.determinize_lattice = True)")
    parser.add_argument("-synthetic", type=int, default=0, help="decode this many seeded synthetic utterances instead")
    parser.add_argument("-synthetic_words", type=int, default=500, help="(synthetic) vocabulary of the word-loop HCLG")
    args = parser.parse_args()

    with open(args.config) as f:
        config = yaml.safe_load(f)
    config["sweep_size"] = args.sweep_size
    print("job starts with config {}".format(json.dumps(config, sort_keys=True, indent=4)))
    dev = th.device("cuda", 0)
    mc = config["model_config"]
    P = mc["label_size"]
    model = lstm.LSTMAM(mc["feat_dim"], P, mc["hidden_size"], mc["num_layers"], mc["dropout"], True).to(dev)
    if args.model_path:
        assert os.path.isfile(args.model_path), "ERROR: model file {} does not exit!".format(args.model_path)
        sd = th.load(args.model_path, map_location="cpu")["model"]
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
        sd = {(k[5:] if k.startswith("nnet.") else k): v for k, v in sd.items()}   # NnetAM(LSTMStack) -> LSTMAM keys
        model.load_state_dict(sd)
        print("=> loaded checkpoint '{}' ".format(args.model_path))

    dc = config["decoder_config"]
    decoder_opts = lattice.LatticeFasterDecoderOptions(beam=dc["beam"], lattice_beam=dc["lattice_beam"],
                                                       max_active=dc["max_active"])
    words = {}
    if args.synthetic:
        tm = synth.transition_model_arrays(P)
        trans_model = lattice.TransitionModel.from_arrays(tm)
        graph = lattice.DecodeGraph(synth.decoding_graph_arcs(args.synthetic_words, P, seed=0))
        log_prior = se.log_prior_from_counts(np.ones(P)).to(dev)
    else:
        HCLG = args.graph_dir + "/HCLG.fst"
        words_txt = args.graph_dir + "/words.txt"
        for path, what in ((HCLG, "HCLG file"), (words_txt, "words.txt file"), (args.trans_model, "trans_model")):
            if not path or not os.path.isfile(path):       # the reference's behaviour: message and exit(0)
                sys.stderr.write("ERROR: The %s %s does not exist!\n" % (what, path))
                sys.exit(0)
        trans_model = lattice.TransitionModel.read(args.trans_model)
        graph = lattice.DecodeGraph(HCLG)
        words = read_words_txt(words_txt)
        log_prior = se.log_prior_from_counts(se.read_kaldi_vector(args.prior_path)).to(dev)
    asr_decoder = lattice.MappedLatticeFasterRecognizer(trans_model, graph, acoustic_scale=dc["acoustic_scale"],
                                                        decoder_opts=decoder_opts)
    transform = None
    if args.transform and os.path.isfile(args.transform):
        transform = fbank.GlobalMeanVarianceNormalization.load(args.transform)
    if args.synthetic:
        source = data.SyntheticSource(P, seed=13)
        utts = [source.draw(float(d)) for d in np.random.default_rng(13).uniform(1.5, 4.0, size=args.synthetic)]
    else:
        source = data.ZipWavSource([dict(wav=args.data_path)])
        utts = [(source._read(z, m), None, None, u) for z, m, u, _, _ in source.items]
    fb = fbank.FbankExtractor()
    model.eval()
    wspec = ("ark,t:" if args.out_file.endswith(".txt") else "ark:") + args.out_file
    n_batches = (len(utts) + args.batch_size - 1) // args.batch_size
    with th.no_grad(), kaldi_io.CompactLatticeWriter(wspec) as lat_out:
        for i in range(n_batches):
            chunk = utts[i * args.batch_size:(i + 1) * args.batch_size]
            lens = [u[0].shape[0] for u in chunk]
            wav = th.from_numpy(np.concatenate([u[0] for u in chunk])).to(dev)
            feats, frames, row_off = fb(wav, lens)
            if transform is not None:
                feats = transform(feats)
            x = fb.pad_roll_subsample(feats, row_off, frames, shift=0, subsample=1, time_major=True)
            loglikes = model.forward_time_major(x).transpose(0, 1) - log_prior      # [B, T, P], time-major storage
            lat = asr_decoder.decode_batch(loglikes, [int(t) for t in frames])      # the whole minibatch in one call
            for j in range(len(chunk)):
                key = chunk[j][3]
                cl = lat.compact_lattice(j, determinize=not args.no_determinize)
                text = " ".join(words.get(w, str(w)) for w in cl["best_words"])
                print(key, text)
                print("Log-like per-frame for utterance {} is {}".format(key, -cl["best_cost"] / frames[j]))
                lat_out[key] = cl
            print("Process batch [{}/{}]".format(i + 1, n_batches))


if __name__ == '__main__':
    main()
