#!/usr/bin/env python
"""Command line of the reference's decode.py (the LibriCSS evaluation helper): dumps the acoustic model's log-likelihoods
(minus log priors when -prior_path is given) into a Kaldi matrix archive, here with fbank + BLSTM forward on the device.
Same flags, including -frame_subsampling_factor and -gpuid; checkpoints of NnetAM(LSTMStack) (keys `nnet.lstm.*`, what the
LibriCSS release ships) load unchanged.  The work is bin/dump_loglikes.py's.

  python decode.py -config configs/ce.yaml -model_path libricss.model.tar -data_path eval.zip -prior_path final.occs \
      -out_file loglikes.ark
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin"))
import dump_loglikes  # noqa: E402

if __name__ == "__main__":
    dump_loglikes.main()
