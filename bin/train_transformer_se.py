#!/usr/bin/env python
"""Sequence-discriminative training (lattice MMI / sMBR / MPFE) of the TransformerAM on MI355X -- command line of
the reference's bin/train_transformer_se.py (flags, YAML schema, checkpoints model.se.{epoch}.tar).  Shares
everything but the model and its attention masks with bin/train_se.py.

  python -m torch.distributed.run --nproc-per-node 8 bin/train_transformer_se.py -config configs/se.yaml \
      -data configs/data.yaml -exp_dir exp/tse -criterion mmi -nlayers 12 -nheads 8 -seed_model exp/tce/model.0.tar ...
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from train_se import main  # noqa: E402

if __name__ == '__main__':
    main("transformer")
