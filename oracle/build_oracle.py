"""Compiles oracle/chain_oracle.c with gcc into oracle/_build/ (checker / CPU-baseline code; never
linked into the product).  There is no reference build here: the reference's arithmetic for this
path lives in Kaldi (C++/CUDA, third party, needs its own build system, OpenFst, BLAS and is absent
from /root/reference), so it is "unbuildable" in the sense of the task rules and oracle/_ref/ stays
empty."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libchain_oracle.so")


def build():
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(HERE, "chain_oracle.c")
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O3", "-march=x86-64-v2", "-fopenmp", "-shared", "-fPIC", src, "-o", LIB, "-lm"])
    return LIB


if __name__ == "__main__":
    print(build())
