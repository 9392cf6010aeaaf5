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
LIB64 = os.path.join(OUT, "libchain_oracle_f64.so")      # the same recursion with double state (parity at bench size)


LAT = os.path.join(OUT, "liblattice_oracle.so")


def build_lattice():
    """oracle/lattice_oracle.c: the C restatement of lattice_ref.py's decoder + lattice MMI (cpu_baseline of `bench.py --se`)."""
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(HERE, "lattice_oracle.c")
    if not os.path.exists(LAT) or os.path.getmtime(LAT) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-march=x86-64-v2", "-ffp-contract=off", "-shared", "-fPIC", src, "-o", LAT, "-lm"])
    return LAT


def build(double=False):
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(HERE, "chain_oracle.c")
    for lib, extra in ((LIB, []), (LIB64, ["-DORC_REAL=double"])):
        if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
            subprocess.check_call(["gcc", "-O3", "-march=x86-64-v2", "-fopenmp", "-shared", "-fPIC"] + extra + [src, "-o", lib, "-lm"])
    return LIB64 if double else LIB


if __name__ == "__main__":
    print(build())
    print(build_lattice())
