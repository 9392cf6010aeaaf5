"""Builds pykaldi2_amd/libpk2hip.so from csrc/*.hip with hipcc for gfx950.

In-tree build (the .so travels to the GPU box with the repo snapshot).
Incremental: a source is recompiled when it or any header is newer than its
object.  `python -m pykaldi2_amd.build [--force]`.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
# experiment builds: PK2_BUILD_TAG=name PK2_EXTRA_FLAGS="-DPK2_DEN_K=4" -> libpk2hip_name.so (select with PK2_LIB)
_TAG = os.environ.get("PK2_BUILD_TAG", "")
OBJ = os.path.join(HERE, "build" + ("_" + _TAG if _TAG else ""))
LIB = os.path.join(HERE, "libpk2hip" + ("_" + _TAG if _TAG else "") + ".so")
ARCH = "gfx950"
# -amdgpu-kernarg-preload-count: gfx950 hands the first kernel arguments to a wave in SGPRs instead of making it
# fetch them from the kernarg segment; the per-time-step kernels (args = two pointers + an index, then one more
# scalar load of the parameter block) start their real loads one memory round trip earlier
# (measured: LSTM forward step 4.40 -> 4.16 us, denominator call 14.67 -> 14.55 ms).
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
         "-mllvm", "-amdgpu-kernarg-preload-count=8",
         "-Wall", "-Wno-unused-function", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC] + \
    os.environ.get("PK2_EXTRA_FLAGS", "").split()
# hipcc defaults to -ffp-contract=fast, which fuses a*b+c across statements and ignores `#pragma clang fp
# contract(off)`; the decoder's costs must round like the oracle's separate float32 operations.
FILE_FLAGS = {"lattice_decode.hip": ["-ffp-contract=off"], "lattice_decode_frames.hip": ["-ffp-contract=off"]}
# experiment builds: PK2_FILE_FLAGS="gemm_f32.hip:-mllvm,-align-loops=64;..." adds flags to single sources
for _spec in os.environ.get("PK2_FILE_FLAGS", "").split(";"):
    if ":" in _spec:
        _f, _fl = _spec.split(":", 1)
        FILE_FLAGS.setdefault(_f, [])
        FILE_FLAGS[_f] = FILE_FLAGS[_f] + _fl.split(",")


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _digest(paths, extra=""):
    import hashlib
    h = hashlib.sha256(extra.encode())
    for p_ in sorted(paths):
        h.update(os.path.basename(p_).encode())
        with open(p_, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def build(force=False, verbose=True):
    """Incremental by CONTENT, not by time stamps (a checkout or a copied tree has arbitrary mtimes -- a
    build that finds a library next to sources it was not made from must not trust it): every object carries a stamp
    `<obj>.sha` = sha256 of its source, all headers and the flags; the library one over the objects' stamps."""
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(ROOT, "include", "pk2hip.h"))
    hdr_sig = _digest(hdrs)
    hipcc = _hipcc()
    jobs, stamps = [], {}
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s[:-4] + ".o")
        sig = _digest([src], hdr_sig + " ".join(FLAGS + FILE_FLAGS.get(s, [])))
        stamps[obj] = sig
        old = None
        if os.path.exists(obj + ".sha"):
            with open(obj + ".sha") as f:
                old = f.read().strip()
        if force or not os.path.exists(obj) or old != sig:
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [hipcc] + FLAGS + FILE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return src, r

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for src, r in ex.map(cc, jobs):
                if verbose and (r.stderr.strip() or r.returncode):
                    sys.stderr.write(r.stderr)
                if r.returncode:
                    raise RuntimeError("hipcc failed on %s" % src)
                with open(os.path.join(OBJ, os.path.basename(src)[:-4] + ".o.sha"), "w") as f:
                    f.write(stamps[os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")])
    objs = [os.path.join(OBJ, s[:-4] + ".o") for s in srcs]
    lib_sig = _digest([], "".join(stamps[o] for o in objs))
    old = None
    if os.path.exists(LIB + ".sha"):
        with open(LIB + ".sha") as f:
            old = f.read().strip()
    if jobs or not os.path.exists(LIB) or force or old != lib_sig:
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            sys.stderr.write(r.stderr)
            raise RuntimeError("link failed")
        with open(LIB + ".sha", "w") as f:
            f.write(lib_sig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
