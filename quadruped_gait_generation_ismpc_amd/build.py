"""Builds the in-tree native libraries with hipcc for gfx950 (no JIT cache, no torch extension):

  libismpc_hip.so   kernels + C ABI of include/ismpc.h, include/ismpc_a.h, include/ismpc_group.h   (csrc/*.hip, csrc/ismpc_tables.cpp)

The C++ MPCSolver drop-in (include/MPCSolver.hpp) is header-only over that ABI.
"""
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
ROOT = os.path.dirname(PKG)
LIB_HIP = os.path.join(PKG, "libismpc_hip.so")
ARCH = "gfx950"


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the ISMPC kernels cannot be built")
    return exe


def _stale(target, sources, flags):
    """Stale = missing, built with other compiler flags (the flag string is kept next to the .so, so that a tuning sweep's variant can never
    be mistaken for the default build), or built from other sources: the sha256 of the sources' CONTENT is kept next to the .so too
    (modification times say nothing after a tree has been copied to another machine)."""
    if not os.path.exists(target):
        return True
    try:
        if open(target + ".flags").read() != flags:
            return True
        return open(target + ".src_sha256").read().strip() != source_sha256(flags)
    except OSError:
        return True


UNITS = ["ismpc_hip.hip", "ismpc_sweep.hip", "ismpc_a_hip.hip", "ismpc_a_wave_rl2.hip", "ismpc_a_wave_rl3.hip", "ismpc_a_wave_rl4.hip", "ismpc_group.hip", "ismpc_tables.cpp"]
HEADERS = ["ismpc_tables.hpp", "ismpc_sweep.hpp", "ismpc_a_dev.hpp", "ismpc_a_wave.hpp"]
PUBLIC = ["ismpc.h", "ismpc_a.h", "ismpc_group.h"]
BASE_FLAGS = ["-O3", "-fno-slp-vectorize", "-std=c++17", "-fPIC"]


def source_sha256(flags=""):
    """sha256 over everything that decides what the library computes: the translation units, their headers, the public headers and the
    compiler flags -- by CONTENT, so the same sources give the same value wherever the tree sits.  (The library's own bytes embed the
    source paths: a build of the same sources in another directory has another lib_sha256.)  Counter summaries under profiles/ carry both;
    bench.py accepts a summary when either equals the loaded library's."""
    import hashlib
    h = hashlib.sha256()
    for name in [os.path.join(CSRC, u) for u in UNITS] + [os.path.join(CSRC, u) for u in HEADERS] + [os.path.join(ROOT, "include", u) for u in PUBLIC]:
        h.update(os.path.basename(name).encode() + b"\0")
        with open(name, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    h.update(" ".join(BASE_FLAGS + [f"--offload-arch={ARCH}"] + flags.split()).encode())
    return h.hexdigest()


def build(force=False, verbose=False, out=None, flags=None):
    """out: alternative output path (tuning sweeps build their variants there and load them with ISMPC_LIB=<path>; the
    in-tree default library is never overwritten by a variant).  flags: extra hipcc flags (default: $ISMPC_HIPCC_FLAGS).
    The translation units are compiled side by side (the Formulation A wave kernels are one unit per rows-per-lane value)."""
    target = out or LIB_HIP
    flags = os.environ.get("ISMPC_HIPCC_FLAGS", "") if flags is None else flags
    if out is None and flags.strip():
        raise RuntimeError("non-default compiler flags need an explicit output path: build(out=..., flags=...) and ISMPC_LIB=<out>")
    hip_src = [os.path.join(CSRC, u) for u in UNITS]
    deps = hip_src + [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(ROOT, "include", h) for h in PUBLIC]
    quiet = None if verbose else subprocess.DEVNULL
    if force or _stale(target, deps, flags):
        objdir = os.path.join(ROOT, "build", "obj", os.path.basename(target))
        os.makedirs(objdir, exist_ok=True)
        # -fno-slp-vectorize: the SLP vectoriser pairs fp32 values into 64-bit register tuples (v_pk_*); in the Formulation A wave
        # kernels that costs far more registers than it saves instructions (<float,3,4,false>: 60 spilled VGPRs with it, 7 without;
        # <float,4,6,true>: 40 -> 0; measured +2-4 % ticks/s, +20 % on trot C=160 fp32)
        common = [_hipcc(), f"--offload-arch={ARCH}"] + BASE_FLAGS + ["-I", os.path.join(ROOT, "include")] + flags.split()
        procs = []
        for src in hip_src:
            obj = os.path.join(objdir, os.path.basename(src) + ".o")
            procs.append((src, obj, subprocess.Popen(common + ["-c", src, "-o", obj], stdout=quiet)))
        objs = []
        for src, obj, pr in procs:
            if pr.wait() != 0:
                raise subprocess.CalledProcessError(pr.returncode, f"hipcc -c {src}")
            objs.append(obj)
        subprocess.check_call([_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC"] + objs + ["-ldl", "-o", target], stdout=quiet)
        with open(target + ".flags", "w") as f:
            f.write(flags)
        with open(target + ".src_sha256", "w") as f:          # what the library was built FROM (bench.py / scripts/pmc_summary.py read it)
            f.write(source_sha256(flags))
    return target


if __name__ == "__main__":
    build(force=True, verbose=True)
    print(LIB_HIP)
