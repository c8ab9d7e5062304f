"""Build libf5hip.so (hand-written HIP for gfx950) in-tree:  python -m eraxvif5tts_amd.build [--force]

hipcc cross-compiles gfx950 without a GPU.  Objects go to eraxvif5tts_amd/build/, the library to
eraxvif5tts_amd/lib/libf5hip.so (git-ignored, but shipped to the GPU box with the tree).
"""
from __future__ import annotations

import concurrent.futures
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
BUILD = os.path.join(HERE, "build")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libf5hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function", "-Wno-unused-variable",
         "-Wno-unused-but-set-variable", "-ffp-contract=off", "-fvisibility=hidden"]


# per-file extra flags: the hand-laid vector stream of the attention kernel must not be re-packed into v_pk_*_f32 by the SLP vectorizer
# (packed fp32 VALU is slower beside MFMAs: /opt/skills/guides/MI355X_MICROARCH.md, cycle constants)
EXTRA_FLAGS = {"attention_pipe.hip": ["-fno-slp-vectorize"], "attention_fast.hip": ["-fno-slp-vectorize"]}


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libf5hip needs the ROCm toolchain (there is no CPU fallback)")


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest():
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)) + ["../../include/f5hip.h"]:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode())
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(EXTRA_FLAGS.items())).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(BUILD, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    stamp = os.path.join(BUILD, "digest.txt")
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB
    hipcc = _hipcc()
    # an object is reused when its own source, every header and its flags are unchanged (per-object stamp next to it)
    hdr = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)) + ["../../include/f5hip.h"]:
        if f.endswith(".h"):
            with open(os.path.join(CSRC, f), "rb") as fh:
                hdr.update(f.encode())
                hdr.update(fh.read())

    def compile_one(src):
        obj = os.path.join(BUILD, src.replace(".hip", ".o"))
        extra = EXTRA_FLAGS.get(src, [])
        h = hdr.copy()
        with open(os.path.join(CSRC, src), "rb") as fh:
            text = fh.read()
        h.update(text)
        for line in text.decode("utf-8", "replace").splitlines():  # a .hip that includes another .hip (second translation unit of one template)
            if line.startswith('#include "') and line.rstrip().endswith('.hip"'):
                with open(os.path.join(CSRC, line.split('"')[1]), "rb") as fh:
                    h.update(fh.read())
        h.update(" ".join(FLAGS + extra).encode())
        ostamp = obj + ".digest"
        if not force and os.path.exists(obj) and os.path.exists(ostamp) and open(ostamp).read() == h.hexdigest():
            return obj
        cmd = [hipcc, *FLAGS, *extra, "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        with open(ostamp, "w") as fh:
            fh.write(h.hexdigest())
        return obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 2)) as ex:
        objs = list(ex.map(compile_one, _sources()))
    r = subprocess.run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", LIB], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as fh:
        fh.write(dig)
    if verbose:
        print(f"built {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
