"""Build recipe of libfdgs.so (hipcc, gfx950 only) -- in-tree so that the built library travels with the repo.

    python -m 4dgaussians_amd.build      (or __graft_entry__.build())
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libfdgs.so")
SOURCES = ["api.hip", "preprocess.hip", "binning.hip", "render.hip", "deform.hip", "regulation.hip", "adam.hip", "knn.hip", "loss.hip", "densify.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _deps(src):
    deps = [os.path.join(CSRC, src)]
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "fdgs.h"))
    return deps


def build(force=False, verbose=False):
    """Compile every HIP translation unit for gfx950 and link libfdgs.so next to this file."""
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    objs, procs = [], []
    for src in SOURCES:
        if not os.path.exists(os.path.join(CSRC, src)):
            continue
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        stale = force or not os.path.exists(obj) or any(os.path.getmtime(d) > os.path.getmtime(obj) for d in _deps(src))
        if stale:
            cmd = [hipcc, *FLAGS, *os.environ.get("FDGS_EXTRA_FLAGS", "").split(), "-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"--- {src} ---\n{out.decode()}\n")
        elif verbose and out:
            sys.stderr.write(out.decode())
    if failed:
        raise RuntimeError("hipcc failed")
    if procs or force or not os.path.exists(LIB):
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
