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
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function",
         "-Rpass-analysis=kernel-resource-usage"]     # (registers / scratch / LDS per kernel -> build/<unit>.resources.txt)


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


def _resource_usage(text):
    """{kernel: {field: value}} from hipcc's -Rpass-analysis=kernel-resource-usage remarks."""
    usage, cur = {}, None
    for line in text.splitlines():
        if "kernel-resource-usage" not in line or "remark:" not in line:
            continue
        body = line.split("remark:", 1)[1].split("[-Rpass", 1)[0].strip()
        if body.startswith("Function Name:"):
            cur = usage.setdefault(body.split(":", 1)[1].strip(), {})
        elif cur is not None and ":" in body:
            k, v = body.rsplit(":", 1)
            cur[k.strip()] = v.strip()
    return usage


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
        text = out.decode()
        usage = _resource_usage(text)
        if usage:
            with open(os.path.join(objdir, src.replace(".hip", ".resources.txt")), "w") as f:
                for name, u in sorted(usage.items()):
                    f.write(f"{name} vgprs={u.get('VGPRs', '?')} agprs={u.get('AGPRs', '?')} scratch={u.get('ScratchSize [bytes/lane]', '?')} lds={u.get('LDS Size [bytes/block]', '?')}\n")
            # the weight-stationary kernels hold their operands in registers for a whole launch: a spilled register's reload is a vector-memory
            # wait for every DMA and store in flight (csrc/deform_fwd_ws.h, deform_bwd_ws.h) -- a build that spills them is a slow build
            for name, u in usage.items():
                if "_ws_kernel" in name and u.get("ScratchSize [bytes/lane]", "0") != "0":
                    sys.stderr.write(f"WARNING: {name} spills ({u['ScratchSize [bytes/lane]']} bytes of scratch per lane)\n")
        text = "\n".join(l for l in text.splitlines() if "kernel-resource-usage" not in l)
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"--- {src} ---\n{text}\n")
        elif verbose and text:
            sys.stderr.write(text)
    if failed:
        raise RuntimeError("hipcc failed")
    if procs or force or not os.path.exists(LIB):
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
