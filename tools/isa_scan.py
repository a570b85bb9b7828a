#!/usr/bin/env python3
"""tools/isa_scan.py [file.s] [kernel-substring] -- per kernel of a gfx950 assembly dump (tools/kernel_regs.sh leaves /tmp/last_kernel.s):
instruction counts, scratch (spill) traffic and where it sits relative to the MFMA stream, s_waitcnt vmcnt(0) count (development aid)."""
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/last_kernel.s"
want = sys.argv[2] if len(sys.argv) > 2 else ""
lines = open(path).read().split("\n")
starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
starts.append(len(lines))
for a, b in zip(starts, starts[1:]):
    name = lines[a].split(":")[0]
    if want not in name:
        continue
    body = lines[a:b]
    end = next((i for i, l in enumerate(body) if l.strip().startswith("s_endpgm")), len(body))
    body = body[:end]
    ins = [l for l in body if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    mf = [i for i, l in enumerate(ins) if "v_mfma" in l]
    sc = [i for i, l in enumerate(ins) if "scratch_" in l]
    vm0 = sum(1 for l in ins if "s_waitcnt" in l and "vmcnt(0)" in l)
    print(f"{name[:80]}: {len(ins)} instr, {len(mf)} mfma, {len(sc)} scratch ops, {vm0} x vmcnt(0)")
    if sc and mf:
        # scratch ops that have MFMAs within 40 instructions on both sides = inside a matrix loop
        hot = [i for i in sc if any(abs(i - m) < 40 for m in mf)]
        print(f"    scratch ops within 40 instr of an MFMA: {len(hot)}  (loads {sum('load' in ins[i] for i in hot)}, stores {sum('store' in ins[i] for i in hot)})")
