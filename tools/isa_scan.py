#!/usr/bin/env python3
"""tools/isa_scan.py [file.s] [kernel-substring] -- per kernel of a gfx950 assembly dump (tools/kernel_regs.sh leaves /tmp/last_kernel.s):
instruction counts, scratch (spill) traffic and where it sits relative to the MFMA stream, s_waitcnt vmcnt(0) count (development aid)."""
import re
import sys

_argv = [x for x in sys.argv[1:] if x != "--ring"]
path = _argv[0] if len(_argv) > 0 else "/tmp/last_kernel.s"
want = _argv[1] if len(_argv) > 1 else ""
lines = open(path).read().split("\n") if "--ring" not in sys.argv else []
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


def ring_check(path, want):
    """--ring: the ring form of deform_fwd16_kernel issues its operand requests from inline assembly (4 x global_load_dwordx4 between
    ASMSTART / ASMEND) and commits them with s_waitcnt vmcnt(0) + 4 x ds_write_b128.  The compiler does not know the results arrive
    late: any instruction that READS or WRITES one of the 16 destination registers between the request and the vmcnt(0) of its commit
    (a register copy at a branch join, a live-range split) would move or clobber data that are not there yet.  Flags every such one."""
    lines = open(path).read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)] + [len(lines)]
    bad = total = 0
    for a, b in zip(starts, starts[1:]):
        name = lines[a].split(":")[0]
        if want not in name:
            continue
        body = lines[a:b]
        i = 0
        while i < len(body):
            if "ASMSTART" in body[i] and i + 4 < len(body) and all("global_load_dwordx4" in body[i + k] for k in range(1, 5)):
                regs = set()
                for k in range(1, 5):
                    m = re.search(r"global_load_dwordx4 v\[(\d+):(\d+)\]", body[i + k])
                    regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
                total += 1
                j = i + 6
                seen_branch = False
                while j < len(body):
                    l = body[j].strip()
                    if l.startswith("s_waitcnt") and "vmcnt(0)" in l:
                        break
                    if l.startswith("s_endpgm"):
                        print(f"  request at line {a + i}: no commit before the end of the kernel"); bad += 1
                        break
                    if l and not l.startswith((";", ".")) and not l.endswith(":"):
                        used = set()
                        for m in re.finditer(r"v\[(\d+):(\d+)\]", l):
                            used.update(range(int(m.group(1)), int(m.group(2)) + 1))
                        for m in re.finditer(r"\bv(\d+)\b", l):
                            used.add(int(m.group(1)))
                        if used & regs:
                            print(f"  request at line {a + i}: line {a + j} touches staging registers before the commit: {l}"); bad += 1
                    j += 1
                i = j
            else:
                i += 1
    print(f"ring check: {total} request sites, {bad} problems")
    return bad


if "--ring" in sys.argv:
    args = [x for x in sys.argv[1:] if x != "--ring"]
    sys.exit(1 if ring_check(args[0], args[1] if len(args) > 1 else "Lb1EEE") else 0)
