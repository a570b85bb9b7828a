"""Instruction mix of one kernel in a hipcc -save-temps .s file:  python tools/isa_kernel_stats.py FILE.s MANGLED_SUBSTRING"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
m = re.search(r"^(\S*%s\S*):" % re.escape(sys.argv[2]), s, re.M)
i = m.start()
j = s.find('.end_amdhsa_kernel', i)
lines = s[i:j].split('\n')
print(m.group(1), len(lines), "lines")
c = collections.Counter()
for l in lines:
    mm = re.match(r'\s+([a-z_0-9]+)', l)
    if mm:
        c[mm.group(1)] += 1
pat = sys.argv[3] if len(sys.argv) > 3 else r'mfma|scratch|ds_|global_|s_waitcnt|accvgpr|s_barrier|buffer_'
for k, v in sorted(c.items()):
    if re.search(pat, k):
        print(f"  {k:32s} {v}")
for n, l in enumerate(lines):
    if re.match(r'\.LBB\d+_\d+:', l):
        print(n, l)
