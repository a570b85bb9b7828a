#!/bin/bash
# tools/kernel_regs.sh [extra hipcc flags] -- register / scratch / LDS use of every kernel of deform.hip's (128, 32) instance (development aid)
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d); cd $T
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -DFDGS_DEV_ONLY_44 "$@" -save-temps -c $R/4dgaussians_amd/csrc/${SRC:-deform.hip} -o x.o 2>&1 | grep -v warning | head -20
python3 - <<'PY'
import re,glob
s=open(glob.glob('*gfx950.s')[0]).read()
for m in re.finditer(r'- \.agpr_count:\s+(\d+).*?\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)', s, flags=re.S):
    import subprocess
    name=subprocess.run(['c++filt',m.group(3)],capture_output=True,text=True).stdout.strip()[:90]
    print(f"{name:92s} vgpr+agpr {m.group(6):>4s} (agpr {m.group(1):>3s}) sgpr {m.group(5):>3s} scratch {m.group(4):>4s} lds {m.group(2):>6s}")
PY
cp *gfx950.s /tmp/last_kernel.s
rm -rf $T
