#!/bin/bash
# Static resource census of the hot kernels (no GPU needed): architectural VGPRs, AGPRs, SGPRs, scratch bytes per lane, occupancy.
#   bash tools/kernel_resources.sh > profiles/kernel_resources_rNN.txt
REPO=$(cd "$(dirname "$0")/.." && pwd)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -I $REPO/include -I $REPO/consistent_depth_amd/csrc -c --cuda-device-only -Rpass-analysis=kernel-resource-usage"
for f in conv_split wgrad_split conv1x1_split wgrad1x1_split loss_sweep; do
  extra=""; [ $f = loss_sweep ] && extra="-fno-slp-vectorize"
  echo "== $f"
  /opt/rocm/bin/hipcc $FLAGS $extra -o /tmp/kr_$f.o $REPO/consistent_depth_amd/csrc/$f.hip 2>&1 | python3 -c "
import re, sys, subprocess
cur = None
rows = []
for line in sys.stdin:
    m = re.search(r'remark: +Function Name: (\S+)', line)
    if m:
        cur = {'name': m.group(1)}; rows.append(cur); continue
    for key, pat in (('vgpr', r'remark: +VGPRs: (\d+)'), ('agpr', r'AGPRs: (\d+)'), ('sgpr', r'TotalSGPRs: (\d+)'), ('scratch', r'ScratchSize \[bytes/lane\]: (\d+)'), ('occ', r'Occupancy \[waves/SIMD\]: (\d+)'), ('lds', r'LDS Size \[bytes/block\]: (\d+)')):
        m = re.search(pat, line)
        if m and cur is not None: cur[key] = int(m.group(1))
names = subprocess.run(['/usr/bin/c++filt'] + [r['name'] for r in rows], capture_output=True, text=True).stdout.split('\n')
for r, n in zip(rows, names):
    n = re.sub(r'\(.*', '', n).replace('void ', '').replace('cd::', '')
    if 'kernel' not in n: continue
    print(f\"{n[:78]:78s} vgpr {r.get('vgpr', 0):4d} agpr {r.get('agpr', 0):4d} sgpr {r.get('sgpr', 0):4d} scratch {r.get('scratch', 0):4d} occ {r.get('occ', 0)} lds {r.get('lds', 0)}\")
"
done
