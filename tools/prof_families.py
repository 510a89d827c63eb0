#!/usr/bin/env python3
"""Kernel time per family from a tools/prof_step_summary.py listing (stdin or file)."""
import re
import sys
from collections import defaultdict

fam, n = defaultdict(float), defaultdict(float)
for line in (open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin):
    m = re.search(r'^\s+(\S.*?)\s+n=\s*(\d+) total_ms=\s*([\d.]+).*per_step_ms=\s*([\d.]+)', line)
    if not m:
        print(line.rstrip())
        continue
    name, per = m.group(1), float(m.group(4))
    k = re.sub(r'<.*', '', name)
    mm = re.match(r'conv_(fwd|wgrad)_kernel<(\d+)', name)
    if mm:
        k = f"conv_{mm.group(1)} k={mm.group(2)}"
    fam[k] += per
    n[k] += int(m.group(2))
for k, v in sorted(fam.items(), key=lambda kv: -kv[1]):
    print(f"{k:44s} {v:7.3f} ms/step   launches {int(n[k])}")
print(f"{'sum':44s} {sum(fam.values()):7.3f} ms/step")
