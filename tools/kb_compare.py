#!/usr/bin/env python
"""A/B of library builds on the UNet's igemm launch shapes: python tools/kb_compare.py libA.so libB.so ...  (ROT=1: cold weights)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cols = []
for lib in sys.argv[1:]:
    env = dict(os.environ, LDMSEG_HIP_LIB=os.path.abspath(lib))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kbench.py"), "igemm"], env=env, capture_output=True, text=True).stdout
    rows = []
    for ln in out.splitlines():
        m = re.match(r"(M=.*?):\s+([\d.]+) us\s+([\d.]+) TF/s\s+(\S+)", ln)
        if m:
            rows.append((re.sub(r"\s+", "", m.group(1).replace(" N=", ",N=").replace(" K=", ",K=")), float(m.group(2)), m.group(4)))
    cols.append(rows)
print(" " * 74 + "  ".join(f"{os.path.basename(l)[:14]:>14s}" for l in sys.argv[1:]))
tot = [0.0] * len(cols)
for i, (name, _, kern) in enumerate(cols[0]):
    vals = [c[i][1] for c in cols]
    for j, v in enumerate(vals):
        tot[j] += v
    print(f"{name[:72]:72s}  " + "  ".join(f"{v:14.1f}" for v in vals) + "  " + kern[:32])
print(f"{'sum':72s}  " + "  ".join(f"{v:14.1f}" for v in tot))
