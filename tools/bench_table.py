#!/usr/bin/env python3
"""markdown rows for DESIGN.md section 7 from a directory of bench JSONs: python tools/bench_table.py profiles/r03"""
import glob
import json
import sys

rows = []
for f in sorted(glob.glob(sys.argv[1] + "/*_bench.json")):
    d = json.load(open(f))
    r, c = d["roofline"], d.get("cpu_baseline") or {}
    rows.append((d["config"]["workload"], r["launches"].replace("fr::", ""), r["kernel_ms"], d["value"] / 1e6, 100 * r["frac"], c.get("value"), c.get("matches_gpu_bytes")))
for w, k, ms, tp, fr, cpu, same in rows:
    print(f"| `{w}` | `{k}` | {ms:.4f} | {tp:.3f} | {fr:.1f} % | {cpu} ({'==' if same else same}) |")
