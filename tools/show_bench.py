#!/usr/bin/env python3
"""one line per bench JSON: python tools/show_bench.py gpurun_out/r3c/bench_*.json"""
import json
import sys

for f in sys.argv[1:]:
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "unreadable:", e)
        continue
    r, c = d["roofline"], d.get("cpu_baseline") or {}
    print(f'{d["config"]["workload"]:62s} {d["value"] / 1e3:9.1f} Gpx/s  step {d["ms_per_step"]:.4f} ms  kernel {r["kernel_ms"]:.4f} ms  '
          f'frac {r["frac"]:.4f}  cpu {c.get("value")} same={c.get("matches_gpu_bytes")}')
    print("      ", r.get("launches"))
