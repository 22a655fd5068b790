#!/usr/bin/env python3
"""profiles/traffic.json from a collect_r02.sh run: usage tools/make_traffic.py gpurun_out/prof_<tag> <build_id>.
HBM bytes per launch of the dominant kernel(s) of a workload = WRITE_SIZE + 2 x FETCH_SIZE (both counters in KiB;
gfx950 tallies the 128-B requests of wide streaming loads at 64 B: MI355X_MICROARCH.md, HBM section), from separate
rocprofv3 --pmc passes, mean per dispatch.  bench.py quotes an entry only while its build_id matches the library's."""
import json
import re
import sys

src, build = sys.argv[1], sys.argv[2]
WL = {   # workload -> (kernel name fragments summed, label, algorithmic bytes per launch)
    "c3_cjk21k_256px_s128_16spp": (["cov4_kernel<4, 32, 4>"], "fr::cov4_kernel<4,32,4>", 20992 * 65536),
    "c3_cjk21k_256px_s128_gray_debug": (["win1_kernel<4, 1, 4>"], "fr::win1_kernel<4,gray_debug,4>", 20992 * 65536),
    "c3_cjk21k_256px_s128_winding_i16": (["win1_kernel<4, 0, 4>"], "fr::win1_kernel<4,winding_i16,4>", 20992 * 65536 * 2),
    "c5_sdf_shard_512px_s64": (["win1_kernel<4, 2, ", "sdf_kernel<false>"], "fr::win1_kernel<4,mask> (sign pass) + fr::sdf_kernel<false>", 512 * 512 * 512),
}
vals = {}
for line in open(f"{src}/pmc_summary.txt"):
    m = re.match(r"pmc_(\S+)_(WRITE_SIZE|FETCH_SIZE)\s+(.*?)\s+(WRITE_SIZE|FETCH_SIZE)\s+(\S+)$", line.rstrip())
    if m:
        vals.setdefault((m.group(1), m.group(2)), []).append((m.group(3), float(m.group(5))))
out = {}
for wl, (frags, label, alg) in WL.items():
    w = sum(v for k, v in vals.get((wl, "WRITE_SIZE"), []) if any(f in k for f in frags)) * 1024
    f = sum(v for k, v in vals.get((wl, "FETCH_SIZE"), []) if any(f in k for f in frags)) * 1024
    if not w:
        continue
    out[wl] = {"build_id": build, "kernel": label, "WRITE_SIZE_bytes": w, "FETCH_SIZE_bytes_raw": f,
               "FETCH_SIZE_correction": "x2 (gfx950 tallies the 128-B requests of wide streaming loads at 64 B: MI355X_MICROARCH.md, HBM section)",
               "hbm_bytes_per_launch": int(w + 2 * f), "algorithmic_bytes_per_launch": alg,
               "source": "profiles/r02/pmc_summary.txt (rocprofv3 --pmc WRITE_SIZE / --pmc FETCH_SIZE, separate passes, mean per dispatch; tools/collect_r02.sh, tools/make_traffic.py)"}
json.dump(out, open("profiles/traffic.json", "w"), indent=1)
for k, v in out.items():
    print(k, v["hbm_bytes_per_launch"], round(v["hbm_bytes_per_launch"] / v["algorithmic_bytes_per_launch"], 3))
