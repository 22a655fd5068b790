#!/usr/bin/env python3
"""profiles/traffic.json from a collect_r03.sh run: usage tools/make_traffic.py gpurun_out/prof_<tag> <build_id> [profiles dir name].
HBM bytes per launch of a workload's kernels = WRITE_SIZE + 2 x FETCH_SIZE (both counters in KiB; gfx950 tallies the
128-B requests of wide streaming loads at 64 B: MI355X_MICROARCH.md, HBM section), from separate rocprofv3 --pmc
passes, mean per dispatch, summed over the kernels one fr_plan_render launches.  bench.py quotes an entry only while
its build_id matches the library's."""
import json
import re
import sys

src, build = sys.argv[1], sys.argv[2]
rdir = sys.argv[3] if len(sys.argv) > 3 else "r03"
FRAGS = ("fr::cov4_kernel<", "fr::win1_kernel<", "fr::sdf_kernel<", "fr::render_kernel<")
WL = {   # workload -> algorithmic bytes per launch
    "c3_cjk21k_256px_s128_16spp": 20992 * 65536,
    "c3_cjk21k_256px_s128_gray_debug": 20992 * 65536,
    "c4_bmp_shard_128px_s32_16spp": 7936 * 16384,
    "c5_sdf_shard_512px_s64": 512 * 512 * 512,
    "real_dejavuserif_italic_whole_font_512px_sdf": None,     # (filled from the bench JSON of the same run)
}
vals = {}
for line in open(f"{src}/pmc_summary.txt"):
    m = re.match(r"pmc_(\S+)_(WRITE_SIZE|FETCH_SIZE)\s+(.*?)\s+(WRITE_SIZE|FETCH_SIZE)\s+(\S+) x(\d+)$", line.rstrip())
    if m:
        vals.setdefault((m.group(1), m.group(2)), []).append((m.group(3), float(m.group(5)), int(m.group(6))))
out = {}
for wl, alg in WL.items():
    if alg is None:
        try:
            alg = json.load(open(f"{src}/{wl}_bench.json"))["roofline"]["algorithmic_bytes_per_launch"]
        except Exception:
            continue
    ks = [(k, v) for k, v, n in vals.get((wl, "WRITE_SIZE"), []) if any(f in k for f in FRAGS) and "render_kernel<1, 1," not in k]
    fs = [(k, v) for k, v, n in vals.get((wl, "FETCH_SIZE"), []) if any(f in k for f in FRAGS) and "render_kernel<1, 1," not in k]
    w = sum(v for _, v in ks) * 1024
    f = sum(v for _, v in fs) * 1024
    if not w:
        continue
    out[wl] = {"build_id": build, "kernels": sorted(k for k, _ in ks), "WRITE_SIZE_bytes": w, "FETCH_SIZE_bytes_raw": f,
               "FETCH_SIZE_correction": "x2 (gfx950 tallies the 128-B requests of wide streaming loads at 64 B: MI355X_MICROARCH.md, HBM section)",
               "hbm_bytes_per_launch": int(w + 2 * f), "algorithmic_bytes_per_launch": alg,
               "ratio": round((w + 2 * f) / alg, 3),
               "source": f"profiles/{rdir}/pmc_summary.txt (rocprofv3 --pmc WRITE_SIZE / --pmc FETCH_SIZE, separate passes, mean per dispatch, summed over the render's kernels; tools/collect_r03.sh, tools/make_traffic.py)"}
json.dump(out, open("profiles/traffic.json", "w"), indent=1)
for k, v in out.items():
    print(k, v["hbm_bytes_per_launch"], v["ratio"])
