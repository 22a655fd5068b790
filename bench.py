#!/usr/bin/env python3
"""bench.py — whole-job throughput of the hot path (AA glyph coverage) on N MI355X.

One "step" = one pass of the hot path over one batch: everything from parsed quadratic contours
resident in HBM to the anti-aliased u8 coverage atlas resident in HBM (root records rebuilt inside
the render kernels every step; the stand-alone precompute kernel runs first for glyphs too large
for that).  Synthetic outlines unless the workload names a font; each rank owns a
contiguous glyph range and its own atlas band — no collective on the data path.

N > 1: the driver launches `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`
(one rank per GPU, RCCL); bench.py never spawns ranks itself.  Weak scaling by default (the workload's
glyph count PER GPU); `--total-glyphs T` fixes the job size instead (strong scaling: configs[3] /
configs[4] are fixed-size jobs).  With N > 1 the optional assembly of the atlas bands is timed too, in both forms —
gather onto rank 0 and all-gather — reported separately under "gather", never part of `value` (--no-gather skips it).

Prints ONE JSON line (rank 0).  See DESIGN.md §7 for every field's derivation."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import numpy as np  # noqa: E402

RAMP_S = 0.25                  # untimed renders before --warmup (reported as "ramp_s" / "ramp_steps")
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
DEJAVU = "/usr/local/lib/python3.10/dist-packages/matplotlib/mpl-data/fonts/ttf/DejaVuSerif-Italic.ttf"

WORKLOADS = {
    # BASELINE.json configs[2]: the largest single-GPU configuration (~21k CJK-like glyphs) — the headline
    "c3_cjk21k_256px_s128_16spp": dict(glyphs=20992, cell=256, segs=128, n=4, cols=64),
    # BASELINE.json configs[1]: 95 glyphs, 128x128 cells, one 2048^2 atlas
    "c2_ascii95_128px_s32_16spp": dict(glyphs=95, cell=128, segs=32, n=4, cols=16),
    "c2_ascii95_real_128px_16spp": dict(glyphs=95, cell=128, segs=0, n=4, cols=16, gen="ascii"),
    # configs[1] without the launch bound: 64 such pages (64 x 95 different glyphs, each page its own 2048^2) stacked in one
    # buffer and rendered by ONE plan = one launch per step
    "c2_ascii95_128px_s32_16spp_x64pages": dict(glyphs=95 * 64, cell=128, segs=32, n=4, cols=16, gen="pages", per_page=95, page_rows=16),
    # BASELINE.json configs[3] per-GPU share (BMP/8 = 7936 glyphs, 128^2, 16 samples)
    "c4_bmp_shard_128px_s32_16spp": dict(glyphs=7936, cell=128, segs=32, n=4, cols=64),
    "c4_bmp_shard_128px_s32_gray_debug": dict(glyphs=7936, cell=128, segs=32, n=1, cols=64, mode="gray_debug"),
    # BASELINE.json configs[4] per-GPU share (4 096 glyphs / 8 = 512, 512^2 cells): build-defined SDF, 1 sample
    "c5_sdf_shard_512px_s64": dict(glyphs=512, cell=512, segs=64, n=1, cols=16, mode="sdf"),
    # SURVEY §8(d) segment list {16, 32, 64, 128, 256}: the C3 shape at the other segment counts
    "c3_cjk21k_256px_s256_16spp": dict(glyphs=20992, cell=256, segs=256, n=4, cols=64),
    "c3_cjk21k_256px_s64_16spp": dict(glyphs=20992, cell=256, segs=64, n=4, cols=64),
    "c3_cjk21k_256px_s32_16spp": dict(glyphs=20992, cell=256, segs=32, n=4, cols=64),
    "c3_cjk21k_256px_s16_16spp": dict(glyphs=20992, cell=256, segs=16, n=4, cols=64),
    # 2 x 2 samples per pixel (cov4_kernel<.., 2>)
    "c3_cjk21k_256px_s128_4spp": dict(glyphs=20992, cell=256, segs=128, n=2, cols=64),
    # glyphs too large for the fast kernels (512 segments > 384): what is left on the general render_kernel
    "big_s512_2048cells_256px_16spp": dict(glyphs=2048, cell=256, segs=512, n=4, cols=64),
    # what is LEFT on the general render_kernel: glyphs of more than 768 segments
    "huge_s1024_512cells_256px_16spp": dict(glyphs=512, cell=256, segs=1024, n=4, cols=32),
    # stroke-dense outlines: 8-16 thin strokes per glyph, 10-30 crossings per ray (synth.stroke_glyph)
    "c3_strokes21k_256px_s128_16spp": dict(glyphs=20992, cell=256, segs=128, n=4, cols=64, gen="stroke"),
    # a real font through the C-side contour producer (fr_font_*): every glyph of DejaVuSerif-Italic the
    # reference could load (~3 000 glyphs, up to 347 segments), at 6 font sizes -> 256^2 cells
    "real_dejavuserif_italic_whole_font_256px_16spp": dict(glyphs=0, cell=256, segs=0, n=4, cols=64, gen="font", font=DEJAVU),
    # the reference's own products (one sample per pixel, the pixel corner): renderGlyph's value map, Image.Winding
    "c3_cjk21k_256px_s128_gray_debug": dict(glyphs=20992, cell=256, segs=128, n=1, cols=64, mode="gray_debug"),
    "c3_cjk21k_256px_s128_winding_i16": dict(glyphs=20992, cell=256, segs=128, n=1, cols=64, mode="winding_i16"),
    # the reference's own product on the real font: renderGlyph's gray map of every DejaVuSerif-Italic glyph, 6 sizes
    "real_dejavuserif_italic_whole_font_256px_gray_debug": dict(glyphs=0, cell=256, segs=0, n=1, cols=64, gen="font", font=DEJAVU, mode="gray_debug"),
    # BASELINE.json configs[4] as stated: per-pixel SDF at 512 x 512 per glyph, the WHOLE font (every loadable glyph once)
    "real_dejavuserif_italic_whole_font_512px_sdf": dict(glyphs=0, cell=512, segs=0, n=1, cols=16, gen="font", font=DEJAVU, mode="sdf", size_factors=(1.0,)),
    # the reference's literal product in batch (render_glyph.zig:11-33): every glyph of the font at EXACTLY renderGlyph's own
    # image size for font_size 64 (47 x 45-ish images), shelf-packed tight into one atlas — ragged cells on the fast kernels
    "real_dejavuserif_italic_renderglyph_dims_size64_gray_debug": dict(glyphs=0, cell=0, segs=0, n=1, cols=0, gen="font_dims", font=DEJAVU, mode="gray_debug", sizes=(64,), atlas_w=4096),
    "real_dejavuserif_italic_renderglyph_dims_size64_16spp": dict(glyphs=0, cell=0, segs=0, n=4, cols=0, gen="font_dims", font=DEJAVU, sizes=(64,), atlas_w=4096),
    # the same product over the sizes a text renderer asks for (12 ... 256): 12 images per glyph
    "real_dejavuserif_italic_renderglyph_dims_sizes12to256_gray_debug": dict(glyphs=0, cell=0, segs=0, n=1, cols=0, gen="font_dims", font=DEJAVU, mode="gray_debug", sizes=(12, 16, 20, 24, 32, 48, 64, 96, 128, 160, 200, 256), atlas_w=8192),
    "real_dejavuserif_italic_renderglyph_dims_sizes12to256_16spp": dict(glyphs=0, cell=0, segs=0, n=4, cols=0, gen="font_dims", font=DEJAVU, sizes=(12, 16, 20, 24, 32, 48, 64, 96, 128, 160, 200, 256), atlas_w=8192),
}


def build_inputs(wl, rank, lo, hi):
    """-> (GlyphSet, jobs, (H, W)) for this rank's glyph range [lo, hi) of the workload"""
    from font_renderer_amd.atlas import atlas_shape, cell_jobs, glyph_dims_jobs
    from font_renderer_amd.synth import stroke_glyphset, synth_glyphset
    cell, cols, gen = wl["cell"], wl["cols"], wl.get("gen", "synth")
    if gen == "font_dims":
        import font_renderer_amd as fr
        font = fr.Font.initTTF(wl["font"])
        gs, kept = font.glyphset()
        upm = font.information.units_per_em
        W = wl["atlas_w"]
        parts, y = [], 0
        for size in wl["sizes"]:                                   # one run of shelves per font size
            jb, h = glyph_dims_jobs(gs, size, upm, W)
            jb["out_y"] += y
            y += h
            parts.append(jb)
        return gs, np.concatenate(parts), (y, W)
    if gen == "font":
        import font_renderer_amd as fr
        font = fr.Font.initTTF(wl["font"])
        gs, kept = font.glyphset()
        upm = font.information.units_per_em
        sizes = [int(cell * f) for f in wl.get("size_factors", (1.0, 0.9, 0.8, 0.7, 0.6, 0.5))]
        jobs = np.concatenate([cell_jobs(gs, cell, s, upm, cols) for s in sizes])
        k = np.arange(len(jobs))
        jobs["out_x"] = (k % cols) * cell
        jobs["out_y"] = (k // cols) * cell
        return gs, jobs, atlas_shape(len(jobs), cell, cols)
    if gen == "ascii":
        import fixtures
        asc = fixtures.load_ascii()
        jobs = cell_jobs(asc.gs, cell, 100, asc.g_upm, cols, first_glyph=0, n_glyphs=95)
        return asc.gs, jobs, atlas_shape(95, cell, cols)
    make = stroke_glyphset if gen == "stroke" else synth_glyphset
    gs = make(hi - lo, wl["segs"], first_index=lo)
    if gen == "pages":
        # pages of `per_page` glyphs on a cols x page_rows cell grid, stacked one under the other in ONE buffer
        per, pr = wl["per_page"], wl["page_rows"]
        n_pages = (hi - lo + per - 1) // per
        parts = []
        for pg in range(n_pages):
            a, b = pg * per, min((pg + 1) * per, hi - lo)
            jb = cell_jobs(gs, cell, cell, 2048, cols, first_glyph=a, n_glyphs=b - a)
            jb["out_y"] += pg * pr * cell
            parts.append(jb)
        return gs, np.concatenate(parts), (n_pages * pr * cell, cols * cell)
    return gs, cell_jobs(gs, cell, cell, 2048, cols), atlas_shape(hi - lo, cell, cols)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20, help="untimed steps before the timed region (after 0.25 s of untimed renders that bring the clocks up)")
    ap.add_argument("--workload", default="c3_cjk21k_256px_s128_16spp", choices=sorted(WORKLOADS))
    ap.add_argument("--glyphs", type=int, default=0, help="override the per-GPU glyph count (smoke runs)")
    ap.add_argument("--total-glyphs", type=int, default=0, help="strong scaling: the whole job's glyph count, sharded over the ranks")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: skip the separately timed all-gather of the atlas bands")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--kmax", type=int, default=0, help="crossings per sample row kept in registers (8/16/32)")
    ap.add_argument("--n", type=int, default=0, help="override samples per axis (experiments)")
    ap.add_argument("--segs", type=int, default=0, help="override segments per glyph (experiments)")
    ap.add_argument("--opt", action="append", default=[], help="library tuning knob key=value (experiments)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: bench.py runs ONE rank per process; launch N > 1 as "
                         f"`python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 "
                         f"--master-port P bench.py --gpus {args.gpus} ...`")

    import torch
    import torch.distributed as dist

    ndev = torch.cuda.device_count()
    if ndev == 0:
        raise SystemExit("bench.py needs a GPU: the library has no CPU path")
    backend = os.environ.get("FR_BENCH_BACKEND", "nccl")      # "gloo": rehearsal of the N > 1 path on one GPU
    if backend == "nccl" and world > ndev:
        raise SystemExit(f"{world} ranks but {ndev} GPUs")
    local = local % ndev
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    import font_renderer_amd as fr
    from font_renderer_amd.shard import shard_range

    wl = dict(WORKLOADS[args.workload])
    if args.glyphs:
        wl["glyphs"] = args.glyphs
    if args.n:
        wl["n"] = args.n
    if args.segs:
        wl["segs"] = args.segs
    cell, S, n, cols = wl["cell"], wl["segs"], wl["n"], wl["cols"]
    mode_name = wl.get("mode", "coverage")
    mode = {"coverage": fr.FR_COVERAGE_U8, "sdf": fr.FR_SDF_U8, "gray_debug": fr.FR_GRAY_DEBUG, "winding_i16": fr.FR_WINDING_I16}[mode_name]
    bpp = 2 if mode == fr.FR_WINDING_I16 else 1
    phase = fr.FR_SAMPLE_CORNER if mode_name in ("gray_debug", "winding_i16") else fr.FR_SAMPLE_CENTER
    strong = args.total_glyphs > 0
    if strong:
        lo, hi = shard_range(args.total_glyphs, rank, world)
        total_glyphs = args.total_glyphs
    else:
        lo, hi = rank * wl["glyphs"], (rank + 1) * wl["glyphs"]
        total_glyphs = wl["glyphs"] * world

    # ---- inputs: this rank's glyph range
    t_gen = time.time()
    gs, jobs, (H, W) = build_inputs(wl, rank, lo, hi)
    t_gen = time.time() - t_gen
    G = len(jobs)
    stream = torch.cuda.Stream()
    ctx = fr.Context(local, stream.cuda_stream)
    if args.kmax:
        ctx.set_option("kmax", args.kmax)
    for kv in args.opt:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    # the atlas band lives inside the gather destination when the gather is timed (no staging copy)
    gbuf = None
    if world > 1 and not args.no_gather:
        # every rank's band has the same shape in every workload here (equal glyph counts, or the same font): W equal slots
        with torch.cuda.stream(stream):
            gbuf = torch.empty((world * H, W), dtype=torch.int16 if bpp == 2 else torch.uint8, device="cuda")
            pad = H
            out = gbuf[rank * pad:rank * pad + H]
            out.zero_()
    else:
        with torch.cuda.stream(stream):
            out = torch.zeros((H, W), dtype=torch.int16 if bpp == 2 else torch.uint8, device="cuda")
    dgs = fr.DeviceGlyphSet(ctx, gs)                       # points -> HBM (+ first precompute)
    plan = fr.Plan(dgs, jobs, mode, n, phase)
    pixels = plan.pixels
    stats = dgs.stats()
    pstats = plan.stats()

    def step():
        # one render = everything from the glyph points to the atlas (the root records are rebuilt
        # inside the render kernels, or by the precompute kernel the library launches with them)
        plan.render(out.data_ptr(), W, H)

    def barrier():
        if world > 1:
            dist.barrier()

    # the GPU's clocks come up over the first ~0.1 s of load (a timed region right after an idle start runs up to 11 %
    # slower than the kernel: DESIGN.md section 7): 0.25 s of untimed renders first, whatever --warmup says
    t_ramp = time.perf_counter()
    ramp_steps = 0
    while time.perf_counter() - t_ramp < RAMP_S:
        for _ in range(8):
            step()
        ramp_steps += 8
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)                                       # HIP events on the LAUNCH stream around the K back-to-back launches
    for _ in range(args.steps):
        step()
    ev1.record(stream)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gpu_ms = ev0.elapsed_time(ev1) / args.steps              # average launch duration over the timed region (sustained: back to back)
    tot_pixels = pixels
    if world > 1:
        dev = "cuda" if backend == "nccl" else "cpu"
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        p = torch.tensor([pixels], dtype=torch.int64, device=dev)
        dist.all_reduce(p, op=dist.ReduceOp.SUM)
        tot_pixels = int(p.item())
    ms_per_step = dt / args.steps * 1e3
    value = tot_pixels / (dt / args.steps) / 1e6            # Mpixel/s, whole job

    # ---- optional assembly of the atlas bands, timed on its own ("gather reported separately")
    gather = None
    if gbuf is not None:
        from font_renderer_amd.shard import gather_bands
        gather = {}
        for form, root in (("to_root0", 0), ("all", None)):
            try:
                with torch.cuda.stream(stream):
                    for _ in range(2):
                        gather_bands(out, gbuf, root=root)
                    torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
                    tg = time.perf_counter()
                    reps = 10
                    for _ in range(reps):
                        gather_bands(out, gbuf, root=root)
                    torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
                    tg = (time.perf_counter() - tg) / reps
                t = torch.tensor([tg], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                tg = float(t.item())
                gbytes = gbuf.numel() * gbuf.element_size()
                moved = gbytes * (world - 1) // world if root is not None else gbytes * (world - 1)
                gather[form] = {"ms": round(tg * 1e3, 4), "atlas_GBps": round(gbytes / tg / 1e9, 2), "atlas_bytes": gbytes,
                                "bytes_over_xgmi": moved,
                                "collective": ("dist.gather onto rank 0 (RCCL: a group of point-to-point transfers, each peer's band over its own link)"
                                               if root is not None else "all_gather_into_tensor (every rank receives the whole atlas)") + "; not part of `value`"}
            except Exception as e:                           # the optional assembly must never cost the throughput line
                gather[form] = {"error": f"{type(e).__name__}: {e}"}

    # ---- roofline of the dominant kernel: its launch duration, HIP events on the launch stream around each launch
    # (fr_plan_render_timed) — what rocprofv3 --kernel-trace reports for the kernel (profiles/r02/*_kernel_stats.csv).
    # Beside it the launch PERIOD of the timed region (the events above / steps): the same number once the clocks are up
    # (they take ~100 launches: with --warmup 5 a 0.29 ms kernel ran at a period of 0.32 ms).
    kms = sorted(plan.render_timed(out.data_ptr(), W, H) for _ in range(max(5, min(args.steps, 20))))
    launches = plan.describe()                                 # the instances this plan launches, as rocprofv3 names them
    k_ms = float(np.mean(kms))
    achieved = pixels * bpp / (k_ms * 1e-3) / 1e9             # GB/s of algorithmic bytes (1 B / pixel; 2 for int16 windings)
    traffic, traffic_src = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath) and not args.glyphs and not strong:
        try:
            ent = json.load(open(tpath)).get(args.workload, {})
            if ent.get("build_id") == fr.build_id():          # counters of THIS kernel build only
                traffic, traffic_src = ent.get("hbm_bytes_per_launch"), ent.get("source")
        except Exception:
            traffic = None
    # the dominant kernel = the launch with the most jobs (SDF: the distance kernel, which follows the sign pass)
    parts = [q.rsplit(" x", 1) for q in launches.split("; ")]
    kname = ("fr::sdf_kernel" + parts[-1][0].split("fr::sdf_kernel")[1]) if mode == fr.FR_SDF_U8 else max(parts, key=lambda q: int(q[1]))[0]
    roofline = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "launches": launches, "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": pixels * bpp, "kernel_ms": round(k_ms, 4), "period_ms": round(float(gpu_ms), 4),
                "frac_of_period": round(pixels * bpp / (gpu_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                "timing": "kernel_ms: HIP events around each fr_plan_render (all of `launches`; == the rocprofv3 kernel duration when there is one launch); period_ms: HIP events around the timed region's back-to-back renders / steps",
                "build_id": fr.build_id()}

    # ---- configs[0]: the reference's own call shape — one glyph, one image (STIX 'A' at 64 -> 47 x 45), latency per call
    c1 = None
    if rank == 0 and world == 1:
        try:
            import ctypes as C
            import fixtures
            from font_renderer_amd import _lib as L
            asc = fixtures.load_ascii()
            i = asc.find("STIX", "A")
            g, info = asc.glyph(i), fr.FontInformation(int(asc.g_upm[i]))
            im = fr.renderGlyph(g, info, 64, ctx=ctx)                  # (also sizes the context's scratch)
            gs1 = fr.GlyphSet([g])
            pts, cs, box = gs1.points_xy, gs1.contour_start, g.box.as_array()
            buf = np.zeros(im.width * im.height, np.uint8)
            lib = L.load_library()
            call = lambda: lib.fr_render_glyph(ctx._h, L.ptr(pts), L.ptr(cs), len(cs) - 1, L.ptr(box), info.units_per_em, 64, fr.FR_GRAY_DEBUG, L.ptr(buf))
            for _ in range(300):
                call()
            tc = time.perf_counter()
            for _ in range(1000):
                rc1 = call()
            us = (time.perf_counter() - tc) / 1000 * 1e6
            assert rc1 == 0 and np.array_equal(buf, im.data)
            c1 = {"config": "configs[0]: STIX 'A', font_size 64 -> 47x45 gray; one fr_render_glyph call (upload, render, download, sync)",
                  "us_per_call": round(us, 1), "size": [im.width, im.height]}
        except Exception as e:                               # fixtures are test data: absent -> no C1 line
            c1 = {"error": str(e)}

    # ---- CPU baseline: the oracle (restated reference algorithm), rank 0, N = 1 only
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle_lib
        from font_renderer_amd.atlas import atlas_shape
        orc = oracle_lib.Oracle()
        threads = min(len(os.sched_getaffinity(0)), 16)   # the GPU box grants 16 cores per GPU
        omode = {"coverage": oracle_lib.COVERAGE_U8, "sdf": oracle_lib.SDF_U8, "gray_debug": oracle_lib.GRAY_DEBUG,
                 "winding_i16": oracle_lib.WINDING_I16}[mode_name]
        odt = np.int16 if bpp == 2 else np.uint8
        center = phase == fr.FR_SAMPLE_CENTER
        Wd = int(W)

        def sample(k, th):
            """the oracle on the first k jobs (their own rectangles of an atlas as wide as the GPU's) -> (seconds, buffer)"""
            rows = int((jobs["out_y"][:k].astype(np.int64) + jobs["h"][:k]).max())
            buf = np.zeros((rows, Wd), odt)
            t = time.perf_counter()
            orc.render_batch(gs, jobs[:k], omode, buf, n, center, th)
            return time.perf_counter() - t, buf

        probe = min(2, G)
        per_job = sample(probe, 1)[0] / probe
        px = jobs["w"].astype(np.int64) * jobs["h"]
        ng = int(max(min(threads, G), min(G, args.cpu_seconds * threads / max(per_job, 1e-9))))
        ct, buf = sample(ng, threads)
        with torch.cuda.stream(stream):
            gpu = out[:buf.shape[0]].cpu().numpy()
        same = all(np.array_equal(gpu[j["out_y"]:j["out_y"] + j["h"], j["out_x"]:j["out_x"] + j["w"]],
                                  buf[j["out_y"]:j["out_y"] + j["h"], j["out_x"]:j["out_x"] + j["w"]]) for j in jobs[:ng])
        shape = f"{cell}x{cell}" if cell else "renderGlyph-sized"
        cpu = {"value": round(float(px[:ng].sum()) / ct / 1e6, 4), "unit": "Mpixel/s", "cores": threads, "kind": "port",
               "sample": f"first {ng} cells of the same workload ({shape}, {n * n} samples/pixel, {int(px[:ng].sum())} pixels), "
                         f"{ct:.1f} s wall on {threads} threads, oracle/fr_oracle.c (C restatement of "
                         f"render_glyph.zig, not the Zig binary)",
               "single_thread_value": round(float(px[:probe].sum()) / probe / per_job / 1e6, 4), "matches_gpu_bytes": bool(same)}

    if rank == 0:
        line = {
            "metric": "Mpixel/s AA glyph coverage", "value": round(value, 1), "unit": "Mpixel/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic" if wl.get("gen", "synth") in ("synth", "stroke") else "real font outlines (matplotlib's DejaVuSerif-Italic / the committed ASCII fixture), random-free",
            "ramp_s": RAMP_S, "ramp_steps": ramp_steps,
            "config": {"workload": args.workload, "cells_per_gpu": G, "glyphs_in_set": len(gs),
                       "cell": f"{cell}x{cell}" if cell else f"renderGlyph's own image per glyph (render_glyph.zig:14-19), font sizes {list(wl['sizes'])}",
                       "segments_per_glyph": S if S else "as in the font", "total_segments_in_set": stats["segments"],
                       "samples_per_pixel": n * n, "mode": mode_name if mode_name != "coverage" else "coverage_u8",
                       "pixels_per_step_per_gpu": pixels, "pixels_per_step_whole_job": tot_pixels,
                       "kernel_split": pstats, "step": "points -> atlas (records rebuilt every render)",
                       "parallelism": f"glyph-sharded x{world}, no collective on the data path"},
            "roofline": roofline, "cpu_baseline": cpu, "c1_single_glyph": c1, "gather": gather,
            "gpixel_per_s": round(value / 1e3, 2), "input_gen_s": round(t_gen, 2),
        }
        print(json.dumps(line), flush=True)
    plan.close(); dgs.close(); ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
