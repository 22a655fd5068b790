#!/usr/bin/env python3
"""bench.py — whole-job throughput of the hot path (AA glyph coverage) on N MI355X.

One "step" = one pass of the hot path over one batch: the per-segment precompute kernel
(root records) + the render kernel, from parsed quadratic contours resident in HBM to the
anti-aliased u8 coverage atlas resident in HBM.  Synthetic outlines; each
rank owns its own contiguous glyph range and atlas band — no collective on the data path
(weak scaling).

Prints ONE JSON line (rank 0).  See DESIGN.md §7 for every field's derivation."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

WORKLOADS = {
    # BASELINE.json configs[2]: the largest single-GPU configuration (~21k CJK-like glyphs)
    "c3_cjk21k_256px_s128_16spp": dict(glyphs=20992, cell=256, segs=128, n=4, cols=64),
    # BASELINE.json configs[1]: 95 glyphs, 128x128 cells, one 2048^2 atlas
    "c2_ascii95_128px_s32_16spp": dict(glyphs=95, cell=128, segs=32, n=4, cols=16),
    # BASELINE.json configs[3] per-GPU share (BMP/8 = 7936 glyphs, 128^2, 16 samples)
    "c4_bmp_shard_128px_s32_16spp": dict(glyphs=7936, cell=128, segs=32, n=4, cols=64),
    # BASELINE.json configs[4] per-GPU share (4 096 glyphs / 8 = 512, 512^2 cells): build-defined SDF, 1 sample
    "c5_sdf_shard_512px_s64": dict(glyphs=512, cell=512, segs=64, n=1, cols=16, mode="sdf"),
    # SURVEY §8(d) segment list {16, 32, 64, 128, 256}: the C3 shape at the other segment counts
    "c3_cjk21k_256px_s256_16spp": dict(glyphs=20992, cell=256, segs=256, n=4, cols=64),
    "c3_cjk21k_256px_s64_16spp": dict(glyphs=20992, cell=256, segs=64, n=4, cols=64),
    "c3_cjk21k_256px_s16_16spp": dict(glyphs=20992, cell=256, segs=16, n=4, cols=64),
    # stroke-dense outlines: 8-16 thin strokes per glyph, 10-30 crossings per ray (synth.stroke_glyph)
    "c3_strokes21k_256px_s128_16spp": dict(glyphs=20992, cell=256, segs=128, n=4, cols=64, gen="stroke"),
    # a real font through the C-side contour producer (fr_font_*): every glyph of DejaVuSerif-Italic the reference
    # could load (~3 000, up to 347 segments), 4 sizes each -> 256^2 cells
    "real_dejavuserif_italic_whole_font_256px_16spp": dict(glyphs=0, cell=256, segs=0, n=4, cols=64, gen="font",
                                                           font="/usr/local/lib/python3.10/dist-packages/matplotlib/mpl-data/fonts/ttf/DejaVuSerif-Italic.ttf"),
    # the reference's own products (one sample per pixel, the pixel corner): renderGlyph's value map and Image.Winding
    "c3_cjk21k_256px_s128_gray_debug": dict(glyphs=20992, cell=256, segs=128, n=1, cols=64, mode="gray_debug"),
    "c3_cjk21k_256px_s128_winding_i16": dict(glyphs=20992, cell=256, segs=128, n=1, cols=64, mode="winding_i16"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c3_cjk21k_256px_s128_16spp", choices=sorted(WORKLOADS))
    ap.add_argument("--glyphs", type=int, default=0, help="override glyph count (smoke runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--kmax", type=int, default=0, help="crossings per sample row kept in registers (8/16/32)")
    ap.add_argument("--n", type=int, default=0, help="override samples per axis (experiments)")
    ap.add_argument("--segs", type=int, default=0, help="override segments per glyph (experiments)")
    ap.add_argument("--opt", action="append", default=[], help="library tuning knob key=value (experiments)")
    ap.add_argument("--no-prep-in-step", action="store_true", help="time the render kernel alone per step")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
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
    from font_renderer_amd.atlas import atlas_shape, cell_jobs
    from font_renderer_amd.synth import synth_glyphset

    wl = dict(WORKLOADS[args.workload])
    if args.glyphs:
        wl["glyphs"] = args.glyphs
    if args.n:
        wl["n"] = args.n
    if args.segs:
        wl["segs"] = args.segs
    G, cell, S, n, cols = wl["glyphs"], wl["cell"], wl["segs"], wl["n"], wl["cols"]
    sdf = wl.get("mode") == "sdf"

    # ---- inputs: this rank's glyph range (weak scaling: G glyphs per GPU)
    t_gen = time.time()
    gs = synth_glyphset(G, S, first_index=rank * G)
    t_gen = time.time() - t_gen
    H, W = atlas_shape(G, cell, cols)
    stream = torch.cuda.Stream()
    ctx = fr.Context(local, stream.cuda_stream)
    if args.kmax:
        ctx.set_option("kmax", args.kmax)
    for kv in args.opt:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    with torch.cuda.stream(stream):
        out = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
    dgs = fr.DeviceGlyphSet(ctx, gs)                       # points -> HBM (+ first precompute)
    jobs = cell_jobs(gs, cell, cell, 2048, cols)
    plan = fr.Plan(dgs, jobs, fr.FR_SDF_U8 if sdf else fr.FR_COVERAGE_U8, n, fr.FR_SAMPLE_CENTER)
    pixels = plan.pixels
    stats = dgs.stats()

    def step():
        # one render = everything from the glyph points to the atlas (the root records are rebuilt
        # inside the render kernel, or by the precompute kernel the library launches with it)
        plan.render(out.data_ptr(), W, H)

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = pixels * world / (dt / args.steps) / 1e6        # Mpixel/s, whole job

    # ---- roofline of the dominant kernel: HIP events on the launch stream, kernel alone
    kms = sorted(plan.render_timed(out.data_ptr(), W, H) for _ in range(max(5, min(args.steps, 20))))
    k_ms = float(np.mean(kms))
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        ev0.record(stream)
        for _ in range(5):
            dgs.prepare()
        ev1.record(stream)
    torch.cuda.synchronize()
    prep_ms = ev0.elapsed_time(ev1) / 5
    achieved = pixels * 1.0 / (k_ms * 1e-3) / 1e9             # GB/s of algorithmic bytes (1 B / pixel)
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(args.workload if not args.glyphs else "", {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": "fr::render_kernel<COVERAGE_U8,1> + fr::sdf_kernel" if sdf else f"fr::render_kernel<COVERAGE_U8,{n}>",
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": traffic, "algorithmic_bytes_per_launch": pixels, "kernel_ms": round(k_ms, 4),
                "prepare_kernel_ms": round(prep_ms, 4)}

    # ---- CPU baseline: the oracle (restated reference algorithm), rank 0, N = 1 only
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle_lib
        orc = oracle_lib.Oracle()
        threads = min(len(os.sched_getaffinity(0)), 16)   # the GPU box grants 16 cores per GPU
        probe = min(2, G)
        buf = np.zeros(atlas_shape(probe, cell, cols), np.uint8)
        t = time.perf_counter()
        omode = oracle_lib.SDF_U8 if sdf else oracle_lib.COVERAGE_U8
        orc.render_batch(gs, jobs[:probe], omode, buf, n, True, 1)
        per_glyph = (time.perf_counter() - t) / probe
        ng = int(max(threads, min(G, args.cpu_seconds * threads / max(per_glyph, 1e-9))))
        unit = max(threads, cols)                      # whole atlas rows, so the sample can be compared with the GPU's bytes
        ng = min(G, (ng // unit) * unit if ng >= unit else ng)
        buf = np.zeros(atlas_shape(ng, cell, cols), np.uint8)
        t = time.perf_counter()
        orc.render_batch(gs, jobs[:ng], omode, buf, n, True, threads)
        ct = time.perf_counter() - t
        with torch.cuda.stream(stream):
            same = bool(np.array_equal(out[:buf.shape[0]].cpu().numpy(), buf)) if (ng % cols == 0 or ng <= cols) else None
        cpu = {"value": round(ng * cell * cell / ct / 1e6, 4), "unit": "Mpixel/s", "cores": threads, "kind": "port",
               "sample": f"first {ng} glyphs of the same workload ({cell}x{cell}, {n * n} samples/pixel), "
                         f"{ct:.1f} s wall on {threads} threads, oracle/fr_oracle.c (C restatement of "
                         f"render_glyph.zig, not the Zig binary)",
               "single_thread_value": round(cell * cell / per_glyph / 1e6, 4), "matches_gpu_bytes": same}

    if rank == 0:
        line = {
            "metric": "Mpixel/s AA glyph coverage", "value": round(value, 1), "unit": "Mpixel/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": args.workload, "glyphs_per_gpu": G, "cell": f"{cell}x{cell}", "segments_per_glyph": S,
                       "samples_per_pixel": n * n, "mode": "sdf_u8" if sdf else "coverage_u8", "pixels_per_step_per_gpu": pixels,
                       "root_records": stats["records"], "step": "points -> atlas (records rebuilt every render)",
                       "parallelism": f"glyph-sharded x{world}, no collective"},
            "roofline": roofline, "cpu_baseline": cpu,
            "gpixel_per_s": round(value / 1e3, 2), "input_gen_s": round(t_gen, 2),
        }
        print(json.dumps(line), flush=True)
    plan.close(); dgs.close(); ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
