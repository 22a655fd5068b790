"""CPU suite: the N > 1 path with world_size-2 gloo.

Each rank owns a contiguous glyph range, renders ITS glyph subset into ITS atlas band (here
with the oracle standing in for the GPU: the sharding/gather logic is what is under test),
then one all_gather assembles the atlas.  The result must equal the unsharded render byte
for byte.  No collective touches the render path itself."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, initfile, outdir, n_glyphs, cell, cols, use_gpu=False):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    from font_renderer_amd.atlas import atlas_shape, cell_jobs
    from font_renderer_amd.shard import gather_atlas, shard_range
    from font_renderer_amd.synth import synth_glyphset
    dist.init_process_group("gloo", init_method=f"file://{initfile}", rank=rank, world_size=world)
    try:
        gs = synth_glyphset(n_glyphs, 24)
        lo, hi = shard_range(n_glyphs, rank, world)
        sub = gs.subset(lo, hi)                                  # only this rank's points travel to "its GPU"
        jobs = cell_jobs(sub, cell, cell, 2048, cols)
        band = np.zeros(atlas_shape(hi - lo, cell, cols), np.uint8)
        if use_gpu:
            # both ranks drive the LIBRARY (each its own fr_ctx) on the one GPU of the box; the gather stays gloo
            import font_renderer_amd as fr
            from font_renderer_amd import render_glyph as rg
            with fr.Context(0) as ctx:
                dgs = fr.DeviceGlyphSet(ctx, sub)
                rg.render_batch(dgs, jobs, fr.FR_COVERAGE_U8, band, 2, fr.FR_SAMPLE_CENTER)
                dgs.close()
        else:
            import oracle_lib
            oracle_lib.Oracle().render_batch(sub, jobs, oracle_lib.COVERAGE_U8, band, 2, True)
        full = gather_atlas(torch.from_numpy(band), n_glyphs, cell, cols)
        np.save(os.path.join(outdir, f"rank{rank}.npy"), full.numpy())
        # the gather-to-root form: equal bands into views of the ROOT's buffer; the other rank gets nothing
        from font_renderer_amd.shard import band_rows, gather_bands
        pad = max(band_rows(n_glyphs, cell, cols, world))
        mine = torch.zeros((pad, cols * cell), dtype=torch.uint8)
        mine[:band.shape[0]] = torch.from_numpy(band)
        root_out = torch.full((world * pad, cols * cell), 0x5b, dtype=torch.uint8)
        res = gather_bands(mine, root_out, root=1)
        assert (res is None) == (rank != 1)
        if rank == 1:
            np.save(os.path.join(outdir, "root1.npy"), res.numpy())
        else:
            assert bool((root_out == 0x5b).all())                # a non-root rank's buffer is left alone
    finally:
        dist.destroy_process_group()


def test_shard_ranges_cover_everything():
    from font_renderer_amd.shard import shard_ranges
    for n in (0, 1, 7, 95, 20992, 63488):
        for w in (1, 2, 3, 8):
            r = shard_ranges(n, w)
            assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def _check_two_rank_gather(oracle, n_glyphs, use_gpu):
    from font_renderer_amd.atlas import atlas_shape, cell_jobs
    from font_renderer_amd.shard import shard_ranges
    from font_renderer_amd.synth import synth_glyphset
    cell, cols, world = 24, 4, 2
    with tempfile.TemporaryDirectory() as d:
        initfile = os.path.join(d, "init")
        mp.spawn(_worker, args=(world, initfile, d, n_glyphs, cell, cols, use_gpu), nprocs=world, join=True)
        got = [np.load(os.path.join(d, f"rank{r}.npy")) for r in range(world)]
        root1 = np.load(os.path.join(d, "root1.npy"))
    assert np.array_equal(got[0], got[1])
    # gather-to-root: rank 1's buffer holds both bands in their (padded) slots
    from font_renderer_amd.shard import band_rows
    rows, pad = band_rows(n_glyphs, cell, cols, world), max(band_rows(n_glyphs, cell, cols, world))
    at = 0
    for r in range(world):
        assert np.array_equal(root1[r * pad:r * pad + rows[r]], got[0][at:at + rows[r]]), r
        at += rows[r]
    # expected: each rank's band is a cols-wide atlas of its own glyphs, bands stacked in rank order
    import oracle_lib as O
    gs = synth_glyphset(n_glyphs, 24)
    bands = []
    for lo, hi in shard_ranges(n_glyphs, world):
        sub = gs.subset(lo, hi)
        b = np.zeros(atlas_shape(hi - lo, cell, cols), np.uint8)
        oracle.render_batch(sub, cell_jobs(sub, cell, cell, 2048, cols), O.COVERAGE_U8, b, 2, True)
        bands.append(b)
    assert np.array_equal(got[0], np.concatenate(bands, 0))
    # and a glyph rendered inside a shard equals the same glyph rendered unsharded
    whole = np.zeros(atlas_shape(n_glyphs, cell, cols), np.uint8)
    oracle.render_batch(gs, cell_jobs(gs, cell, cell, 2048, cols), O.COVERAGE_U8, whole, 2, True)
    assert np.array_equal(whole[:cell, :cell], got[0][:cell, :cell])
    lo1 = shard_ranges(n_glyphs, world)[1][0]
    r1_row0 = atlas_shape(lo1, cell, cols)[0]
    gy, gx = (lo1 // cols) * cell, (lo1 % cols) * cell
    assert np.array_equal(whole[gy:gy + cell, gx:gx + cell], got[0][r1_row0:r1_row0 + cell, :cell])


@pytest.mark.parametrize("n_glyphs", [22, 17])
def test_two_rank_gloo_gather_equals_unsharded(oracle, n_glyphs):
    """22 glyphs: 11 per rank, equal bands with ragged last rows (the gathered buffer is the atlas as it is);
    17 glyphs: 8 + 9 -> bands of 2 and 3 cell rows (the short band's slot is trimmed, in place)"""
    _check_two_rank_gather(oracle, n_glyphs, False)


@pytest.mark.gpu
def test_two_ranks_render_with_the_library_on_one_gpu(oracle):
    """the N > 1 path with the product in it: two gloo ranks, each with its own fr_ctx on the box's one GPU,
    render their shards through libfr_raster and gather; equal to the oracle's unsharded bytes"""
    _check_two_rank_gather(oracle, 22, True)


@pytest.mark.gpu
def test_allgather_bands_through_the_c_abi_with_a_one_rank_communicator():
    """fr_allgather_bands (include/fr_raster.h): the optional RCCL assembly of row bands at the C boundary.  One GPU
    is all this box has, so the communicator has ONE rank — RCCL's own C API creates it (ncclGetUniqueId /
    ncclCommInitRank through ctypes, as a C or Zig host would), the library finds that RCCL in the process, runs the
    in-place all-gather on the context's stream and leaves the rendered band as it was."""
    import ctypes as C
    import font_renderer_amd as fr
    from font_renderer_amd import _lib as L
    from font_renderer_amd.atlas import atlas_shape, cell_jobs
    from font_renderer_amd.synth import synth_glyphset
    rccl = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), mode=C.RTLD_GLOBAL)

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    uid, comm = UniqueId(), C.c_void_p()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    torch.cuda.set_device(0)
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        stream = torch.cuda.Stream()
        ctx = fr.Context(0, stream.cuda_stream)
        gs = synth_glyphset(8, 32, first_index=4242)
        jobs = cell_jobs(gs, 128, 128, 2048, 4)
        H, W = atlas_shape(8, 128, 4)
        with torch.cuda.stream(stream):
            band = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
        plan = fr.Plan(fr.DeviceGlyphSet(ctx, gs), jobs, fr.FR_COVERAGE_U8, 4, fr.FR_SAMPLE_CENTER)
        plan.render(band.data_ptr(), W, H)
        ctx.sync()
        before = band.cpu().numpy().copy()
        lib = L.load_library()
        L.check(lib.fr_allgather_bands(ctx._h, comm, C.c_void_p(band.data_ptr()), band.numel()))
        ctx.sync()
        assert np.array_equal(band.cpu().numpy(), before) and before.max() == 255
        # the gather-to-root form on the same communicator (root 0 = the only rank: nothing to receive), and root < 0 = all-gather
        L.check(lib.fr_gather_bands(ctx._h, comm, C.c_void_p(band.data_ptr()), band.numel(), 0))
        L.check(lib.fr_gather_bands(ctx._h, comm, C.c_void_p(band.data_ptr()), band.numel(), -1))
        ctx.sync()
        assert np.array_equal(band.cpu().numpy(), before)
        assert lib.fr_gather_bands(ctx._h, comm, C.c_void_p(band.data_ptr()), band.numel(), 1) != 0      # root 1 of 1 rank
        # a context without a communicator is refused, not dereferenced
        assert lib.fr_allgather_bands(ctx._h, None, C.c_void_p(band.data_ptr()), band.numel()) != 0
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)
