//! fr_raster.zig — Zig 0.15 binding of include/fr_raster.h for nyasyamorina/font-renderer.
//!
//! UNVERIFIED BY A COMPILER: no Zig toolchain exists in the build image (`zig version` -> not found), so
//! this file has never been compiled.  What IS checked (tests/test_abi.py::test_zig_binding_matches_the_header):
//! every function of the header is declared here with the same name, the same number of parameters and the
//! same integer / float / pointer widths, and the enums carry the header's values.  It is the binding a
//! maintainer drops into `src/tools/` next to the file it replaces (`src/tools/render_glyph.zig`); the C side
//! it declares is exercised by the Python / C++ hosts and the GPU parity tests.
//!
//! Drop-in contract: `renderGlyph` keeps the reference's signature
//!   pub fn renderGlyph(glyph: Glyph, font_info: Font.Information, font_size: u16) !Image.Gray
//! (src/tools/render_glyph.zig:11).  Memory stays with the reference's allocator: the image is
//! `Image.Gray.init` (src/tools/Image.zig:58-61, helpers.alloc) and the library only fills it.
const std = @import("std");

const Font = @import("../font/Font.zig");
const Glyph = @import("../font/Glyph.zig");
const Image = @import("Image.zig");
const Point = @import("geometry.zig").Point;

pub const fr_ctx = opaque {};
pub const fr_glyphset = opaque {};
pub const fr_plan = opaque {};
pub const fr_font = opaque {};

pub const Status = enum(c_int) { ok = 0, invalid = -1, hip = -2, nomem = -3, unsupported = -4 };
pub const Mode = enum(i32) { winding_i16 = 0, gray_debug = 1, mask_nonzero = 2, coverage_u8 = 3, sdf_u8 = 4 };
pub const SamplePhase = enum(i32) { corner = 0, center = 1 };
pub const FR_FONT_ALLOW_HINTED: u32 = 1;

pub const RasterParams = extern struct {
    mode: i32,
    samples_per_axis: i32,
    sample_phase: i32,
    reserved: i32 = 0,
};

pub const Job = extern struct {
    glyph: u32,
    min_x: i32,
    max_y: i32,
    w: u32,
    h: u32,
    out_x: u32,
    out_y: u32,
    scale: f32,
};

// ---- library / context
pub extern "c" fn fr_abi_version() c_int;
pub extern "c" fn fr_last_error() [*:0]const u8;
pub extern "c" fn fr_build_id() [*:0]const u8;
pub extern "c" fn fr_ctx_create(device: c_int, hip_stream: ?*anyopaque, out: *?*fr_ctx) c_int;
pub extern "c" fn fr_ctx_destroy(ctx: ?*fr_ctx) void;
pub extern "c" fn fr_ctx_sync(ctx: *fr_ctx) c_int;
pub extern "c" fn fr_ctx_set_option(ctx: *fr_ctx, key: [*:0]const u8, value: i64) c_int;
// ---- glyph sets
pub extern "c" fn fr_glyphset_create(ctx: *fr_ctx, points_xy: [*]const i16, contour_start: [*]const u32, n_contours: u32, glyph_start: [*]const u32, n_glyphs: u32, out: *?*fr_glyphset) c_int;
pub extern "c" fn fr_glyphset_destroy(gs: ?*fr_glyphset) void;
pub extern "c" fn fr_glyphset_prepare(gs: *fr_glyphset) c_int;
pub extern "c" fn fr_glyphset_stats(gs: *const fr_glyphset, n_segments: ?*u64, n_records: ?*u64) c_int;
// ---- batched rasterization
pub extern "c" fn fr_plan_create(ctx: *fr_ctx, gs: *const fr_glyphset, jobs: [*]const Job, n_jobs: u32, params: *const RasterParams, out: *?*fr_plan) c_int;
pub extern "c" fn fr_plan_destroy(plan: ?*fr_plan) void;
pub extern "c" fn fr_plan_render(plan: *fr_plan, out_dev: *anyopaque, out_stride: usize, out_rows: usize) c_int;
pub extern "c" fn fr_plan_render_timed(plan: *fr_plan, out_dev: *anyopaque, out_stride: usize, out_rows: usize, ms: *f32) c_int;
pub extern "c" fn fr_plan_pixels(plan: *const fr_plan) u64;
pub extern "c" fn fr_plan_stats(plan: *const fr_plan, n_jobs_cov4: ?*u32, n_jobs_general: ?*u32) c_int;
pub extern "c" fn fr_plan_describe(plan: *const fr_plan, buf: [*]u8, cap: usize) c_int;
pub extern "c" fn fr_allgather_bands(ctx: *fr_ctx, nccl_comm: *anyopaque, atlas_dev: *anyopaque, band_bytes: usize) c_int;
pub extern "c" fn fr_gather_bands(ctx: *fr_ctx, nccl_comm: *anyopaque, atlas_dev: *anyopaque, band_bytes: usize, root: c_int) c_int;
pub extern "c" fn fr_render_batch(ctx: *fr_ctx, gs: *const fr_glyphset, jobs: [*]const Job, n_jobs: u32, params: *const RasterParams, out_host: *anyopaque, out_stride: usize, out_rows: usize) c_int;
// ---- renderGlyph drop-in
pub extern "c" fn fr_render_glyph_dims(box: *const [4]i16, units_per_em: u16, font_size: u16, min_corner: *[2]i16, max_corner: *[2]i16, width: *u16, height: *u16, scale: ?*f32) c_int;
pub extern "c" fn fr_render_glyph(ctx: *fr_ctx, points_xy: [*]const i16, contour_start: [*]const u32, n_contours: u32, box: *const [4]i16, units_per_em: u16, font_size: u16, mode: i32, out_host: *anyopaque) c_int;
// ---- exact-integer path
pub extern "c" fn fr_glyph_info_init(ctx: *fr_ctx, points_xy: [*]const i16, contour_start: [*]const u32, n_contours: u32, curve_type: [*]u8, include_p0: [*]u8) c_int;
pub extern "c" fn fr_winding_in_glyph(ctx: *fr_ctx, points_xy: [*]const i16, contour_start: [*]const u32, n_contours: u32, query_xy: [*]const i16, n_query: u32, out_winding: [*]i16) c_int;
pub extern "c" fn fr_winding_lattice(ctx: *fr_ctx, points_xy: [*]const i16, contour_start: [*]const u32, n_contours: u32, box: *const [4]i16, out_host: [*]i16) c_int;
pub extern "c" fn fr_glyph_debug_render(ctx: *fr_ctx, points_xy: [*]const i16, contour_start: [*]const u32, n_contours: u32, box: *const [4]i16, winding_scale: u8, rgb_host: [*]u8) c_int;
// build-defined (no reference counterpart): the exact-integer path on a K-times refined lattice
pub extern "c" fn fr_exact_lattice(ctx: *fr_ctx, points_xy: [*]const i16, contour_start: [*]const u32, n_contours: u32, k: u32, x0: i32, y0: i32, w: u32, h: u32, out_host: [*]i16) c_int;
pub extern "c" fn fr_exact_coverage(ctx: *fr_ctx, points_xy: [*]const i16, contour_start: [*]const u32, n_contours: u32, k: u32, x0: i32, y0: i32, w_px: u32, h_px: u32, n: u32, out_host: [*]u8) c_int;
// ---- atlas layout (host side)
pub extern "c" fn fr_atlas_layout(boxes: [*]const i16, n_glyphs: u32, first_glyph: u32, units_per_em: [*]const u16, n_upm: u32, font_size: u16, cell: u32, cols: u32, rows_per_page: u32, jobs_out: [*]Job, page_of_job: ?[*]u32, n_pages: ?*u32) c_int;
pub extern "c" fn fr_atlas_layout_glyph_dims(boxes: [*]const i16, n_glyphs: u32, first_glyph: u32, units_per_em: [*]const u16, n_upm: u32, font_size: u16, atlas_w: u32, @"align": u32, jobs_out: [*]Job, atlas_h: ?*u32) c_int;
// ---- contour producer (host side; the Zig host has font/Font.zig and does not need it)
pub extern "c" fn fr_font_open(ttf_bytes: *const anyopaque, len: usize, flags: u32, out: *?*fr_font) c_int;
pub extern "c" fn fr_font_close(font: ?*fr_font) void;
pub extern "c" fn fr_font_info(font: *const fr_font, units_per_em: ?*u16, num_glyphs: ?*u16, y0_baseline: ?*c_int) c_int;
pub extern "c" fn fr_font_char_to_glyph(font: *const fr_font, codepoint: u32, glyph_index: *u16) c_int;
pub extern "c" fn fr_font_glyph_advance(font: *const fr_font, glyph_index: u16, advance_width: *i16) c_int;
pub extern "c" fn fr_font_glyph_measure(font: *fr_font, glyph_index: u16, n_contours: *u32, n_points: *u32, box: *[4]i16) c_int;
pub extern "c" fn fr_font_glyph_fill(font: *fr_font, glyph_index: u16, points_xy: [*]i16, contour_start: [*]u32) c_int;
// ---- QOI writer (host side; byte-compatible with tools/qoi.zig, which the Zig host keeps)
pub extern "c" fn fr_qoi_bound(width: u32, height: u32) usize;
pub extern "c" fn fr_qoi_encode_rgb(rgb: [*]const u8, width: u32, height: u32, out: [*]u8, cap: usize, n_out: *usize) c_int;
pub extern "c" fn fr_qoi_encode_gray(gray: [*]const u8, width: u32, height: u32, stride: usize, out: [*]u8, cap: usize, n_out: *usize) c_int;
// ---- self-tests of the two arithmetic shortcuts (device-side, exhaustive)
pub extern "c" fn fr_selftest_division(d_lo: u32, d_hi: u32, mismatches: *u64, bad_divisor: ?*u32, bad_x_bits: ?*u32) c_int;
pub extern "c" fn fr_selftest_sqrt(mismatches: *u64, bad_x_bits: ?*u32) c_int;

pub const Error = error{ RasterFailed, OutOfMemory };

var g_ctx: ?*fr_ctx = null;

fn context() Error!*fr_ctx {
    if (g_ctx == null) {
        if (fr_ctx_create(0, null, &g_ctx) != 0) return error.RasterFailed;
    }
    return g_ctx.?;
}

/// Releases the process-wide context (its stream, device arena and pinned staging block).  Call once from the
/// host's shutdown path, next to Font.deinit (src/main.zig:33); a later renderGlyph creates a fresh one.
pub fn deinit() void {
    if (g_ctx) |c| fr_ctx_destroy(c);
    g_ctx = null;
}

/// Flat view of a Glyph: every contour's points live in ONE allocation in contour order
/// (src/font/Glyph.zig:89-96), and Point(i16) is an extern struct {x, y} (geometry.zig:7-11),
/// so `contours[0].points.ptr` is already the i16 (x,y) array the C ABI wants — no copy.
/// Only the contour offsets are built here.
fn flatten(glyph: Glyph, starts: []u32) [*]const i16 {
    starts[0] = 0;
    for (glyph.contours, 0..) |contour, i| starts[i + 1] = starts[i] + @as(u32, @intCast(contour.points.len));
    if (glyph.contours.len == 0) return @ptrCast(&[_]i16{ 0, 0 });
    return @ptrCast(glyph.contours[0].points.ptr);
}

fn boxOf(glyph: Glyph) [4]i16 {
    return .{ glyph.box.x_min, glyph.box.y_min, glyph.box.x_max, glyph.box.y_max };
}

/// Drop-in for src/tools/render_glyph.zig:11 — same signature, same bytes.
pub fn renderGlyph(glyph: Glyph, font_info: Font.Information, font_size: u16) !Image.Gray {
    const helpers = @import("../helpers.zig");
    const ctx = try context();
    const box = boxOf(glyph);
    var mn: [2]i16 = undefined;
    var mx: [2]i16 = undefined;
    var w: u16 = 0;
    var h: u16 = 0;
    if (fr_render_glyph_dims(&box, font_info.units_per_em, font_size, &mn, &mx, &w, &h, null) != 0) return error.RasterFailed;

    var im: Image.Gray = .init(w, h); // render_glyph.zig:22 — the caller's allocator owns the pixels
    errdefer im.deinit();

    const starts = helpers.alloc(u32, glyph.contours.len + 1);
    defer helpers.allocator.free(starts);
    const pts = flatten(glyph, starts);
    const rc = fr_render_glyph(ctx, pts, starts.ptr, @intCast(glyph.contours.len), &box, font_info.units_per_em, font_size, @intFromEnum(Mode.gray_debug), im.data.ptr);
    if (rc != 0) return error.RasterFailed;
    return im;
}

/// The same grid as renderGlyph, holding the winding numbers themselves: Image.Winding
/// (src/tools/Image.zig:85-130; `scaler` / `overflow_color` are the display parameters of its colour map).
pub fn renderGlyphWinding(glyph: Glyph, font_info: Font.Information, font_size: u16, scaler: u8, overflow_color: u8) !Image.Winding {
    const helpers = @import("../helpers.zig");
    const ctx = try context();
    const box = boxOf(glyph);
    var mn: [2]i16 = undefined;
    var mx: [2]i16 = undefined;
    var w: u16 = 0;
    var h: u16 = 0;
    if (fr_render_glyph_dims(&box, font_info.units_per_em, font_size, &mn, &mx, &w, &h, null) != 0) return error.RasterFailed;

    var im: Image.Winding = .init(w, h, scaler, overflow_color); // Image.zig:101-104
    errdefer im.deinit();

    const starts = helpers.alloc(u32, glyph.contours.len + 1);
    defer helpers.allocator.free(starts);
    const pts = flatten(glyph, starts);
    const rc = fr_render_glyph(ctx, pts, starts.ptr, @intCast(glyph.contours.len), &box, font_info.units_per_em, font_size, @intFromEnum(Mode.winding_i16), im.data.ptr);
    if (rc != 0) return error.RasterFailed;
    return im;
}

/// Image.GlyphDebug.render's lattice (src/tools/Image.zig:227-236) through the exact-integer
/// path: fills `out` ((x_max-x_min+3) * (y_max-y_min+3) i16) with windingInGlyph values.
pub fn windingLattice(glyph: Glyph, out: []i16) !void {
    const helpers = @import("../helpers.zig");
    const ctx = try context();
    const box = boxOf(glyph);
    const starts = helpers.alloc(u32, glyph.contours.len + 1);
    defer helpers.allocator.free(starts);
    const pts = flatten(glyph, starts);
    if (fr_winding_lattice(ctx, pts, starts.ptr, @intCast(glyph.contours.len), &box, out.ptr) != 0) return error.RasterFailed;
}

/// Drop-in for Image.GlyphDebug.render (src/tools/Image.zig:220-240): the coloured lattice with the glyph's points
/// marked, written into the Image.RGB the reference's own init allocates (Image.zig:181-190).
pub fn glyphDebugRender(glyph: Glyph, winding_scale: u8) !Image.GlyphDebug {
    const helpers = @import("../helpers.zig");
    const ctx = try context();
    const box = boxOf(glyph);
    var im: Image.GlyphDebug = .init(glyph.box, winding_scale, 150, .{ 255, 255, 0 }, .{ 0, 255, 255 });
    errdefer im.rgb.deinit();
    const starts = helpers.alloc(u32, glyph.contours.len + 1);
    defer helpers.allocator.free(starts);
    const pts = flatten(glyph, starts);
    if (fr_glyph_debug_render(ctx, pts, starts.ptr, @intCast(glyph.contours.len), &box, winding_scale, @ptrCast(im.rgb.data.ptr)) != 0) return error.RasterFailed;
    return im;
}

/// A batch of glyphs into one atlas page (build-defined cell grid: fr_atlas_layout): `glyphs` are flattened into
/// one points array (one copy — they live in separate allocations of Font.glyphs), rendered as n x n-sample
/// coverage into a `cols * cell` wide Image.Gray of ceil(len / cols) cell rows.
pub fn renderAtlas(glyphs: []const Glyph, font_info: Font.Information, font_size: u16, cell: u32, cols: u32, samples_per_axis: i32) !Image.Gray {
    const helpers = @import("../helpers.zig");
    const ctx = try context();
    var n_points: usize = 0;
    var n_contours: usize = 0;
    for (glyphs) |g| {
        n_contours += g.contours.len;
        for (g.contours) |c| n_points += c.points.len;
    }
    const pts = helpers.alloc(i16, 2 * @max(n_points, 1));
    defer helpers.allocator.free(pts);
    const cstart = helpers.alloc(u32, n_contours + 1);
    defer helpers.allocator.free(cstart);
    const gstart = helpers.alloc(u32, glyphs.len + 1);
    defer helpers.allocator.free(gstart);
    const boxes = helpers.alloc(i16, 4 * @max(glyphs.len, 1));
    defer helpers.allocator.free(boxes);
    var p: usize = 0;
    var c_i: usize = 0;
    cstart[0] = 0;
    gstart[0] = 0;
    for (glyphs, 0..) |g, gi| {
        for (g.contours) |c| {
            for (c.points) |pt| {
                pts[2 * p] = pt.x;
                pts[2 * p + 1] = pt.y;
                p += 1;
            }
            c_i += 1;
            cstart[c_i] = @intCast(p);
        }
        gstart[gi + 1] = @intCast(c_i);
        const b = boxOf(g);
        @memcpy(boxes[4 * gi .. 4 * gi + 4], &b);
    }
    const jobs = helpers.alloc(Job, @max(glyphs.len, 1));
    defer helpers.allocator.free(jobs);
    const upm = [1]u16{font_info.units_per_em};
    if (fr_atlas_layout(boxes.ptr, @intCast(glyphs.len), 0, &upm, 1, font_size, cell, cols, 0, jobs.ptr, null, null) != 0) return error.RasterFailed;

    var gs: ?*fr_glyphset = null;
    if (fr_glyphset_create(ctx, pts.ptr, cstart.ptr, @intCast(n_contours), gstart.ptr, @intCast(glyphs.len), &gs) != 0) return error.RasterFailed;
    defer fr_glyphset_destroy(gs);

    const rows: u32 = @intCast((glyphs.len + cols - 1) / cols);
    var im: Image.Gray = .init(@intCast(cols * cell), @intCast(rows * cell));
    errdefer im.deinit();
    @memset(im.data, 0);
    const prm = RasterParams{ .mode = @intFromEnum(Mode.coverage_u8), .samples_per_axis = samples_per_axis, .sample_phase = @intFromEnum(SamplePhase.center) };
    if (fr_render_batch(ctx, gs.?, jobs.ptr, @intCast(glyphs.len), &prm, im.data.ptr, cols * cell, rows * cell) != 0) return error.RasterFailed;
    return im;
}
