//! fr_raster.zig — Zig 0.15 binding of include/fr_raster.h for nyasyamorina/font-renderer.
//!
//! UNVERIFIED: no Zig toolchain exists in the build image (`zig version` -> not found), so
//! this file has never been compiled.  It is the binding a maintainer drops into
//! `src/tools/` next to the file it replaces (`src/tools/render_glyph.zig`); the C side it
//! declares is exercised by the Python/C++ hosts and the GPU parity tests.
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

pub const Mode = enum(i32) { winding_i16 = 0, gray_debug = 1, mask_nonzero = 2, coverage_u8 = 3 };
pub const SamplePhase = enum(i32) { corner = 0, center = 1 };

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

pub extern "c" fn fr_abi_version() c_int;
pub extern "c" fn fr_last_error() [*:0]const u8;
pub extern "c" fn fr_ctx_create(device: c_int, hip_stream: ?*anyopaque, out: *?*fr_ctx) c_int;
pub extern "c" fn fr_ctx_destroy(ctx: ?*fr_ctx) void;
pub extern "c" fn fr_ctx_sync(ctx: *fr_ctx) c_int;
pub extern "c" fn fr_ctx_set_option(ctx: *fr_ctx, key: [*:0]const u8, value: i64) c_int;
pub extern "c" fn fr_glyphset_create(ctx: *fr_ctx, points_xy: [*]const i16, contour_start: [*]const u32, n_contours: u32, glyph_start: [*]const u32, n_glyphs: u32, out: *?*fr_glyphset) c_int;
pub extern "c" fn fr_glyphset_destroy(gs: ?*fr_glyphset) void;
pub extern "c" fn fr_glyphset_prepare(gs: *fr_glyphset) c_int;
pub extern "c" fn fr_glyphset_stats(gs: *const fr_glyphset, n_segments: ?*u64, n_records: ?*u64) c_int;
pub extern "c" fn fr_plan_create(ctx: *fr_ctx, gs: *const fr_glyphset, jobs: [*]const Job, n_jobs: u32, params: *const RasterParams, out: *?*fr_plan) c_int;
pub extern "c" fn fr_plan_destroy(plan: ?*fr_plan) void;
pub extern "c" fn fr_plan_render(plan: *fr_plan, out_dev: *anyopaque, out_stride: usize, out_rows: usize) c_int;
pub extern "c" fn fr_plan_render_timed(plan: *fr_plan, out_dev: *anyopaque, out_stride: usize, out_rows: usize, ms: *f32) c_int;
pub extern "c" fn fr_plan_pixels(plan: *const fr_plan) u64;
pub extern "c" fn fr_render_batch(ctx: *fr_ctx, gs: *const fr_glyphset, jobs: [*]const Job, n_jobs: u32, params: *const RasterParams, out_host: *anyopaque, out_stride: usize, out_rows: usize) c_int;
pub extern "c" fn fr_render_glyph_dims(box: *const [4]i16, units_per_em: u16, font_size: u16, min_corner: *[2]i16, max_corner: *[2]i16, width: *u16, height: *u16, scale: ?*f32) c_int;
pub extern "c" fn fr_render_glyph(ctx: *fr_ctx, points_xy: [*]const i16, contour_start: [*]const u32, n_contours: u32, box: *const [4]i16, units_per_em: u16, font_size: u16, mode: i32, out_host: *anyopaque) c_int;
pub extern "c" fn fr_glyph_info_init(ctx: *fr_ctx, points_xy: [*]const i16, contour_start: [*]const u32, n_contours: u32, curve_type: [*]u8, include_p0: [*]u8) c_int;
pub extern "c" fn fr_winding_in_glyph(ctx: *fr_ctx, points_xy: [*]const i16, contour_start: [*]const u32, n_contours: u32, query_xy: [*]const i16, n_query: u32, out_winding: [*]i16) c_int;
pub extern "c" fn fr_winding_lattice(ctx: *fr_ctx, points_xy: [*]const i16, contour_start: [*]const u32, n_contours: u32, box: *const [4]i16, out_host: [*]i16) c_int;
// build-defined (no reference counterpart): the exact-integer path on a K-times refined lattice
pub extern "c" fn fr_exact_lattice(ctx: *fr_ctx, points_xy: [*]const i16, contour_start: [*]const u32, n_contours: u32, k: u32, x0: i32, y0: i32, w: u32, h: u32, out_host: [*]i16) c_int;
pub extern "c" fn fr_exact_coverage(ctx: *fr_ctx, points_xy: [*]const i16, contour_start: [*]const u32, n_contours: u32, k: u32, x0: i32, y0: i32, w_px: u32, h_px: u32, n: u32, out_host: [*]u8) c_int;

pub const Error = error{ RasterFailed, OutOfMemory };

var g_ctx: ?*fr_ctx = null;

fn context() Error!*fr_ctx {
    if (g_ctx == null) {
        if (fr_ctx_create(0, null, &g_ctx) != 0) return error.RasterFailed;
    }
    return g_ctx.?;
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

/// Drop-in for src/tools/render_glyph.zig:11 — same signature, same bytes.
pub fn renderGlyph(glyph: Glyph, font_info: Font.Information, font_size: u16) !Image.Gray {
    const helpers = @import("../helpers.zig");
    const ctx = try context();
    const box = [4]i16{ glyph.box.x_min, glyph.box.y_min, glyph.box.x_max, glyph.box.y_max };
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

/// Image.GlyphDebug.render's lattice (src/tools/Image.zig:227-236) through the exact-integer
/// path: fills `out` ((x_max-x_min+3) * (y_max-y_min+3) i16) with windingInGlyph values.
pub fn windingLattice(glyph: Glyph, out: []i16) !void {
    const helpers = @import("../helpers.zig");
    const ctx = try context();
    const box = [4]i16{ glyph.box.x_min, glyph.box.y_min, glyph.box.x_max, glyph.box.y_max };
    const starts = helpers.alloc(u32, glyph.contours.len + 1);
    defer helpers.allocator.free(starts);
    const pts = flatten(glyph, starts);
    if (fr_winding_lattice(ctx, pts, starts.ptr, @intCast(glyph.contours.len), &box, out.ptr) != 0) return error.RasterFailed;
}
