//! mint_vectors.zig — mints GOLDEN VECTORS FROM THE REFERENCE ITSELF (nyasyamorina/font-renderer, Zig >= 0.15.1).
//!
//! Why this exists: the reference ships no tests, no fixtures and no golden outputs, and the image this rasterizer was
//! built in has no Zig toolchain — so oracle/fr_oracle.c is pinned to the Zig TEXT only ("parity unpinned", DESIGN.md §2).
//! This program closes that gap on any machine that has Zig 0.15: it runs the reference's OWN `renderGlyph`
//! (src/tools/render_glyph.zig:11-33) and the lattice of `Image.GlyphDebug.render` (src/tools/Image.zig:220-240 ->
//! `GlyphInfo.init` / `windingInGlyph`, render_glyph.zig:110-247) on the committed fixture and writes their outputs in the
//! format tests/test_zig_vectors.py compares the oracle with, byte for byte.  Nothing here is a restatement: every value
//! written comes out of the reference's functions.
//!
//! It must be compiled INSIDE a checkout of the reference (Zig modules cannot import files above their root directory):
//!
//!     cp tools/zig_mint/mint_vectors.zig  $REF/src/mint_vectors.zig
//!     cd $REF && zig run -O ReleaseSafe src/mint_vectors.zig -- $REPO/tests/golden/ascii_glyphs.bin $REPO/tests/golden
//!     cd $REPO && python -m pytest tests/test_zig_vectors.py -q          # oracle == Zig, or the first differing byte
//!
//! (ascii_glyphs.bin is tests/golden/ascii_glyphs.npz re-emitted by tools/zig_mint/export_fixture.py; layout below.)
//! Known obstacles, from reading the source:
//!   * `windingInGlyph` / `GlyphInfo.init` are dead code in the reference and Zig analyses lazily, so they have never been
//!     type-checked by its authors' builds: render_glyph.zig:136 and :175 apply `^` to two `bool`s, which Zig 0.15 rejects
//!     ("invalid operands to binary bitwise expression") — if so, set MINT_LATTICE = false below (renderGlyph vectors are
//!     still minted), or change `(a) ^ (b)` to `(a) != (b)` in a private copy and say so next to the vectors.
//!   * helpers.zig imports c/vk.zig (a @cImport of the Vulkan / GLFW headers) at file scope; nothing this program calls
//!     references it, so lazy analysis should leave it alone.  If your Zig insists, add `-I $VULKAN_SDK/include -lc`.
//!
//! Output files (all little-endian):
//!   zig_render_glyph.bin  "FRZG1\0\0\0" | u32 count | count x { u32 glyph, u16 font_size, u16 width, u16 height, u8[w*h] }
//!                         — glyph i of the fixture at font size (64, 33, 100, 17)[i % 4] (what the GPU suite renders)
//!   zig_lattice.bin       "FRZL1\0\0\0" | u32 count | count x { u32 glyph, u32 W, u32 H, i16[W*H] }
//!                         — windingInGlyph at (x_min + w - 1, y_max - h + 1), W = x_max - x_min + 3, H = y_max - y_min + 3
const std = @import("std");

const Font = @import("font/Font.zig");
const Glyph = @import("font/Glyph.zig");
const helpers = @import("helpers.zig");
const render_glyph = @import("tools/render_glyph.zig");
const Point = @import("tools/geometry.zig").Point(i16);

/// false: skip the integer-path lattice (see the header) and mint only the renderGlyph images
const MINT_LATTICE = true;
/// fixture indices whose GlyphDebug lattice is minted (== LATTICE_GLYPHS of tests/test_zig_vectors.py):
/// STIXGeneral 'A' (33), '&' (6), 's' (83), '8' (24) and DejaVuSerif-Italic 'i' (168) — about 4 MB of int16 in all
const LATTICE_GLYPHS = [_]u32{ 6, 24, 33, 83, 168 };
const SIZES = [_]u16{ 64, 33, 100, 17 };

const Reader = struct {
    bytes: []const u8,
    at: usize = 0,

    fn int(self: *Reader, comptime T: type) T {
        const n = @sizeOf(T);
        const v = std.mem.readInt(T, self.bytes[self.at..][0..n], .little);
        self.at += n;
        return v;
    }
};

fn put(list: *std.ArrayList(u8), gpa: std.mem.Allocator, comptime T: type, v: T) !void {
    var buf: [@sizeOf(T)]u8 = undefined;
    std.mem.writeInt(T, &buf, v, .little);
    try list.appendSlice(gpa, &buf);
}

pub fn main() !void {
    var dbg: std.heap.DebugAllocator(.{}) = .init;
    defer _ = dbg.deinit();
    const gpa = dbg.allocator();
    helpers.allocator = gpa; // what src/main.zig:22-23 does in safe builds

    const args = try std.process.argsAlloc(gpa);
    defer std.process.argsFree(gpa, args);
    if (args.len != 3) {
        std.debug.print("usage: mint_vectors <ascii_glyphs.bin> <out dir>\n", .{});
        return error.BadUsage;
    }
    const data = try std.fs.cwd().readFileAlloc(gpa, args[1], 1 << 28);
    defer gpa.free(data);
    if (!std.mem.eql(u8, data[0..8], "FRFX1\x00\x00\x00")) return error.BadFixture;
    var rd: Reader = .{ .bytes = data, .at = 8 };
    const n_glyphs = rd.int(u32);

    var images: std.ArrayList(u8) = .empty;
    defer images.deinit(gpa);
    try images.appendSlice(gpa, "FRZG1\x00\x00\x00");
    try put(&images, gpa, u32, n_glyphs);
    var lattices: std.ArrayList(u8) = .empty;
    defer lattices.deinit(gpa);
    try lattices.appendSlice(gpa, "FRZL1\x00\x00\x00");
    try put(&lattices, gpa, u32, if (MINT_LATTICE) LATTICE_GLYPHS.len else 0);

    var gi: u32 = 0;
    while (gi < n_glyphs) : (gi += 1) {
        // ---- one fixture glyph -> the reference's Glyph (font/Glyph.zig:11-24)
        const upm = rd.int(u16);
        const box: Glyph.Box = .{ .x_min = rd.int(i16), .y_min = rd.int(i16), .x_max = rd.int(i16), .y_max = rd.int(i16) };
        const n_contours = rd.int(u32);
        const contours = try gpa.alloc(Glyph.Contour, n_contours);
        defer gpa.free(contours);
        var made: usize = 0;
        defer for (contours[0..made]) |c| gpa.free(c.points);
        for (contours) |*c| {
            const n_points = rd.int(u32);
            const pts = try gpa.alloc(Point, n_points);
            for (pts) |*p| p.* = .{ .x = rd.int(i16), .y = rd.int(i16) };
            c.* = .{ .points = pts };
            made += 1;
        }
        const glyph: Glyph = .{ .box = box, .contours = contours };

        // ---- renderGlyph (render_glyph.zig:11-33): the image the reference itself allocates and fills
        const font_size = SIZES[gi % SIZES.len];
        const info: Font.Information = .{ .units_per_em = upm, .y0_baseline = true, .loca_format = @enumFromInt(0) };
        var im = try render_glyph.renderGlyph(glyph, info, font_size);
        defer im.deinit();
        try put(&images, gpa, u32, gi);
        try put(&images, gpa, u16, font_size);
        try put(&images, gpa, u16, @intCast(im.width));
        try put(&images, gpa, u16, @intCast(im.height));
        try images.appendSlice(gpa, im.data);

        // ---- the lattice Image.GlyphDebug.render walks (Image.zig:227-236), windings as windingInGlyph returns them
        if (MINT_LATTICE) {
            for (LATTICE_GLYPHS) |want| {
                if (want != gi) continue;
                var glyph_info = render_glyph.GlyphInfo.init(glyph);
                defer glyph_info.deinit();
                const W: u32 = @intCast(@as(i32, box.x_max) - box.x_min + 3);
                const H: u32 = @intCast(@as(i32, box.y_max) - box.y_min + 3);
                try put(&lattices, gpa, u32, gi);
                try put(&lattices, gpa, u32, W);
                try put(&lattices, gpa, u32, H);
                var h: u32 = 0;
                while (h < H) : (h += 1) {
                    var w: u32 = 0;
                    while (w < W) : (w += 1) {
                        const y = box.y_max - @as(i16, @intCast(h)) + 1;
                        const x = box.x_min + @as(i16, @intCast(w)) - 1;
                        try put(&lattices, gpa, i16, render_glyph.windingInGlyph(glyph, glyph_info, .{ .x = x, .y = y }));
                    }
                }
            }
        }
    }

    var dir = try std.fs.cwd().openDir(args[2], .{});
    defer dir.close();
    try dir.writeFile(.{ .sub_path = "zig_render_glyph.bin", .data = images.items });
    if (MINT_LATTICE) try dir.writeFile(.{ .sub_path = "zig_lattice.bin", .data = lattices.items });
    std.debug.print("minted {d} renderGlyph images ({d} bytes){s}\n", .{ n_glyphs, images.items.len, if (MINT_LATTICE) " and the GlyphDebug lattices" else "" });
}
