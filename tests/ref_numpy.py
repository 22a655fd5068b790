"""A second, independent restatement of glyphWindingAt
(/root/reference/src/tools/render_glyph.zig:35-73) in numpy float32, vectorised over
sample points.  Exists only to cross-validate oracle/fr_oracle.c: two restatements
written separately from the Zig text must agree bit for bit."""
import numpy as np

F = np.float32


def winding_at(points_xy: np.ndarray, contour_start: np.ndarray, cx: np.ndarray, cy: np.ndarray) -> np.ndarray:
    cx = np.asarray(cx, F); cy = np.asarray(cy, F)
    w = np.zeros(np.broadcast(cx, cy).shape, np.int32)
    two, zero, one = F(2), F(0), F(1)
    with np.errstate(all="ignore"):
        for c in range(len(contour_start) - 1):
            p = points_xy[int(contour_start[c]):int(contour_start[c + 1])].astype(F)
            for k in range(len(p) // 2):
                p0x, p0y = p[2 * k]; p1x, p1y = p[2 * k + 1]; p2x, p2y = p[2 * k + 2]
                a = F(F(p0y - F(two * p1y)) + p2y)
                ax = F(F(p0x - F(two * p1x)) + p2x)
                bx = F(two * F(p1x - p0x))
                if a == zero:
                    if p2y == p0y:
                        continue
                    t = (cy - p0y) / F(p2y - p0y)
                    ok = ~((t < zero) | (t >= one))
                    xx = (ax * t + bx) * t + p0x
                    ok = ok & ~(xx < cx)
                    w += np.where(ok, -1 if p0y < p2y else 1, 0)
                    continue
                delta = cy * a + F(p1y * p1y) - F(p0y * p2y)
                good = ~(delta < zero)
                sq = np.sqrt(np.where(good, delta, zero)).astype(F)
                b = F(p0y - p1y)
                for t in ((b + sq) / a, (b - sq) / a):
                    ok = good & ~((t < zero) | (t >= one))
                    xx = (ax * t + bx) * t + p0x
                    ok = ok & ~(xx < cx)
                    dy = a * t + F(p1y - p0y)
                    w += np.where(ok, np.where(dy > zero, -1, 1), 0)
    return w.astype(np.int16)


def render_glyph(points_xy, contour_start, box, upm, font_size):
    """renderGlyph, render_glyph.zig:11-33 -> (winding int16 (H,W), gray u8 (H,W))"""
    scale = F(font_size) / F(upm)
    b = np.asarray(box, np.int16).astype(F) * scale
    mn = (int(np.floor(b[0])), int(np.floor(b[1])))
    mx = (int(np.ceil(b[2])), int(np.ceil(b[3])))
    W, H = mx[0] - mn[0] + 1, mx[1] - mn[1] + 1
    cx = (np.arange(W, dtype=np.int32) + mn[0]).astype(F) / scale
    cy = (mx[1] - np.arange(H, dtype=np.int32)).astype(F) / scale
    wd = winding_at(points_xy, contour_start, cx[None, :], cy[:, None])
    gray = np.clip(wd.astype(np.int32) * 20 + 100, 0, 255).astype(np.uint8)
    return wd, gray
