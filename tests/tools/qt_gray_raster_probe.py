"""
TEST TOOL (run with /opt/conda/bin/python3.9 = PyQt5 5.9.7) -- groundwork for render_human on jumper (not in the product yet):
what QPainter::drawEllipse(QRectF) does under QPainter::Antialiasing with a brush and NoPen (jumper.cpp:134-169 draws its
compass and the agent's shadow that way).  QRasterPaintEngine::drawEllipse leaves its midpoint fast path when antialiasing is on:
QPaintEngineEx::drawEllipse builds the 4-cubic path (qt_curves_for_arc, 13 points), QRasterPaintEngine::fill maps it to an
outline in 26.6 fixed point (QOutlineMapper) and hands it to Qt's copy of FreeType's gray raster (qgrayraster.c): exact area
coverage accumulated in cells with 8 sub-pixel bits, cubic arcs flattened by recursive subdivision, coverage = area >> 9 clamped
to 255 (non-zero winding), spans blended with comp_func_solid_SourceOver.

model_fill_ellipse() restates that chain; main() compares it with Qt on random and knife-edge rects.
usage: qt_gray_raster_probe.py [n_cases] [seed]
"""
import math
import os
import sys

os.environ["QT_QPA_PLATFORM"] = "offscreen"
import numpy as np

CW = CH = 96
PIXEL_BITS = 8
ONE_PIXEL = 1 << PIXEL_BITS


def TRUNC(x):
    return x >> PIXEL_BITS


def SUBPIXELS(x):
    return x << PIXEL_BITS


def UPSCALE(x):
    return x << (PIXEL_BITS - 6)


# knobs the probe decides between (see main): how QOutlineMapper rounds to 26.6, and which cubic flattener the raster has
TO_26_6 = "round"      # "round" | "trunc" | "floor"
CUBIC = "hain"         # "hain" (FreeType >= 2.4.5) | "levels" (FreeType 2.3.x)


def to_26_6(v):
    f = v * 64.0
    if TO_26_6 == "round":
        return int(math.floor(f + 0.5)) if f >= 0 else -int(math.floor(-f + 0.5))  # qRound
    if TO_26_6 == "trunc":
        return int(f)
    return int(math.floor(f))


class Raster:
    """the cell accumulator of qgrayraster.c (gray_set_cell / gray_render_scanline / gray_render_line), cells kept in a dict"""

    def __init__(self, w, h):
        self.w, self.h = w, h
        self.cells = {}  # (y, x) -> [area, cover]
        self.x = self.y = 0
        self.last_ey = 0
        self.ex = self.ey = 0
        self.area = self.cover = 0
        self.invalid = True

    # --- cells
    def record(self):
        if not self.invalid and (self.area or self.cover):
            c = self.cells.setdefault((self.ey, self.ex), [0, 0])
            c[0] += self.area
            c[1] += self.cover

    def set_cell(self, ex, ey):
        # all cells left of the clip region are merged into column -1 (min_ex = 0), those right of it / outside the rows dropped
        if ex > self.w:
            ex = self.w
        if ex < 0:
            ex = -1
        if ex != self.ex or ey != self.ey:
            self.record()
            self.area = self.cover = 0
        self.ex, self.ey = ex, ey
        self.invalid = not (0 <= ey < self.h and ex < self.w)

    def start_cell(self, ex, ey):
        if ex > self.w:
            ex = self.w
        if ex < 0:
            ex = -1
        self.area = self.cover = 0
        self.ex, self.ey = ex, ey
        self.last_ey = SUBPIXELS(ey)
        self.invalid = False
        self.set_cell(ex, ey)

    # --- lines
    def render_scanline(self, ey, x1, y1, x2, y2):
        dx = x2 - x1
        ex1 = TRUNC(x1); ex2 = TRUNC(x2)
        fx1 = x1 - SUBPIXELS(ex1); fx2 = x2 - SUBPIXELS(ex2)
        if y1 == y2:
            self.set_cell(ex2, ey)
            return
        if ex1 == ex2:
            delta = y2 - y1
            self.area += (fx1 + fx2) * delta
            self.cover += delta
            return
        p = (ONE_PIXEL - fx1) * (y2 - y1)
        first = ONE_PIXEL; incr = 1
        if dx < 0:
            p = fx1 * (y2 - y1)
            first = 0; incr = -1; dx = -dx
        delta, mod = divmod(p, dx)  # floor division, 0 <= mod < dx (the C code repairs its truncating / and % to the same)
        self.area += (fx1 + first) * delta
        self.cover += delta
        ex1 += incr
        self.set_cell(ex1, ey)
        y1 += delta
        if ex1 != ex2:
            p = ONE_PIXEL * (y2 - y1 + delta)
            lift, rem = divmod(p, dx)
            mod -= dx
            while ex1 != ex2:
                delta = lift
                mod += rem
                if mod >= 0:
                    mod -= dx
                    delta += 1
                self.area += ONE_PIXEL * delta
                self.cover += delta
                y1 += delta
                ex1 += incr
                self.set_cell(ex1, ey)
        delta = y2 - y1
        self.area += (fx2 + ONE_PIXEL - first) * delta
        self.cover += delta

    def render_line(self, to_x, to_y):
        ey1 = TRUNC(self.last_ey); ey2 = TRUNC(to_y)
        fy1 = self.y - self.last_ey; fy2 = to_y - SUBPIXELS(ey2)
        dx = to_x - self.x; dy = to_y - self.y
        lo, hi = (ey1, ey2) if ey1 <= ey2 else (ey2, ey1)
        if lo >= self.h or hi < 0:
            pass  # vertical clipping: nothing of it in the rows
        elif ey1 == ey2:
            self.render_scanline(ey1, self.x, fy1, to_x, fy2)
        elif dx == 0:
            ex = TRUNC(self.x)
            two_fx = (self.x - SUBPIXELS(ex)) << 1
            first = ONE_PIXEL; incr = 1
            if dy < 0:
                first = 0; incr = -1
            delta = first - fy1
            self.area += two_fx * delta
            self.cover += delta
            ey1 += incr
            self.set_cell(ex, ey1)
            delta = first + first - ONE_PIXEL
            area = two_fx * delta
            while ey1 != ey2:
                self.area += area
                self.cover += delta
                ey1 += incr
                self.set_cell(ex, ey1)
            delta = fy2 - ONE_PIXEL + first
            self.area += two_fx * delta
            self.cover += delta
        else:
            p = (ONE_PIXEL - fy1) * dx
            first = ONE_PIXEL; incr = 1
            if dy < 0:
                p = fy1 * dx
                first = 0; incr = -1; dy = -dy
            delta, mod = divmod(p, dy)
            x = self.x + delta
            self.render_scanline(ey1, self.x, fy1, x, first)
            ey1 += incr
            self.set_cell(TRUNC(x), ey1)
            if ey1 != ey2:
                p = ONE_PIXEL * dx
                lift, rem = divmod(p, dy)
                mod -= dy
                while ey1 != ey2:
                    delta = lift
                    mod += rem
                    if mod >= 0:
                        mod -= dy
                        delta += 1
                    x2 = x + delta
                    self.render_scanline(ey1, x, ONE_PIXEL - first, x2, first)
                    x = x2
                    ey1 += incr
                    self.set_cell(TRUNC(x), ey1)
            self.render_scanline(ey1, x, ONE_PIXEL - first, to_x, fy2)
        self.x, self.y = to_x, to_y
        self.last_ey = SUBPIXELS(ey2)

    def move_to(self, x26, y26):
        self.record()
        x = UPSCALE(x26); y = UPSCALE(y26)
        self.start_cell(TRUNC(x), TRUNC(y))
        self.x, self.y = x, y

    # --- cubic arcs
    def cubic_to(self, c1, c2, to, cubic_level):
        arc = [(UPSCALE(to[0]), UPSCALE(to[1])), (UPSCALE(c2[0]), UPSCALE(c2[1])), (UPSCALE(c1[0]), UPSCALE(c1[1])), (self.x, self.y)]
        if CUBIC == "hain":
            stack = [arc]
            while stack:
                a = stack[-1]
                dx = a[3][0] - a[0][0]; dy = a[3][1] - a[0][1]
                ax, ay = abs(dx), abs(dy)
                L = ax + (3 * ay >> 3) if ax > ay else ay + (3 * ax >> 3)  # QT_FT_HYPOT
                split = False
                if L > 32767:
                    split = True
                else:
                    s_limit = L * (ONE_PIXEL // 6)
                    dx1 = a[1][0] - a[0][0]; dy1 = a[1][1] - a[0][1]
                    dx2 = a[2][0] - a[0][0]; dy2 = a[2][1] - a[0][1]
                    if abs(dy * dx1 - dx * dy1) > s_limit or abs(dy * dx2 - dx * dy2) > s_limit:
                        split = True
                    elif dx1 * (dx1 - dx) + dy1 * (dy1 - dy) > 0 or dx2 * (dx2 - dx) + dy2 * (dy2 - dy) > 0:
                        split = True
                if split:
                    lo, hi = split_cubic(a)
                    stack[-1] = hi   # base[3..6]: the half that ends at the old arc[3] (drawn second)
                    stack.append(lo)  # the half that is drawn first is on top ... see split_cubic
                    continue
                self.render_line(a[0][0], a[0][1])
                stack.pop()
        else:
            # FreeType 2.3.x: a fixed number of halvings from the size of the arc
            mid = lambda k: (self.x >> 0, 0)
            dx = (self.x >> (PIXEL_BITS - 6)) + to[0] - (((self.x >> (PIXEL_BITS - 6)) + to[0] + 3 * (c1[0] + c2[0])) // 8 << 1)
            dy = (self.y >> (PIXEL_BITS - 6)) + to[1] - (((self.y >> (PIXEL_BITS - 6)) + to[1] + 3 * (c1[1] + c2[1])) // 8 << 1)
            d = max(abs(dx), abs(dy))
            level = 1
            d //= cubic_level
            while d > 0:
                d >>= 2
                level += 1
            def rec(a, lvl):
                if lvl > 1:
                    lo, hi = split_cubic(a)
                    rec(lo, lvl - 1)
                    rec(hi, lvl - 1)
                else:
                    to_x, to_y = a[0]
                    mx = (self.x + to_x + 3 * (a[1][0] + a[2][0])) // 8
                    my = (self.y + to_y + 3 * (a[1][1] + a[2][1])) // 8
                    self.render_line(mx, my)
                    self.render_line(to_x, to_y)
            rec(arc, level)


def split_cubic(b):
    """gray_split_cubic on base[0..3] = (end, c2, c1, start): returns (first half to draw, second half), each in the same layout"""
    x = [p[0] for p in b]; y = [p[1] for p in b]
    def halves(v):
        v0, v1, v2, v3 = v
        a = (v0 + v1) // 2; bb = (v3 + v2) // 2; c = (v1 + v2) // 2
        a2 = (a + c) // 2; b2 = (bb + c) // 2; m = (a2 + b2) // 2
        # base[0..3] = v0, a, a2, m (the half next to the END point); base[3..6] = m, b2, bb, v3 (next to the start)
        return (v0, a, a2, m), (m, b2, bb, v3)
    ex, sx = halves(x); ey, sy = halves(y)
    end_half = list(zip(ex, ey)); start_half = list(zip(sx, sy))
    return start_half, end_half  # drawn first: the half that starts at the current point


def qt_arc_points(rx, ry, rw, rh):
    """qpainterpath.cpp qt_curves_for_arc(rect, 0, -360) / QPaintEngineEx::drawEllipse: start point + 4 cubics (12 points)"""
    K = 0.5522847498
    x, y, w, h = rx, ry, rw, rh
    w2, h2 = w / 2, h / 2
    w2k, h2k = w2 * K, h2 * K
    pts = [
        (x + w, y + h2),
        (x + w, y + h2 + h2k), (x + w2 + w2k, y + h), (x + w2, y + h),          # 0 -> 270
        (x + w2 - w2k, y + h), (x, y + h2 + h2k), (x, y + h2),                    # 270 -> 180
        (x, y + h2 - h2k), (x + w2 - w2k, y), (x + w2, y),                        # 180 -> 90
        (x + w2 + w2k, y), (x + w, y + h2 - h2k), (x + w, y + h2),                # 90 -> 0
    ]
    return pts


def bez_split(b):
    """qbezier_p.h QBezier::split -> (first, second)"""
    x1, y1, x2, y2, x3, y3, x4, y4 = b
    c = (x2 + x3) * .5
    fx2 = (x1 + x2) * .5; sx3 = (x3 + x4) * .5
    fx3 = (fx2 + c) * .5; sx2 = (sx3 + c) * .5
    mx = (fx3 + sx2) * .5
    c = (y2 + y3) / 2
    fy2 = (y1 + y2) * .5; sy3 = (y3 + y4) * .5
    fy3 = (fy2 + c) * .5; sy2 = (sy3 + c) * .5
    my = (fy3 + sy2) * .5
    return (x1, y1, fx2, fy2, fx3, fy3, mx, my), (mx, my, sx2, sy2, sx3, sy3, x4, y4)


def bez_add_to_polygon(b, out, thr=0.25):
    """qbezier.cpp QBezier::addToPolygon (QOutlineMapper::curveTo flattens every curve with m_curve_threshold = 0.25)"""
    stack = [(b, 9)]
    while stack:
        bz, lvl = stack[-1]
        x1, y1, x2, y2, x3, y3, x4, y4 = bz
        y4y1 = y4 - y1; x4x1 = x4 - x1
        l = abs(x4x1) + abs(y4y1)
        if l > 1.:
            d = abs(x4x1 * (y1 - y2) - y4y1 * (x1 - x2)) + abs(x4x1 * (y1 - y3) - y4y1 * (x1 - x3))
        else:
            d = abs(x1 - x2) + abs(y1 - y2) + abs(x1 - x3) + abs(y1 - y3)
            l = 1.
        if d < thr * l or lvl == 0:
            out.append((x4, y4))
            stack.pop()
        else:
            first, second = bez_split(bz)
            stack[-1] = (second, lvl - 1)
            stack.append((first, lvl - 1))


def model_fill_ellipse(dst, rx, ry, rw, rh, color):
    ch, cw = dst.shape
    ap = qt_arc_points(rx, ry, rw, rh)
    poly = [ap[0]]
    for k in range(4):
        x0, y0 = poly[-1]
        bez_add_to_polygon((x0, y0) + ap[1 + 3 * k] + ap[2 + 3 * k] + ap[3 + 3 * k], poly)
    if poly[-1] != poly[0]:
        poly.append(poly[0])  # closeSubpath
    pts = [(to_26_6(px), to_26_6(py)) for (px, py) in poly]
    r = Raster(cw, ch)
    r.move_to(*pts[0])
    for (qx, qy) in pts[1:]:
        r.render_line(UPSCALE(qx), UPSCALE(qy))
    r.record()
    # sweep
    rows = {}
    for (y, x), (area, cover) in r.cells.items():
        rows.setdefault(y, []).append((x, area, cover))
    sa = (color >> 24) & 255
    for y, cells in rows.items():
        if not (0 <= y < ch):
            continue
        cells.sort()
        cover = 0
        x = 0
        def hline(x0, area, count):
            cov = area >> (PIXEL_BITS * 2 + 1 - 8)
            if cov < 0:
                cov = -cov
            if cov >= 256:
                cov = 255
            if cov:
                for xx in range(max(x0, 0), min(x0 + count, cw)):
                    dst[y, xx] = source_over(int(dst[y, xx]), color, cov)
        for (cx, area, cv) in cells:
            if cx > x and cover != 0:
                hline(x, cover * (ONE_PIXEL * 2), cx - x)
            cover += cv
            a = cover * (ONE_PIXEL * 2) - area
            if a != 0 and cx >= 0:
                hline(cx, a, 1)
            x = cx + 1
        if cover != 0:
            hline(x, cover * (ONE_PIXEL * 2), cw - x)


def byte_mul(x, a):
    t = (x & 0xff00ff) * a
    t = (t + ((t >> 8) & 0xff00ff) + 0x800080) >> 8
    t &= 0xff00ff
    x = ((x >> 8) & 0xff00ff) * a
    x = (x + ((x >> 8) & 0xff00ff) + 0x800080)
    x &= 0xff00ff00
    return (x | t) & 0xffffffff


def source_over(d, s, ca):
    if ca != 255:
        s = byte_mul(s, ca)
    return (s + byte_mul(d, 255 - (s >> 24))) & 0xffffffff


# ---- the outline: a pen of width 1 under Antialiasing is "fast": QCosmeticStroker::drawPath with drawLineAA ---------------------
def c_int(v):
    return int(v)


def fixdiv(x, y):  # F16Dot16FixedDiv: C division truncates
    n = x << 16
    q = abs(n) // abs(y)
    return q if (n >= 0) == (y > 0) else -q


class CosmeticAA:
    def __init__(self, dst, color):
        self.dst = dst
        ch, cw = dst.shape
        self.cw, self.ch = cw, ch
        self.color = color
        self.xmin, self.xmax, self.ymin, self.ymax = -1.0, cw + 1.0, -1.0, ch + 1.0  # setup(): device rect widened by one pixel

    def pixel(self, x, y, coverage):
        if x < 0 or x > self.cw - 1 or y < 0 or y > self.ch - 1:
            return
        c = byte_mul(self.color, coverage)  # drawPixelARGB32
        d = int(self.dst[y, x])
        self.dst[y, x] = (c + byte_mul(d, 255 - (c >> 24))) & 0xffffffff

    def clip_line(self, x1, y1, x2, y2):
        """QCosmeticStroker::clipLine -> (clipped?, x1, y1, x2, y2)"""
        if x1 < self.xmin:
            if x2 <= self.xmin: return True, x1, y1, x2, y2
            y1 += (y2 - y1) / (x2 - x1) * (self.xmin - x1); x1 = self.xmin
        elif x1 > self.xmax:
            if x2 >= self.xmax: return True, x1, y1, x2, y2
            y1 += (y2 - y1) / (x2 - x1) * (self.xmax - x1); x1 = self.xmax
        if x2 < self.xmin:
            y2 += (y2 - y1) / (x2 - x1) * (self.xmin - x2); x2 = self.xmin
        elif x2 > self.xmax:
            y2 += (y2 - y1) / (x2 - x1) * (self.xmax - x2); x2 = self.xmax
        if y1 < self.ymin:
            if y2 <= self.ymin: return True, x1, y1, x2, y2
            x1 += (x2 - x1) / (y2 - y1) * (self.ymin - y1); y1 = self.ymin
        elif y1 > self.ymax:
            if y2 >= self.ymax: return True, x1, y1, x2, y2
            x1 += (x2 - x1) / (y2 - y1) * (self.ymax - y1); y1 = self.ymax
        if y2 < self.ymin:
            x2 += (x2 - x1) / (y2 - y1) * (self.ymin - y2); y2 = self.ymin
        elif y2 > self.ymax:
            x2 += (x2 - x1) / (y2 - y1) * (self.ymax - y2); y2 = self.ymax
        return False, x1, y1, x2, y2

    def line(self, rx1, ry1, rx2, ry2, caps=0):
        """drawLineAA<NoDasher>"""
        clipped, rx1, ry1, rx2, ry2 = self.clip_line(rx1, ry1, rx2, ry2)
        if clipped:
            return
        x1 = c_int(rx1 * 64.); y1 = c_int(ry1 * 64.); x2 = c_int(rx2 * 64.); y2 = c_int(ry2 * 64.)
        dx = x2 - x1; dy = y2 - y1
        if abs(dx) < abs(dy):
            xinc = fixdiv(dx, dy)
            if y1 > y2:
                y1, y2 = y2, y1; x1, x2 = x2, x1
                caps = ((caps & 1) << 1) | ((caps & 2) >> 1)
            x = (x1 - 32) << 10
            x -= (((y1 & 63) - 32) * xinc) >> 6
            if caps & 1: y1 -= 32; x -= xinc >> 1
            if caps & 2: y2 += 32
            y = y1 >> 6; ys = y2 >> 6
            if y == ys:
                a_start = y2 - y1; a_end = 0
            else:
                a_start = 64 - (y1 & 63); a_end = y2 & 63
            al = (x >> 8) & 255
            self.pixel(x >> 16, y, (255 - al) * a_start >> 6); self.pixel((x >> 16) + 1, y, al * a_start >> 6)
            x += xinc; y += 1
            if y < ys:
                while True:
                    al = (x >> 8) & 255
                    self.pixel(x >> 16, y, 255 - al); self.pixel((x >> 16) + 1, y, al)
                    x += xinc
                    y += 1
                    if not (y < ys): break
            if a_end:
                al = (x >> 8) & 255
                self.pixel(x >> 16, y, (255 - al) * a_end >> 6); self.pixel((x >> 16) + 1, y, al * a_end >> 6)
        else:
            if not dx:
                return
            yinc = fixdiv(dy, dx)
            if x1 > x2:
                y1, y2 = y2, y1; x1, x2 = x2, x1
                caps = ((caps & 1) << 1) | ((caps & 2) >> 1)
            y = (y1 - 32) << 10
            y -= (((x1 & 63) - 32) * yinc) >> 6
            if caps & 1: x1 -= 32; y -= yinc >> 1
            if caps & 2: x2 += 32
            x = x1 >> 6; xs = x2 >> 6
            if x == xs:
                a_start = x2 - x1; a_end = 0
            else:
                a_start = 64 - (x1 & 63); a_end = x2 & 63
            al = (y >> 8) & 255
            self.pixel(x, y >> 16, (255 - al) * a_start >> 6); self.pixel(x, (y >> 16) + 1, al * a_start >> 6)
            y += yinc; x += 1
            if x < xs:
                while True:
                    al = (y >> 8) & 255
                    self.pixel(x, y >> 16, 255 - al); self.pixel(x, (y >> 16) + 1, al)
                    y += yinc
                    x += 1
                    if not (x < xs): break
            if a_end:
                al = (y >> 8) & 255
                self.pixel(x, y >> 16, (255 - al) * a_end >> 6); self.pixel(x, (y >> 16) + 1, al * a_end >> 6)

    def cubic(self, pts, level, caps):
        """renderCubicSubdivision: pts = [p4, p3, p2, p1] (points[0] is the end)"""
        if level:
            dx = pts[3][0] - pts[0][0]; dy = pts[3][1] - pts[0][1]
            ln = .25 * (abs(dx) + abs(dy))
            if abs(dx * (pts[0][1] - pts[2][1]) - dy * (pts[0][0] - pts[2][0])) >= ln or abs(dx * (pts[0][1] - pts[1][1]) - dy * (pts[0][0] - pts[1][0])) >= ln:
                # splitCubic: points[0..3] <- half next to the end, points[3..6] <- half next to the start
                def halves(v0, v1, v2, v3):
                    a = (v0 + v1) / 2; b = (v3 + v2) / 2; c = (v1 + v2) / 2
                    a2 = (a + c) / 2; b2 = (b + c) / 2; m = (a2 + b2) / 2
                    return (v0, a, a2, m), (m, b2, b, v3)
                ex, sx = halves(pts[0][0], pts[1][0], pts[2][0], pts[3][0]); ey, sy = halves(pts[0][1], pts[1][1], pts[2][1], pts[3][1])
                self.cubic(list(zip(sx, sy)), level - 1, caps & 1)
                self.cubic(list(zip(ex, ey)), level - 1, caps & 2)
                return
        self.line(pts[3][0], pts[3][1], pts[0][0], pts[0][1], caps)


def model_stroke_ellipse(dst, rx, ry, rw, rh, color):
    ap = qt_arc_points(rx, ry, rw, rh)
    st = CosmeticAA(dst, color)
    for k in range(4):
        st.cubic([ap[3 * k + 3], ap[3 * k + 2], ap[3 * k + 1], ap[3 * k]], 6, 0)


def model_draw_ellipse(dst, rx, ry, rw, rh, pen, brush):
    """QPainter::drawEllipse(QRectF) under Antialiasing with a brush and a solid pen of width 1 (set_pen_brush_color, BAG:972-977)"""
    if brush is not None:
        model_fill_ellipse(dst, rx, ry, rw, rh, brush)
    if pen is not None:
        model_stroke_ellipse(dst, rx, ry, rw, rh, pen)


def qt_draw_ellipse(dst0, rx, ry, rw, rh, pen, brush):
    from PyQt5.QtGui import QImage, QPainter, QColor, QGuiApplication, QBrush, QPen
    from PyQt5.QtCore import QRectF, Qt
    global _app
    _app = QGuiApplication.instance() or QGuiApplication([])
    h, w = dst0.shape
    raw = np.ascontiguousarray(dst0.astype(np.uint32)).tobytes()
    img = QImage(raw, w, h, w * 4, QImage.Format_RGB32).copy()
    p = QPainter(img)
    p.setRenderHint(QPainter.Antialiasing, True)
    p.setRenderHint(QPainter.SmoothPixmapTransform, True)
    qc = lambda c: QColor((c >> 16) & 255, (c >> 8) & 255, c & 255, c >> 24)
    p.setBrush(QBrush(qc(brush)) if brush is not None else Qt.NoBrush)
    p.setPen(QPen(qc(pen), 1) if pen is not None else Qt.NoPen)
    p.drawEllipse(QRectF(rx, ry, rw, rh))
    p.end()
    ptr = img.constBits(); ptr.setsize(w * h * 4)
    return np.frombuffer(bytes(ptr), np.uint32).reshape(h, w).copy()


def probe_outlined(n, seed):
    rng = np.random.RandomState(seed)
    miss = worst = 0
    for case in range(n):
        dst0 = np.full((CH, CW), 0xff000000, np.uint32)
        rw = rng.uniform(2, 80); rh = rng.uniform(2, 80)
        rx = rng.uniform(-rw * 0.3, CW - rw * 0.7); ry = rng.uniform(-rh * 0.3, CH - rh * 0.7)
        if case % 4 == 0:
            rx, ry, rw, rh = float(int(rx)), float(int(ry)), float(int(rw) + 1), float(int(rh) + 1)
        if case % 4 == 1:
            rx, ry, rw, rh = [float(np.float32(v)) for v in (rx, ry, rw, rh)]
        pen = 0xffa8a69e
        brush = None if case % 3 == 0 else 0xffa8a69e
        got = qt_draw_ellipse(dst0, rx, ry, rw, rh, pen, brush)
        want = dst0.copy(); model_draw_ellipse(want, rx, ry, rw, rh, pen, brush)
        d = np.abs(got.view(np.uint8).reshape(CH, CW, 4)[..., :3].astype(int) - want.view(np.uint8).reshape(CH, CW, 4)[..., :3].astype(int))
        if d.max() > 0:
            miss += 1; worst = max(worst, int(d.max()))
            ys, xs = np.nonzero(d.max(axis=2))
            if miss <= 10:
                print(f"case {case} rect ({rx:.4f},{ry:.4f},{rw:.4f},{rh:.4f}) brush {brush is not None}: {len(ys)} px differ, max {d.max()}, first (x={xs[0]},y={ys[0]}) got {got[ys[0], xs[0]] & 0xffffff:06x} want {want[ys[0], xs[0]] & 0xffffff:06x}")
    print(f"outlined antialiased ellipses (pen width 1): {n} cases, {miss} with differences, worst {worst}")


def qt_fill_ellipse(dst0, rx, ry, rw, rh, c, int_rect=False):
    from PyQt5.QtGui import QImage, QPainter, QColor, QGuiApplication, QBrush
    from PyQt5.QtCore import QRectF, QRect, Qt
    global _app
    _app = QGuiApplication.instance() or QGuiApplication([])
    h, w = dst0.shape
    raw = np.ascontiguousarray(dst0.astype(np.uint32)).tobytes()
    img = QImage(raw, w, h, w * 4, QImage.Format_RGB32).copy()
    p = QPainter(img)
    p.setRenderHint(QPainter.Antialiasing, True)
    p.setRenderHint(QPainter.SmoothPixmapTransform, True)
    p.setBrush(QBrush(QColor((c >> 16) & 255, (c >> 8) & 255, c & 255, c >> 24)))
    p.setPen(Qt.NoPen)
    p.drawEllipse(QRect(int(rx), int(ry), int(rw), int(rh)) if int_rect else QRectF(rx, ry, rw, rh))
    p.end()
    ptr = img.constBits(); ptr.setsize(w * h * 4)
    return np.frombuffer(bytes(ptr), np.uint32).reshape(h, w).copy()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    miss = worst = 0
    for case in range(n):
        dst0 = np.full((CH, CW), 0xff000000, np.uint32)
        rw = rng.uniform(1, 80); rh = rng.uniform(1, 80)
        rx = rng.uniform(-rw * 0.4, CW - rw * 0.6); ry = rng.uniform(-rh * 0.4, CH - rh * 0.6)
        if case % 4 == 0:
            rx, ry, rw, rh = float(int(rx)), float(int(ry)), float(int(rw) + 1), float(int(rh) + 1)
        if case % 4 == 1:
            rx, ry, rw, rh = [float(np.float32(v)) for v in (rx, ry, rw, rh)]
        color = 0xffffffff if case % 3 else 0x78ffffff  # (the brush colour as QColor(r, g, b, a); jumper's shadow is QColor(255, 255, 255, 120))
        got = qt_fill_ellipse(dst0, rx, ry, rw, rh, color)
        a_ = color >> 24
        premul = (byte_mul(color & 0xffffff, a_) & 0xffffff) | (a_ << 24)  # qPremultiply
        want = dst0.copy(); model_fill_ellipse(want, rx, ry, rw, rh, premul)
        d = np.abs((got & 255).astype(int) - (want & 255).astype(int))
        if d.max() > 0:
            miss += 1; worst = max(worst, int(d.max()))
            ys, xs = np.nonzero(d)
            if miss <= 12:
                print(f"case {case} rect ({rx:.4f},{ry:.4f},{rw:.4f},{rh:.4f}): {len(ys)} px differ, max {d.max()}, first (x={xs[0]},y={ys[0]}) got {got[ys[0], xs[0]] & 255} want {want[ys[0], xs[0]] & 255}")
    print(f"filled antialiased ellipses: {n} cases, {miss} with differences, worst {worst} (26.6 conversion: {TO_26_6}, cubic flattener: {CUBIC})")


if __name__ == "__main__":
    main()
    probe_outlined(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
