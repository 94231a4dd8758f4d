// pg_aapath.h -- Qt 5.9's ANTIALIASED path route, as far as jumper's compass needs it under render_human (reference
// src/games/jumper.cpp:134-169 drawn by Game::render_to_buf(..., antialias = true), src/game.cpp:84-87):
//   drawEllipse(QRectF / QRect) with a brush : QPaintEngineEx::drawEllipse -> QRasterPaintEngine::fill -> QOutlineMapper (every cubic of
//       qt_curves_for_arc flattened by QBezier::addToPolygon(.25), points to 26.6 by qRound) -> qgrayraster.c, Qt's copy of
//       FreeType's gray raster: signed area / cover accumulated per pixel cell with 8 sub-pixel bits, coverage =
//       |area| >> 9 capped at 255 (non-zero winding), spans blended with the brush colour;
//   ... with a pen of width 1               : a "fast" pen even when antialiased -> QCosmeticStroker::drawPath with drawLineAA: the
//       path's cubics subdivided as in the aliased stroker (pg_qtpath.h), every segment walked in 16.16 with its coverage split
//       between the two pixels it passes between, first / last pixel weighted by the part of the segment inside them.
// The wide compass needle (rasterizeLine with a square cap) lives in pg_human.h next to the other rasterizeLine users.
// Pinned against PyQt5 5.9.7 by tests/tools/qt_gray_raster_probe.py (200 filled, 100 outlined ellipses incl. clipped, tiny,
// integer and translucent ones: 0 misses).  Scalar code on 32-bit integers and doubles, callable from wave-uniform device code.
#pragma once
#include "pg_qtpath.h"

namespace pgamd {
namespace aapath {

constexpr int ONE_PIXEL = 256;  // PIXEL_BITS = 8
PG_QT_FN int trunc8(int x) { return x >> 8; }
// p / d and p % d as ftgrays repairs them: floor division for a positive divisor
PG_QT_FN void floor_divmod(int p, int d, int &q, int &m) {
    q = p / d;
    m = p % d;
    if (m < 0) {
        q--;
        m += d;
    }
}

// The gray raster's accumulators for ONE pixel row: the outline is walked completely (the walk is incremental along every line),
// cells of other rows are dropped.  Column ex (-1 = everything left of the frame) lives at index ex + 1.
struct RowCells {
    int *area, *cover;
    int w, row;
    int ex, ey, a, c;
    bool valid;
    int x, y, last_ey;  // current point, 24.8
    int min_x, max_x;   // cell columns touched in this row
    int n;              // points seen (flatten sink)

    PG_QT_FN void init(int *area_, int *cover_, int w_, int row_) {
        area = area_; cover = cover_; w = w_; row = row_;
        ex = ey = a = c = 0;
        valid = false;
        x = y = last_ey = 0;
        min_x = w_;
        max_x = -2;
        n = 0;
    }
    PG_QT_FN void record() {
        if (valid && (a | c)) {
            area[ex + 1] += a;
            cover[ex + 1] += c;
            if (ex < min_x) min_x = ex;
            if (ex > max_x) max_x = ex;
        }
    }
    PG_QT_FN void set_cell(int nex, int ney) {  // gray_set_cell
        if (nex > w) nex = w;
        if (nex < 0) nex = -1;
        if (nex != ex || ney != ey) {
            record();
            a = c = 0;
        }
        ex = nex;
        ey = ney;
        valid = ney == row && nex < w;
    }
    PG_QT_FN void scanline(int sy, int x1, int y1, int x2, int y2) {  // gray_render_scanline
        int dx = x2 - x1;
        int ex1 = trunc8(x1);
        const int ex2 = trunc8(x2);
        const int fx1 = x1 - (ex1 << 8), fx2 = x2 - (ex2 << 8);
        if (y1 == y2) {
            set_cell(ex2, sy);
            return;
        }
        if (ex1 == ex2) {
            const int delta = y2 - y1;
            a += (fx1 + fx2) * delta;
            c += delta;
            return;
        }
        int p = (ONE_PIXEL - fx1) * (y2 - y1), first = ONE_PIXEL, incr = 1;
        if (dx < 0) {
            p = fx1 * (y2 - y1);
            first = 0;
            incr = -1;
            dx = -dx;
        }
        int delta, mod;
        floor_divmod(p, dx, delta, mod);
        a += (fx1 + first) * delta;
        c += delta;
        ex1 += incr;
        set_cell(ex1, sy);
        y1 += delta;
        if (ex1 != ex2) {
            int lift, rem;
            floor_divmod(ONE_PIXEL * (y2 - y1 + delta), dx, lift, rem);
            mod -= dx;
            while (ex1 != ex2) {
                delta = lift;
                mod += rem;
                if (mod >= 0) {
                    mod -= dx;
                    delta++;
                }
                a += ONE_PIXEL * delta;
                c += delta;
                y1 += delta;
                ex1 += incr;
                set_cell(ex1, sy);
            }
        }
        delta = y2 - y1;
        a += (fx2 + ONE_PIXEL - first) * delta;
        c += delta;
    }
    PG_QT_FN void line_to(int to_x, int to_y) {  // gray_render_line
        int ey1 = trunc8(last_ey);
        const int ey2 = trunc8(to_y);
        const int fy1 = y - last_ey, fy2 = to_y - (ey2 << 8);
        int dx = to_x - x, dy = to_y - y;
        const int lo = ey1 < ey2 ? ey1 : ey2, hi = ey1 < ey2 ? ey2 : ey1;
        if (row < lo || row > hi) {
            // nothing of this line in the row: only the current cell moves on (it is always the cell of the current point)
            set_cell(trunc8(to_x), ey2);
        } else if (ey1 == ey2) {
            scanline(ey1, x, fy1, to_x, fy2);
        } else if (dx == 0) {
            const int cx = trunc8(x), two_fx = (x - (cx << 8)) << 1;
            int first = ONE_PIXEL, incr = 1;
            if (dy < 0) {
                first = 0;
                incr = -1;
            }
            int delta = first - fy1;
            a += two_fx * delta;
            c += delta;
            ey1 += incr;
            set_cell(cx, ey1);
            delta = first + first - ONE_PIXEL;
            const int ar = two_fx * delta;
            while (ey1 != ey2) {
                a += ar;
                c += delta;
                ey1 += incr;
                set_cell(cx, ey1);
            }
            delta = fy2 - ONE_PIXEL + first;
            a += two_fx * delta;
            c += delta;
        } else {
            int p = (ONE_PIXEL - fy1) * dx, first = ONE_PIXEL, incr = 1;
            if (dy < 0) {
                p = fy1 * dx;
                first = 0;
                incr = -1;
                dy = -dy;
            }
            int delta, mod;
            floor_divmod(p, dy, delta, mod);
            int cx = x + delta;
            scanline(ey1, x, fy1, cx, first);
            ey1 += incr;
            set_cell(trunc8(cx), ey1);
            if (ey1 != ey2) {
                int lift, rem;
                floor_divmod(ONE_PIXEL * dx, dy, lift, rem);
                mod -= dy;
                while (ey1 != ey2) {
                    delta = lift;
                    mod += rem;
                    if (mod >= 0) {
                        mod -= dy;
                        delta++;
                    }
                    const int x2 = cx + delta;
                    scanline(ey1, cx, ONE_PIXEL - first, x2, first);
                    cx = x2;
                    ey1 += incr;
                    set_cell(trunc8(cx), ey1);
                }
            }
            scanline(ey1, cx, ONE_PIXEL - first, to_x, fy2);
        }
        x = to_x;
        y = to_y;
        last_ey = ey2 << 8;
    }
    PG_QT_FN void move_to(int x26, int y26) {  // gray_move_to + gray_start_cell
        record();
        x = x26 << 2;
        y = y26 << 2;
        int sx = trunc8(x);
        const int sy = trunc8(y);
        if (sx > w) sx = w;
        if (sx < 0) sx = -1;
        a = c = 0;
        ex = sx;
        ey = sy;
        last_ey = sy << 8;
        valid = sy == row && sx < w;
    }
    // sink of qtpath::flatten: the outline's points as QOutlineMapper hands them on (26.6 by qRound)
    PG_QT_FN void point(double px, double py) {
        const int qx = qtpath::q_round(px * 64), qy = qtpath::q_round(py * 64);
        if (n == 0) move_to(qx, qy);
        else line_to(qx << 2, qy << 2);
        n++;
    }
    // gray_hline's coverage of a cell / a run between cells (non-zero winding)
    PG_QT_FN static int coverage(int area_) {
        int cv = area_ >> 9;  // PIXEL_BITS * 2 + 1 - 8
        if (cv < 0) cv = -cv;
        if (cv >= 256) cv = 255;
        return cv;
    }
};

// the outline's vertical extent in pixel rows (a sink for qtpath::flatten)
struct RowExtent {
    int min_y, max_y, n;
    PG_QT_FN void point(double, double py) {
        const int q = qtpath::q_round(py * 64);
        if (n == 0 || q < min_y) min_y = q;
        if (n == 0 || q > max_y) max_y = q;
        n++;
    }
};

// ---- the pen: QCosmeticStroker with drawLineAA.  Sink: pixel(x, y, coverage 0..255), already clipped to the frame ----------------------
PG_QT_FN int fixdiv16(int x, int y) { return (int)(((long long)x << 16) / y); }  // F16Dot16FixedDiv

template <class Sink>
struct CosmeticAA {
    Sink &sink;
    double xmin, xmax, ymin, ymax;  // QCosmeticStroker::setup: the device rect widened by one pixel
    PG_QT_FN bool clip_line(double &x1, double &y1, double &x2, double &y2) const {  // QCosmeticStroker::clipLine
        if (x1 < xmin) {
            if (x2 <= xmin) return true;
            y1 += (y2 - y1) / (x2 - x1) * (xmin - x1);
            x1 = xmin;
        } else if (x1 > xmax) {
            if (x2 >= xmax) return true;
            y1 += (y2 - y1) / (x2 - x1) * (xmax - x1);
            x1 = xmax;
        }
        if (x2 < xmin) {
            y2 += (y2 - y1) / (x2 - x1) * (xmin - x2);
            x2 = xmin;
        } else if (x2 > xmax) {
            y2 += (y2 - y1) / (x2 - x1) * (xmax - x2);
            x2 = xmax;
        }
        if (y1 < ymin) {
            if (y2 <= ymin) return true;
            x1 += (x2 - x1) / (y2 - y1) * (ymin - y1);
            y1 = ymin;
        } else if (y1 > ymax) {
            if (y2 >= ymax) return true;
            x1 += (x2 - x1) / (y2 - y1) * (ymax - y1);
            y1 = ymax;
        }
        if (y2 < ymin) {
            x2 += (x2 - x1) / (y2 - y1) * (ymin - y2);
            y2 = ymin;
        } else if (y2 > ymax) {
            x2 += (x2 - x1) / (y2 - y1) * (ymax - y2);
            y2 = ymax;
        }
        return false;
    }
    PG_QT_FN void line(double rx1, double ry1, double rx2, double ry2, int caps) {  // drawLineAA<NoDasher>
        if (clip_line(rx1, ry1, rx2, ry2)) return;
        int x1 = (int)(rx1 * 64.), y1 = (int)(ry1 * 64.), x2 = (int)(rx2 * 64.), y2 = (int)(ry2 * 64.);
        const int dx = x2 - x1, dy = y2 - y1;
        if (qtpath::i_abs(dx) < qtpath::i_abs(dy)) {
            const int xinc = fixdiv16(dx, dy);
            if (y1 > y2) {
                int t = y1; y1 = y2; y2 = t;
                t = x1; x1 = x2; x2 = t;
                caps = ((caps & 1) << 1) | ((caps & 2) >> 1);
            }
            int x = (int)((unsigned)(x1 - 32) << 10);
            x -= (((y1 & 63) - 32) * xinc) >> 6;
            if (caps & 1) { y1 -= 32; x -= xinc >> 1; }
            if (caps & 2) y2 += 32;
            int y = y1 >> 6;
            const int ys = y2 >> 6;
            int a_start, a_end;
            if (y == ys) { a_start = y2 - y1; a_end = 0; }
            else { a_start = 64 - (y1 & 63); a_end = y2 & 63; }
            int al = (x >> 8) & 255;
            sink.pixel(x >> 16, y, (255 - al) * a_start >> 6);
            sink.pixel((x >> 16) + 1, y, al * a_start >> 6);
            x += xinc;
            ++y;
            if (y < ys) {
                do {
                    al = (x >> 8) & 255;
                    sink.pixel(x >> 16, y, 255 - al);
                    sink.pixel((x >> 16) + 1, y, al);
                    x += xinc;
                } while (++y < ys);
            }
            if (a_end) {
                al = (x >> 8) & 255;
                sink.pixel(x >> 16, y, (255 - al) * a_end >> 6);
                sink.pixel((x >> 16) + 1, y, al * a_end >> 6);
            }
        } else {
            if (!dx) return;
            const int yinc = fixdiv16(dy, dx);
            if (x1 > x2) {
                int t = y1; y1 = y2; y2 = t;
                t = x1; x1 = x2; x2 = t;
                caps = ((caps & 1) << 1) | ((caps & 2) >> 1);
            }
            int y = (int)((unsigned)(y1 - 32) << 10);
            y -= (((x1 & 63) - 32) * yinc) >> 6;
            if (caps & 1) { x1 -= 32; y -= yinc >> 1; }
            if (caps & 2) x2 += 32;
            int x = x1 >> 6;
            const int xs = x2 >> 6;
            int a_start, a_end;
            if (x == xs) { a_start = x2 - x1; a_end = 0; }
            else { a_start = 64 - (x1 & 63); a_end = x2 & 63; }
            int al = (y >> 8) & 255;
            sink.pixel(x, y >> 16, (255 - al) * a_start >> 6);
            sink.pixel(x, (y >> 16) + 1, al * a_start >> 6);
            y += yinc;
            ++x;
            if (x < xs) {
                do {
                    al = (y >> 8) & 255;
                    sink.pixel(x, y >> 16, 255 - al);
                    sink.pixel(x, (y >> 16) + 1, al);
                    y += yinc;
                } while (++x < xs);
            }
            if (a_end) {
                al = (y >> 8) & 255;
                sink.pixel(x, y >> 16, (255 - al) * a_end >> 6);
                sink.pixel(x, (y >> 16) + 1, al * a_end >> 6);
            }
        }
    }
    // renderCubicSubdivision with an explicit stack: p[0] is the END of the arc, p[3] its start; at most 6 levels
    PG_QT_FN void cubic(double sx, double sy, double c1x, double c1y, double c2x, double c2y, double ex_, double ey_) {
        double px[3 * 6 + 4], py[3 * 6 + 4];
        int level[7], base[7], cap[7];
        px[3] = sx; py[3] = sy; px[2] = c1x; py[2] = c1y; px[1] = c2x; py[1] = c2y; px[0] = ex_; py[0] = ey_;
        // depth-first: the half next to the start (points + 3) is drawn first
        int top = 0;
        base[0] = 0;
        level[0] = 6;
        cap[0] = 0;
        // an explicit stack of (base, level); splitting an arc at `b` leaves the end half at b and the start half at b + 3
        while (top >= 0) {
            const int b = base[top], lv = level[top];
            bool split = false;
            if (lv) {
                const double dx = px[b + 3] - px[b], dy = py[b + 3] - py[b];
                const double len = .25 * (qtpath::q_abs(dx) + qtpath::q_abs(dy));
                split = qtpath::q_abs(dx * (py[b] - py[b + 2]) - dy * (px[b] - px[b + 2])) >= len ||
                        qtpath::q_abs(dx * (py[b] - py[b + 1]) - dy * (px[b] - px[b + 1])) >= len;
            }
            if (split) {
                // splitCubic(points + b)
                for (int k = 0; k < 2; k++) {
                    double *v = k ? py : px;
                    v[b + 6] = v[b + 3];
                    double aa = v[b + 1], cc = v[b + 2];
                    v[b + 1] = (v[b] + aa) / 2;
                    v[b + 5] = (v[b + 3] + cc) / 2;
                    const double mid = (aa + cc) / 2;
                    v[b + 2] = (v[b + 1] + mid) / 2;
                    v[b + 4] = (v[b + 5] + mid) / 2;
                    v[b + 3] = (v[b + 2] + v[b + 4]) / 2;
                }
                // replace this entry by the end half (drawn second), push the start half (drawn first)
                level[top] = lv - 1;
                base[top + 1] = b + 3;
                level[top + 1] = lv - 1;
                cap[top + 1] = 0;
                top++;
                continue;
            }
            line(px[b + 3], py[b + 3], px[b], py[b], 0);
            top--;
        }
    }
    // QCosmeticStroker::drawPath on the closed 4-cubic ellipse path (closed: no caps)
    PG_QT_FN void ellipse(double x, double y, double w, double h) {
        qtpath::Arc a;
        qtpath::arc_points(x, y, w, h, a);
        for (int k = 0; k < 4; k++) cubic(a.x[3 * k], a.y[3 * k], a.x[3 * k + 1], a.y[3 * k + 1], a.x[3 * k + 2], a.y[3 * k + 2], a.x[3 * k + 3], a.y[3 * k + 3]);
    }
};

}  // namespace aapath
}  // namespace pgamd
