// pg_qtpath.h -- Qt 5.9's non-antialiased path route for QPainter::drawEllipse(QRectF) on a rect that is not integer
// aligned (identity transform, pen width <= 1).  Reference call sites: src/games/jumper.cpp:137-142 (compass in easy mode
// and without center_agent), src/assetgen.cpp:99-105 (generated assets).
//
// QRasterPaintEngine::drawEllipse only takes the midpoint algorithm (pg_render.h exec_ellipse) when the rect equals its
// integer truncation; otherwise QPaintEngineEx::drawEllipse builds qt_curves_for_arc(rect, 0, -360) -- a start point and
// four cubics -- and draws the path:
//   brush: QRasterPaintEngine::fill(QVectorPath).  Culled unless controlPointRect().toRect() (5.9 rounds x, y, w, h one by
//          one) intersects the device rect.  QOutlineMapper::curveTo flattens each cubic with QBezier::addToPolygon
//          (threshold .25), the points become 26.6 fixed point by qRound(v * 64), and QRasterizer's QScanConverter walks
//          every line in 16.16 sampled at the pixel centres; a row is filled between its two crossings.
//   pen:   QCosmeticStroker::drawPath: calculateLastPoint() on the last two points of the closed path, renderCubic (at most
//          6 levels of subdivision, flatness .25 * (|dx| + |dy|)), drawLine<drawPixel, NoDasher> per segment with the
//          stroker's duplicate-pixel / drop-out control between consecutive segments.
// Qt's sources are not on disk; the algorithm is pinned against PyQt5 5.9.7 by tests/tools/qt_path_probe.py (tens of
// thousands of random, knife-edge, tiny and partly-outside rects; pen, brush and both: 0 misses).
//
// Everything here is scalar code on doubles and 32-bit integers, callable from host code and from wave-uniform device code.
// The caller supplies a sink:  void span(int y, int x0, int x1)  (brush, x1 exclusive, already clipped to the canvas)  and
// void pixel(int x, int y)  (pen, already clipped).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define PG_QT_FN __host__ __device__ inline
#else
#define PG_QT_FN inline
#endif

namespace pgamd {
namespace qtpath {

constexpr double KAPPA = 0.5522847498;  // QT_PATH_KAPPA
constexpr int INT_MIN_ = -2147483647 - 1;

PG_QT_FN int q_round(double d) { return d >= 0.0 ? (int)(d + 0.5) : (int)(d - (double)((int)(d - 1)) + 0.5) + (int)(d - 1); }  // qglobal.h qRound
PG_QT_FN double q_abs(double v) { return v < 0 ? -v : v; }
PG_QT_FN int i_abs(int v) { return v < 0 ? -v : v; }

struct Arc {  // qpainterpath.cpp qt_curves_for_arc(rect, 0, -360): start point + 4 x (cp1, cp2, end)
    double x[13], y[13];
};
PG_QT_FN void arc_points(double x, double y, double w, double h, Arc &a) {
    const double w2 = w / 2, w2k = w2 * KAPPA, h2 = h / 2, h2k = h2 * KAPPA;
    a.x[0] = x + w;         a.y[0] = y + h2;
    a.x[1] = x + w;         a.y[1] = y + h2 + h2k;
    a.x[2] = x + w2 + w2k;  a.y[2] = y + h;
    a.x[3] = x + w2;        a.y[3] = y + h;
    a.x[4] = x + w2 - w2k;  a.y[4] = y + h;
    a.x[5] = x;             a.y[5] = y + h2 + h2k;
    a.x[6] = x;             a.y[6] = y + h2;
    a.x[7] = x;             a.y[7] = y + h2 - h2k;
    a.x[8] = x + w2 - w2k;  a.y[8] = y;
    a.x[9] = x + w2;        a.y[9] = y;
    a.x[10] = x + w2 + w2k; a.y[10] = y;
    a.x[11] = x + w;        a.y[11] = y + h2 - h2k;
    a.x[12] = x + w;        a.y[12] = y + h2;
}

// QRasterPaintEngine::drawEllipse's test for the midpoint route (rects here hold float-valued doubles, for which QRectF's
// fuzzy comparison is the exact one)
PG_QT_FN bool is_integer_rect(double x, double y, double w, double h) {
    return (double)(int)x == x && (double)(int)y == y && (double)(int)w == w && (double)(int)h == h;
}

// ---------------------------------------------------------------------------------------------------- brush
struct Bez {
    double x1, y1, x2, y2, x3, y3, x4, y4;
};
PG_QT_FN void bez_split(const Bez &b, Bez &first, Bez &second) {  // qbezier_p.h QBezier::split
    double c = (b.x2 + b.x3) * .5;
    first.x2 = (b.x1 + b.x2) * .5;
    second.x3 = (b.x3 + b.x4) * .5;
    first.x1 = b.x1;
    second.x4 = b.x4;
    first.x3 = (first.x2 + c) * .5;
    second.x2 = (second.x3 + c) * .5;
    first.x4 = second.x1 = (first.x3 + second.x2) * .5;
    c = (b.y2 + b.y3) / 2;
    first.y2 = (b.y1 + b.y2) * .5;
    second.y3 = (b.y3 + b.y4) * .5;
    first.y1 = b.y1;
    second.y4 = b.y4;
    first.y3 = (first.y2 + c) * .5;
    second.y2 = (second.y3 + c) * .5;
    first.y4 = second.y1 = (first.y3 + second.y2) * .5;
}

// One line of the flattened outline, from (ax, ay) to (bx, by) in 26.6: QScanConverter::mergeLine.  For every row whose
// centre it covers the sink's cross(y, x) is called with the 16.16 walker's integer part.
template <class Rows>
PG_QT_FN void merge_line(Rows &rows, int ax, int ay, int bx, int by, int top, int bot) {
    if (ay > by) {
        int t = ax; ax = bx; bx = t;
        t = ay; ay = by; by = t;
    }
    int itop = (ay + 32) >> 6, ibot = (by - 32) >> 6;
    if (itop < top) itop = top;
    if (ibot > bot) ibot = bot;
    if (itop > ibot) return;
    int xfp = 32768 + ax * 1024, slope = 0;
    if (bx != ax) {
        const double s = (double)(bx - ax) / (double)(by - ay);
        slope = (int)(s * 65536.);
        xfp += (int)(((long long)slope * (long long)((itop << 16) + 32768 - (ay << 10))) >> 16);
    }
    for (int y = itop; y <= ibot; y++) {
        rows.cross(y, xfp >> 16);
        xfp += slope;
    }
}

// Walks the flattened outline (QBezier::addToPolygon per cubic, in path order) and hands every point to `f(x, y)`.
template <class F>
PG_QT_FN void flatten(const Arc &a, F &f) {
    double lx = a.x[0], ly = a.y[0];
    f.point(lx, ly);
    for (int k = 0; k < 4; k++) {
        Bez stack[10];
        int levels[10];
        stack[0] = Bez{lx, ly, a.x[1 + 3 * k], a.y[1 + 3 * k], a.x[2 + 3 * k], a.y[2 + 3 * k], a.x[3 + 3 * k], a.y[3 + 3 * k]};
        levels[0] = 9;
        int top = 0;
        while (top >= 0) {
            Bez &b = stack[top];
            const double y4y1 = b.y4 - b.y1, x4x1 = b.x4 - b.x1;
            double l = q_abs(x4x1) + q_abs(y4y1), d;
            if (l > 1.) {
                d = q_abs(x4x1 * (b.y1 - b.y2) - y4y1 * (b.x1 - b.x2)) + q_abs(x4x1 * (b.y1 - b.y3) - y4y1 * (b.x1 - b.x3));
            } else {
                d = q_abs(b.x1 - b.x2) + q_abs(b.y1 - b.y2) + q_abs(b.x1 - b.x3) + q_abs(b.y1 - b.y3);
                l = 1.;
            }
            if (d < .25 * l || levels[top] == 0) {
                lx = b.x4;
                ly = b.y4;
                f.point(lx, ly);
                --top;
            } else {
                const Bez whole = b;
                bez_split(whole, stack[top + 1], stack[top]);
                levels[top + 1] = --levels[top];
                ++top;
            }
        }
    }
    if (lx != a.x[0] || ly != a.y[0]) f.point(a.x[0], a.y[0]);  // QOutlineMapper::closeSubpath
}

// pass 1 of the fill: vertical extent of the outline in 26.6;  pass 2: the lines
struct ExtentPass {
    int min_y, max_y, n;
    PG_QT_FN void point(double, double y) {
        const int q = q_round(y * 64);
        if (n == 0 || q < min_y) min_y = q;
        if (n == 0 || q > max_y) max_y = q;
        n++;
    }
};
template <class Rows>
struct LinePass {
    Rows &rows;
    int top, bot, px, py, n;
    PG_QT_FN void point(double x, double y) {
        const int qx = q_round(x * 64), qy = q_round(y * 64);
        if (n) merge_line(rows, px, py, qx, qy, top, bot);
        px = qx;
        py = qy;
        n++;
    }
};

// QRasterPaintEngine::fill's cull (see the header comment)
PG_QT_FN bool fill_culled(double x, double y, double w, double h, int cw, int ch) {
    const double l = x, r = x + w, t = y, b = y + h;  // QVectorPath::controlPointRect: min / max over the 13 points
    const int x1 = q_round(l), y1 = q_round(t), x2 = x1 + q_round(r - l) - 1, y2 = y1 + q_round(b - t) - 1;
    if (x2 == x1 - 1 && y2 == y1 - 1) return true;
    return x1 > cw - 1 || 0 > x2 || y1 > ch - 1 || 0 > y2;
}

// Brush part.  `rows` receives cross(y, x) twice for every row the (convex) outline covers, for rows top..bot that the call
// returns through `top` / `bot` (top > bot: nothing to fill); the caller fills [min, max) of a row's two crossings.
template <class Rows>
PG_QT_FN void fill_crossings(Rows &rows, double x, double y, double w, double h, int cw, int ch, int &top, int &bot) {
    top = 0;
    bot = -1;
    if (fill_culled(x, y, w, h, cw, ch)) return;
    Arc a;
    arc_points(x, y, w, h, a);
    ExtentPass e{0, 0, 0};
    flatten(a, e);
    int t = (e.min_y + 32) >> 6, b = (e.max_y - 32) >> 6;  // QRasterizer::rasterize
    if (t < 0) t = 0;
    if (b > ch - 1) b = ch - 1;
    if (t > b) return;
    top = t;
    bot = b;
    LinePass<Rows> lp{rows, t, b, 0, 0, 0};
    flatten(a, lp);
}

// ---------------------------------------------------------------------------------------------------- pen
enum { TB = 1, BT = 2, LR = 4, RL = 8 };
template <class Sink>
struct Cosmetic {  // QCosmeticStroker's state between segments
    Sink &sink;
    int cw, ch;
    int last_dir, last_x, last_y, last_axis_aligned;

    PG_QT_FN static int fixdiv(int x, int y) { return (int)(((long long)x << 16) / y); }  // F16Dot16FixedDiv
    PG_QT_FN void put(int x, int y) {
        if (x >= 0 && x < cw && y >= 0 && y < ch) sink.pixel(x, y);
    }
    // QCosmeticStroker::clipLine against the device rect widened by one pixel (setup())
    PG_QT_FN bool clip_line(double &x1, double &y1, double &x2, double &y2) {
        const double xmin = -1., xmax = cw + 1., ymin = -1., ymax = ch + 1.;
        if (x1 < xmin) {
            if (x2 <= xmin) { last_x = INT_MIN_; return true; }
            y1 += (y2 - y1) / (x2 - x1) * (xmin - x1);
            x1 = xmin;
        } else if (x1 > xmax) {
            if (x2 >= xmax) { last_x = INT_MIN_; return true; }
            y1 += (y2 - y1) / (x2 - x1) * (xmax - x1);
            x1 = xmax;
        }
        if (x2 < xmin) {
            last_x = INT_MIN_;
            y2 += (y2 - y1) / (x2 - x1) * (xmin - x2);
            x2 = xmin;
        } else if (x2 > xmax) {
            last_x = INT_MIN_;
            y2 += (y2 - y1) / (x2 - x1) * (xmax - x2);
            x2 = xmax;
        }
        if (y1 < ymin) {
            if (y2 <= ymin) { last_x = INT_MIN_; return true; }
            x1 += (x2 - x1) / (y2 - y1) * (ymin - y1);
            y1 = ymin;
        } else if (y1 > ymax) {
            if (y2 >= ymax) { last_x = INT_MIN_; return true; }
            x1 += (x2 - x1) / (y2 - y1) * (ymax - y1);
            y1 = ymax;
        }
        if (y2 < ymin) {
            last_x = INT_MIN_;
            x2 += (x2 - x1) / (y2 - y1) * (ymin - y2);
            y2 = ymin;
        } else if (y2 > ymax) {
            last_x = INT_MIN_;
            x2 += (x2 - x1) / (y2 - y1) * (ymax - y2);
            y2 = ymax;
        }
        return false;
    }
    // calculateLastPoint: where the closing segment of the path will end, so that the first one joins it
    PG_QT_FN void calculate_last_point(double rx1, double ry1, double rx2, double ry2) {
        last_x = INT_MIN_;
        last_y = INT_MIN_;
        if (clip_line(rx1, ry1, rx2, ry2)) return;
        int x1 = (int)(rx1 * 64.), y1 = (int)(ry1 * 64.), x2 = (int)(rx2 * 64.), y2 = (int)(ry2 * 64.);
        const int dx = i_abs(x2 - x1), dy = i_abs(y2 - y1);
        if (dx < dy) {
            bool swapped = false;
            if (y1 > y2) {
                swapped = true;
                int t = y1; y1 = y2; y2 = t;
                t = x1; x1 = x2; x2 = t;
            }
            const int xinc = fixdiv(x2 - x1, y2 - y1);
            int x = x1 * 1024;
            const int y = (y1 + 32) >> 6, ys = (y2 + 32) >> 6, rnd = xinc > 0 ? 32 : 0;
            if (y != ys) {
                x += (int)(((long long)((y * 64) + rnd - y1) * xinc) >> 6);
                if (swapped) {
                    last_x = x >> 16; last_y = y; last_dir = BT;
                } else {
                    last_x = (x + (ys - y - 1) * xinc) >> 16; last_y = ys - 1; last_dir = TB;
                }
                last_axis_aligned = i_abs(xinc) < (1 << 14);
            }
        } else {
            if (!dx) return;
            bool swapped = false;
            if (x1 > x2) {
                swapped = true;
                int t = y1; y1 = y2; y2 = t;
                t = x1; x1 = x2; x2 = t;
            }
            const int yinc = fixdiv(y2 - y1, x2 - x1);
            int y = y1 * 1024;
            const int x = (x1 + 32) >> 6, xs = (x2 + 32) >> 6, rnd = yinc > 0 ? 32 : 0;
            if (x != xs) {
                y += (int)(((long long)((x * 64) + rnd - x1) * yinc) >> 6);
                if (swapped) {
                    last_x = x; last_y = y >> 16; last_dir = RL;
                } else {
                    last_x = xs - 1; last_y = (y + (xs - x - 1) * yinc) >> 16; last_dir = LR;
                }
                last_axis_aligned = i_abs(yinc) < (1 << 14);
            }
        }
    }
    // drawLine<drawPixel, NoDasher>
    PG_QT_FN void line(double rx1, double ry1, double rx2, double ry2, int caps) {
        if (clip_line(rx1, ry1, rx2, ry2)) return;
        int x1 = (int)(rx1 * 64.), y1 = (int)(ry1 * 64.), x2 = (int)(rx2 * 64.), y2 = (int)(ry2 * 64.);
        const int dx = i_abs(x2 - x1), dy = i_abs(y2 - y1);
        int nlx = last_x, nly = last_y;  // "QCosmeticStroker::Point last = stroker->lastPixel"
        const int lpx = last_x, lpy = last_y;
        if (dx < dy) {
            int dir = TB;
            bool swapped = false;
            if (y1 > y2) {
                swapped = true;
                int t = y1; y1 = y2; y2 = t;
                t = x1; x1 = x2; x2 = t;
                caps = ((caps & 1) << 1) | ((caps & 2) >> 1);
                dir = BT;
            }
            const int xinc = fixdiv(x2 - x1, y2 - y1);
            int x = x1 * 1024;
            if ((last_dir ^ 3) == dir) caps |= swapped ? 2 : 1;  // the path turned around: cap towards the previous segment
            if (caps & 1) {  // capAdjust
                y1 -= 32;
                x -= xinc >> 1;
            }
            if (caps & 2) y2 += 32;
            int y = (y1 + 32) >> 6, ys = (y2 + 32) >> 6;
            const int rnd = xinc > 0 ? 32 : 0;
            if ((caps & 1) && lpy == y + 1) y++;  // "capAdjust made us round away from what calculateLastPoint gave us"
            if (y != ys) {
                x += (int)(((long long)((y * 64) + rnd - y1) * xinc) >> 6);
                int fx = x >> 16, fy = y;
                nlx = (x + (ys - y - 1) * xinc) >> 16;
                nly = ys - 1;
                if (swapped) {
                    int t = fx; fx = nlx; nlx = t;
                    t = fy; fy = nly; nly = t;
                }
                const int axis_aligned = i_abs(xinc) < (1 << 14);
                if (lpx > INT_MIN_) {
                    if (fx == lpx && fy == lpy) {  // remove duplicated pixel
                        if (swapped) {
                            --ys;
                        } else {
                            ++y;
                            x += xinc;
                        }
                    } else if (last_dir != dir && ((axis_aligned && last_axis_aligned && lpx != fx && lpy != fy) || (i_abs(lpx - fx) > 1 || i_abs(lpy - fy) > 1))) {  // missing pixel: insert it
                        if (swapped) {
                            ++ys;
                        } else {
                            --y;
                            x -= xinc;
                        }
                    } else if (last_dir == dir && (i_abs(lpx - fx) <= 1 && i_abs(lpy - fy) > 1)) {
                        x += xinc >> 1;
                        if (swapped) nlx = x >> 16;
                        else nlx = (x + (ys - y - 1) * xinc) >> 16;
                    }
                }
                last_dir = dir;
                last_axis_aligned = axis_aligned;
                do {
                    put(x >> 16, y);
                    x += xinc;
                } while (++y < ys);
            }
        } else {
            if (!dx) return;
            int dir = LR;
            bool swapped = false;
            if (x1 > x2) {
                swapped = true;
                int t = y1; y1 = y2; y2 = t;
                t = x1; x1 = x2; x2 = t;
                caps = ((caps & 1) << 1) | ((caps & 2) >> 1);
                dir = RL;
            }
            const int yinc = fixdiv(y2 - y1, x2 - x1);
            int y = y1 * 1024;
            if ((last_dir ^ 0xc) == dir) caps |= swapped ? 2 : 1;
            if (caps & 1) {
                x1 -= 32;
                y -= yinc >> 1;
            }
            if (caps & 2) x2 += 32;
            int x = (x1 + 32) >> 6, xs = (x2 + 32) >> 6;
            const int rnd = yinc > 0 ? 32 : 0;
            if ((caps & 1) && lpx == x + 1) x++;
            if (x != xs) {
                y += (int)(((long long)((x * 64) + rnd - x1) * yinc) >> 6);
                int fx = x, fy = y >> 16;
                nlx = xs - 1;
                nly = (y + (xs - x - 1) * yinc) >> 16;
                if (swapped) {
                    int t = fx; fx = nlx; nlx = t;
                    t = fy; fy = nly; nly = t;
                }
                const int axis_aligned = i_abs(yinc) < (1 << 14);
                if (lpx > INT_MIN_) {
                    if (fx == lpx && fy == lpy) {
                        if (swapped) {
                            --xs;
                        } else {
                            ++x;
                            y += yinc;
                        }
                    } else if (last_dir != dir && ((axis_aligned && last_axis_aligned && lpx != fx && lpy != fy) || (i_abs(lpx - fx) > 1 || i_abs(lpy - fy) > 1))) {
                        if (swapped) {
                            ++xs;
                        } else {
                            --x;
                            y -= yinc;
                        }
                    } else if (last_dir == dir && (i_abs(lpx - fx) <= 1 && i_abs(lpy - fy) > 1)) {  // sic: the vertical branch's test (probe)
                        y += yinc >> 1;
                        if (swapped) nly = y >> 16;
                        else nly = (y + (xs - x - 1) * yinc) >> 16;
                    }
                }
                last_dir = dir;
                last_axis_aligned = axis_aligned;
                do {
                    put(x, y >> 16);
                    y += yinc;
                } while (++x < xs);
            }
        }
        last_x = nlx;
        last_y = nly;
    }
    // renderCubic / renderCubicSubdivision with an explicit stack: p[3] = start ... p[0] = end of the piece on top
    PG_QT_FN void cubic(double x1, double y1, double x2, double y2, double x3, double y3, double x4, double y4) {
        double px[3 * 6 + 4], py[3 * 6 + 4];
        int off[8], lvl[8];  // pieces still to draw, first-to-last order = pop order
        px[3] = x1; py[3] = y1;
        px[2] = x2; py[2] = y2;
        px[1] = x3; py[1] = y3;
        px[0] = x4; py[0] = y4;
        int n = 0;
        off[n] = 0;
        lvl[n] = 6;
        n++;
        while (n > 0) {
            --n;
            const int o = off[n];
            int level = lvl[n];
            double *p = px + o, *q = py + o;
            bool split = false;
            if (level) {
                const double dx = p[3] - p[0], dy = q[3] - q[0];
                const double len = .25 * (q_abs(dx) + q_abs(dy));
                split = q_abs(dx * (q[0] - q[2]) - dy * (p[0] - p[2])) >= len || q_abs(dx * (q[0] - q[1]) - dy * (p[0] - p[1])) >= len;
            }
            if (split) {
                for (int k = 0; k < 2; k++) {  // splitCubic
                    double *v = k ? q : p, a, b, c, d;
                    v[6] = v[3];
                    c = v[1];
                    d = v[2];
                    v[1] = a = (v[0] + c) * .5;
                    v[5] = b = (v[3] + d) * .5;
                    c = (c + d) * .5;
                    v[2] = a = (a + c) * .5;
                    v[4] = b = (b + c) * .5;
                    v[3] = (a + b) * .5;
                }
                --level;
                off[n] = o;  // second half (drawn after ...
                lvl[n] = level;
                n++;
                off[n] = o + 3;  // ... the first half)
                lvl[n] = level;
                n++;
            } else {
                line(p[3], q[3], p[0], q[0], 0);
            }
        }
    }
};

// Pen part: QCosmeticStroker::drawPath on the closed 4-cubic path.
template <class Sink>
PG_QT_FN void stroke_ellipse(Sink &sink, double x, double y, double w, double h, int cw, int ch) {
    Arc a;
    arc_points(x, y, w, h, a);
    Cosmetic<Sink> s{sink, cw, ch, LR, INT_MIN_, INT_MIN_, 0};
    s.calculate_last_point(a.x[11], a.y[11], a.x[12], a.y[12]);
    for (int k = 0; k < 4; k++) s.cubic(a.x[3 * k], a.y[3 * k], a.x[3 * k + 1], a.y[3 * k + 1], a.x[3 * k + 2], a.y[3 * k + 2], a.x[3 * k + 3], a.y[3 * k + 3]);
}

}  // namespace qtpath
}  // namespace pgamd
