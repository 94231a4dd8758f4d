// pg_bgpaint.h -- use_generated_assets: the episode's 500 x 500 background (reference BAG:58-63,769-773: AssetGen bggen(&rand_gen);
// bggen.generate_resource(main_bg_image) inside game_reset) painted on the device by one wavefront per env.
//
// The reset path (Env::bag_game_reset) runs the generator without a painter to consume exactly its rand_gen draws and leaves
// a request {level seed, draws made since the reseed} for the env; this kernel re-seeds a private MT19937 from the level
// seed, skips those draws and runs the same generator (pg_assetgen.h) with a painter on the env's canvas in HBM
// (DevCtx::gen_bg).  The generator is scalar (wave-uniform) code; the painter's pixel work is laid over the lanes:
//   fillRect        rows x 64-pixel chunks (a full-width rect is one linear range)
//   ellipse brush   the outline is flattened once into LDS (pg_qtpath.h flatten), then 64 rows at a time: lane = row, each
//                   lane walks the segments for its own two crossings (QScanConverter::mergeLine's arithmetic), and the
//                   rows' spans are written chunk by chunk
//   ellipse pen     QCosmeticStroker's pixels are collected 64 at a time and written together
#pragma once
#include "pg_assetgen.h"
#include "pg_defs.h"
#include "wave.h"

namespace pgamd {

constexpr int BGP_MAX_POINTS = 600;

struct BgPaintLds {
    uint32_t mt[2][MT_N];          // the generator state and its twist target
    int px[BGP_MAX_POINTS], py[BGP_MAX_POINTS];  // flattened outline in 26.6
    uint32_t pen[64];              // pending pen pixels (y * 512 + x)
};

// std::mt19937 for one wave: state in LDS, twist lane-parallel between two buffers (as Env::mt_twist)
struct WaveMT {
    BgPaintLds *s;
    int cur, idx;
    PG_DEV void seed(int seed_) {
        cur = 0;
        uint32_t x = (uint32_t)seed_;
        for (int i = 0; i < MT_N; i++) {
            if (i > 0) x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)i;
            PG_FOR_LANES(l) {
                if (l == 0) s->mt[0][i] = x;
            }
        }
        PG_SYNC();
        idx = MT_N;
    }
    PG_DEV void twist() {
        const uint32_t *src = s->mt[cur];
        uint32_t *dst = s->mt[cur ^ 1];
        for (int base = 0; base < 227; base += 64) {
            PG_FOR_LANES(l) {
                const int k = base + l;
                if (k < 227) {
                    const uint32_t y = (src[k] & 0x80000000u) | (src[k + 1] & 0x7fffffffu);
                    dst[k] = src[k + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
                }
            }
        }
        PG_SYNC();
        for (int lo = 227; lo < 623; lo += 227) {  // [227, 454), [454, 623): new[k - 227] comes from the phase before
            const int hi = lo + 227 < 623 ? lo + 227 : 623;
            for (int base = lo; base < hi; base += 64) {
                PG_FOR_LANES(l) {
                    const int k = base + l;
                    if (k < hi) {
                        const uint32_t y = (src[k] & 0x80000000u) | (src[k + 1] & 0x7fffffffu);
                        dst[k] = dst[k - 227] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
                    }
                }
            }
            PG_SYNC();
        }
        {
            const uint32_t y = (src[623] & 0x80000000u) | (dst[0] & 0x7fffffffu);
            const uint32_t v = dst[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            PG_FOR_LANES(l) {
                if (l == 0) dst[623] = v;
            }
        }
        PG_SYNC();
        cur ^= 1;
        idx = 0;
    }
    PG_DEV uint32_t u32() {
        if (idx >= MT_N) twist();
        uint32_t z = (uint32_t)PG_UNIFORM_I(s->mt[cur][idx]);
        idx++;
        z ^= (z >> 11);
        z ^= (z << 7) & 0x9d2c5680u;
        z ^= (z << 15) & 0xefc60000u;
        z ^= (z >> 18);
        return z;
    }
};

struct BgPainter {
    uint32_t *cv;  // [GEN_BG_DIM][GEN_BG_DIM] 0xAARRGGBB
    BgPaintLds *s;
    int npen;
    int error;
    static constexpr int W = GEN_BG_DIM, H = GEN_BG_DIM;

    PG_DEV static uint32_t veil(uint32_t c, uint32_t d) { return c + byte_mul_(d, 255u - (c >> 24)); }
    PG_DEV static uint32_t byte_mul_(uint32_t x, uint32_t a) {  // Qt BYTE_MUL
        uint32_t t = (x & 0xff00ffu) * a;
        t = (t + ((t >> 8) & 0xff00ffu) + 0x800080u) >> 8;
        t &= 0xff00ffu;
        x = ((x >> 8) & 0xff00ffu) * a;
        x = (x + ((x >> 8) & 0xff00ffu) + 0x800080u);
        x &= 0xff00ff00u;
        return x | t;
    }
    // pixels [i0, i1) of the linear canvas
    PG_DEV void range(int i0, int i1, uint32_t c, bool over) {
        for (int b = i0; b < i1; b += 64) {
            PG_FOR_LANES(l) {
                if (b + l < i1) cv[b + l] = over ? veil(c, cv[b + l]) : c;
            }
        }
    }
    PG_DEV void span(int y, int x0, int x1, uint32_t c) {  // [x0, x1) of row y, clipped
        if (y < 0 || y >= H) return;
        if (x0 < 0) x0 = 0;
        if (x1 > W) x1 = W;
        if (x1 > x0) range(y * W + x0, y * W + x1, c, false);
    }
    PG_DEV void fill(double x, double y, double ww, double hh, uint32_t c, bool over) {  // QPainter::fillRect(QRectF, QColor)
        int x1 = qtpath::q_round(x), x2 = qtpath::q_round(x + ww), y1 = qtpath::q_round(y), y2 = qtpath::q_round(y + hh);
        if (x2 < x1) { const int t = x1; x1 = x2; x2 = t; }
        if (y2 < y1) { const int t = y1; y1 = y2; y2 = t; }
        if (x1 < 0) x1 = 0;
        if (y1 < 0) y1 = 0;
        if (x2 > W) x2 = W;
        if (y2 > H) y2 = H;
        if (x2 <= x1 || y2 <= y1) return;
        if (x1 == 0 && x2 == W) {
            range(y1 * W, y2 * W, c, over);
        } else {
            for (int yy = y1; yy < y2; yy++) range(yy * W + x1, yy * W + x2, c, over);
        }
        PG_SYNC();
    }
    // ---- ellipse ----
    struct PointSink {  // qtpath::flatten -> 26.6 points in LDS
        BgPainter &p;
        int n, min_y, max_y;
        PG_DEV void point(double x, double y) {
            const int qx = qtpath::q_round(x * 64), qy = qtpath::q_round(y * 64);
            if (n == 0 || qy < min_y) min_y = qy;
            if (n == 0 || qy > max_y) max_y = qy;
            if (n < BGP_MAX_POINTS) {
                PG_FOR_LANES(l) {
                    if (l == 0) {
                        p.s->px[n] = qx;
                        p.s->py[n] = qy;
                    }
                }
            } else {
                p.error = 1;
            }
            n++;
        }
    };
    PG_DEV void pixel(int x, int y) {  // QCosmeticStroker's drawPixel (clipped by the stroker), opaque pen
        PG_FOR_LANES(l) {
            if (l == 0) s->pen[npen] = (uint32_t)(y * 512 + x);
        }
        npen++;
        if (npen == 64) flush_pen();
    }
    uint32_t pen_px;
    PG_DEV void flush_pen() {
        PG_SYNC();
        PG_FOR_LANES(l) {
            if (l < npen) {
                const uint32_t v = s->pen[l];
                cv[(int)(v >> 9) * W + (int)(v & 511u)] = pen_px;
            }
        }
        PG_SYNC();
        npen = 0;
    }
    PG_DEV void ellipse(double x, double y, double ww, double hh, uint32_t pen_c, uint32_t brush_c) {  // QPainter::drawEllipse(QRectF)
        if (ww < 0) { x += ww; ww = -ww; }
        if (hh < 0) { y += hh; hh = -hh; }
        if ((ww > hh ? ww : hh) < 32767 && ww > 0 && hh > 0 && qtpath::is_integer_rect(x, y, ww, hh)) {
            assetgen::midpoint_ellipse(*this, (int)x, (int)y, (int)ww, (int)hh, pen_c, brush_c);
            PG_SYNC();
            return;
        }
        if (ww == 0 && hh == 0) return;
        if (!qtpath::fill_culled(x, y, ww, hh, W, H)) {
            qtpath::Arc a;
            qtpath::arc_points(x, y, ww, hh, a);
            PointSink ps{*this, 0, 0, 0};
            qtpath::flatten(a, ps);
            PG_SYNC();
            const int n = ps.n < BGP_MAX_POINTS ? ps.n : BGP_MAX_POINTS;
            int top = (ps.min_y + 32) >> 6, bot = (ps.max_y - 32) >> 6;  // QRasterizer::rasterize
            if (top < 0) top = 0;
            if (bot > H - 1) bot = H - 1;
            for (int rb = top; rb <= bot; rb += 64) {
                PG_LANE_VAR(int, xa);
                PG_LANE_VAR(int, xb);
                PG_LANE_VAR(int, cnt);
                PG_FOR_LANES(l) {
                    const int yrow = rb + l;
                    int lo = 0, hi = 0, c = 0;
                    if (yrow <= bot) {
                        for (int i = 0; i + 1 < n; i++) {  // QScanConverter::mergeLine for this lane's row
                            int ax = s->px[i], ay = s->py[i], bx = s->px[i + 1], by = s->py[i + 1];
                            if (ay > by) {
                                int t = ax; ax = bx; bx = t;
                                t = ay; ay = by; by = t;
                            }
                            int itop = (ay + 32) >> 6, ibot = (by - 32) >> 6;
                            if (itop < top) itop = top;
                            if (ibot > bot) ibot = bot;
                            if (yrow < itop || yrow > ibot) continue;
                            int xfp = 32768 + ax * 1024;
                            if (bx != ax) {
                                const double sl = (double)(bx - ax) / (double)(by - ay);
                                const int slope = (int)(sl * 65536.);
                                xfp += (int)(((long long)slope * (long long)((itop << 16) + 32768 - (ay << 10))) >> 16);
                                xfp += slope * (yrow - itop);
                            }
                            const int xi = xfp >> 16;
                            if (c == 0) lo = hi = xi;
                            else {
                                if (xi < lo) lo = xi;
                                if (xi > hi) hi = xi;
                            }
                            c++;
                        }
                    }
                    PG_LV(xa, l) = lo;
                    PG_LV(xb, l) = hi;
                    PG_LV(cnt, l) = c;
                }
                const int nrows = bot - rb + 1 < 64 ? bot - rb + 1 : 64;
                for (int r = 0; r < nrows; r++) {
                    if (PG_READLANE(cnt, r) == 2) span(rb + r, PG_READLANE(xa, r), PG_READLANE(xb, r), brush_c);
                }
            }
            PG_SYNC();
        }
        pen_px = pen_c;
        npen = 0;
        qtpath::stroke_ellipse(*this, x, y, ww, hh, W, H);
        if (npen > 0) flush_pen();
    }
};

// one env's background: seed, skip, generate
PG_DEV void paint_background(uint32_t *canvas, int level_seed, int skip_draws, BgPaintLds *lds, int *error) {
    WaveMT mt{lds, 0, MT_N};
    mt.seed(level_seed);
    for (int i = 0; i < skip_draws; i++) (void)mt.u32();
    BgPainter painter{canvas, lds, 0, 0, 0u};
    assetgen::Gen<WaveMT, BgPainter> gen{mt, painter};
    gen.generate_resource(GEN_BG_DIM, GEN_BG_DIM, 1, 50, true);
    if (painter.error) *error = 1;
}

}  // namespace pgamd
