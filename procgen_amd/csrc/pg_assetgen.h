// pg_assetgen.h -- AssetGen (reference src/assetgen.cpp): the procedurally painted sprites and backgrounds of
// use_generated_assets.  The generator is a stream of rand_gen draws that decides rects and colours, and three painter calls:
// fillRect(QRectF, opaque colour), fillRect(QRectF, colour with alpha 200) and drawEllipse(QRectF) with a brush and a width-1
// pen.  This header holds the generator (scalar code, host and device) over two policies:
//   Rng     : uint32_t u32()                                   -- one std::mt19937 draw
//   Painter : void fill(x, y, w, h, px, over)                  -- [qRound(l), qRound(r)) x [qRound(t), qRound(b)), px premultiplied;
//                                                                  over = SourceOver (only the alpha-200 veil), else overwrite
//             void ellipse(x, y, w, h, pen_px, brush_px)       -- QPainter::drawEllipse(QRectF), opaque colours
// Who runs it: the host paints the 64 x 64 sprite of every object type once per handle (seed fixed_asset_seed + type, BAG:100-107);
// the reset path on the device runs it WITHOUT a painter to consume exactly the draws of the episode's 500 x 500 background
// (BAG:769-773), and the background kernel (kernels.hip paint_backgrounds) runs it again with a painter on the env's canvas in HBM.
// Float / double promotions follow the reference expressions; Qt's colour arithmetic is restated from qrgba64.h.
#pragma once
#include <stdint.h>

#include "pg_qtpath.h"

namespace pgamd {
namespace assetgen {

struct NoPainter {
    PG_QT_FN void fill(double, double, double, double, uint32_t, bool) {}
    PG_QT_FN void ellipse(double, double, double, double, uint32_t, uint32_t) {}
};

// QColor(r, g, b, 200) as the raster engine hands it to the span filler: qPremultiply(QRgba64).toArgb32() (qrgba64.h)
PG_QT_FN uint32_t premul_alpha200(uint32_t rgb) {
    const uint32_t a16 = 200u * 257u;
    uint32_t out = 0;
    for (int sh = 16; sh >= 0; sh -= 8) {
        const uint32_t c16 = ((rgb >> sh) & 0xffu) * 257u;
        uint32_t x = c16 * a16;
        x = (x + (x >> 16) + 0x8000u) >> 16;  // div_65535
        x += 128;                              // div_257
        x = (x - (x >> 8)) >> 8;
        out |= x << sh;
    }
    uint32_t al = a16 + 128;
    al = (al - (al >> 8)) >> 8;
    return out | (al << 24);
}

struct Rect {
    double x, y, w, h;
};

template <class Rng, class Painter>
struct Gen {
    Rng &rng;
    Painter &p;
    struct ColorGen {  // assetgen.cpp:3-29
        float rgb_start[3], rgb_len[3], p_rect;
    };
    PG_QT_FN float rand01() { return (float)((double)rng.u32() / 4294967296.0); }  // randgen.cpp:19-23
    PG_QT_FN int randn(int high) { return (int)(rng.u32() % (uint32_t)high); }
    PG_QT_FN int randint(int low, int high) { return (int)((uint32_t)low + (rng.u32() % (uint32_t)(high - low))); }
    PG_QT_FN bool randbool() { return (double)rand01() > .5; }
    PG_QT_FN void roll(ColorGen &c) {
        for (int i = 0; i < 3; i++) c.rgb_len[i] = rand01();
        for (int i = 0; i < 3; i++) c.rgb_start[i] = rand01() * (1 - c.rgb_len[i]);
        c.p_rect = rand01();
    }
    PG_QT_FN uint32_t rand_color(const ColorGen &c) {  // -> 0xffRRGGBB
        uint32_t px = 0xff000000u;
        for (int i = 0; i < 3; i++) {
            const int ch = (int)(255 * (rand01() * c.rgb_len[i] + c.rgb_start[i]));
            px |= ((uint32_t)ch & 0xffu) << (16 - 8 * i);
        }
        return px;
    }
    PG_QT_FN Rect choose_sub_rect(const Rect &rect, float min_dim, float max_dim) {  // assetgen.cpp:35-52
        const int w = (int)rect.w, h = (int)rect.h;
        const int smaller = (w > h) ? h : w;
        const float del_dim = max_dim - min_dim;
        const float rdx = (rand01() * del_dim + min_dim) * smaller;
        const float rdy = (rand01() * del_dim + min_dim) * smaller;
        const float rx_off = rand01() * (w - rdx);
        const float ry_off = rand01() * (h - rdy);
        return Rect{(double)rx_off + rect.x, (double)ry_off + rect.y, (double)rdx, (double)rdy};
    }
    PG_QT_FN void paint_shape(const Rect &main_rect, const ColorGen &cgen) {  // assetgen.cpp:77-107 (split_rect :54-75)
        const int k = randn(10);
        const int num_splits = (k * k) / 50 + 1;
        const bool is_horizontal = randbool();
        const float x = (float)main_rect.x, y = (float)main_rect.y, w = (float)main_rect.w, h = (float)main_rect.h;
        const float dw = w / num_splits, dh = h / num_splits;
        const bool use_rect = randbool();
        const bool regen_colors = randbool();
        uint32_t c1 = rand_color(cgen);
        uint32_t c2 = rand_color(cgen);
        for (int i = 0; i < num_splits; i++) {
            Rect r;
            if (is_horizontal) r = Rect{(double)(x + i * dw), (double)y, (double)dw, (double)h};
            else r = Rect{(double)x, (double)(y + i * dh), (double)w, (double)dh};
            if (regen_colors) {
                c1 = rand_color(cgen);
                c2 = rand_color(cgen);
            }
            if (use_rect) p.fill(r.x, r.y, r.w, r.h, c1, false);
            else p.ellipse(r.x, r.y, r.w, r.h, c2, c1);
        }
    }
    template <int DEPTH>
    PG_QT_FN void paint_rect_resource(const Rect &rect, int num_recurse, int blotch_scale) {  // assetgen.cpp:109-138
        ColorGen cgen;
        roll(cgen);
        const uint32_t bgcolor = rand_color(cgen);
        p.fill(rect.x, rect.y, rect.w, rect.h, bgcolor, false);
        const float scale = (float)(.3 + .7 * (double)rand01());
        const float max_rand_dim = (float)(.5 * (double)scale);
        const float min_rand_dim = (float)(.05 * (double)scale);
        const int num_blotches = randint(blotch_scale, 2 * blotch_scale);
        const float p_recurse = (float)((double)rand01() * .75);
        for (int j = 0; j < num_blotches; j++) {
            const Rect dst3 = choose_sub_rect(rect, min_rand_dim, max_rand_dim);
            bool recurse = false;
            if constexpr (DEPTH > 0) recurse = (num_recurse > 0) && (rand01() < p_recurse);
            if (recurse) {
                if constexpr (DEPTH > 0) paint_rect_resource<DEPTH - 1>(dst3, num_recurse - 1, 10);
            } else {
                paint_shape(dst3, cgen);
            }
        }
        p.fill(rect.x, rect.y, rect.w, rect.h, premul_alpha200(bgcolor), true);
    }
    PG_QT_FN Rect create_bar(const Rect &rect, bool is_horizontal) {  // assetgen.cpp:140-155
        const float k1 = (float)(.45 + (double)rand01() * .4);
        const float k2 = (float)(.45 + (double)rand01() * .4);
        const float w = (float)(rect.w * (double)k1 * (double)k1);
        const float h = (float)(rect.h * (double)k2 * (double)k2);
        const float pct = rand01();
        if (!is_horizontal) return Rect{0, (rect.h - (double)h) * (double)pct, rect.w, (double)h};
        return Rect{(rect.h - (double)w) * (double)pct, 0, (double)w, rect.h};
    }
    PG_QT_FN void paint_shape_resource(const Rect &rect) {  // assetgen.cpp:157-190
        ColorGen cgen;
        roll(cgen);
        const bool horizontal_first = randbool();
        const int nbar1 = randn(3) / 2 + 1;
        const int nbar2 = randn(3) / 2 + 1;
        p.fill(rect.x, rect.y, rect.w, rect.h, 0u, false);  // CompositionMode_Source, QColor(0, 0, 0, 0)
        for (int i = 0; i < nbar1; i++) {
            const Rect c1 = create_bar(rect, horizontal_first);
            paint_shape(c1, cgen);
        }
        for (int i = 0; i < nbar2; i++) {
            const Rect c2 = create_bar(rect, !horizontal_first);
            paint_shape(c2, cgen);
        }
        const int num_blotches = randint(1, 5);
        for (int j = 0; j < num_blotches; j++) {
            const Rect d = choose_sub_rect(rect, 0.1f, 0.6f);
            paint_shape(d, cgen);
        }
    }
    // AssetGen::generate_resource (assetgen.cpp:192-201); num_recurse is 0 (sprites) or 1 (backgrounds)
    PG_QT_FN void generate_resource(int w, int h, int num_recurse, int blotch_scale, bool is_rect) {
        const Rect rect{0, 0, (double)w, (double)h};
        if (!is_rect) paint_shape_resource(rect);
        else if (num_recurse > 0) paint_rect_resource<1>(rect, num_recurse, blotch_scale);
        else paint_rect_resource<0>(rect, 0, blotch_scale);
    }
};

// ---- a painter over a w x h canvas of 0xAARRGGBB words in ordinary memory (host; tests) ----------------------------------
// spans of the integer-rect ellipse (QRasterPaintEngine's drawEllipse_midpoint_i / drawEllipsePoints, Qt 5.9.7) in the order Qt emits them
template <class Sink>
PG_QT_FN void midpoint_ellipse(Sink &s, int rx, int ry, int rw, int rh, uint32_t pen_px, uint32_t brush_px) {
    if (rw <= 0 || rh <= 0) return;
    auto points = [&](int px_, int py_, int length) {
        if (length == 0) return;
        const int midx = rx + (rw + 1) / 2, midy = ry + (rh + 1) / 2;
        const int x = px_ + midx, y = midy - py_;
        const int o0x = midx + midx - x - (length - 1) - (rw & 1);
        const int o0len = length < x - o0x ? length : x - o0x;
        const int o2y = midy + midy - y - (rh & 1);
        if (o0x + o0len < x) {
            const int f0x = o0x + o0len - 1, f0len = x - f0x > 0 ? x - f0x : 0;
            s.span(y, f0x, f0x + f0len, brush_px);
            if (!(y >= o2y)) s.span(o2y, f0x, f0x + f0len, brush_px);
        }
        s.span(y, o0x, o0x + o0len, pen_px);
        s.span(y, x, x + length, pen_px);
        if (!(y >= o2y)) {
            s.span(o2y, o0x, o0x + o0len, pen_px);
            s.span(o2y, x, x + length, pen_px);
        }
    };
    const double a = rw / 2.0, b = rh / 2.0;
    double d = b * b - (a * a * b) + 0.25 * a * a;
    int x = 0, y = (rh + 1) / 2, startx = x;
    while (a * a * (2 * y - 1) > 2 * b * b * (x + 1)) {
        if (d < 0) {
            d += b * b * (2 * x + 3);
            ++x;
        } else {
            d += b * b * (2 * x + 3) + a * a * (-2 * y + 2);
            points(startx, y, x - startx + 1);
            startx = ++x;
            --y;
        }
    }
    points(startx, y, x - startx + 1);
    d = b * b * (x + 0.5) * (x + 0.5) + a * a * ((y - 1) * (y - 1) - b * b);
    const int miny = rh & 1;
    while (y > miny) {
        if (d < 0) {
            d += b * b * (2 * x + 2) + a * a * (-2 * y + 3);
            ++x;
        } else {
            d += a * a * (-2 * y + 3);
        }
        --y;
        points(x, y, 1);
    }
}

struct MemPainter {
    uint32_t *px;
    int w, h;
    int *cnt, *xa;  // per-row scratch of the path route (h ints each)
    uint32_t cur_brush;
    static uint32_t byte_mul(uint32_t x, uint32_t a) {  // Qt BYTE_MUL
        uint32_t t = (x & 0xff00ffu) * a;
        t = (t + ((t >> 8) & 0xff00ffu) + 0x800080u) >> 8;
        t &= 0xff00ffu;
        x = ((x >> 8) & 0xff00ffu) * a;
        x = (x + ((x >> 8) & 0xff00ffu) + 0x800080u);
        x &= 0xff00ff00u;
        return x | t;
    }
    void span(int y, int x0, int x1, uint32_t c) {
        if (y < 0 || y >= h) return;
        if (x0 < 0) x0 = 0;
        if (x1 > w) x1 = w;
        for (int x = x0; x < x1; x++) px[y * w + x] = c;
    }
    void fill(double x, double y, double ww, double hh, uint32_t c, bool over) {
        int x1 = qtpath::q_round(x), x2 = qtpath::q_round(x + ww), y1 = qtpath::q_round(y), y2 = qtpath::q_round(y + hh);
        if (x2 < x1) { const int t = x1; x1 = x2; x2 = t; }
        if (y2 < y1) { const int t = y1; y1 = y2; y2 = t; }
        if (x1 < 0) x1 = 0;
        if (y1 < 0) y1 = 0;
        if (x2 > w) x2 = w;
        if (y2 > h) y2 = h;
        for (int yy = y1; yy < y2; yy++)
            for (int xx = x1; xx < x2; xx++) {
                uint32_t *d = &px[yy * w + xx];
                *d = over ? c + byte_mul(*d, 255u - (c >> 24)) : c;
            }
    }
    // sink of qtpath::fill_crossings / stroke_ellipse
    void cross(int y, int x) {
        if (cnt[y]++ == 0) {
            xa[y] = x;
            return;
        }
        span(y, xa[y] < x ? xa[y] : x, xa[y] < x ? x : xa[y], cur_brush);
    }
    uint32_t cur_pen;
    void pixel(int x, int y) { px[y * w + x] = cur_pen; }
    void ellipse(double x, double y, double ww, double hh, uint32_t pen_px, uint32_t brush_px) {
        if (ww < 0) { x += ww; ww = -ww; }  // QPainter::drawEllipse: rect.normalized()
        if (hh < 0) { y += hh; hh = -hh; }
        if ((ww > hh ? ww : hh) < 32767 && ww > 0 && hh > 0 && qtpath::is_integer_rect(x, y, ww, hh)) {
            midpoint_ellipse(*this, (int)x, (int)y, (int)ww, (int)hh, pen_px, brush_px);
            return;
        }
        if (ww == 0 && hh == 0) return;  // qt_curves_for_arc: rect.isNull()
        for (int i = 0; i < h; i++) cnt[i] = 0;
        cur_brush = brush_px;
        cur_pen = pen_px;
        int top, bot;
        qtpath::fill_crossings(*this, x, y, ww, hh, w, h, top, bot);
        qtpath::stroke_ellipse(*this, x, y, ww, hh, w, h);
    }
};

}  // namespace assetgen
}  // namespace pgamd
