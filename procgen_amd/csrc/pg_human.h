// pg_human.h -- the render_human frame: 512 x 512 x 3, antialiased, smooth pixmap transform (reference src/vecgame.cpp:270-282,
// 363-376: VecGame::observe renders every env a second time at RENDER_RES with Game::render_to_buf(..., antialias = true),
// src/game.cpp:77-91).  Not a hot path -- the reference draws these frames one env after the other on the Python thread, for
// interactive.py and gym's render_mode="rgb_array" -- but a second rasterizer: with QPainter::Antialiasing and
// SmoothPixmapTransform every drawImage / fillRect of game_draw (BAG:979-1012) leaves Qt's fast paths and goes through
//   coverage : QRasterizer::rasterizeLine(a, b, h / w), antialiased (16.16 fixed point, 8-bit coverage per pixel),
//   sampling : fetchTransformedBilinearARGB32PM over the spans of a row that touch (one run per row),
//   blend    : comp_func_SourceOver (comp_func_Source for an RGB32 source) with const_alpha = (coverage * intOpacity) >> 8.
// All three are restated here and pinned against PyQt5 5.9.7 by tests/tools/qt_smooth_aa_probe.py; whole frames are compared
// with the compiled reference (tests/golden/render_human.npz).
//
// Execution: one wavefront per (env, band of HUMAN_BAND rows); the band lives in LDS as 0xffRRGGBB words.  Drawables are
// walked in the painter's order in wave-uniform code (rect -> line -> clip -> coverage set-up in doubles, as Qt does it);
// the lanes take the pixels of a row's run, 64 at a time.  Game policies are the ones the 64 x 64 renderer uses
// (image_for_type, theme_for_grid_obj, adjusted_image_rect, tile_aspect_ratio, draw_overlay, ...).
#pragma once
#include "pg_render.h"
#include "pg_aapath.h"

namespace pgamd {

constexpr int HUMAN_BAND = 32;                        // rows per workgroup (HUMAN_RES, HUMAN_BYTES: pg_defs.h)
constexpr int HUMAN_BANDS = HUMAN_RES / HUMAN_BAND;

struct HumanLds {
    uint32_t fb[HUMAN_BAND * HUMAN_RES];
    int area[HUMAN_RES + 2], cover[HUMAN_RES + 2];  // the gray raster's cells of one pixel row (pg_aapath.h; jumper's compass)
};

#if defined(PGAMD_WAVE_EMU) && defined(PG_HUMAN_TRACE)
inline int *pg_human_trace_xy() {
    static int xy[2] = {-1, -1};
    static bool init = false;
    if (!init) {
        init = true;
        if (const char *e = getenv("PG_HUMAN_TRACE")) sscanf(e, "%d,%d", &xy[0], &xy[1]);
    }
    return xy;
}
#endif
namespace human {

PG_DEV int f16(double v) { return (int)(v * 65536.0); }                                     // FloatToQ16Dot16
PG_DEV int mul16(int a, int b) { return (int)(((long long)a * (long long)b) >> 16); }       // Q16Dot16Multiply
PG_DEV uint32_t bmul(uint32_t x, uint32_t a) {                                           // qdrawhelper_p.h BYTE_MUL
    uint32_t t = (x & 0xff00ffu) * a;
    t = (t + ((t >> 8) & 0xff00ffu) + 0x800080u) >> 8;
    t &= 0xff00ffu;
    x = ((x >> 8) & 0xff00ffu) * a;
    x = (x + ((x >> 8) & 0xff00ffu) + 0x800080u);
    x &= 0xff00ff00u;
    return x | t;
}
PG_DEV uint32_t interpolate_pixel_255(uint32_t x, uint32_t a, uint32_t y, uint32_t b) {     // qdrawhelper_p.h INTERPOLATE_PIXEL_255
    uint32_t t = (x & 0xff00ffu) * a + (y & 0xff00ffu) * b;
    t = (t + ((t >> 8) & 0xff00ffu) + 0x800080u) >> 8;
    t &= 0xff00ffu;
    x = ((x >> 8) & 0xff00ffu) * a + ((y >> 8) & 0xff00ffu) * b;
    x = (x + ((x >> 8) & 0xff00ffu) + 0x800080u);
    x &= 0xff00ff00u;
    return x | t;
}
// per channel (a * (256 - d) + b * d) >> 8 -- INTERPOLATE_PIXEL_256 and the intermediate-buffer form of the scale-up fetch
PG_DEV uint32_t lerp256(uint32_t a, uint32_t b, uint32_t dd) {
    const uint32_t id = 256u - dd;
    const uint32_t rb = (((a & 0xff00ffu) * id + (b & 0xff00ffu) * dd) >> 8) & 0xff00ffu;
    const uint32_t ag = (((a >> 8) & 0xff00ffu) * id + ((b >> 8) & 0xff00ffu) * dd) & 0xff00ff00u;
    return rb | ag;
}
PG_DEV uint32_t interp8(uint32_t tl, uint32_t tr, uint32_t bl, uint32_t br, uint32_t dx, uint32_t dy) {  // 8-bit distances: rows first, then columns
    return lerp256(lerp256(tl, bl, dy), lerp256(tr, br, dy), dx);
}
PG_DEV uint32_t interp16(uint32_t tl, uint32_t tr, uint32_t bl, uint32_t br, uint32_t dx, uint32_t dy) {  // interpolate_4_pixels_16 (distances 0..16)
    const uint32_t dxy = dx * dy;
    const uint32_t w0 = 256u - 16u * dx - 16u * dy + dxy, w1 = 16u * dx - dxy, w2 = 16u * dy - dxy, w3 = dxy;
    const uint32_t rb = ((tl & 0xff00ffu) * w0 + (tr & 0xff00ffu) * w1 + (bl & 0xff00ffu) * w2 + (br & 0xff00ffu) * w3) >> 8;
    const uint32_t ag = ((tl >> 8) & 0xff00ffu) * w0 + ((tr >> 8) & 0xff00ffu) * w1 + ((bl >> 8) & 0xff00ffu) * w2 + ((br >> 8) & 0xff00ffu) * w3;
    return (rb & 0xff00ffu) | (ag & 0xff00ff00u);
}

// What QRasterizer::rasterizeLine leaves of an axis-aligned line: columns [iLeft, iRight] with the 16.16 coverage factors of the
// left / inner / right columns, rows [iTop, iBottom] of the line's 16.16 extent [yPa, yPb].
struct AxisCoverage {
    int iLeft, iRight, covLeft, covRight;  // covLeft: coverage factor (x 255, 16.16) of column iLeft; inner columns 255 << 16
    int iTop, iBottom, yPa, yPb;
    bool single;                           // iLeft == iRight (Qt adds the two partial widths there)
};

}  // namespace human

// a game whose overlay needs other draws at 512 pixels than at 64 declares HAS_HUMAN_OVERLAY and draw_overlay_human(r)
template <class Game, class = void>
struct GameHasHumanOverlay {
    static constexpr bool value = false;
};
template <class Game>
struct GameHasHumanOverlay<Game, decltype((void)Game::HAS_HUMAN_OVERLAY)> {
    static constexpr bool value = Game::HAS_HUMAN_OVERLAY;
};

template <class Game>
struct HumanRenderer {
    static constexpr int FRAME_W = HUMAN_RES, FRAME_H = HUMAN_RES;
    const DevCtx &d;
    GameOptions opt;  // this env's options (pg_defs.h env_options)
    const int env;
    HumanLds *lds;
    uint32_t *fb;
    EnvHdr G;
    const uint32_t *ge;
    int ecap;
    const typename Game::cell_t *gg;
    int row0, row1;  // this workgroup's band

    PG_DEV HumanRenderer(const DevCtx &d_, int env_, HumanLds *lds_, int band) : d(d_), env(env_), lds(lds_), fb(lds_->fb) {
        ge = d.ents + ent_table_base(env, d.ent_cap);
        ecap = d.ent_cap;
        gg = reinterpret_cast<const typename Game::cell_t *>(d.grid + (size_t)env * d.grid_bytes);
        row0 = band * HUMAN_BAND;
        row1 = row0 + HUMAN_BAND;
    }

    // the accessors the game policies use (same names as Renderer / Env)
    PG_DEV float ef(int field, int i) const { return __builtin_bit_cast(float, ge[(uint32_t)(field * ecap + i)]); }
    PG_DEV uint32_t meta(int i) const { return ge[(uint32_t)(EF_META * ecap + i)]; }
    PG_DEV float ex(int i) const { return ef(EF_X, i); }
    PG_DEV float ey(int i) const { return ef(EF_Y, i); }
    PG_DEV float evx(int i) const { return ef(EF_VX, i); }
    PG_DEV float evy(int i) const { return ef(EF_VY, i); }
    PG_DEV float erx(int i) const { return ef(EF_RX, i); }
    PG_DEV float ery(int i) const { return ef(EF_RY, i); }
    PG_DEV int etype(int i) const { return meta_type(meta(i)); }
    PG_DEV void fail(int code, int line = __builtin_LINE()) {
        if (G.error == 0) G.error = pg_error_word(code, line);
    }
    PG_DEV int get_obj(int x, int y) const {  // BAG:180-185
        if (!(0 <= y && y < G.main_height && 0 <= x && x < G.main_width)) return G.out_of_bounds_object;
        return (int)gg[y * G.main_width + x];
    }
    PG_DEV RectD get_screen_rect(float x, float y, float dx, float dy, float render_eps) const {  // BAG:799-801
        RectD r;
        r.x = (double)((x - render_eps) * G.unit - G.x_off);
        r.y = (double)((G.view_dim - y - render_eps) * G.unit + G.y_off);
        r.w = (double)((dx + 2 * render_eps) * G.unit);
        r.h = (double)((dy + 2 * render_eps) * G.unit);
        return r;
    }
    PG_DEV RectD get_abs_rect(float x, float y, float dx, float dy) const {  // BAG:803-805
        RectD r;
        r.x = (double)(x * G.unit);
        r.y = (double)(y * G.unit);
        r.w = (double)(dx * G.unit);
        r.h = (double)(dy * G.unit);
        return r;
    }

    // ---- QRasterizer::rasterizeLine, antialiased, clip = the frame ------------------------------------------------------------
    // the common head: clips the line to the (widened) frame and rescales the relative width; false: nothing to draw
    PG_DEV static bool clip_line(double ax, double ay, double bx, double by, double &width, double &pax, double &pay, double &pbx, double &pby, bool square_cap = false) {
        const int cw = HUMAN_RES, ch = HUMAN_RES;
        if ((ax == bx && ay == by) || width == 0) return false;
        pax = ax; pay = ay; pbx = bx; pby = by;
        if (square_cap) {  // the line grows by half its width at either end
            const double ddx = pbx - pax, ddy = pby - pay;
            pax -= (0.5 * width) * ddx; pay -= (0.5 * width) * ddy;
            pbx += (0.5 * width) * ddx; pby += (0.5 * width) * ddy;
        }
        const double offx = pg_fabs(by - ay) * width * 0.5, offy = pg_fabs(bx - ax) * width * 0.5;
        const double cl = 0 - offx, ct = 0 - offy, cr = (cw - 1) + 1 + offx, cb = (ch - 1) + 1 + offy;
        const bool a_in = cl <= pax && pax <= cr && ct <= pay && pay <= cb, b_in = cl <= pbx && pbx <= cr && ct <= pby && pby <= cb;
        if (!a_in || !b_in) {
            double t1 = 0, t2 = 1;
            for (int i = 0; i < 2; i++) {
                const double o = i ? pay : pax, dd = i ? pby - pay : pbx - pax, low = i ? ct : cl, high = i ? cb : cr;
                if (dd == 0) {
                    if (o <= low || o >= high) return false;
                    continue;
                }
                const double d_inv = 1 / dd;
                double t_low = (low - o) * d_inv, t_high = (high - o) * d_inv;
                if (t_low > t_high) { const double t = t_low; t_low = t_high; t_high = t; }
                if (t1 < t_low) t1 = t_low;
                if (t2 > t_high) t2 = t_high;
                if (t1 >= t2) return false;
            }
            const double npax = pax + (pbx - pax) * t1, npay = pay + (pby - pay) * t1, npbx = pax + (pbx - pax) * t2, npby = pay + (pby - pay) * t2;
            pax = npax; pay = npay; pbx = npbx; pby = npby;
        }
        const double d0x = ax - bx, d0y = ay - by, w0 = d0x * d0x + d0y * d0y;
        const double dx = pax - pbx, dy = pay - pby, w = dx * dx + dy * dy;
        if (w == 0) return false;
        width *= pg_sqrt(w0 / w);
        return true;
    }
    PG_DEV static bool q26_equal(double p, double q) { return (int)((p - q) * 64) == 0; }  // q26Dot6Compare
    // 0: nothing; 1: axis-aligned (cov filled in); 2: a general line (pa, pb, width returned for the trapezoid walker)
    PG_DEV static int rasterize_line(double ax, double ay, double bx, double by, double width, human::AxisCoverage &cov, double (&gl)[5], bool square_cap = false) {
        using namespace human;
        const int cw = HUMAN_RES, ch = HUMAN_RES;
        double pax, pay, pbx, pby;
        if (!clip_line(ax, ay, bx, by, width, pax, pay, pbx, pby, square_cap)) return 0;
        if (q26_equal(pay, pby)) {
            if (q26_equal(pax, pbx)) return 0;
            const double x = (pax + pbx) * 0.5, dx = pg_fabs(pbx - pax) * 0.5, y = pay, dy = width * dx;
            pax = x; pay = y - dy;
            pbx = x; pby = y + dy;
            width = 1 / width;
        }
        if (!q26_equal(pax, pbx)) {
            gl[0] = pax; gl[1] = pay; gl[2] = pbx; gl[3] = pby; gl[4] = width;
            return 2;
        }
        if (pay > pby) { const double t = pay; pay = pby; pby = t; }
        const double dy = pby - pay, half = 0.5 * width * dy;
        double left = pax - half, right = pax + half;
        left = left < 0 ? 0 : (left > cw ? cw : left);
        right = right < 0 ? 0 : (right > cw ? cw : right);
        pay = pay < 0 ? 0 : (pay > ch ? ch : pay);
        pby = pby < 0 ? 0 : (pby > ch ? ch : pby);
        if (q26_equal(left, right) || q26_equal(pay, pby)) return 0;
        cov.iLeft = (int)left;
        cov.iRight = (int)right;
        const int leftWidth = ((cov.iLeft + 1) << 16) - f16(left), rightWidth = f16(right) - (cov.iRight << 16);
        cov.single = cov.iLeft == cov.iRight;
        cov.covLeft = (cov.single ? leftWidth + rightWidth : leftWidth) * 255;
        cov.covRight = rightWidth * 255;
        cov.iTop = (int)pay;
        cov.iBottom = (int)pby;
        cov.yPa = f16(pay);
        cov.yPb = f16(pby);
        return 1;
    }

    // ---- source of a drawImage: bilinear texture fetch -------------------------------------------------------------------------
    struct Texture {
        const uint32_t *base;  // first pixel: the atlas image, or (use_generated_assets) the env's background canvas
        int w, h;
        bool argb32;     // use_generated_assets: a 64 x 64 Format_ARGB32 sprite (NOT premultiplied): Qt's generic bilinear fetch converts every
                         // texel (qPremultiply) and uses one formula for all pixels of a run (no SSE2 head / groups / tail)
        bool mirrored;   // a reflected sprite (BAG:121 keeps a mirrored copy): column x of it is column w - 1 - x of the atlas image
        bool rgb32;      // no alpha channel (backgrounds): at opacity 1 getOperator turns SourceOver into Source (QSpanData::initTexture: hasAlpha = image.hasAlphaChannel() || intOpacity != 256)
        double m11, m12, m21, m22, dx, dy;  // QSpanData::setupMatrix: inverse of translate(1/65536) * painter matrix * rect mapping
    };
    PG_DEV static uint32_t q_premultiply(uint32_t p) {  // qrgb.h qPremultiply
        const uint32_t a = p >> 24;
        if (a == 255) return p;
        if (a == 0) return 0;
        uint32_t t = (p & 0xff00ffu) * a;
        t = ((t + ((t >> 8) & 0xff00ffu) + 0x800080u) >> 8) & 0xff00ffu;
        uint32_t g = ((p >> 8) & 0xffu) * a;
        g = (g + ((g >> 8) & 0xffu) + 0x80u) & 0xff00u;
        return g | t | (a << 24);
    }
    PG_DEV uint32_t texel(const Texture &t, int x, int y) const {
        const uint32_t p = t.base[(uint32_t)(y * t.w + (t.mirrored ? t.w - 1 - x : x))];
        return t.argb32 ? q_premultiply(p) : p;
    }
    // the texture of a draw: generated sprites are ARGB32, the generated background is the env's own 500 x 500 RGB32 canvas
    PG_DEV void bind_texture(Texture &tx, const ImgDesc im, bool mirrored, bool rgb32) const {
        const bool gen = d.gen_bg != nullptr;
        tx.base = (gen && rgb32) ? d.gen_bg + (size_t)env * GEN_BG_WORDS : d.pixels + im.off;
        tx.w = im.w;
        tx.h = im.h;
        tx.argb32 = gen && !rgb32;
        tx.mirrored = mirrored;
        tx.rgb32 = rgb32;
    }
    // fetchTransformedBilinearARGB32PM<BlendTransformedBilinear> for pixel b of the run that starts at column x0 of row y.  The
    // spans of a row that touch are fetched as one run, and the pixel's place in the run selects the code path of Qt's SSE2 build:
    //   * 8-bit distances, rows blended first, throughout: scaling up on x (0 < fdx <= 1) or zooming more than 8 times;
    //   * otherwise a scalar head (8-bit) while a coordinate pair is clamped at the image border, then groups of FOUR pixels with
    //     rounded 4-bit distances for as long as the whole group stays inside the image, then a scalar tail (8-bit).
    // [lo, hi]: the pixels b of [0, length) whose coordinate f0 + b * fd lies in [0, (n - 1) << 16), i.e. whose pair is not clamped
    PG_DEV static void unclamped_range(long long f0, long long fd, int n, int length, long long &lo, long long &hi) {
        const long long lim = (long long)(n - 1) << 16;
        lo = 0;
        hi = (long long)length - 1;
        if (fd == 0) {
            if (!(f0 >= 0 && f0 < lim)) hi = -1;
            return;
        }
        // floor / ceil division by a positive divisor
        auto fdiv = [](long long a, long long q) { return a >= 0 ? a / q : -((-a + q - 1) / q); };
        auto cdiv = [](long long a, long long q) { return a >= 0 ? (a + q - 1) / q : -((-a) / q); };
        if (fd > 0) {
            const long long l2 = cdiv(-f0, fd), h2 = fdiv(lim - 1 - f0, fd);  // f0 + b fd >= 0 ; f0 + b fd <= lim - 1
            if (l2 > lo) lo = l2;
            if (h2 < hi) hi = h2;
        } else {
            const long long q = -fd;
            const long long l2 = cdiv(f0 - (lim - 1), q), h2 = fdiv(f0, q);  // f0 - b q <= lim - 1 ; f0 - b q >= 0
            if (l2 > lo) lo = l2;
            if (h2 < hi) hi = h2;
        }
    }
    PG_DEV uint32_t fetch(const Texture &t, int y, int x0, int length, int b) const {
        using namespace human;
        const double cx = x0 + 0.5, cy = y + 0.5;
        // QSpanData::setupMatrix: the 16.16 walk below is only taken while the inverse matrix is small ("fast_matrix"); a sprite scaled
        // down ~20 times a few hundred pixels from the origin has an inverse translation beyond 1e4 and gets floating-point source
        // coordinates per pixel with 8-bit distances instead (jumper's memory mode without center_agent: an 11-pixel agent)
        if (!(t.m11 * t.m11 + t.m21 * t.m21 < 1e4 && t.m12 * t.m12 + t.m22 * t.m22 < 1e4 && pg_fabs(t.dx) < 1e4 && pg_fabs(t.dy) < 1e4)) {
            double sfx = t.m21 * cy + t.m11 * cx + t.dx, sfy = t.m22 * cy + t.m12 * cx + t.dy;
            for (int i = 0; i < b; i++) {  // (accumulated as Qt does: fx += fdx per pixel)
                sfx += t.m11;
                sfy += t.m12;
            }
            const double px = sfx - 0.5, py = sfy - 0.5;
            int x1 = (int)px - (px < 0 ? 1 : 0), y1 = (int)py - (py < 0 ? 1 : 0), x2, y2;
            const int distx = (int)((px - x1) * 256), disty = (int)((py - y1) * 256);
            if (x1 < 0) x1 = x2 = 0;
            else if (x1 >= t.w - 1) x1 = x2 = t.w - 1;
            else x2 = x1 + 1;
            if (y1 < 0) y1 = y2 = 0;
            else if (y1 >= t.h - 1) y1 = y2 = t.h - 1;
            else y2 = y1 + 1;
            return interp8(texel(t, x1, y1), texel(t, x2, y1), texel(t, x1, y2), texel(t, x2, y2), (uint32_t)distx, (uint32_t)disty);
        }
        const int fdx = (int)(t.m11 * 65536.0), fdy = (int)(t.m12 * 65536.0);
        const int fx0 = (int)((t.m21 * cy + t.m11 * cx + t.dx) * 65536.0) - 32768;
        const int fy0 = (int)((t.m22 * cy + t.m12 * cx + t.dy) * 65536.0) - 32768;
        const int fx = fx0 + b * fdx, fy = fy0 + b * fdy;
        bool eight;
        if (fdy == 0) eight = (fdx > 0 && fdx <= 65536) || (fdx < 0 && fdx > -(65536 / 8)) || pg_fabs(t.m22) < (1. / 8.);
        else eight = pg_fabs(t.m11) < (1. / 8.) || pg_fabs(t.m22) < (1. / 8.);
        bool four_bit = !eight && t.argb32;  // (the generic fetch of a non-premultiplied source: one formula for the whole run)
        if (!eight && !t.argb32) {
            long long xl, xh, yl = 0, yh = (long long)length - 1;
            unclamped_range(fx0, fdx, t.w, length, xl, xh);
            if (fdy != 0) unclamped_range(fy0, fdy, t.h, length, yl, yh);
            const long long jl = xl > yl ? xl : yl, jh = xh < yh ? xh : yh;
            if (jl <= jh) {
                const long long head = jl;
                const long long fxh = (long long)fx0 + head * fdx, fyh = (long long)fy0 + head * fdy;
                long long bounded = length;
                if (fdx > 0) { const long long v = head + (((long long)(t.w - 1) << 16) - fxh) / fdx; if (v < bounded) bounded = v; }
                else if (fdx < 0) { const long long v = head + (0 - fxh) / fdx; if (v < bounded) bounded = v; }
                if (fdy > 0) { const long long v = head + (((long long)(t.h - 1) << 16) - fyh) / fdy; if (v < bounded) bounded = v; }
                else if (fdy < 0) { const long long v = head + (0 - fyh) / fdy; if (v < bounded) bounded = v; }
                bounded -= 3;
                if (bounded > head) {
                    const long long groups = (bounded - head + 3) / 4;
                    four_bit = b >= head && b < head + groups * 4;
                }
            }
        }
        int x1 = fx >> 16, x2, y1 = fy >> 16, y2;
        if (x1 < 0) x1 = x2 = 0;
        else if (x1 >= t.w - 1) x1 = x2 = t.w - 1;
        else x2 = x1 + 1;
        if (y1 < 0) y1 = y2 = 0;
        else if (y1 >= t.h - 1) y1 = y2 = t.h - 1;
        else y2 = y1 + 1;
        const uint32_t tl = texel(t, x1, y1), tr = texel(t, x2, y1), bl = texel(t, x1, y2), br = texel(t, x2, y2);
        if (four_bit) return interp16(tl, tr, bl, br, (uint32_t)(((fx & 0xffff) + 0x800) >> 12), (uint32_t)(((fy & 0xffff) + 0x800) >> 12));
        return interp8(tl, tr, bl, br, (uint32_t)((fx & 0xffff) >> 8), (uint32_t)((fy & 0xffff) >> 8));
    }

    // blend one fetched (or solid) source pixel into the band: comp_func_SourceOver / comp_func_Source with const_alpha
    PG_DEV static uint32_t blend(uint32_t dst, uint32_t s, uint32_t ca, bool source_mode) {
        using namespace human;
        if (source_mode) return ca == 255 ? s : interpolate_pixel_255(s, ca, dst, 255 - ca);
        if (ca != 255) s = bmul(s, ca);
        return s + bmul(dst, 255 - (s >> 24));
    }

    // The spans of an axis-aligned line, row by row: [left column][inner columns][right column], spans of coverage 0 dropped,
    // the ones that touch merged into a run for the fetch.  src(y, x0, length, b) yields the source pixel.
    template <class Src>
    PG_DEV void fill_axis(const human::AxisCoverage &c, int io, bool source_mode, Src src) {
        using namespace human;
        int ya = c.iTop, yb = c.iBottom;
        if (ya < row0) ya = row0;
        if (yb > row1 - 1) yb = row1 - 1;
        if (yb > HUMAN_RES - 1) yb = HUMAN_RES - 1;
        // QSpanBuffer hands the spans of a draw to the blend function in batches of 256 (SPAN_BUFFER_SIZE), and touching spans only
        // form a run within a batch: the index of a row's first span in the draw decides where a run is cut.  Rows between the
        // first and the last all have the same spans.
        auto spans_of = [&c](int rowHeight, int (&sx)[3], int (&sl)[3], int (&sc)[3]) {
            using namespace human;
            int n = 0;
            {
                int cvL = mul16(rowHeight, c.covLeft) >> 16;
                if (c.single) {
                    // (Qt adds the two partial widths of a rect inside one pixel column: up to 2 x 255, kept in QT_FT_Span's unsigned char)
                    if (cvL) { sx[n] = c.iLeft; sl[n] = 1; sc[n] = cvL & 0xff; n++; }
                } else {
                    const int cvM = mul16(rowHeight, 255 << 16) >> 16, cvR = mul16(rowHeight, c.covRight) >> 16;
                    const bool left_full = c.covLeft == 65536 * 255;  // leftWidth == 1: Qt folds the inner columns into the first span
                    if (left_full) {
                        if (cvL) { sx[n] = c.iLeft; sl[n] = c.iRight - c.iLeft; sc[n] = cvL; n++; }
                    } else {
                        if (cvL) { sx[n] = c.iLeft; sl[n] = 1; sc[n] = cvL; n++; }
                        if (c.iRight - c.iLeft > 1 && cvM) { sx[n] = c.iLeft + 1; sl[n] = c.iRight - c.iLeft - 1; sc[n] = cvM; n++; }
                    }
                    if (c.covRight != 0 && cvR) { sx[n] = c.iRight; sl[n] = 1; sc[n] = cvR; n++; }
                }
            }
            return n;
        };
        auto row_height = [&c](int y) {
            const int yFP = y << 16;
            const int hi = yFP + 65536 < c.yPb ? yFP + 65536 : c.yPb, lo = yFP > c.yPa ? yFP : c.yPa;
            return hi - lo;
        };
        int n_top, n_mid;
        {
            int tx[3], tl[3], tc[3];
            n_top = spans_of(row_height(c.iTop), tx, tl, tc);
            n_mid = spans_of(65536, tx, tl, tc);
        }
        for (int y = ya; y <= yb; y++) {
            const int rowHeight = row_height(y);
            int sx[3], sl[3], sc[3];
            const int n = spans_of(rowHeight, sx, sl, sc);
            const int first_span = y == c.iTop ? 0 : n_top + (y - c.iTop - 1) * n_mid;  // index of this row's first span in the draw
            int i = 0;
            while (i < n) {
                int j = i + 1, right = sx[i] + sl[i];
                while (j < n && sx[j] == right && ((first_span + j) >> 8) == ((first_span + i) >> 8)) { right += sl[j]; j++; }
                const int x0 = sx[i], length = right - x0;
                // per-pixel coverage of the run: the spans i..j-1 (at most three)
                const int e0 = sx[i] + sl[i], c0 = sc[i];
                const int e1 = (i + 1 < j) ? sx[i + 1] + sl[i + 1] : e0, c1 = (i + 1 < j) ? sc[i + 1] : c0;
                const int c2 = (i + 2 < j) ? sc[i + 2] : c1;
                uint32_t *rowp = fb + (y - row0) * HUMAN_RES;
                for (int base = 0; base < length; base += 64) {
                    PG_FOR_LANES(l) {
                        const int b = base + l;
                        if (b < length) {
                            const int x = x0 + b;
                            const int cv = x < e0 ? c0 : (x < e1 ? c1 : c2);
                            const uint32_t ca = (uint32_t)((cv * io) >> 8);
                            const uint32_t s = src(y, x0, length, b);
#if defined(PGAMD_WAVE_EMU) && defined(PG_HUMAN_TRACE)
                            if (x == pg_human_trace_xy()[0] && y == pg_human_trace_xy()[1])
                                fprintf(stderr, "trace (%d,%d): run x0 %d len %d b %d cov %d io %d src %08x dst %08x -> %08x mode %d\n", x, y, x0, length, b, cv, io, s, rowp[x], blend(rowp[x], s, ca, source_mode), (int)source_mode);
#endif
                            rowp[x] = blend(rowp[x], s, ca, source_mode);
                        }
                    }
                }
                i = j;
            }
        }
        PG_SYNC();
    }

    // ---- a general (turned) line: QRasterizer::rasterizeLine's antialiased trapezoid walker ------------------------------------------
    PG_DEV static double safe_div(double x, double y) { return y == 0 ? (x > 0 ? 1e9 : -1e9) : x / y; }                       // qSafeDivide
    PG_DEV static int sf16(double x) { return human::f16(x < -32768.0 ? -32768.0 : (x > 32767.0 ? 32767.0 : x)); }             // qSafeFloatToQ16Dot16
    PG_DEV static int fmul16(int a, int b) { return (int)((uint32_t)a * (uint32_t)b) >> 16; }                                  // Q16Dot16FastMultiply (32-bit product)
    PG_DEV static int intersect_pixel_fp(int x, int top, int bottom, int leftIntersectX, int rightIntersectX, int slope, int invSlope) {
        using namespace human;
        const int leftX = x << 16, rightX = (x << 16) + 65536;
        int leftIntersectY, rightIntersectY;
        if (slope > 0) {
            leftIntersectY = top + mul16(leftX - leftIntersectX, invSlope);
            rightIntersectY = leftIntersectY + invSlope;
        } else {
            leftIntersectY = top + mul16(leftX - rightIntersectX, invSlope);
            rightIntersectY = leftIntersectY + invSlope;
        }
        if (leftIntersectX >= leftX && rightIntersectX <= rightX) return mul16(bottom - top, leftIntersectX - leftX + ((rightIntersectX - leftIntersectX) >> 1));
        if (leftIntersectX >= rightX) return bottom - top;
        if (leftIntersectX >= leftX) {
            if (slope > 0) return (bottom - top) - fmul16((rightX - leftIntersectX) >> 1, rightIntersectY - top);
            return (bottom - top) - fmul16((rightX - leftIntersectX) >> 1, bottom - rightIntersectY);
        }
        if (rightIntersectX <= leftX) return 0;
        if (rightIntersectX <= rightX) {
            if (slope > 0) return fmul16((rightIntersectX - leftX) >> 1, bottom - leftIntersectY);
            return fmul16((rightIntersectX - leftX) >> 1, leftIntersectY - top);
        }
        if (slope > 0) return (bottom - rightIntersectY) + ((rightIntersectY - leftIntersectY) >> 1);
        return (rightIntersectY - top) + ((leftIntersectY - rightIntersectY) >> 1);
    }
    // (Qt 5.9 calls intersectPixelFP without checking that the part of the row it is asked about is non-empty: a side corner just above
    // a clipped first row contributes a negative exclusion there.  Later Qt versions guard these calls; 5.9.7 is what the reference ships.)
    struct GenRow {  // one row of the walker: everything the coverage of a pixel depends on
        int yFP, iLeftFP, iRightFP;
        int rowTop, rowBottom, rowBottomLeft, rowBottomRight, rowTopLeft, rowTopRight, rowHeight;
        int topLeftAf, topLeftBf, topRightAf, topRightBf, bottomLeftAf, bottomLeftBf, bottomRightAf, bottomRightBf;
        int tlFP, trFP, blFP, brFP, itlFP, itrFP, iblFP, ibrFP;
        int leftMin, leftMax, rightMin, rightMax;
        PG_DEV int right_excluded(int x) const {
            int e = 0;
            if (yFP <= iRightFP) e += (rowBottomRight - rowTop) - intersect_pixel_fp(x, rowTop, rowBottomRight, topRightAf, bottomRightAf, trFP, itrFP);
            if (yFP >= iRightFP) e += (rowBottom - rowTopRight) - intersect_pixel_fp(x, rowTopRight, rowBottom, bottomRightBf, topRightBf, brFP, ibrFP);
            return e;
        }
        // x in [leftMin, rightMax].  Normally 0..255; on a degenerate first row (a side corner just above the clip, see above) the unguarded
        // exclusions can push it outside: Qt tests the int for zero and then keeps its low byte (QT_FT_Span::coverage is an unsigned char)
        PG_DEV int coverage(int x) const {
            int cov16;
            if (x <= leftMax) {
                int excluded = 0;
                if (yFP <= iLeftFP) excluded += intersect_pixel_fp(x, rowTop, rowBottomLeft, bottomLeftAf, topLeftAf, tlFP, itlFP);
                if (yFP >= iLeftFP) excluded += intersect_pixel_fp(x, rowTopLeft, rowBottom, topLeftBf, bottomLeftBf, blFP, iblFP);
                if (x >= rightMin) excluded += right_excluded(x);
                cov16 = rowHeight - excluded;
            } else if (x < rightMin) {
                cov16 = rowHeight;
            } else {
                cov16 = rowHeight - right_excluded(x);
            }
            return (255 * cov16) >> 16;
        }
    };
    // gl = {pa.x, pa.y, pb.x, pb.y, width} as rasterize_line left them
    template <class Src>
    PG_DEV void fill_general(const double (&gl)[5], int io, bool source_mode, Src src) {
        using namespace human;
        double pax = gl[0], pay = gl[1], pbx = gl[2], pby = gl[3];
        const double width = gl[4];
        if (pay > pby) {
            double t = pax; pax = pbx; pbx = t;
            t = pay; pay = pby; pby = t;
        }
        const double dlx = (pbx - pax) * (0.5 * width), dly = (pby - pay) * (0.5 * width);
        const double perpx = dly, perpy = -dlx;
        double tx, ty, lx, ly, rx, ry, bx, by;
        if (pax < pbx) {
            tx = pax + perpx; ty = pay + perpy; lx = pax - perpx; ly = pay - perpy; rx = pbx + perpx; ry = pby + perpy; bx = pbx - perpx; by = pby - perpy;
        } else {
            tx = pax - perpx; ty = pay - perpy; lx = pbx - perpx; ly = pby - perpy; rx = pax + perpx; ry = pay + perpy; bx = pbx + perpx; by = pby + perpy;
        }
        // snapTo26Dot6Grid: the four corners, DOWN to multiples of 1/64
        tx = pg_floor(tx * 64) * (1 / 64.); ty = pg_floor(ty * 64) * (1 / 64.);
        lx = pg_floor(lx * 64) * (1 / 64.); ly = pg_floor(ly * 64) * (1 / 64.);
        rx = pg_floor(rx * 64) * (1 / 64.); ry = pg_floor(ry * 64) * (1 / 64.);
        bx = pg_floor(bx * 64) * (1 / 64.); by = pg_floor(by * 64) * (1 / 64.);
        const int clipB = HUMAN_RES - 1, clipR = HUMAN_RES - 1;
        const double topBound = ty < 0 ? 0 : (ty > clipB ? clipB : ty), bottomBound = by < 0 ? 0 : (by > clipB ? clipB : by);
        const double tlS = safe_div(lx - tx, ly - ty), blS = safe_div(bx - lx, by - ly), trS = safe_div(rx - tx, ry - ty), brS = safe_div(bx - rx, by - ry);
        GenRow g;
        g.tlFP = sf16(tlS); g.trFP = sf16(trS); g.blFP = sf16(blS); g.brFP = sf16(brS);
        g.itlFP = sf16(safe_div(1, tlS)); g.itrFP = sf16(safe_div(1, trS)); g.iblFP = sf16(safe_div(1, blS)); g.ibrFP = sf16(safe_div(1, brS));
        const int iTopFP = (int)topBound << 16, iBottomFP = (int)bottomBound << 16;
        g.iLeftFP = (int)ly << 16;
        g.iRightFP = (int)ry << 16;
        int leftAf = sf16(tx + ((int)topBound - ty) * tlS), rightAf = sf16(tx + ((int)topBound - ty) * trS), leftBf = 0, rightBf = 0;
        if (g.iLeftFP < iTopFP) leftBf = sf16(lx + ((int)topBound - ly) * blS);
        if (g.iRightFP < iTopFP) rightBf = sf16(rx + ((int)topBound - ry) * brS);
        const int yTopFP = sf16(ty), yLeftFP = sf16(ly), yRightFP = sf16(ry), yBottomFP = sf16(by);
        int rowTop = iTopFP > yTopFP ? iTopFP : yTopFP;
        int topLeftAf = leftAf + mul16(g.tlFP, rowTop - iTopFP), topRightAf = rightAf + mul16(g.trFP, rowTop - iTopFP);
        int span_idx = 0;  // spans of this draw so far
        for (int yFP = iTopFP; yFP <= iBottomFP; yFP += 65536) {
            const int y = yFP >> 16;
            if (y >= row1) break;
            g.yFP = yFP;
            g.rowTop = rowTop;
            g.rowBottomLeft = yFP + 65536 < yLeftFP ? yFP + 65536 : yLeftFP;
            g.rowBottomRight = yFP + 65536 < yRightFP ? yFP + 65536 : yRightFP;
            g.rowTopLeft = yFP > yLeftFP ? yFP : yLeftFP;
            g.rowTopRight = yFP > yRightFP ? yFP : yRightFP;
            g.rowBottom = yFP + 65536 < yBottomFP ? yFP + 65536 : yBottomFP;
            g.topLeftAf = topLeftAf;
            g.topRightAf = topRightAf;
            if (yFP == g.iLeftFP) {
                leftBf = sf16(lx + (y - ly) * blS);
                g.topLeftBf = leftBf + mul16(g.blFP, g.rowTopLeft - yFP);
                g.bottomLeftAf = leftAf + mul16(g.tlFP, g.rowBottomLeft - yFP);
            } else {
                g.topLeftBf = leftBf;
                g.bottomLeftAf = leftAf + g.tlFP;
            }
            if (yFP == g.iRightFP) {
                rightBf = sf16(rx + (y - ry) * brS);
                g.topRightBf = rightBf + mul16(g.brFP, g.rowTopRight - yFP);
                g.bottomRightAf = rightAf + mul16(g.trFP, g.rowBottomRight - yFP);
            } else {
                g.topRightBf = rightBf;
                g.bottomRightAf = rightAf + g.trFP;
            }
            if (yFP == iBottomFP) {
                g.bottomLeftBf = leftBf + mul16(g.blFP, g.rowBottom - yFP);
                g.bottomRightBf = rightBf + mul16(g.brFP, g.rowBottom - yFP);
            } else {
                g.bottomLeftBf = leftBf + g.blFP;
                g.bottomRightBf = rightBf + g.brFP;
            }
            auto bound = [clipR](int v) { return v < 0 ? 0 : (v > clipR ? clipR : v); };
            auto mx = [](int a, int b) { return a > b ? a : b; };
            auto mn = [](int a, int b) { return a < b ? a : b; };
            if (yFP < g.iLeftFP) { g.leftMin = g.bottomLeftAf >> 16; g.leftMax = g.topLeftAf >> 16; }
            else if (yFP == g.iLeftFP) { g.leftMin = mx(g.bottomLeftAf, g.topLeftBf) >> 16; g.leftMax = mx(g.topLeftAf, g.bottomLeftBf) >> 16; }
            else { g.leftMin = g.topLeftBf >> 16; g.leftMax = g.bottomLeftBf >> 16; }
            g.leftMin = bound(g.leftMin);
            g.leftMax = bound(g.leftMax);
            if (yFP < g.iRightFP) { g.rightMin = g.topRightAf >> 16; g.rightMax = g.bottomRightAf >> 16; }
            else if (yFP == g.iRightFP) { g.rightMin = mn(g.topRightAf, g.bottomRightBf) >> 16; g.rightMax = mx(g.bottomRightAf, g.topRightBf) >> 16; }
            else { g.rightMin = g.bottomRightBf >> 16; g.rightMax = g.topRightBf >> 16; }
            g.rightMin = bound(g.rightMin);
            g.rightMax = bound(g.rightMax);
            if (g.leftMax > g.rightMax) g.leftMax = g.rightMax;
            if (g.rightMin < g.leftMin) g.rightMin = g.leftMin;
            g.rowHeight = g.rowBottom - g.rowTop;
            {
                // the spans of this row in Qt's order: single pixels leftMin..leftMax, the full span up to rightMin, single pixels up to
                // rightMax; spans of coverage 0 are dropped, the ones that touch form a run for the fetch -- within one batch of 256
                // spans of the draw (QSpanBuffer's flush), which is why rows above this band are walked too: they count
                const bool draw = y >= row0;
                const int last = g.rightMax > g.leftMax ? g.rightMax : g.leftMax;
                const int mid_cov = (255 * g.rowHeight) >> 16;
                int run = -1;  // first column of the open run
                uint32_t *rowp = fb + (draw ? (y - row0) : 0) * HUMAN_RES;
                auto flush = [&](int xe) {  // the run [run, xe)
                    if (!draw || run < 0 || xe <= run) return;
                    const int x0 = run, length = xe - run;
                    for (int base = 0; base < length; base += 64) {
                        PG_FOR_LANES(l) {
                            const int b = base + l;
                            if (b < length) {
                                const int x = x0 + b;
                                const int cv = g.coverage(x) & 0xff;
                                const uint32_t ca = (uint32_t)((cv * io) >> 8);
                                rowp[x] = blend(rowp[x], src(y, x0, length, b), ca, source_mode);
                            }
                        }
                    }
                };
                auto span = [&](int x, bool nonzero) {  // the next span of the draw starts at column x
                    if (nonzero) {
                        if ((span_idx & 255) == 0 && run >= 0) {  // a new batch: the open run ends here
                            flush(x);
                            run = -1;
                        }
                        span_idx++;
                        if (run < 0) run = x;
                    } else {
                        flush(x);
                        run = -1;
                    }
                };
                int x = g.leftMin;
                while (x <= last) {
                    if (x > g.leftMax && x < g.rightMin) {  // the full span
                        span(x, mid_cov != 0);
                        x = g.rightMin;
                        continue;
                    }
                    span(x, g.coverage(x) != 0);
                    x++;
                }
                flush(last + 1);
            }
            leftAf += g.tlFP;
            leftBf += g.blFP;
            rightAf += g.trFP;
            rightBf += g.brFP;
            topLeftAf = leftAf;
            topRightAf = rightAf;
            rowTop = yFP + 65536;
        }
        PG_SYNC();
    }

    // ---- antialiased path draws (pg_aapath.h): jumper's compass ------------------------------------------------------------------------------
    // QPainter::drawEllipse with a brush, NoPen part: Qt's gray raster over the flattened outline, row by row of this band
    PG_DEV void aa_fill_ellipse(const RectD &r, uint32_t premul) {
        if (qtpath::fill_culled(r.x, r.y, r.w, r.h, HUMAN_RES, HUMAN_RES)) return;
        qtpath::Arc arc;
        qtpath::arc_points(r.x, r.y, r.w, r.h, arc);
        aapath::RowExtent ext{0, 0, 0};
        qtpath::flatten(arc, ext);
        int ya = ext.min_y >> 6, yb = (ext.max_y + 63) >> 6;
        if (ya < row0) ya = row0;
        if (yb > row1 - 1) yb = row1 - 1;
        int *area = lds->area, *cover = lds->cover;
        for (int y = ya; y <= yb; y++) {
            for (int base = 0; base < HUMAN_RES + 2; base += 64) {
                PG_FOR_LANES(l) {
                    if (base + l < HUMAN_RES + 2) {
                        area[base + l] = 0;
                        cover[base + l] = 0;
                    }
                }
            }
            PG_SYNC();
            aapath::RowCells cells;
            cells.init(area, cover, HUMAN_RES, y);
            qtpath::flatten(arc, cells);  // (the outline is closed: its last point is its first)
            cells.record();
            PG_SYNC();
            // gray_sweep of this row
            uint32_t *rowp = fb + (y - row0) * HUMAN_RES;
            auto hline = [&](int x0, int ar, int count) {
                const int cv = aapath::RowCells::coverage(ar);
                if (!cv) return;
                int xa = x0 < 0 ? 0 : x0, xb = x0 + count > HUMAN_RES ? HUMAN_RES : x0 + count;
                for (int base = xa; base < xb; base += 64) {
                    PG_FOR_LANES(l) {
                        const int x = base + l;
                        if (x < xb) rowp[x] = blend(rowp[x], premul, (uint32_t)cv, false);
                    }
                }
            };
            int cov = 0, x = 0;
            for (int cx = cells.min_x; cx <= cells.max_x; cx++) {
                const int ca = area[cx + 1], cc = cover[cx + 1];
                if (!(ca | cc)) continue;
                if (cx > x && cov != 0) hline(x, cov * (aapath::ONE_PIXEL * 2), cx - x);
                cov += cc;
                const int ar = cov * (aapath::ONE_PIXEL * 2) - ca;
                if (ar != 0 && cx >= 0) hline(cx, ar, 1);
                x = cx + 1;
            }
            if (cov != 0) hline(x, cov * (aapath::ONE_PIXEL * 2), HUMAN_RES - x);
            PG_SYNC();
        }
    }
    // ... its pen of width 1: QCosmeticStroker's antialiased line walker over the subdivided cubics
    struct PenSink {
        HumanRenderer &r;
        uint32_t color;
        PG_DEV void pixel(int x, int y, int coverage) {
            if (x < 0 || x > HUMAN_RES - 1 || y < r.row0 || y > r.row1 - 1) return;
            uint32_t *p = r.fb + (y - r.row0) * HUMAN_RES + x;
            const uint32_t c = human::bmul(color, (uint32_t)coverage);  // drawPixelARGB32
            PG_FOR_LANES(l) {
                if (l == 0) *p = c + human::bmul(*p, 255 - (c >> 24));
            }
            PG_SYNC();
        }
    };
    PG_DEV void aa_stroke_ellipse(const RectD &r, uint32_t color) {
        PenSink sink{*this, color};
        aapath::CosmeticAA<PenSink> st{sink, -1.0, HUMAN_RES + 1.0, -1.0, HUMAN_RES + 1.0};
        st.ellipse(r.x, r.y, r.w, r.h);
    }
    // QPainter::drawLine(int, int, int, int) with a solid pen wider than 1 (QRasterPaintEngine::stroke, LinesHint): rasterizeLine with QPen's
    // default square cap; a zero-length line is a pen-wide dash
    PG_DEV void aa_wide_line(int x1, int y1, int x2, int y2, double pen_width, uint32_t color) {
        human::AxisCoverage c;
        double gl[5];
        int kind;
        if (x1 == x2 && y1 == y2) {
            kind = rasterize_line(x1 - pen_width * 0.5, (double)y1, x1 + pen_width * 0.5, (double)y1, 1.0, c, gl);
        } else {
            const double dx = (double)(x2 - x1), dy = (double)(y2 - y1);
            kind = rasterize_line((double)x1, (double)y1, (double)x2, (double)y2, pen_width / pg_sqrt(dx * dx + dy * dy), c, gl, true);
        }
        auto src = [color](int, int, int, int) { return color; };
        if (kind == 1) fill_axis(c, 256, false, src);
        else if (kind == 2) fill_general(gl, 256, false, src);
    }

    // QPainter::fillRect(QRectF, QColor) under Antialiasing (untransformed painter): the rect's mid line, width h / w
    PG_DEV void exec_fill(const RectD &r, uint32_t color) {
        human::AxisCoverage c;
        double gl[5];
        const double l = r.x, t = r.y, rr = r.x + r.w, bb = r.y + r.h;
        // QRectF::normalized is what fillRect hands on; the rects drawn here have positive extents
        const int kind = rasterize_line((l + l) * 0.5, (t + bb) * 0.5, (rr + rr) * 0.5, (t + bb) * 0.5, r.h / r.w, c, gl);
        if (kind != 1) return;
        // comp_func_solid_SourceOver / _Source: colour * coverage + dst * (255 - coverage), the two products rounded on their own
        fill_axis(c, 256, false, [color](int, int, int, int) { return color; });
    }

    // QPainter::drawImage(QRectF, QImage), painter untransformed
    PG_DEV void draw_image_rect(const ImgDesc im, bool mirrored, bool rgb32, const RectD &r, float opacity) {
        if (!(r.w > 0) || !(r.h > 0)) return;  // QRectF::isEmpty
        if (r.w == (double)im.w && r.h == (double)im.h) {
            // not stretched (translate only): QRasterPaintEngine::drawImage takes fillRect_normalized over the ROUNDED rect with the
            // untransformed image filler -- neither antialiasing nor filtering (e.g. a 512 x 512 background over the 512 frame)
            double o = (double)opacity;
            if (o < 0) o = 0;
            if (o > 1) o = 1;
            const int io = (int)(o * 256);
            const uint32_t ca = (uint32_t)((255 * io) >> 8);
            const bool source_mode = rgb32 && io == 256;
            Texture btx;
            bind_texture(btx, im, mirrored, rgb32);
            const int x1 = q_round(r.x), y1 = q_round(r.y), x2 = q_round(r.x + r.w), y2 = q_round(r.y + r.h);
            int ya = y1 < row0 ? row0 : y1, yb = y2 > row1 ? row1 : y2;
            const int xa = x1 < 0 ? 0 : x1, xb = x2 > HUMAN_RES ? HUMAN_RES : x2;
            for (int y = ya; y < yb; y++) {
                const int sy = y - y1;
                if (sy < 0 || sy >= (int)im.h) continue;
                uint32_t *rowp = fb + (y - row0) * HUMAN_RES;
                for (int base = xa; base < xb; base += 64) {
                    PG_FOR_LANES(l) {
                        const int x = base + l, sx = x - x1;
                        if (x < xb && sx >= 0 && sx < (int)im.w) {
                            const uint32_t s = texel(btx, sx, sy);
                            rowp[x] = blend(rowp[x], s, ca, source_mode);
                        }
                    }
                }
            }
            PG_SYNC();
            return;
        }
        human::AxisCoverage c;
        double gl[5];
        const double l = r.x, t = r.y, rr = r.x + r.w, bb = r.y + r.h;
        const int kind = rasterize_line((l + l) * 0.5, (t + bb) * 0.5, (rr + rr) * 0.5, (t + bb) * 0.5, r.h / r.w, c, gl);
        if (kind != 1) return;
#if defined(PGAMD_WAVE_EMU) && defined(PG_HUMAN_TRACE)
        if (!rgb32 && pg_human_trace_xy()[1] >= row0 && pg_human_trace_xy()[1] < row1 && pg_human_trace_xy()[0] >= c.iLeft && pg_human_trace_xy()[0] <= c.iRight && pg_human_trace_xy()[1] >= c.iTop && pg_human_trace_xy()[1] <= c.iBottom)
            fprintf(stderr, "sprite %dx%d mirrored %d rect %.17g %.17g %.17g %.17g opacity %g\n", (int)im.w, (int)im.h, (int)mirrored, r.x, r.y, r.w, r.h, (double)opacity);
        if (rgb32 && row0 == 0) fprintf(stderr, "bg rect %.17g %.17g %.17g %.17g -> iLeft %d iRight %d covL %d covR %d top %d bottom %d yPa %d yPb %d unit %.9g\n", r.x, r.y, r.w, r.h, c.iLeft, c.iRight, c.covLeft, c.covRight, c.iTop, c.iBottom, c.yPa, c.yPb, (double)G.unit);
#endif
        if (c.iBottom < row0 || c.iTop >= row1) return;
        Texture tx;
        bind_texture(tx, im, mirrored, rgb32);
        {
            const double scx = r.w / (double)im.w, scy = r.h / (double)im.h, dd = 1.0 / 65536;
            const double m11 = 1.0 * scx, m22 = 1.0 * scy, m31 = dd * scx + r.x, m32 = dd * scy + r.y;
            tx.m11 = 1.0 / m11;
            tx.m22 = 1.0 / m22;
            tx.m12 = 0;
            tx.m21 = 0;
            tx.dx = -m31 * tx.m11;
            tx.dy = -m32 * tx.m22;
        }
        double o = (double)opacity;
        if (o < 0) o = 0;
        if (o > 1) o = 1;
        const int io = (int)(o * 256);
        fill_axis(c, io, rgb32 && io == 256, [this, &tx](int y, int x0, int length, int b) { return fetch(tx, y, x0, length, b); });
    }

    // tile_image BAG:840-869
    PG_DEV void tile_image(const ImgDesc im, bool mirrored, bool rgb32, const RectD &rect, float tile_ratio, float opacity) {
        if (tile_ratio != 0) {
            if (tile_ratio < 0) {
                tile_ratio = -1 * tile_ratio;
                int num_tiles = (int)(rect.h / (rect.w * (double)tile_ratio));
                if (num_tiles < 1) num_tiles = 1;
                const float tile_height = (float)(rect.h / num_tiles), tile_width = (float)rect.w;
                for (int i = 0; i < num_tiles; i++) {
                    const RectD tr = {rect.x, rect.y + (double)(tile_height * i), (double)tile_width, (double)tile_height};
                    if (tr.y + tr.h < row0 - 2 || tr.y > row1 + 2) continue;
                    draw_image_rect(im, mirrored, rgb32, tr, opacity);
                }
            } else {
                int num_tiles = (int)(rect.w / (rect.h * (double)tile_ratio));
                if (num_tiles < 1) num_tiles = 1;
                const float tile_width = (float)(rect.w / num_tiles), tile_height = (float)rect.h;
                if (rect.y + rect.h < row0 - 2 || rect.y > row1 + 2) return;
                for (int i = 0; i < num_tiles; i++) {
                    const RectD tr = {rect.x + (double)(tile_width * i), rect.y, (double)tile_width, (double)tile_height};
                    if (tr.x + tr.w < -2 || tr.x > HUMAN_RES + 2) continue;
                    draw_image_rect(im, mirrored, rgb32, tr, opacity);
                }
            }
        } else {
            draw_image_rect(im, mirrored, rgb32, rect, opacity);
        }
    }

    // draw_image BAG:877-913
    PG_DEV void draw_image(RectD base_rect, float rotation, bool is_reflected, int base_type, int theme, float alpha, float tile_ratio) {
        const int img_type = Game::image_for_type(*this, base_type);
        if (img_type < 0) return;
        if (opt.use_monochrome_assets || img_type >= USE_ASSET_THRESHOLD) {  // draw_grid_obj BAG:915-919, color_for_type BAG:455-481
            if constexpr (GameHasGridFills<Game>::value) {  // a game's own draw_grid_obj (chaser.cpp:112-119)
                if (Game::is_grid_fill(*this, img_type)) {
                    RectD out;
                    uint32_t color;
                    Game::grid_fill(*this, img_type, base_rect, out, color);
                    exec_fill(out, color);
                    return;
                }
            }
            if (img_type == SPACE) return;
            if (!opt.use_monochrome_assets || img_type >= 64) {
                fail(PGE_UNSUPPORTED_DRAW);
                return;
            }
            int th = theme;
            if (opt.restrict_themes && !Game::should_preserve_type_themes(img_type)) th = 0;
            const int k = 4, kcubed = 64, chunk = 64;
            int new_type = (29 * (img_type + 1)) % kcubed;
            new_type = (new_type + 19 * th) % kcubed;
            const uint32_t cr = (uint32_t)(chunk * (new_type / (k * k) + 1) - 1), cg = (uint32_t)(chunk * ((new_type / k) % k + 1) - 1), cb = (uint32_t)(chunk * (new_type % k + 1) - 1);
            exec_fill(base_rect, 0xff000000u | (cr << 16) | (cg << 8) | cb);
            return;
        }
        const RectD rect = Game::adjusted_image_rect(img_type, base_rect);
        int mt = theme;
        if (opt.restrict_themes && !Game::should_preserve_type_themes(img_type)) mt = 0;  // BAG:450-453
        const int img = (mt >= 0 && mt < MAX_IMAGE_THEMES) ? (int)d.assets->type_theme_img[img_type][mt] : -1;
        if (img < 0) {
            fail(PGE_THEME);
            return;
        }
        const ImgDesc im = d.assets->img[img];
        if (rotation == 0) {
            tile_image(im, is_reflected, false, rect, tile_ratio, alpha);
        } else {
            draw_image_rotated(im, is_reflected, rect, rotation, alpha);
        }
    }
    // BAG:902-906: p.translate(cx, cy); p.rotate(rotation * 180 / PI); p.drawImage(QRectF(-w/2, -h/2, w, h), img)
    PG_DEV void draw_image_rotated(const ImgDesc im, bool mirrored, const RectD &adjusted, float rotation, float opacity) {
        const double w = adjusted.w, h = adjusted.h;
        if (!(w > 0) || !(h > 0)) return;
        const double cx = adjusted.x + adjusted.w / 2, cy = adjusted.y + adjusted.h / 2;
        const double a = (double)(rotation * 180 / PG_PI);
        double sina = 0, cosa = 0;  // QTransform::rotate
        if (a == 0) cosa = 1;
        else if (a == 90. || a == -270.) sina = 1.;
        else if (a == 270. || a == -90.) sina = -1.;
        else if (a == 180.) cosa = -1.;
        else {
            const double b = 0.017453292519943295769 * a;
            sina = pg_sin_d(b);
            cosa = pg_cos_d(b);
        }
        const double m11 = cosa, m12 = sina, m21 = -sina, m22 = cosa, mdx = cx, mdy = cy;
        // QTransform::type(): 3 rotate, 2 scale, 1 translate, 0 none
        int typ;
        if (!q_fuzzy_is_null(m12) || !q_fuzzy_is_null(m21)) typ = 3;
        else if (!q_fuzzy_is_null(m11 - 1) || !q_fuzzy_is_null(m22 - 1)) typ = 2;
        else if (!q_fuzzy_is_null(mdx) || !q_fuzzy_is_null(mdy)) typ = 1;
        else typ = 0;
        auto map = [&](double x, double y, double &ox, double &oy) {  // QTransform::map
            if (typ == 0) { ox = x; oy = y; }
            else if (typ == 1) { ox = x + mdx; oy = y + mdy; }
            else if (typ == 2) { ox = m11 * x + mdx; oy = m22 * y + mdy; }
            else { ox = m11 * x + m21 * y + mdx; oy = m12 * x + m22 * y + mdy; }
        };
        const double rx = -w / 2, ry = -h / 2;
        // the texture matrix: copy = matrix; copy.translate(r.x, r.y); copy.scale(r.w / sw, r.h / sh); inverse of translate(1/65536) * copy
        Texture tx;
        bind_texture(tx, im, mirrored, false);
        {
            double c11 = m11, c12 = m12, c21 = m21, c22 = m22, cdx = mdx, cdy = mdy;
            int ctyp;
            if (typ == 0) { cdx = rx; cdy = ry; ctyp = 1; }
            else if (typ == 1) { cdx += rx; cdy += ry; ctyp = 1; }
            else if (typ == 2) { cdx += rx * c11; cdy += ry * c22; ctyp = 2; }
            else { cdx += rx * c11 + ry * c21; cdy += ry * c22 + rx * c12; ctyp = 3; }
            const double scx = w / (double)im.w, scy = h / (double)im.h;
            if (ctyp == 3) { c12 *= scx; c21 *= scy; }
            c11 *= scx;
            c22 *= scy;
            if (ctyp < 2) ctyp = 2;
            const double dd = 1.0 / 65536;
            if (ctyp == 2) {
                const double p11 = 1.0 * c11, p22 = 1.0 * c22, p31 = dd * c11 + cdx, p32 = dd * c22 + cdy;
                tx.m11 = 1. / p11;
                tx.m22 = 1. / p22;
                tx.m12 = tx.m21 = 0;
                tx.dx = -p31 * tx.m11;
                tx.dy = -p32 * tx.m22;
            } else {
                const double p11 = 1.0 * c11 + 0.0 * c21, p12 = 1.0 * c12 + 0.0 * c22, p21 = 0.0 * c11 + 1.0 * c21, p22 = 0.0 * c12 + 1.0 * c22;
                const double p31 = dd * c11 + dd * c21 + cdx, p32 = dd * c12 + dd * c22 + cdy;
                const double dtr = p11 * p22 - p12 * p21, dinv = 1.0 / dtr;
                tx.m11 = p22 * dinv;
                tx.m12 = -p12 * dinv;
                tx.m21 = -p21 * dinv;
                tx.m22 = p11 * dinv;
                tx.dx = (p21 * p32 - p22 * p31) * dinv;
                tx.dy = (p12 * p31 - p11 * p32) * dinv;
            }
        }
        double o = (double)opacity;
        if (o < 0) o = 0;
        if (o > 1) o = 1;
        const int io = (int)(o * 256);
        double ax, ay, bx, by;
        const double l = rx, t = ry, rr = rx + w, bb = ry + h;
        map((l + l) * 0.5, (t + bb) * 0.5, ax, ay);
        map((rr + rr) * 0.5, (t + bb) * 0.5, bx, by);
        human::AxisCoverage c;
        double gl[5];
        const int kind = rasterize_line(ax, ay, bx, by, h / w, c, gl);
        auto srcf = [this, &tx](int y, int x0, int length, int b) { return fetch(tx, y, x0, length, b); };
        if (kind == 1) {
            if (c.iBottom < row0 || c.iTop >= row1) return;
            fill_axis(c, io, false, srcf);
        } else if (kind == 2) {
            fill_general(gl, io, false, srcf);
        }
    }
    // jumper's compass (drawEllipse / drawLine under Antialiasing: Qt's gray raster and its antialiased cosmetic stroker) is not
    // restated: libenv_make refuses render_human for jumper, these only keep the policy's draw_overlay compiling
    PG_DEV void exec_ellipse(int, int, int, int, bool, uint32_t, uint32_t) { fail(PGE_UNSUPPORTED_DRAW); }
    PG_DEV void exec_row_masks(const uint32_t *, int, int, uint32_t, uint32_t) { fail(PGE_UNSUPPORTED_DRAW); }
    PG_DEV void exec_line(int, int, int, int, uint32_t) { fail(PGE_UNSUPPORTED_DRAW); }

    PG_DEV void draw_entities(int render_z) {  // BAG:1052-1066
        const int n = G.n_ents;
        for (int i = 0; i < n; i++) {
            const uint32_t mm = meta(i);
            if (meta_render_z(mm) != render_z) continue;
            if (!Game::should_draw_entity(*this, i)) continue;
            const float x = ex(i), y = ey(i), rx = erx(i), ry = ery(i);
            RectD r1;  // get_object_rect BAG:811-817
            if (mm & MF_ABS_COORDS) r1 = get_abs_rect(G.view_dim * (x - rx), G.view_dim * (y + ry), 2 * G.view_dim * rx, 2 * G.view_dim * ry);
            else r1 = get_screen_rect(x - rx, y + ry, 2 * rx, 2 * ry, 0);
            // nothing of it in this band?  (the sprite's rect may be adjusted or turned: its circumcircle, generously)
            {
                const double rad = (pg_fabs(r1.w) + pg_fabs(r1.h)) * 2 + 4, cy = r1.y + r1.h / 2;
                if (cy + rad < row0 || cy - rad > row1) continue;
            }
            const float tile_ratio = Game::tile_aspect_ratio(*this, i);
            draw_image(r1, ef(EF_ROTATION, i), (mm & MF_REFLECTED) != 0, meta_image_type(mm), meta_image_theme(mm), ef(EF_ALPHA, i), tile_ratio);
        }
    }

    // game_draw BAG:1009-1012 at rect_height = 512
    PG_DEV void render_band() {
        {
            const EnvHdr *h = d.hdr + env;
#define PG_X(type, name) G.name = h->name;
            PG_HDR_FIELDS(PG_X)
#undef PG_X
        }
        opt = env_options(d.opt, G.opt_bits, G.opt_debug_mode);
        // prepare_for_drawing(512) BAG:819-838: centre, visibility and view_dim are what the step kernel left for the 64-pixel frame
        {
            const float raw_unit = 64 / G.visibility;
            G.unit = (float)((double)raw_unit * ((double)(float)HUMAN_RES / 64.0));
            G.view_dim = (float)(64.0 / (double)raw_unit);
            G.x_off = G.unit * (G.center_x - G.view_dim / 2);
            G.y_off = G.unit * (G.center_y - G.view_dim / 2);
        }
        for (int base = 0; base < HUMAN_BAND * HUMAN_RES; base += 64) {
            PG_FOR_LANES(l) { fb[base + l] = 0xff000000u; }  // p.fillRect(rect, QColor(0, 0, 0))
        }
        PG_SYNC();
        // draw_background BAG:979-1007
        if (opt.use_backgrounds) {
            const int bgi = (int)d.assets->bg_img[G.background_index];
            const ImgDesc bim = d.assets->img[bgi];
            if constexpr (GameCustomBackground<Game>::value) {
                RectD rects[4];
                const int nr = Game::background_rects(*this, rects);
                _Pragma("unroll") for (int k = 0; k < 4; k++)  // constant indices: a running index would put the array in scratch memory
                    if (k < nr && rects[k].w > 0) draw_image_rect(bim, false, true, rects[k], 1.0f);
            } else {
                const RectD main_rect = get_screen_rect(0, (float)G.main_height, (float)G.main_width, (float)G.main_height, 0);
                if (G.bg_tile_ratio < 0) {
                    tile_image(bim, false, true, main_rect, G.bg_tile_ratio, 1.0f);
                } else {
                    const float bgw = (float)bim.w, bgh = (float)bim.h;
                    const float bg_ar = bgw / bgh;
                    const float world_ar = (float)(G.main_width * 1.0 / G.main_height);
                    const float extra_w = bg_ar - world_ar;
                    const float offset_x = G.bg_pct_x * extra_w;
                    draw_image_rect(bim, false, true, adjust_rect(main_rect, (double)(-offset_x), 0, (double)(bg_ar / world_ar), 1), 1.0f);
                }
            }
        }
        // draw_foreground BAG:921-970
        draw_entities(-1);
        if constexpr (GameDrawsGrid<Game>::value) {
            int low_x, high_x, low_y, high_y;
            if (Game::center_agent(opt)) {
                const float margin = (float)(G.visibility / 2.0 + 1);
                low_x = (int)(G.center_x - margin);
                high_x = (int)(G.center_x + margin);
                low_y = (int)(G.center_y - margin);
                high_y = (int)(G.center_y + margin);
            } else {
                low_x = 0;
                high_x = G.main_width - 1;
                low_y = 0;
                high_y = G.main_height - 1;
            }
            for (int x = low_x; x <= high_x; x++) {
                for (int y = low_y; y <= high_y; y++) {
                    const int type = get_obj(x, y);
                    if (type == INVALID_OBJ) continue;
                    const RectD r2 = get_screen_rect((float)x, (float)(y + 1), 1, 1, RENDER_EPS);
                    if (r2.y + 2 * r2.h < row0 - 2 || r2.y - r2.h > row1 + 2) continue;  // (adjusted_image_rect stretches a cell image by less than its own height)
                    const int theme = Game::theme_for_grid_obj(*this, type);
                    draw_image(r2, 0, false, type, theme, 1.0f, 0.0f);
                }
            }
        }
        draw_entities(0);
        draw_entities(1);
        if (G.has_useful_vel_info && opt.paint_vel_info) {  // BAG:960-969, to_shade reference src/qt-utils.h:21-28
            const float infodim = (float)(HUMAN_RES * .2);
            const int ag = G.agent;
            int s1 = (int)((float)(.5 * (double)evx(ag) / (double)G.maxspeed + .5) * 255);
            int s2 = (int)((float)(.5 * (double)evy(ag) / (double)G.max_jump + .5) * 255);
            s1 = s1 < 0 ? 0 : (s1 > 255 ? 255 : s1);
            s2 = s2 < 0 ? 0 : (s2 > 255 ? 255 : s2);
            const RectD d2 = {0, 0, (double)infodim, (double)infodim}, d3 = {(double)infodim, 0, (double)infodim, (double)infodim};
            exec_fill(d2, 0xff000000u | ((uint32_t)s1 << 16) | ((uint32_t)s1 << 8) | (uint32_t)s1);
            exec_fill(d3, 0xff000000u | ((uint32_t)s2 << 16) | ((uint32_t)s2 << 8) | (uint32_t)s2);
        }
        if constexpr (GameHasHumanOverlay<Game>::value) Game::draw_overlay_human(*this);  // (draws that need the antialiased path route)
        else if constexpr (GameHasOverlay<Game>::value) Game::draw_overlay(*this);
        PG_SYNC();
        store_band();
        if (G.error) {
#if defined(PGAMD_WAVE_EMU)
            pg_report_error(d, env, G.error, ERR_KIND_HUMAN, 0, 0);
#else
            if (PG_LANE_ID() == 0) pg_report_error(d, env, G.error, ERR_KIND_HUMAN, 0, 0);
#endif
        }
        // get_state serializes the camera scalars of the LAST frame drawn, which is this one (reference src/vecgame.cpp:363-376)
        if (row0 == 0) {
            EnvHdr *h = d.hdr + env;
            PG_FOR_LANES(l) {
                if (l == 0) {
                    h->unit = G.unit;
                    h->x_off = G.x_off;
                    h->y_off = G.y_off;
                }
            }
        }
    }

    // bgr32_to_rgb888 (reference src/game.cpp:8-23) into the env's info "rgb" frame
    PG_DEV void store_band() {
        uint32_t *out = reinterpret_cast<uint32_t *>(d.human + (size_t)env * HUMAN_BYTES + (size_t)row0 * HUMAN_RES * 3);
        for (int base = 0; base < HUMAN_BAND * HUMAN_RES; base += 256) {
            PG_FOR_LANES(l) {
                const uint32_t *p = &fb[base + 4 * l];
                const uint32_t p0 = p[0], p1 = p[1], p2 = p[2], p3 = p[3];
                uint32_t *o = out + (base / 4) * 3 + 3 * l;
                o[0] = pg_perm(p1, p0, 0x06000102u);
                o[1] = pg_perm(p2, p1, 0x05060001u);
                o[2] = pg_perm(p3, p2, 0x04050600u);
            }
        }
    }
};

}  // namespace pgamd
