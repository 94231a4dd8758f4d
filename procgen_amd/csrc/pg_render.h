// pg_render.h -- the 64x64x3 frame of one environment, produced by FOUR wavefronts (one 256-thread workgroup):
// wave b rasterizes rows [16b, 16b+16) into its own 4 KB LDS band and writes that band of the RGB888 observation
// with fully coalesced stores.  The waves never synchronize with each other.
//
// Replaces Game::render_to_buf + BasicAbstractGame::game_draw (reference src/game.cpp:77-91, BAG:799-1012,
// "BAG" = reference src/basic-abstract-game.cpp) and the Qt 5.9 raster engine calls they make: non-antialiased
// QPainter::drawImage(QRectF, QImage) = qt_scale_image_32 (16.16 fixed-point nearest sampling, premultiplied
// SourceOver, const alpha = int(opacity*256)) and fillRect.  The painter's order of the reference is kept:
// background, z=-1 entities, grid cells (x-major), z=0 entities, z=1 entities.
//
// Two phases per batch of <= 64 drawables: (1) lane-parallel set-up (each lane turns one drawable into a draw
// command -- the fp64 rect / fixed-point step arithmetic runs 64-wide), (2) the commands that touch this
// wave's band are executed in order; lanes cover pixels.  Texel fetches of up to 8 small commands (or 8 pixels
// per lane of a large one) are issued before the first blend so that several HBM/L2 round trips overlap.
#pragma once
#include "pg_env.h"

namespace pgamd {

constexpr int BAND_ROWS = 16;
constexpr int NUM_BANDS = RES_H / BAND_ROWS;

struct DrawCmd {  // uniform (scalar) view of one command
    int tx1, ty1, w, h;
    uint32_t basex, srcy0, ix, iy;
    uint32_t img;  // image index | mirrored<<12 | const_alpha(0..256)<<16
};

template <class Game>
struct Renderer {
    const DevCtx &d;
    const int env;
    const int band;
    uint32_t *fb;  // this wave's band: BAND_ROWS x 64 words of 0xffRRGGBB
    EnvHdr G;
    const uint32_t *ge;  // this env's entity table in HBM
    int ecap;
    const typename Game::cell_t *gg;
    int row0, row1;  // band rows [row0, row1)

    PG_DEV Renderer(const DevCtx &d_, int env_, int band_, uint32_t *fb_) : d(d_), env(env_), band(band_), fb(fb_) {
        ge = d.ents + (size_t)env * EF_COUNT * d.ent_cap;
        ecap = d.ent_cap;
        gg = reinterpret_cast<const typename Game::cell_t *>(d.grid + (size_t)env * d.grid_bytes);
        row0 = band * BAND_ROWS;
        row1 = row0 + BAND_ROWS;
    }

    // entity accessors with the names the game policies use (HBM reads; the table was written by the step kernel)
    PG_DEV float ef(int field, int i) const { return __builtin_bit_cast(float, ge[field * ecap + i]); }
    PG_DEV uint32_t meta(int i) const { return ge[EF_META * ecap + i]; }
    PG_DEV float ex(int i) const { return ef(EF_X, i); }
    PG_DEV float ey(int i) const { return ef(EF_Y, i); }
    PG_DEV float evx(int i) const { return ef(EF_VX, i); }
    PG_DEV float evy(int i) const { return ef(EF_VY, i); }
    PG_DEV float erx(int i) const { return ef(EF_RX, i); }
    PG_DEV float ery(int i) const { return ef(EF_RY, i); }
    PG_DEV int etype(int i) const { return meta_type(meta(i)); }
    PG_DEV void fail(int code) {
        if (G.error == 0) G.error = code;
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

    // ---- command set-up (lane-local) ----------------------------------------------------------------------
    // geom word: tx1 | ty1<<7 | w<<14 | h<<21, 0 = nothing to draw in this band
    PG_DEV void cmd_image(int img_index, bool mirrored, RectD tr, float opacity, uint32_t &geom, uint32_t &basex_o, uint32_t &srcy_o,
                          uint32_t &ix_o, uint32_t &iy_o, uint32_t &img_o) const {
        geom = 0;
        basex_o = srcy_o = ix_o = iy_o = img_o = 0;
        const ImgDesc im = d.assets->img[img_index];
        const double sx = tr.w / (double)im.w;
        const double sy = tr.h / (double)im.h;
        const int ix = (int)(65536 / sx);
        const int iy = (int)(65536 / sy);
        int tx1 = q_round(tr.x), tx2 = q_round(tr.x + tr.w), ty1 = q_round(tr.y), ty2 = q_round(tr.y + tr.h);
        if (tx1 < 0) tx1 = 0;
        if (ty1 < 0) ty1 = 0;
        if (tx2 > RES_W) tx2 = RES_W;
        if (ty2 > RES_H) ty2 = RES_H;
        int w = tx2 - tx1, h = ty2 - ty1;
        if (w <= 0 || h <= 0) return;
        // Qt 5.9: qCeil(...) - 1 (pinned with tests/tools/qt_drawimage_probe.py)
        const uint32_t basex = (uint32_t)((int)pg_ceil((tx1 + 0.5 - tr.x) * ix) - 1);
        const uint32_t srcy = (uint32_t)((int)pg_ceil((ty1 + 0.5 - tr.y) * iy) - 1);
        const int yend = (int)((srcy + (uint32_t)iy * (uint32_t)(h - 1)) >> 16);
        if (yend < 0 || yend >= (int)im.h) --h;
        const int xend = (int)((basex + (uint32_t)ix * (uint32_t)(w - 1)) >> 16);
        if (xend < 0 || xend >= (int)im.w) --w;
        if (w <= 0 || h <= 0) return;
        if (ty1 >= row1 || ty1 + h <= row0) return;  // does not touch this wave's band
        double o = (double)opacity;  // QPainter::setOpacity clamps to [0,1]; intOpacity = int(opacity * 256)
        if (o < 0) o = 0;
        if (o > 1) o = 1;
        const int io = (int)(o * 256);
        geom = (uint32_t)tx1 | ((uint32_t)ty1 << 7) | ((uint32_t)w << 14) | ((uint32_t)h << 21);
        basex_o = basex;
        srcy_o = srcy;
        ix_o = (uint32_t)ix;
        iy_o = (uint32_t)iy;
        img_o = (uint32_t)img_index | ((mirrored ? 1u : 0u) << 12) | ((uint32_t)io << 16);
    }
    // draw_image BAG:877-913 for one drawable (lane-local); returns the image index or -1
    PG_DEV int resolve_image(int base_type, int theme, float rotation, float tile_ratio, RectD &rect) {
        const int img_type = Game::image_for_type(*this, base_type);
        if (img_type < 0) return -1;
        if (d.opt.use_monochrome_assets || img_type >= USE_ASSET_THRESHOLD) {
            if (img_type != SPACE) fail(PGE_UNSUPPORTED_DRAW);  // colored grid squares: not on the default-option path yet
            return -1;
        }
        rect = Game::adjusted_image_rect(img_type, rect);
        int mt = theme;
        if (d.opt.restrict_themes && !Game::should_preserve_type_themes(img_type)) mt = 0;  // BAG:450-453
        const int img = (mt >= 0 && mt < MAX_IMAGE_THEMES) ? (int)d.assets->type_theme_img[img_type][mt] : -1;
        if (img < 0) {
            fail(PGE_THEME);
            return -1;
        }
        if (rotation != 0 || tile_ratio != 0) {
            fail(PGE_UNSUPPORTED_DRAW);
            return -1;
        }
        return img;
    }

    // ---- command execution ----------------------------------------------------------------------------------
    PG_DEV static DrawCmd unpack(uint32_t geom, uint32_t basex, uint32_t srcy0, uint32_t ix, uint32_t iy, uint32_t img) {
        DrawCmd c;
        c.tx1 = (int)(geom & 0x7fu);
        c.ty1 = (int)((geom >> 7) & 0x7fu);
        c.w = (int)((geom >> 14) & 0x7fu);
        c.h = (int)((geom >> 21) & 0x7fu);
        c.basex = basex;
        c.srcy0 = srcy0;
        c.ix = ix;
        c.iy = iy;
        c.img = img;
        return c;
    }
    PG_DEV static uint32_t blend(uint32_t sp, uint32_t dst, int io, uint32_t ca) {
        if (io != 256) sp = byte_mul(sp, ca);
        return sp + byte_mul(dst, 255u - (sp >> 24));
    }
    // one command of any size: every lane takes up to 8 pixels per round, all 8 texel fetches before the blends
    PG_DEV void exec_large(const DrawCmd &c) {
        const ImgDesc im = d.assets->img[c.img & 0xfffu];
        const uint32_t *src = d.pixels + im.off;
        const bool mirrored = ((c.img >> 12) & 1u) != 0;
        const int io = (int)(c.img >> 16);
        const uint32_t ca = (uint32_t)((io * 255) >> 8);
        const int y0 = c.ty1 > row0 ? c.ty1 : row0;
        const int y1 = (c.ty1 + c.h) < row1 ? (c.ty1 + c.h) : row1;
        const int npix = c.w * (y1 - y0);
        const uint32_t inv = (uint32_t)(((1u << 20) + (uint32_t)c.w - 1u) / (uint32_t)c.w);  // p / w == (p * inv) >> 20 for p < 4096
        for (int base = 0; base < npix; base += 512) {
            PG_FOR_LANES(l) {
                uint32_t tex[8];
                int fbi[8];
                _Pragma("unroll") for (int j = 0; j < 8; j++) {
                    const int p = base + j * 64 + l;
                    fbi[j] = -1;
                    tex[j] = 0;
                    if (p < npix) {
                        const int pyb = (int)(((uint32_t)p * inv) >> 20);
                        const int px = p - pyb * c.w;
                        const int y = y0 + pyb;
                        const int sxp = (int)((c.basex + (uint32_t)px * c.ix) >> 16);
                        const int syp = (int)((c.srcy0 + (uint32_t)(y - c.ty1) * c.iy) >> 16);
                        tex[j] = src[syp * (int)im.w + (mirrored ? ((int)im.w - 1 - sxp) : sxp)];
                        fbi[j] = (y - row0) * RES_W + c.tx1 + px;
                    }
                }
                _Pragma("unroll") for (int j = 0; j < 8; j++) {
                    if (fbi[j] >= 0) fb[fbi[j]] = blend(tex[j], fb[fbi[j]], io, ca);
                }
            }
        }
        PG_SYNC();
    }
    // up to 8 commands of at most 8x8 pixels each: lane = (row, column) of the footprint; the 8 texel fetches are
    // issued together, the blends then run command by command (two commands may touch the same pixel from
    // different lanes, so each blend is its own lane section)
    PG_DEV void exec_small_group(const DrawCmd (&c)[8], int count) {
        PG_LANE_ARR(uint32_t, tex, 8);
        PG_LANE_ARR(int, fbi, 8);
        PG_FOR_LANES(l) {
            const int lx = l & 7, ly = l >> 3;
            _Pragma("unroll") for (int g = 0; g < 8; g++) {
                PG_LA(fbi, g, l) = -1;
                PG_LA(tex, g, l) = 0;
                if (g < count) {
                    const int y = c[g].ty1 + ly;
                    if (lx < c[g].w && ly < c[g].h && y >= row0 && y < row1) {
                        const ImgDesc im = d.assets->img[c[g].img & 0xfffu];
                        const int sxp = (int)((c[g].basex + (uint32_t)lx * c[g].ix) >> 16);
                        const int syp = (int)((c[g].srcy0 + (uint32_t)ly * c[g].iy) >> 16);
                        const bool mirrored = ((c[g].img >> 12) & 1u) != 0;
                        PG_LA(tex, g, l) = d.pixels[im.off + (uint32_t)(syp * (int)im.w + (mirrored ? ((int)im.w - 1 - sxp) : sxp))];
                        PG_LA(fbi, g, l) = (y - row0) * RES_W + c[g].tx1 + lx;
                    }
                }
            }
        }
        _Pragma("unroll") for (int g = 0; g < 8; g++) {
            if (g < count) {
                const int io = (int)(c[g].img >> 16);
                const uint32_t ca = (uint32_t)((io * 255) >> 8);
                PG_FOR_LANES(l) {
                    const int fi = PG_LA(fbi, g, l);
                    if (fi >= 0) fb[fi] = blend(PG_LA(tex, g, l), fb[fi], io, ca);
                }
            }
        }
        PG_SYNC();
    }

    // one draw command per lane, produced by a set-up section and consumed by run_batch()
    struct CmdRegs {
        PG_LANE_VAR(uint32_t, geom);
        PG_LANE_VAR(uint32_t, basex);
        PG_LANE_VAR(uint32_t, srcy);
        PG_LANE_VAR(uint32_t, ix);
        PG_LANE_VAR(uint32_t, iy);
        PG_LANE_VAR(uint32_t, img);
    };
    PG_DEV static DrawCmd read_cmd(const CmdRegs &r, int k) {
        return unpack(PG_READLANE(r.geom, k), PG_READLANE(r.basex, k), PG_READLANE(r.srcy, k), PG_READLANE(r.ix, k), PG_READLANE(r.iy, k),
                      PG_READLANE(r.img, k));
    }
    // executes the commands in lane order; runs of small commands go eight at a time
    PG_DEV void run_batch(const CmdRegs &r, uint64_t lane_mask = ~0ull) {
        uint64_t valid = PG_BALLOT(l, PG_LV(r.geom, l) != 0) & lane_mask;
        const uint64_t small = PG_BALLOT(l, PG_LV(r.geom, l) != 0 && ((PG_LV(r.geom, l) >> 14) & 0x7fu) <= 8u && ((PG_LV(r.geom, l) >> 21) & 0x7fu) <= 8u);
        while (valid) {
            const int k = pg_ctz64(valid);
            if ((small >> k) & 1ull) {
                DrawCmd c[8];
                int count = 0;
                bool open = true;
                _Pragma("unroll") for (int g = 0; g < 8; g++) {
                    c[g].w = 0;
                    c[g].h = 0;
                    const int kk = pg_ctz64(valid);
                    open = open && valid != 0 && ((small >> (kk & 63)) & 1ull);
                    if (open) {
                        valid &= valid - 1;
                        c[g] = read_cmd(r, kk);
                        count = g + 1;
                    }
                }
                exec_small_group(c, count);
            } else {
                valid &= valid - 1;
                exec_large(read_cmd(r, k));
            }
        }
    }

    // draw_entities BAG:1052-1066.  Commands of 64 entities are set up once (they do not depend on the layer) and
    // then executed per render_z layer through a lane mask.
    PG_DEV void setup_entities(int base, CmdRegs &r, uint64_t (&zmask)[3]) {
        const int n = G.n_ents;
        for (int z = 0; z < 3; z++) zmask[z] = PG_BALLOT(l, (base + l) < n && meta_render_z(meta(base + l)) == z - 1);
        PG_FOR_LANES(l) {
            const int i = base + l;
            PG_LV(r.geom, l) = 0;
            PG_LV(r.basex, l) = PG_LV(r.srcy, l) = PG_LV(r.ix, l) = PG_LV(r.iy, l) = PG_LV(r.img, l) = 0;
            if (i < n && Game::should_draw_entity(*this, i)) {
                const uint32_t mm = meta(i);
                const float x = ex(i), y = ey(i), rx = erx(i), ry = ery(i);
                RectD r1;  // get_object_rect BAG:811-817
                if (mm & MF_ABS_COORDS) {
                    const float vd = G.view_dim;
                    r1.x = (double)((vd * (x - rx)) * G.unit);
                    r1.y = (double)((vd * (y + ry)) * G.unit);
                    r1.w = (double)((2 * vd * rx) * G.unit);
                    r1.h = (double)((2 * vd * ry) * G.unit);
                } else {
                    r1 = get_screen_rect(x - rx, y + ry, 2 * rx, 2 * ry, 0);
                }
                const int im = resolve_image(meta_image_type(mm), meta_image_theme(mm), ef(EF_ROTATION, i), Game::tile_aspect_ratio(*this, i), r1);
                if (im >= 0) cmd_image(im, (mm & MF_REFLECTED) != 0, r1, ef(EF_ALPHA, i), PG_LV(r.geom, l), PG_LV(r.basex, l), PG_LV(r.srcy, l), PG_LV(r.ix, l), PG_LV(r.iy, l), PG_LV(r.img, l));
            }
        }
    }
    PG_DEV void draw_entities(int render_z) {  // general path (more than 64 entities): one set-up per layer and chunk
        const int n = G.n_ents;
        for (int base = 0; base < n; base += 64) {
            const uint64_t any = PG_BALLOT(l, (base + l) < n && meta_render_z(meta(base + l)) == render_z);
            if (!any) continue;
            CmdRegs r;
            uint64_t zmask[3];
            setup_entities(base, r, zmask);
            run_batch(r, zmask[render_z + 1]);
        }
    }

    // game_draw BAG:1009-1012 (draw_background BAG:979-1007 + draw_foreground BAG:921-970), restricted to the band
    PG_DEV void render_band() {
        {
            const EnvHdr *h = d.hdr + env;
#define PG_X(type, name) G.name = h->name;
            PG_HDR_FIELDS(PG_X)
#undef PG_X
        }
        for (int base = 0; base < BAND_ROWS * RES_W; base += 64) {
            PG_FOR_LANES(l) { fb[base + l] = 0xff000000u; }  // p.fillRect(rect, QColor(0,0,0))
        }
        PG_SYNC();
        if (d.opt.use_backgrounds) {
            const RectD main_rect = get_screen_rect(0, (float)G.main_height, (float)G.main_width, (float)G.main_height, 0);
            const int bgi = (int)d.assets->bg_img[G.background_index];
            if (G.bg_tile_ratio < 0) fail(PGE_UNSUPPORTED_DRAW);
            const ImgDesc bim = d.assets->img[bgi];
            const float bgw = (float)bim.w, bgh = (float)bim.h;
            const float bg_ar = bgw / bgh;
            const float world_ar = (float)(G.main_width * 1.0 / G.main_height);
            const float extra_w = bg_ar - world_ar;
            const float offset_x = G.bg_pct_x * extra_w;
            const RectD bg_rect = adjust_rect(main_rect, (double)(-offset_x), 0, (double)(bg_ar / world_ar), 1);
            uint32_t geom, basex, srcy, ix, iy, img;  // wave-uniform: every lane computes the same command
            cmd_image(bgi, false, bg_rect, 1.0f, geom, basex, srcy, ix, iy, img);
            if (geom != 0) exec_large(unpack(geom, basex, srcy, ix, iy, img));
        }
        // common case (<= 64 entities): their commands are built once and kept in registers across the tile pass
        const bool one_chunk = G.n_ents <= 64;
        CmdRegs er;
        uint64_t ezmask[3] = {0, 0, 0};
        if (one_chunk) {
            setup_entities(0, er, ezmask);
            if (ezmask[0]) run_batch(er, ezmask[0]);
        } else {
            draw_entities(-1);
        }
        int low_x, high_x, low_y, high_y;
        if (Game::center_agent(d.opt)) {
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
        {
            // only cell rows whose (inflated) rect can reach this band: screen y falls as cell y grows.
            // Conservative by a full cell either side; cells outside the range draw nothing into the band, and
            // dropping them keeps the x-major order of the rest (BAG:941-955).
            const float inv_unit = 1.0f / G.unit;
            const int cy_hi = (int)pg_ceil((double)(G.view_dim - ((float)row0 - G.y_off) * inv_unit)) + 1;
            const int cy_lo = (int)pg_floor((double)(G.view_dim - ((float)row1 - G.y_off) * inv_unit)) - 2;
            if (cy_lo > low_y) low_y = cy_lo;
            if (cy_hi < high_y) high_y = cy_hi;
        }
        const int ny = high_y - low_y + 1;
        const int ncell = ny > 0 ? (high_x - low_x + 1) * ny : 0;
        const uint32_t ny_inv = ny > 0 ? (uint32_t)(((1u << 20) + (uint32_t)ny - 1u) / (uint32_t)ny) : 0u;
        if (ncell > 4096) fail(PGE_ASSERT);
        for (int base = 0; base < ncell; base += 64) {
            CmdRegs r;
            PG_FOR_LANES(l) {
                const int cidx = base + l;
                PG_LV(r.geom, l) = 0;
                PG_LV(r.basex, l) = PG_LV(r.srcy, l) = PG_LV(r.ix, l) = PG_LV(r.iy, l) = PG_LV(r.img, l) = 0;
                if (cidx < ncell) {
                    const int cx = (int)(((uint32_t)cidx * ny_inv) >> 20);  // cidx / ny (exact for cidx < 4096)
                    const int x = low_x + cx, y = low_y + (cidx - cx * ny);
                    const int type = get_obj(x, y);
                    if (type != INVALID_OBJ && type != SPACE) {
                        const int theme = Game::theme_for_grid_obj(*this, type);
                        RectD r2 = get_screen_rect((float)x, (float)(y + 1), 1, 1, RENDER_EPS);
                        const int im = resolve_image(type, theme, 0.0f, 0.0f, r2);
                        if (im >= 0) cmd_image(im, false, r2, 1.0f, PG_LV(r.geom, l), PG_LV(r.basex, l), PG_LV(r.srcy, l), PG_LV(r.ix, l), PG_LV(r.iy, l), PG_LV(r.img, l));
                    }
                }
            }
            run_batch(r);
        }
        if (one_chunk) {
            if (ezmask[1]) run_batch(er, ezmask[1]);
            if (ezmask[2]) run_batch(er, ezmask[2]);
        } else {
            draw_entities(0);
            draw_entities(1);
        }
        if (G.has_useful_vel_info && d.opt.paint_vel_info) fail(PGE_UNSUPPORTED_DRAW);
        PG_SYNC();
        store_band();
    }

    // bgr32_to_rgb888 + the ob write of Game::observe (reference src/game.cpp:8-23,159): 4 pixels -> 3 dwords per
    // lane; each wave-wide store instruction covers 768 contiguous bytes of the observation buffer.
    PG_DEV void store_band() {
        uint32_t *out = reinterpret_cast<uint32_t *>(d.obs + (size_t)env * OBS_BYTES + (size_t)row0 * RES_W * 3);
        for (int base = 0; base < BAND_ROWS * RES_W; base += 256) {
            PG_FOR_LANES(l) {
                const uint32_t *p = &fb[base + 4 * l];
                const uint32_t p0 = p[0], p1 = p[1], p2 = p[2], p3 = p[3];
                // bytes: R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3   (pixel word = 0xffRRGGBB)
                const uint32_t r0 = (p0 >> 16) & 0xff, g0 = (p0 >> 8) & 0xff, b0 = p0 & 0xff;
                const uint32_t r1 = (p1 >> 16) & 0xff, g1 = (p1 >> 8) & 0xff, b1 = p1 & 0xff;
                const uint32_t r2 = (p2 >> 16) & 0xff, g2 = (p2 >> 8) & 0xff, b2 = p2 & 0xff;
                const uint32_t r3 = (p3 >> 16) & 0xff, g3 = (p3 >> 8) & 0xff, b3 = p3 & 0xff;
                uint32_t *o = out + (base / 4) * 3 + 3 * l;
                o[0] = r0 | (g0 << 8) | (b0 << 16) | (r1 << 24);
                o[1] = g1 | (b1 << 8) | (r2 << 16) | (g2 << 24);
                o[2] = b2 | (r3 << 8) | (g3 << 16) | (b3 << 24);
            }
        }
        if (G.error) {
#if defined(PGAMD_WAVE_EMU)
            if (d.error) *d.error |= G.error;
#else
            if (PG_LANE_ID() == 0) atomicOr(d.error, G.error);
#endif
        }
    }
};

}  // namespace pgamd
