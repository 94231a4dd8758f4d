// pg_render.h -- the 64x64x3 frame of one environment, produced by ONE wavefront in four passes: pass b rasterizes
// rows [16b, 16b+16) into a 4 KB LDS band and writes that band of the RGB888 observation with fully coalesced
// stores.  The frame-level set-up (header, entity draw commands, per-column / per-row tile geometry, background
// command) is done once and kept in registers / 0.5 KB of LDS across the passes, so a workgroup needs only 4.5 KB
// of LDS and the CU runs at its 32-wave occupancy limit.
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
#include "pg_math.h"

namespace pgamd {

// Lane sections of the renderer take their lane id through pg_lane_opaque() (wave.h): what they derive from it is recomputed where it
// is used instead of being hoisted to the top of the kernel and held in registers for a whole frame.  Measured on coinrun (round 4,
// same box): 134 -> 106 VGPRs, steps/s equal to the hoisting build under an occupancy hint -- the headroom keeps the kernel at four
// waves per SIMD while its code changes (-DPG_RENDER_HOIST: the other form).
#if !defined(PG_RENDER_HOIST)
#define PG_R_LANES(l) PG_FOR_LANES_NOHOIST(l)
#else
#define PG_R_LANES(l) PG_FOR_LANES(l)
#endif

#ifndef PG_BAND_ROWS
#define PG_BAND_ROWS 16
#endif
constexpr int BAND_ROWS = PG_BAND_ROWS;  // rows per pass (64 / BAND_ROWS passes per frame)
constexpr int NUM_BANDS = RES_H / BAND_ROWS;

struct DrawCmd {  // uniform (scalar) view of one command
    int tx1, ty1, w, h;
    uint32_t basex, srcy0, ix, iy;
    uint32_t src;  // first pixel of the source image in the atlas blob
    uint32_t aux;  // source width (13 bits) | mirrored<<13 | opaque<<14 | rotated<<15 | const_alpha(0..256)<<16
    uint32_t e0, e1;  // Renderer<Game, GEN = true> only (see cmd_image_generic)
};
// Commands of a GEN renderer (use_generated_assets): a sprite is a 64 x 64 QImage::Format_ARGB32, which Qt draws through its
// generic span route (tests/tools/qt_generic_image_probe.py).  For the untransformed painter the command keeps the pixel
// box in geom, fx at the box's first column in basex and its per-column step in ix (16.16, fetchTransformed), and the two
// doubles of the row mapping fy(y) = int((i22 * (y + .5) + idy) * 65536): i22 in (srcy0, iy), idy in (e0, e1).  The
// background stays an RGB32 image on the fast path; aux bit 27 says its pixels are the env's canvas (DevCtx::gen_bg).
PG_DEV bool cmd_bgcanvas(uint32_t aux) { return ((aux >> 27) & 1u) != 0; }
// A rotated command (aux bit 15) keeps its bounding box in geom, its source height in iy, the index of its
// parameter record (RenderLds::rot) in basex; srcy0 / ix are unused.
// A tiled command (aux bit 25) keeps the bounding box of the entity's rect in geom and the entity's lane in basex;
// its tiles are generated when it is executed (exec_tiled).
PG_DEV int cmd_src_w(uint32_t aux) { return (int)(aux & 0x1fffu); }
PG_DEV bool cmd_mirrored(uint32_t aux) { return ((aux >> 13) & 1u) != 0; }
PG_DEV bool cmd_opaque(uint32_t aux) { return ((aux >> 14) & 1u) != 0; }
PG_DEV bool cmd_rotated(uint32_t aux) { return ((aux >> 15) & 1u) != 0; }
PG_DEV bool cmd_tiled(uint32_t aux) { return ((aux >> 25) & 1u) != 0; }
PG_DEV bool cmd_fill(uint32_t aux) { return ((aux >> 26) & 1u) != 0; }  // solid colour (draw_grid_obj): src holds the colour, no texture  // entity drawn as a row / column of tiles (tile_image BAG:840-865)
PG_DEV int cmd_alpha(uint32_t aux) { return (int)(aux >> 16); }
PG_DEV uint32_t cmd_aux(int src_w, bool mirrored, bool opaque, int io) {
    return (uint32_t)src_w | ((mirrored ? 1u : 0u) << 13) | (((opaque && io == 256) ? 1u : 0u) << 14) | ((uint32_t)io << 16);
}

// optional policy constants (defaults when a game does not declare them)
template <class Game, class = void>
struct GameUsesRotation {
    static constexpr bool value = false;
};
template <class Game>
struct GameUsesRotation<Game, decltype((void)Game::USES_ROTATION)> {
    static constexpr bool value = Game::USES_ROTATION;
};
// the cells of this game's window that hold an image mostly lie beside the screen (fruitbot: the out-of-bounds wall columns either side of its
// 20-column world) or nowhere (dodgeball: a world of SPACE): build_pull_tables checks every cell's column / row for pixels before it looks at it
template <class Game, class = void>
struct GameGridRarelyOnScreen {
    static constexpr bool value = false;
};
template <class Game>
struct GameGridRarelyOnScreen<Game, decltype((void)Game::GRID_RARELY_ON_SCREEN)> {
    static constexpr bool value = Game::GRID_RARELY_ON_SCREEN;
};
template <class Game, class = void>
struct GameDrawsGrid {
    static constexpr bool value = true;
};
template <class Game>
struct GameDrawsGrid<Game, decltype((void)Game::DRAWS_GRID)> {
    static constexpr bool value = Game::DRAWS_GRID;
};
// most cells the pull form of the grid pass may see at once (window columns x rows): sizes the LDS cell table, which
// bounds how many frames a CU renders at a time.  A policy whose centred window is small says so (PULL_CELLS).
template <class Game, class = void>
struct GamePullCells {
    static constexpr int value = 1024;
};
template <class Game>
struct GamePullCells<Game, decltype((void)Game::PULL_CELLS)> {
    static constexpr int value = Game::PULL_CELLS;
};
// the game's cell images share one size (PULL_SINGLE_SIZE in a policy: coinrun, climber): the arena carries no size-class tables, which is
// what lets seven instead of five render waves fit a SIMD's share of LDS; a frame that does show a second size takes the per-cell path
template <class Game, class = void>
struct GamePullSingle {
    static constexpr bool value = false;
};
template <class Game>
struct GamePullSingle<Game, decltype((void)Game::PULL_SINGLE_SIZE)> {
    static constexpr bool value = Game::PULL_SINGLE_SIZE;
};
// register sets (64 draw commands each) the frame's visible entities are packed into; a policy that routinely shows
// more than 64 entities asks for two (RENDER_CMD_SETS)
template <class Game, class = void>
struct GameRenderCmdSets {
    static constexpr int value = 1;
};
template <class Game>
struct GameRenderCmdSets<Game, decltype((void)Game::RENDER_CMD_SETS)> {
    static constexpr int value = Game::RENDER_CMD_SETS;
};
// rows per fetch batch of full-width draws (background, tile stage 1): a whole band by default; a policy whose renderer
// is short of registers takes half a band (WIDE_ROWS = 8) to keep a third wave per SIMD
template <class Game, class = void>
struct GameWideRows {
    static constexpr int value = BAND_ROWS;
};
template <class Game>
struct GameWideRows<Game, decltype((void)Game::WIDE_ROWS)> {
    static constexpr int value = Game::WIDE_ROWS;
};
// the game draws its background as a column of tiles (bg_tile_ratio < 0 in its constructor; fruitbot)
template <class Game, class = void>
struct GameTiledBackground {
    static constexpr bool value = false;
};
template <class Game>
struct GameTiledBackground<Game, decltype((void)Game::TILED_BACKGROUND)> {
    static constexpr bool value = Game::TILED_BACKGROUND;
};
template <class Game, class = void>
struct GameUsesTiledEntities {
    static constexpr bool value = false;
};
template <class Game>
struct GameUsesTiledEntities<Game, decltype((void)Game::USES_TILED_ENTITIES)> {
    static constexpr bool value = Game::USES_TILED_ENTITIES;
};
template <class Game, class = void>
struct GameHasGridFills {
    static constexpr bool value = false;
};
template <class Game>
struct GameHasGridFills<Game, decltype((void)Game::HAS_GRID_FILLS)> {
    static constexpr bool value = Game::HAS_GRID_FILLS;
};
template <class Game, class = void>
struct GameHasOverlay {
    static constexpr bool value = false;
};
template <class Game>
struct GameHasOverlay<Game, decltype((void)Game::HAS_OVERLAY)> {
    static constexpr bool value = Game::HAS_OVERLAY;
};
template <class Game, class = void>
struct GameCustomBackground {
    static constexpr bool value = false;
};
template <class Game>
struct GameCustomBackground<Game, decltype((void)Game::CUSTOM_BACKGROUND)> {
    static constexpr bool value = Game::CUSTOM_BACKGROUND;
};

// parameter record of one rotated command (qt_transform_image): inverse mapping + three trapezoids
constexpr int ROT_WORDS = 24;  // u0 v0 dudx dudy dvdx dvdy | 3 x (from_y to_y x_l dx_l x_r dx_r)
// Rotation / tile records per render workgroup: a pool of 16 (round 5; 64 = a record per entity lane, the form of rounds 1-4).  Records go
// only to the turned / tiled entities that can reach the rows being drawn, 16 at a time; a frame with more of them falls back to the
// per-band path, which cuts its chunks into windows of at most that many (0-0.2 % of the frames).  It takes 4.6 KB off the arena of the games
// with rotation (14.3 -> 9.7 KB: 11 -> 16 frames per CU).  Same-box A/B against 64 (profiles/r05_rot_pool_ab.txt, M steps/s): caveflyer
// 38.6 -> 48.0, heist 34.4 -> 37.3, starpilot 64.6 -> 70.1, plunder 71.8 -> 76.6, leaper 25.1 -> 27.0, dodgeball 24.6 -> 26.1, fruitbot
// 18.3 -> 20.2; bossfight keeps 64 through its factor (a third of its frames show more than 16 turned bullets).  -DPG_ROT_POOL=2 is the
// emulation tests' way to force every fallback (tests/test_kernel_logic_emu.py).
#ifndef PG_ROT_POOL
#define PG_ROT_POOL 16
#endif
static_assert(PG_ROT_POOL >= 1 && PG_ROT_POOL <= 64, "records are addressed by six bits");
// a policy with many turned sprites on screen (bossfight's bullets, starpilot) asks for a multiple: ROT_POOL_FACTOR
template <class Game, class = void>
struct GameRotPool {
    static constexpr int value = PG_ROT_POOL;
};
template <class Game>
struct GameRotPool<Game, decltype((void)Game::ROT_POOL_FACTOR)> {
    static constexpr int value = PG_ROT_POOL * Game::ROT_POOL_FACTOR < 64 ? PG_ROT_POOL * Game::ROT_POOL_FACTOR : 64;
};

// The raster kernel of this display-list game keeps the pull form's column / row / type tables in vector registers (RenderLdsT, raster_env):
// the games whose cell images share one size -- their short path has no other reader of these tables.
template <class Game>
struct GameRasterRegTabs {
    static constexpr bool value = GameDrawsGrid<Game>::value && GamePullSingle<Game>::value;
};
// ... and then its band buffer lies where those tables would (RenderLdsT::ci); so does the one of a game that draws no grid (bigfish: 5472 ->
// 4432 bytes, four LDS granules instead of five).  Neither has rotation records, which lie between the tables and the band buffer.
template <class Game>
struct GameRasterBandOverTables {
    static constexpr bool value = (GameRasterRegTabs<Game>::value || !GameDrawsGrid<Game>::value) && !GameUsesRotation<Game>::value;
};

// LDS arena of one render workgroup (one wave)
template <class Game>
struct RenderLdsT {
    // Tables that are never alive together share their words (the arena bounds how many frames a CU renders at a time: 8 KB is the
    // step from four to five waves per SIMD): the per-cell path's axis table `ax` (setup_tile_axes) lies over ci -- a frame is drawn in
    // pull form or cell by cell --, typesz (set-up only) over words 128..191 of the band buffer (idle during the set-up; build_pull_tables
    // keeps its scratch in words 0..127, the entity commands are staged there only after the pull tables are done); see Renderer::ax / typesz.
    // ---- the pull form's tables.  For a display-list game (pg_prep.h) this block IS the table part of an env's frame record: the prep
    // kernel builds it here and copies it out word by word, the raster kernel copies it back in; [seamcols, srcx) is all a frame with one cell
    // image size needs (TAB_SINGLE_WORDS), the size-class tables follow
    uint8_t seamcols[64];            // screen columns covered by two cell columns
    // window cell -> grid object type (CELL8_NONE: nothing to draw); the type's image is typeany[type].  One byte per cell (round 4; a word
    // per cell with the image in it cost the games with large windows 4 KB of the arena, i.e. two of eleven resident frames per CU)
    uint8_t cellimg[GameDrawsGrid<Game>::value ? GamePullCells<Game>::value : 4];
    uint32_t fillcmd[GameHasGridFills<Game>::value ? 2 * 256 : 1];  // solid-colour cells of a pull-form frame: (geom, colour) pairs
    // The three tables a register-table rasterizer (GameRasterRegTabs, raster_env) holds in five vector registers -- word k in lane k, read
    // with ds_bpermute / v_readlane -- instead of in the arena: its band buffer starts HERE (Renderer::fb), its arena ends 1280 bytes earlier.
    // 1280 bytes is the LDS allocation granule of gfx950 (profiles/r06_lds_granule.txt): coinrun's 5968-byte arena is five of them, 25 frames
    // per CU; without the three tables it is four, 32 frames per CU = eight waves per SIMD.
    // (a game that draws no grid -- bigfish, starpilot, plunder, bossfight -- has neither: bossfight's arena 11 632 -> 10 640 bytes, nine LDS granules
    // instead of ten, 14 frames per CU instead of 12)
    alignas(16) uint32_t ci[GameDrawsGrid<Game>::value ? 2 : 1][GameDrawsGrid<Game>::value ? 64 : 4];  // screen column -> the (at most two) cell columns covering it
    uint32_t ri[GameDrawsGrid<Game>::value ? 2 : 1][GameDrawsGrid<Game>::value ? 64 : 4];              // screen row    -> the (at most two) cell rows covering it
    uint32_t typeany[GameDrawsGrid<Game>::value ? 64 : 1];    // grid object type -> cell image of any size: atlas offset | size class<<27 | opaque<<31 (pull form)
    static constexpr bool SIZE_CLASSES = GameDrawsGrid<Game>::value && !GamePullSingle<Game>::value;
    uint8_t srcx[SIZE_CLASSES ? 3 : 1][SIZE_CLASSES ? 2 : 1][SIZE_CLASSES ? 64 : 4];    // size classes 1..3: screen column -> source column, per covering slot
    uint16_t srcyw[SIZE_CLASSES ? 3 : 1][SIZE_CLASSES ? 2 : 1][SIZE_CLASSES ? 64 : 2];  // size classes 1..3: screen row -> source row * image width
    // ---- end of the record's table part
    uint32_t rot[GameUsesRotation<Game>::value ? GameRotPool<Game>::value * ROT_WORDS : 1];  // rotated commands of the current 64-entity chunk, by lane (by pool slot: ROT_POOL)
    // the band being rasterized, 0xffRRGGBB (+ a dump row for masked-off lanes).  Last: the prep kernel of a display-list game (pg_prep.h),
    // which only builds tables -- and uses the first PREP_FB_WORDS words of the band buffer as their scratch -- allocates the arena up to there
    alignas(16) uint32_t fb[BAND_ROWS * RES_W + 64];
    static constexpr int PREP_FB_WORDS = 192;
    // grid object type -> cell image of this frame (build_type_table), read by the per-cell path of render_env only.  Behind the band buffer:
    // the raster kernel of a display-list game allocates the arena up to here (one more workgroup per CU for coinrun), the prep kernel up to
    // PREP_FB_WORDS of the band buffer -- neither touches it (Renderer::keep_typeimg)
    uint32_t typeimg[GameDrawsGrid<Game>::value ? 64 : 1];
};
// Frame record of a display-list game (pg_prep.h): what prep<Game> leaves in HBM for raster<Game>, per env.  Words:
//   [0, 16)          header: flags, dims (window rows | visible commands << 8 | grid fills << 16), the pull form's column-seam / row-seam /
//                    row-any masks, the background command (7 words), the atlas' reference cell width
//   [CMD, TAB)       up to MAX_CMDS entity commands in draw order, 8 words each: geom basex srcy ix iy src aux | render_z + 1
//   [TAB, ...)       the pull form's tables exactly as they lie in the render arena (RenderLdsT [ci, typeimg))
template <class Game>
struct FrameRec {
    // MAX_CMDS: three register sets' worth -- coinrun's enemies leave eight trail sprites each, and 1 frame in 10 000 shows more than 64 sprites
    enum : int { FLAGS = 0, DIMS = 1, COLSEAM = 2, ROWSEAM = 4, ROWANY = 6, BG = 8, REF_W = 15, HDR_WORDS = 16, CMD = 16, CMD_WORDS = 8, MAX_CMDS = 192, TAB = CMD + MAX_CMDS * CMD_WORDS };
    enum : uint32_t { F_FAST = 1u, F_PULL = 2u, F_MULTI = 4u };
    static constexpr int LDS_TAB_WORD0 = (int)(offsetof(RenderLdsT<Game>, seamcols) / 4);
    static constexpr int TAB_SINGLE_WORDS = (int)((offsetof(RenderLdsT<Game>, srcx) - offsetof(RenderLdsT<Game>, seamcols) + 3) / 4);
    static constexpr int TAB_WORDS = (int)((offsetof(RenderLdsT<Game>, rot) - offsetof(RenderLdsT<Game>, seamcols) + 3) / 4);
    // the table block's words of ci / ri / typeany (what a register-table rasterizer loads one word per lane; the words before ci go to LDS)
    static constexpr int TAB_CI = (int)((offsetof(RenderLdsT<Game>, ci) - offsetof(RenderLdsT<Game>, seamcols)) / 4);
    static constexpr int TAB_RI = (int)((offsetof(RenderLdsT<Game>, ri) - offsetof(RenderLdsT<Game>, seamcols)) / 4);
    static constexpr int TAB_TYPEANY = (int)((offsetof(RenderLdsT<Game>, typeany) - offsetof(RenderLdsT<Game>, seamcols)) / 4);
    static constexpr int WORDS = (TAB + (GameDrawsGrid<Game>::value ? TAB_WORDS : 0) + 3) / 4 * 4;  // (records start on 16-byte boundaries)
    static_assert(offsetof(RenderLdsT<Game>, seamcols) % 4 == 0 && offsetof(RenderLdsT<Game>, ci) % 4 == 0 && offsetof(RenderLdsT<Game>, rot) % 4 == 0, "the table block is whole words");
};
template <bool GEN>
struct CmdExtra {};
template <>
struct CmdExtra<true> {
    PG_LANE_VAR(uint32_t, e0);
    PG_LANE_VAR(uint32_t, e1);
};
constexpr uint32_t CELL_NONE = 0xffffffffu;
constexpr uint8_t CELL8_NONE = 0xffu, CELL8_FILL = 0xfeu;  // RenderLdsT::cellimg codes beside the type ids 0..63
template <int N>
struct PgInt {
    static constexpr int value = N;
};
constexpr uint32_t CELL_FILL = 0xfffffffdu;  // build_pull_tables, transient: a solid-colour cell (draw_grid_obj override) waiting for its command
constexpr uint32_t TYPE_SLOW = 0xfffffffeu;  // typeimg: this type needs the per-cell path (fill, odd image size, adjusted rect, missing asset)

// x86 double -> int32 conversion (cvttsd2si): out-of-range and NaN give INT_MIN.  Qt's edge walkers convert
// unbounded slopes this way; the GPU's conversion saturates instead.
PG_DEV int d2i_x86(double v) { return (v > -2147483649.0 && v < 2147483648.0) ? (int)v : (int)0x80000000; }
PG_DEV bool q_fuzzy_is_null(double v) { return (v < 0 ? -v : v) <= 0.000000000001; }

template <class Game, bool GEN = false>
struct Renderer {
    const DevCtx &d;
    GameOptions opt;  // this env's options (pg_defs.h env_options), bound when the header has been read
    const int env;
    typedef RenderLdsT<Game> RenderLds;
    static constexpr int FRAME_W = RES_W, FRAME_H = RES_H;  // (pg_human.h's renderer draws the same policies at 512 x 512)
    static constexpr int WIDE_ROWS = GameWideRows<Game>::value;
    static_assert(BAND_ROWS % WIDE_ROWS == 0, "fetch batches tile the band");
    RenderLds *lds;
    uint32_t *fb;  // the band being rasterized: BAND_ROWS x 64 words of 0xffRRGGBB
    uint32_t *ax;  // tile-axis scratch (see setup_tile_axes): 128 words over ci
    uint32_t *typeany;  // = lds->typeany
    uint32_t *typesz;   // its width<<16 | height; over fb[128..191]
    EnvHdr G;
    const uint32_t *ge;  // this env's entity table in HBM
    int ecap;
    const typename Game::cell_t *gg;
    int row0, row1;  // band rows [row0, row1)
    bool keep_typeimg = true;  // false in the prep kernel: its arena ends before RenderLdsT::typeimg
    // register-table rasterizer (GameRasterRegTabs): ci[0], ci[1], ri[0], ri[1], typeany of the frame's record, word k in lane k
    PG_LANE_VAR(uint32_t, tr_ci0);
    PG_LANE_VAR(uint32_t, tr_ci1);
    PG_LANE_VAR(uint32_t, tr_ri0);
    PG_LANE_VAR(uint32_t, tr_ri1);
    PG_LANE_VAR(uint32_t, tr_ty);

    PG_DEV Renderer(const DevCtx &d_, int env_, RenderLds *lds_) : d(d_), env(env_), lds(lds_), fb(lds_->fb), ax(&lds_->ci[0][0]), typeany(lds_->typeany), typesz(lds_->fb + 128) {
        ge = d.ents + ent_table_base(env, d.ent_cap);
        ecap = d.ent_cap;
        gg = reinterpret_cast<const typename Game::cell_t *>(d.grid + (size_t)env * d.grid_bytes);
        row0 = 0;
        row1 = RES_H;
    }

    // entity accessors with the names the game policies use (HBM reads; the table was written by the step kernel)
    PG_DEV float ef(int field, int i) const { return __builtin_bit_cast(float, ge[(uint32_t)(field * ecap + i)]); }
    PG_DEV uint32_t meta(int i) const { return ge[(uint32_t)(EF_META * ecap + i)]; }
    PG_DEV float ex(int i) const { return ef(EF_X, i); }
    PG_DEV float ey(int i) const { return ef(EF_Y, i); }
    PG_DEV float evx(int i) const { return ef(EF_VX, i); }
    PG_DEV float evy(int i) const { return ef(EF_VY, i); }
    PG_DEV float erx(int i) const { return ef(EF_RX, i); }
    PG_DEV float ery(int i) const { return ef(EF_RY, i); }
    PG_DEV int etype(int i) const { return meta_type(meta(i)); }
    PG_DEV void fail(int code) {  // (the code alone: a line number per call site, as the step kernels record, costs this kernel the registers it does not have)
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
    PG_DEV static int opacity_to_io(float opacity) {  // QPainter::setOpacity clamps to [0,1]; intOpacity = int(opacity * 256)
        double o = (double)opacity;
        if (o < 0) o = 0;
        if (o > 1) o = 1;
        return (int)(o * 256);
    }
    // qt_scale_image_32bit.  tr.w / tr.h may be negative (a 180 degree rotation arrives as a negative scale).
    PG_DEV void cmd_image(const ImgDesc im, bool mirrored, RectD tr, float opacity, uint32_t &geom, uint32_t &basex_o, uint32_t &srcy_o,
                          uint32_t &ix_o, uint32_t &iy_o, uint32_t &src_o, uint32_t &aux_o) const {
        cmd_image_fast(im, mirrored, tr, opacity, geom, basex_o, srcy_o, ix_o, iy_o, src_o, aux_o);
    }
    // (type, theme) -> image descriptor with ONE load: the host resolved the type -> theme -> image indirection into
    // GameAssetsDev::type_theme_desc.  Branch-free (out-of-range arguments read entry 0 and report IMG_NONE), so that a set-up
    // section can request the descriptors of all its drawables before it needs the first one.
    PG_DEV ImgDesc desc_for(int img_type, int theme) const {
        int mt = theme;
        if (opt.restrict_themes && !Game::should_preserve_type_themes(img_type)) mt = 0;  // BAG:450-453
        const bool ok = img_type >= 0 && img_type < MAX_ASSETS && mt >= 0 && mt < MAX_IMAGE_THEMES;
        ImgDesc r = d.assets->type_theme_desc[ok ? img_type : 0][ok ? mt : 0];
        if (!ok) r.off = IMG_NONE;
        return r;
    }
    // the descriptor resolve_image() will want for a drawable of this base type and theme
    PG_DEV ImgDesc request_desc(int base_type, int theme) {
        const int img_type = Game::image_for_type(*this, base_type);
        return desc_for((img_type >= 0 && img_type < USE_ASSET_THRESHOLD) ? img_type : -1, theme);
    }
    PG_DEV void cmd_image_desc(const ImgDesc im, bool mirrored, RectD tr, float opacity, uint32_t &geom, uint32_t &basex_o, uint32_t &srcy_o,
                               uint32_t &ix_o, uint32_t &iy_o, uint32_t &src_o, uint32_t &aux_o) const {
        static_assert(!GEN, "a GEN renderer's sprites go through cmd_image_generic, its background through cmd_image_fast");
        cmd_image_fast(im, mirrored, tr, opacity, geom, basex_o, srcy_o, ix_o, iy_o, src_o, aux_o);
    }
    PG_DEV void cmd_image_fast(const ImgDesc im, bool mirrored, RectD tr, float opacity, uint32_t &geom, uint32_t &basex_o, uint32_t &srcy_o,
                               uint32_t &ix_o, uint32_t &iy_o, uint32_t &src_o, uint32_t &aux_o) const {
        geom = 0;
        basex_o = srcy_o = ix_o = iy_o = src_o = aux_o = 0;
        const double sx = tr.w / (double)im.w;
        const double sy = tr.h / (double)im.h;
        const int ix = (int)(65536 / sx);
        const int iy = (int)(65536 / sy);
        const double right = tr.x + tr.w, bottom = tr.y + tr.h;
        int tx1 = q_round(tr.x), tx2 = q_round(right), ty1 = q_round(tr.y), ty2 = q_round(bottom);
        if (tx2 < tx1) { const int t = tx1; tx1 = tx2; tx2 = t; }
        if (ty2 < ty1) { const int t = ty1; ty1 = ty2; ty2 = t; }
        if (tx1 < 0) tx1 = 0;
        if (ty1 < 0) ty1 = 0;
        if (tx2 > RES_W) tx2 = RES_W;
        if (ty2 > RES_H) ty2 = RES_H;
        int w = tx2 - tx1, h = ty2 - ty1;
        if (w <= 0 || h <= 0) return;
        // Qt 5.9: qCeil(...) - 1 for positive scales (pinned with tests/tools/qt_drawimage_probe.py), qFloor(...) + 1
        // from the far edge for negative ones (tests/tools/qt_rotate_probe.py)
        uint32_t basex, srcy;
        if (sx < 0) basex = (uint32_t)im.w * 65536u + (uint32_t)((int)pg_floor((tx1 + 0.5 - right) * ix) + 1);
        else basex = (uint32_t)((int)pg_ceil((tx1 + 0.5 - tr.x) * ix) - 1);
        if (sy < 0) srcy = (uint32_t)im.h * 65536u + (uint32_t)((int)pg_floor((ty1 + 0.5 - bottom) * iy) + 1);
        else srcy = (uint32_t)((int)pg_ceil((ty1 + 0.5 - tr.y) * iy) - 1);
        if ((int)(srcy >> 16) >= (int)im.h && iy < 0) { srcy += (uint32_t)iy; --h; }
        if ((int)(basex >> 16) >= (int)im.w && ix < 0) { basex += (uint32_t)ix; --w; }
        const int yend = (int)((srcy + (uint32_t)iy * (uint32_t)(h - 1)) >> 16);
        if (yend < 0 || yend >= (int)im.h) --h;
        const int xend = (int)((basex + (uint32_t)ix * (uint32_t)(w - 1)) >> 16);
        if (xend < 0 || xend >= (int)im.w) --w;
        if (w <= 0 || h <= 0) return;
        if (ty1 >= row1 || ty1 + h <= row0) return;  // does not touch this wave's band
        const int io = opacity_to_io(opacity);
        geom = (uint32_t)tx1 | ((uint32_t)ty1 << 7) | ((uint32_t)w << 14) | ((uint32_t)h << 21);
        basex_o = basex;
        srcy_o = srcy;
        ix_o = (uint32_t)ix;
        iy_o = (uint32_t)iy;
        src_o = im.off;
        aux_o = cmd_aux((int)im.w, mirrored, im.opaque != 0, io);
    }

    // ---- GEN: Qt's generic span route for QImage::Format_ARGB32 sources (reference BAG:102-107; oracle draw_image_generic) -------
    PG_DEV static bool q26_equal(double p, double q) { return (int)((p - q) * 64) == 0; }  // qrasterizer.cpp q26Dot6Compare
    // QRasterizer::rasterizeLine(a, b, width), not antialiased, clip = the frame.  0: nothing; 1: the pixel box (x1, x2, y1, y2
    // inclusive) of an axis-aligned line; 2: the four corners (top, right, bottom, left) in 26.6 for the scan converter
    PG_DEV static int rasterize_line(double ax, double ay, double bx, double by, double width, int (&box)[4], int (&qx)[4], int (&qy)[4]) {
        const int cw = RES_W, ch = RES_H;
        if ((ax == bx && ay == by) || width == 0) return 0;
        double pax = ax, pay = ay, pbx = bx, pby = by;
        const double offx = pg_fabs(by - ay) * width * 0.5, offy = pg_fabs(bx - ax) * width * 0.5;
        const double cl = 0 - offx, ct = 0 - offy, cr = (cw - 1) + 1 + offx, cb = (ch - 1) + 1 + offy;
        const bool a_in = cl <= pax && pax <= cr && ct <= pay && pay <= cb, b_in = cl <= pbx && pbx <= cr && ct <= pby && pby <= cb;
        if (!a_in || !b_in) {
            double t1 = 0, t2 = 1;
            for (int i = 0; i < 2; i++) {
                const double o = i ? pay : pax, dd = i ? pby - pay : pbx - pax, low = i ? ct : cl, high = i ? cb : cr;
                if (dd == 0) {
                    if (o <= low || o >= high) return 0;
                    continue;
                }
                const double d_inv = 1 / dd;
                double t_low = (low - o) * d_inv, t_high = (high - o) * d_inv;
                if (t_low > t_high) { const double t = t_low; t_low = t_high; t_high = t; }
                if (t1 < t_low) t1 = t_low;
                if (t2 > t_high) t2 = t_high;
                if (t1 >= t2) return 0;
            }
            const double npax = pax + (pbx - pax) * t1, npay = pay + (pby - pay) * t1, npbx = pax + (pbx - pax) * t2, npby = pay + (pby - pay) * t2;
            pax = npax; pay = npay; pbx = npbx; pby = npby;
        }
        {
            const double d0x = ax - bx, d0y = ay - by, w0 = d0x * d0x + d0y * d0y;
            const double dx = pax - pbx, dy = pay - pby, w = dx * dx + dy * dy;
            if (w == 0) return 0;
            width *= pg_sqrt(w0 / w);
        }
        if (q26_equal(pay, pby)) {
            if (q26_equal(pax, pbx)) return 0;
            const double x = (pax + pbx) * 0.5, dx = pg_fabs(pbx - pax) * 0.5, y = pay, dy = width * dx;
            pax = x; pay = y - dy;
            pbx = x; pby = y + dy;
            width = 1 / width;
        }
        if (q26_equal(pax, pbx)) {
            if (pay > pby) { double t = pax; pax = pbx; pbx = t; t = pay; pay = pby; pby = t; }
            const double dy = pby - pay, half = 0.5 * width * dy;
            double left = pax - half, right = pax + half;
            left = left < 0 ? 0 : (left > cw ? cw : left);
            right = right < 0 ? 0 : (right > cw ? cw : right);
            pay = pay < 0 ? 0 : (pay > ch ? ch : pay);
            pby = pby < 0 ? 0 : (pby > ch ? ch : pby);
            if (q26_equal(left, right) || q26_equal(pay, pby)) return 0;
            box[0] = (int)(left + 0.5);
            box[1] = right < 0.5 ? -1 : (int)(right - 0.5);
            box[2] = (int)(pay + 0.5);
            box[3] = pby < 0.5 ? -1 : (int)(pby - 0.5);
            return (box[1] >= box[0] && box[3] >= box[2]) ? 1 : 0;
        }
        if (pay > pby) { double t = pax; pax = pbx; pbx = t; t = pay; pay = pby; pby = t; }
        const double dlx = (pbx - pax) * (0.5 * width), dly = (pby - pay) * (0.5 * width);
        const double perpx = dly, perpy = -dlx;
        double cxs[4], cys[4];  // top, right, bottom, left
        if (pax < pbx) {
            cxs[0] = pax + perpx; cys[0] = pay + perpy; cxs[3] = pax - perpx; cys[3] = pay - perpy;
            cxs[1] = pbx + perpx; cys[1] = pby + perpy; cxs[2] = pbx - perpx; cys[2] = pby - perpy;
        } else {
            cxs[0] = pax - perpx; cys[0] = pay - perpy; cxs[3] = pbx - perpx; cys[3] = pby - perpy;
            cxs[1] = pax + perpx; cys[1] = pay + perpy; cxs[2] = pbx + perpx; cys[2] = pby + perpy;
        }
        for (int i = 0; i < 4; i++) {
            qx[i] = (int)pg_floor(cxs[i] * 64.);
            qy[i] = (int)pg_floor(cys[i] * 64.);
        }
        return 2;
    }
    // the span [x1, x2] (inclusive, clipped to the frame) the scan converter gives row y of the four-corner polygon; false: none
    PG_DEV static bool polygon_row(const int (&qx)[4], const int (&qy)[4], int y, int &x1, int &x2) {
        int cnt = 0, lo = 0, hi = 0;
        for (int i = 0; i < 4; i++) {  // QScanConverter::mergeLine, evaluated for this row
            int a_x = qx[i], a_y = qy[i], b_x = qx[(i + 1) & 3], b_y = qy[(i + 1) & 3];
            if (a_y > b_y) {
                int t = a_x; a_x = b_x; b_x = t;
                t = a_y; a_y = b_y; b_y = t;
            }
            int itop = (a_y + 32) >> 6, ibot = (b_y - 32) >> 6;
            if (itop < 0) itop = 0;
            if (ibot > RES_H - 1) ibot = RES_H - 1;
            if (y < itop || y > ibot) continue;
            int xfp = 32768 + a_x * 1024;
            if (b_x != a_x) {
                const int slope = (int)((double)(b_x - a_x) / (double)(b_y - a_y) * 65536.);
                xfp += (int)(((long long)slope * (long long)((itop << 16) + 32768 - (a_y << 10))) >> 16);
                xfp += slope * (y - itop);
            }
            const int xi = xfp >> 16;
            if (cnt == 0) lo = hi = xi;
            else {
                if (xi < lo) lo = xi;
                if (xi > hi) hi = xi;
            }
            cnt++;
        }
        x1 = lo < 0 ? 0 : lo;
        x2 = (hi > RES_W ? RES_W : hi) - 1;
        return cnt >= 2 && x2 >= x1;
    }
    // p.drawImage(tr, sprite) with the untransformed painter (lane-local set-up)
    PG_DEV void cmd_image_generic(const ImgDesc im, bool mirrored, RectD tr, float opacity, uint32_t &geom, uint32_t &fx0_o, uint32_t &i22lo, uint32_t &fdx_o,
                                  uint32_t &i22hi, uint32_t &src_o, uint32_t &aux_o, uint32_t &idylo, uint32_t &idyhi) const {
        geom = 0;
        fx0_o = i22lo = fdx_o = i22hi = src_o = aux_o = idylo = idyhi = 0;
        if (!(tr.w > 0 && tr.h > 0)) return;  // drawImage: r.isEmpty()
        const double l = tr.x, t = tr.y, rr = tr.x + tr.w, b = tr.y + tr.h;
        int box[4], qx[4], qy[4];
        if (rasterize_line((l + l) * 0.5, (t + b) * 0.5, (rr + rr) * 0.5, (t + b) * 0.5, tr.h / tr.w, box, qx, qy) != 1) return;
        if (box[2] >= row1 || box[3] < row0) return;  // does not touch this wave's band
        // QSpanData::setupMatrix: inverse of translate(1/65536, 1/65536) * translate(tr.x, tr.y) * scale(tr.w / sw, tr.h / sh)
        const double dlt = 1.0 / 65536;
        const double c11 = 1.0 * (tr.w / (double)im.w), c22 = 1.0 * (tr.h / (double)im.h);
        const double p11 = 1.0 * c11, p22 = 1.0 * c22, p31 = dlt * c11 + tr.x, p32 = dlt * c22 + tr.y;
        const double i11 = 1. / p11, i22 = 1. / p22, idx = -p31 * i11, idy = -p32 * i22;
        const int fx0 = (int)((i11 * ((double)box[0] + 0.5) + idx) * 65536.);
        const int fdx = (int)(i11 * 65536.);
        geom = (uint32_t)box[0] | ((uint32_t)box[2] << 7) | ((uint32_t)(box[1] - box[0] + 1) << 14) | ((uint32_t)(box[3] - box[2] + 1) << 21);
        fx0_o = (uint32_t)fx0;
        fdx_o = (uint32_t)fdx;
        const uint64_t b22 = __builtin_bit_cast(uint64_t, i22), bdy = __builtin_bit_cast(uint64_t, idy);
        i22lo = (uint32_t)b22;
        i22hi = (uint32_t)(b22 >> 32);
        idylo = (uint32_t)bdy;
        idyhi = (uint32_t)(bdy >> 32);
        src_o = im.off;
        aux_o = cmd_aux((int)im.w, mirrored, im.opaque != 0, opacity_to_io(opacity));
    }
    // source texel of frame pixel (column offset lx from the command's box, row y)
    PG_DEV void sample_xy(const DrawCmd &c, int lx, int y, int sw, int &sxp, int &syp) const {
        if (GEN && !cmd_bgcanvas(c.aux)) {
            int px = ((int)c.basex + lx * (int)c.ix) >> 16;
            const double i22 = words_to_double(c.srcy0, c.iy), idy = words_to_double(c.e0, c.e1);
            int py = (int)((i22 * ((double)y + 0.5) + idy) * 65536.) >> 16;
            sxp = px < 0 ? 0 : (px > sw - 1 ? sw - 1 : px);
            syp = py < 0 ? 0 : (py > 63 ? 63 : py);  // generated sprites are 64 x 64
        } else {
            sxp = (int)((c.basex + (uint32_t)lx * c.ix) >> 16);
            syp = (int)((c.srcy0 + (uint32_t)(y - c.ty1) * c.iy) >> 16);
        }
    }

    // BAG:902-906: p.translate(cx, cy); p.rotate(rotation * 180 / PI); p.drawImage(QRectF(-w/2, -h/2, w, h), img).
    // QTransform::rotate special-cases 90 / 180 / 270 degrees; QTransform::type() then routes the draw to the
    // scale path (sine fuzzy-null: a 180 degree turn is a negative scale) or to qt_transform_image, whose set-up
    // (inverse 16.16 mapping, three trapezoids with 16.16 edge walkers) is written to the lane's LDS record.
    PG_DEV void cmd_image_rotated(int lane, const ImgDesc imdesc, bool mirrored, RectD adjusted, float rotation, float opacity, uint32_t &geom,
                                  uint32_t &basex_o, uint32_t &srcy_o, uint32_t &ix_o, uint32_t &iy_o, uint32_t &src_o, uint32_t &aux_o) {
        geom = 0;
        basex_o = srcy_o = ix_o = iy_o = src_o = aux_o = 0;
        const double cx = adjusted.x + adjusted.w / 2, cy = adjusted.y + adjusted.h / 2;
        const double a = (double)(rotation * 180 / PG_PI);
        const RectD r = {-adjusted.w / 2, -adjusted.h / 2, adjusted.w, adjusted.h};
        double sina = 0, cosa = 0;
        if (a == 0) cosa = 1;
        else if (a == 90. || a == -270.) sina = 1.;
        else if (a == 270. || a == -90.) sina = -1.;
        else if (a == 180.) cosa = -1.;
        else {
            const double b = 0.017453292519943295769 * a;
            sina = pg_sin_d(b);
            cosa = pg_cos_d(b);
        }
        const double m11 = cosa, m12 = sina, m21 = -sina, m22 = cosa;
        if (q_fuzzy_is_null(m12) && q_fuzzy_is_null(m21)) {  // TxScale or below: qt_mapRect_non_normalizing + scale path
            double x1, y1, x2, y2;
            if (!q_fuzzy_is_null(m11 - 1) || !q_fuzzy_is_null(m22 - 1)) {
                x1 = m11 * r.x + cx;
                y1 = m22 * r.y + cy;
                x2 = m11 * (r.x + r.w) + cx;
                y2 = m22 * (r.y + r.h) + cy;
            } else if (!q_fuzzy_is_null(cx) || !q_fuzzy_is_null(cy)) {
                x1 = r.x + cx;
                y1 = r.y + cy;
                x2 = (r.x + r.w) + cx;
                y2 = (r.y + r.h) + cy;
            } else {
                x1 = r.x;
                y1 = r.y;
                x2 = r.x + r.w;
                y2 = r.y + r.h;
            }
            const RectD tr = {x1, y1, x2 - x1, y2 - y1};
            cmd_image(imdesc, mirrored, tr, opacity, geom, basex_o, srcy_o, ix_o, iy_o, src_o, aux_o);
            return;
        }
        if constexpr (!GameUsesRotation<Game>::value) {
            fail(PGE_UNSUPPORTED_DRAW);
            return;
        } else {
            const ImgDesc im = imdesc;
            double vx[4], vy[4], vu[4], vv[4];
            {
                const double L = r.x, T = r.y, R = r.x + r.w, B = r.y + r.h;
                const double px[4] = {L, R, R, L}, py[4] = {T, T, B, B};
                const double pu[4] = {0, (double)im.w, (double)im.w, 0}, pv[4] = {0, 0, (double)im.h, (double)im.h};
                int topmost = 0;
                double tx[4], ty[4];
                for (int i = 0; i < 4; i++) {
                    tx[i] = m11 * px[i] + m21 * py[i] + cx;
                    ty[i] = m12 * px[i] + m22 * py[i] + cy;
                }
                for (int i = 1; i < 4; i++)
                    if (ty[i] < ty[topmost]) topmost = i;
                for (int i = 0; i < 4; i++) {
                    const int k = (topmost + i) & 3;
                    vx[i] = tx[k];
                    vy[i] = ty[k];
                    vu[i] = pu[k];
                    vv[i] = pv[k];
                }
            }
            {
                const double dx1 = vx[1] - vx[0], dy1 = vy[1] - vy[0], dx2 = vx[3] - vx[0], dy2 = vy[3] - vy[0];
                if (dx1 * dy2 - dx2 * dy1 > 0) {
                    double t;
                    t = vx[1]; vx[1] = vx[3]; vx[3] = t;
                    t = vy[1]; vy[1] = vy[3]; vy[3] = t;
                    t = vu[1]; vu[1] = vu[3]; vu[3] = t;
                    t = vv[1]; vv[1] = vv[3]; vv[3] = t;
                }
            }
            const double ux = vx[1] - vx[0], uy = vy[1] - vy[0], uu = vu[1] - vu[0], uv = vv[1] - vv[0];
            const double wx = vx[2] - vx[0], wy = vy[2] - vy[0], wu = vu[2] - vu[0], wv = vv[2] - vv[0];
            const double det = ux * wy - uy * wx;
            if (det == 0) return;
            const double det_inv = 1 / det;
            const double i11 = (uu * wy - uy * wu) * det_inv;
            const double i12 = (ux * wu - uu * wx) * det_inv;
            const double i21 = (uv * wy - uy * wv) * det_inv;
            const double i22 = (ux * wv - uv * wx) * det_inv;
            const double mdx = vu[0] - i11 * vx[0] - i12 * vy[0];
            const double mdy = vv[0] - i21 * vx[0] - i22 * vy[0];
            uint32_t *rp = &lds->rot[lane * ROT_WORDS];
            rp[0] = (uint32_t)(d2i_x86(pg_ceil((0.5 * i11 + 0.5 * i12 + mdx) * 0x10000)) - 1);  // u0
            rp[1] = (uint32_t)(d2i_x86(pg_ceil((0.5 * i21 + 0.5 * i22 + mdy) * 0x10000)) - 1);  // v0
            rp[2] = (uint32_t)d2i_x86(i11 * 0x10000);                                            // dudx
            rp[3] = (uint32_t)d2i_x86(i12 * 0x10000);                                            // dudy
            rp[4] = (uint32_t)d2i_x86(i21 * 0x10000);                                            // dvdx
            rp[5] = (uint32_t)d2i_x86(i22 * 0x10000);                                            // dvdy
            // trapezoids: top-left, bottom-left, top-right, bottom-right vertices and the top / bottom y of each, picked by
            // value (index tables would be addressed dynamically, i.e. live in scratch memory):
            //   vy[1] <  vy[3]: (0,1,0,3 | 0,1) (1,2,0,3 | 1,3) (1,2,3,2 | 3,2)
            //   vy[1] >= vy[3]: (0,1,0,3 | 0,3) (0,1,3,2 | 3,1) (1,2,3,2 | 1,2)
            const bool lf = vy[1] < vy[3];
            const double tlx[3] = {vx[0], lf ? vx[1] : vx[0], vx[1]}, tly[3] = {vy[0], lf ? vy[1] : vy[0], vy[1]};
            const double blx[3] = {vx[1], lf ? vx[2] : vx[1], vx[2]}, bly[3] = {vy[1], lf ? vy[2] : vy[1], vy[2]};
            const double trx[3] = {vx[0], lf ? vx[0] : vx[3], vx[3]}, try_[3] = {vy[0], lf ? vy[0] : vy[3], vy[3]};
            const double brx[3] = {vx[3], lf ? vx[3] : vx[2], vx[2]}, bry[3] = {vy[3], lf ? vy[3] : vy[2], vy[2]};
            const double ytv[3] = {vy[0], lf ? vy[1] : vy[3], lf ? vy[3] : vy[1]}, ybv[3] = {lf ? vy[1] : vy[3], lf ? vy[3] : vy[1], vy[2]};
            int ymin = RES_H, ymax = 0;
            _Pragma("unroll") for (int t = 0; t < 3; t++) {
                int from_y = q_round(ytv[t]), to_y = q_round(ybv[t]);
                if (from_y < 0) from_y = 0;
                if (to_y > RES_H) to_y = RES_H;
                uint32_t *tp = rp + 6 + 6 * t;
                if (from_y >= to_y) {
                    tp[0] = tp[1] = tp[2] = tp[3] = tp[4] = tp[5] = 0;
                    continue;
                }
                const double left_slope = (blx[t] - tlx[t]) / (bly[t] - tly[t]);
                const double right_slope = (brx[t] - trx[t]) / (bry[t] - try_[t]);
                tp[0] = (uint32_t)from_y;
                tp[1] = (uint32_t)to_y;
                tp[2] = (uint32_t)d2i_x86((tlx[t] + (0.5 + from_y - tly[t]) * left_slope + 0.5) * 0x10000);
                tp[3] = (uint32_t)d2i_x86(left_slope * 0x10000);
                tp[4] = (uint32_t)d2i_x86((trx[t] + (0.5 + from_y - try_[t]) * right_slope + 0.5) * 0x10000);
                tp[5] = (uint32_t)d2i_x86(right_slope * 0x10000);
                if (from_y < ymin) ymin = from_y;
                if (to_y > ymax) ymax = to_y;
            }
            if (ymin >= ymax) return;
            // bounding columns (conservative; the per-row spans decide coverage)
            double xmin = vx[0], xmax = vx[0];
            for (int i = 1; i < 4; i++) {
                if (vx[i] < xmin) xmin = vx[i];
                if (vx[i] > xmax) xmax = vx[i];
            }
            int bx1 = xmin < 1.0 ? 0 : (xmin > 63.0 ? 63 : (int)xmin - 1);
            int bx2 = xmax > 62.0 ? RES_W : (xmax < 0.0 ? 1 : (int)xmax + 2);
            if (bx1 < 0) bx1 = 0;
            if (bx2 > RES_W) bx2 = RES_W;
            if (ymin >= row1 || ymax <= row0) return;
            geom = (uint32_t)bx1 | ((uint32_t)ymin << 7) | ((uint32_t)(bx2 - bx1) << 14) | ((uint32_t)(ymax - ymin) << 21);
            basex_o = (uint32_t)lane;
            iy_o = (uint32_t)im.h;
            src_o = im.off;
            aux_o = cmd_aux((int)im.w, mirrored, false, opacity_to_io(opacity)) | (1u << 15);
        }
    }
    // p.fillRect(QRectF, QColor) as a draw command (lane-local)
    PG_DEV void cmd_fill_rect(const RectD &fr, uint32_t color, uint32_t &geom, uint32_t &src_o, uint32_t &aux_o) const {
        int x1 = q_round(fr.x), x2 = q_round(fr.x + fr.w), y1 = q_round(fr.y), y2 = q_round(fr.y + fr.h);
        if (x2 < x1) { const int t = x1; x1 = x2; x2 = t; }
        if (y2 < y1) { const int t = y1; y1 = y2; y2 = t; }
        if (x1 < 0) x1 = 0;
        if (y1 < 0) y1 = 0;
        if (x2 > RES_W) x2 = RES_W;
        if (y2 > RES_H) y2 = RES_H;
        if (x2 > x1 && y2 > y1 && !(y1 >= row1 || y2 <= row0)) {
            geom = (uint32_t)x1 | ((uint32_t)y1 << 7) | ((uint32_t)(x2 - x1) << 14) | ((uint32_t)(y2 - y1) << 21);
            src_o = color;
            aux_o = cmd_aux(1, false, true, 256) | (1u << 26);
        }
    }
    // draw_image BAG:877-913 for one drawable (lane-local); returns the image index or -1
    // returns the image index, -1 (nothing to draw) or IMG_FILL: draw_grid_obj paints the drawable's base rect with
    // color_for_type (BAG:455-481,915-919; use_monochrome_assets) -- the colour is returned in *fill_color
    static constexpr int IMG_FILL = -2;
    // pre: the descriptor request_desc(base_type, theme) returned earlier (a set-up section requests all its descriptors before
    // it needs the first), or null: looked up here.  >= 0: *desc is the image.
    PG_DEV int resolve_image(int base_type, int theme, float rotation, float tile_ratio, RectD &rect, uint32_t *fill_color, ImgDesc *desc, const ImgDesc *pre = nullptr) {
        const int img_type = Game::image_for_type(*this, base_type);
        if (img_type < 0) return -1;
        if (opt.use_monochrome_assets || img_type >= USE_ASSET_THRESHOLD) {
            if (img_type == SPACE) return -1;
            if (!opt.use_monochrome_assets || img_type >= 64 || !fill_color) {  // fassert(false) BAG:477 / fassert(type < kcubed) BAG:465
                fail(PGE_UNSUPPORTED_DRAW);
                return -1;
            }
            int th = theme;
            if (opt.restrict_themes && !Game::should_preserve_type_themes(img_type)) th = 0;
            const int k = 4, kcubed = 64, chunk = 64;
            int new_type = (29 * (img_type + 1)) % kcubed;
            new_type = (new_type + 19 * th) % kcubed;
            const uint32_t cr = (uint32_t)(chunk * (new_type / (k * k) + 1) - 1), cg = (uint32_t)(chunk * ((new_type / k) % k + 1) - 1), cb = (uint32_t)(chunk * (new_type % k + 1) - 1);
            *fill_color = 0xff000000u | (cr << 16) | (cg << 8) | cb;
            return IMG_FILL;
        }
        rect = Game::adjusted_image_rect(img_type, rect);
        const ImgDesc im = pre ? *pre : desc_for(img_type, theme);
        if (im.off == IMG_NONE) {
            fail(PGE_THEME);
            return -1;
        }
        (void)rotation;
        if (tile_ratio != 0) {
            fail(PGE_UNSUPPORTED_DRAW);
            return -1;
        }
        *desc = im;
        return 0;
    }

    // ---- separable geometry of grid cells -------------------------------------------------------------------
    // Every cell rect has the same size and an x that depends only on the cell column (y: only on the row), so for
    // cells whose image has the reference dimensions (ref_w x ref_h, the game's common sprite size) the Qt
    // stepping arithmetic is done once per column and once per row instead of once per cell:
    //   lanes 0..ncol-1 : column c -> tx1 | w<<8 (valid bit 16), basex      lanes 32..32+nrow-1 : row r -> ty1 | h<<8, srcy
    // (same formulas as cmd_image, evaluated on one axis).  ix / iy are wave-uniform.
    PG_DEV void axis_params(double pos, double len, int src_len, int limit, uint32_t &packed, uint32_t &base, int &step_out) const {
        const double sc = len / (double)src_len;
        const int step = (int)(65536 / sc);
        step_out = step;
        int t1 = q_round(pos), t2 = q_round(pos + len);
        if (t1 < 0) t1 = 0;
        if (t2 > limit) t2 = limit;
        int n = t2 - t1;
        packed = 0;
        base = 0;
        if (n <= 0) return;
        const uint32_t b = (uint32_t)((int)pg_ceil((t1 + 0.5 - pos) * step) - 1);
        const int end = (int)((b + (uint32_t)step * (uint32_t)(n - 1)) >> 16);
        if (end < 0 || end >= src_len) --n;
        if (n <= 0) return;
        packed = (uint32_t)t1 | ((uint32_t)n << 8) | (1u << 16);
        base = b;
    }
    PG_DEV void setup_tile_axes(int low_x, int ncol, int low_y, int nrow, int ref_w, int ref_h, int &ix_out, int &iy_out) {
        int ixu = 0, iyu = 0;
        {
            // wave-uniform steps (all cells share the rect size)
            const RectD r0 = get_screen_rect(0.0f, 1.0f, 1, 1, RENDER_EPS);
            ixu = (int)(65536 / (r0.w / (double)ref_w));
            iyu = (int)(65536 / (r0.h / (double)ref_h));
        }
        PG_R_LANES(l) {
            uint32_t packed = 0, base = 0;
            int step = 0;
            if (l < 32) {
                if (l < ncol) {
                    const RectD r = get_screen_rect((float)(low_x + l), 1.0f, 1, 1, RENDER_EPS);
                    axis_params(r.x, r.w, ref_w, RES_W, packed, base, step);
                }
            } else if (l - 32 < nrow) {
                const RectD r = get_screen_rect(0.0f, (float)(low_y + (l - 32) + 1), 1, 1, RENDER_EPS);
                axis_params(r.y, r.h, ref_h, RES_H, packed, base, step);
            }
            ax[l] = packed;
            ax[64 + l] = base;
        }
        PG_SYNC();
        ix_out = ixu;
        iy_out = iyu;
    }

    // ---- grid cells, pull form ---------------------------------------------------------------------------------
    // When every drawn cell of the window uses an image of the reference size with an unadjusted rect (the normal
    // case), the cells are not executed as ~150 separate 5x5 blits: each screen column knows the (at most two)
    // cell columns covering it and the source column they sample, each screen row likewise, and a pixel composites
    // its covering cells in the reference's x-major draw order: (c0,r0), (c0,r1), (c1,r0), (c1,r1).  Stage 1 does
    // (c0,r0) for all pixels with lane = screen column; the seam stages touch only the doubly covered columns/rows.
    // entry: valid<<31 | cell index<<12 | source coordinate
    // Grid object type -> image, once per frame: the asset tables live in HBM and resolving a cell costs two dependent
    // loads; a frame shows a few hundred cells of a handful of types.  Lane t resolves type t without raising errors
    // (types a level does not use may have no asset); anything unusual is left to the per-cell path (TYPE_SLOW).
    // lane l's request for build_type_table (issued ahead, render_env)
    PG_DEV ImgDesc request_type_desc(int type) {
        if constexpr (GameDrawsGrid<Game>::value) return request_desc(type, Game::theme_for_grid_obj(*this, type));
        else return ImgDesc{IMG_NONE, 0, 0, 0};
    }
    PG_DEV void build_type_table(PG_LANE_REF(const ImgDesc, pre_td)) {
        if constexpr (GameDrawsGrid<Game>::value) {
            const int ref_w = d.assets->ref_w, ref_h = d.assets->ref_h;
            PG_R_LANES(l) {
                uint32_t v = TYPE_SLOW, any = TYPE_SLOW, sz = 0;
                const int type = l;
                bool is_fill = false;
                if constexpr (GameHasGridFills<Game>::value) is_fill = Game::is_grid_fill(*this, type);
                if (!is_fill && !opt.use_monochrome_assets) {
                    const int img_type = Game::image_for_type(*this, type);
                    if (img_type < 0 || img_type == SPACE) {
                        v = any = CELL_NONE;
                    } else if (img_type < USE_ASSET_THRESHOLD) {
                        const ImgDesc imd = PG_LV(pre_td, l);  // = desc_for(img_type, Game::theme_for_grid_obj(*this, type))
                        if (imd.off != IMG_NONE) {
                            const RectD probe = {1.0, 2.0, 3.0, 5.0};
                            const RectD adj = Game::adjusted_image_rect(img_type, probe);
                            const bool same_rect = adj.x == probe.x && adj.y == probe.y && adj.w == probe.w && adj.h == probe.h;
                            if (same_rect && imd.off < 0x7ffffffu) {  // any size: the pull form keeps size classes (build_pull_tables)
                                any = imd.off | (imd.opaque ? (1u << 31) : 0u);
                                sz = ((uint32_t)imd.w << 16) | (uint32_t)imd.h;
                                if ((int)imd.w == ref_w && (int)imd.h == ref_h) v = any;
                            }
                        }
                    }
                }
                if (keep_typeimg) lds->typeimg[l] = v;
                typeany[l] = any;
                typesz[l] = sz;
            }
            PG_SYNC();
        }
    }
    // Cells of one frame may use images of several sizes (maze: 128x128 sand + 27x27 cheese; miner: 16x16 dirt, 70x70
    // boulders, 72x51 gems): a size class k has its own source-coordinate tables.  Class 0 is the reference size and lives
    // in ci / ri (covered<<31 | class-0 sample valid<<30 | cell index<<12 | source coordinate); classes 1..3 keep only
    // the source coordinates (srcx: column, 0xff = none; srcyw: row * image width, 0xffff = none).  Which cell covers a
    // pixel depends on the rect alone; whether that cell has a sample there is per class (Qt drops a last sample
    // that would fall outside the source).  Returns false when the frame needs the per-cell path: solid-colour
    // cells, adjusted rects, more than four sizes, three cells over one pixel.
    // the grid objects of window cells [q * 64 + l], q < 4 (x-major, as build_pull_tables walks them): its first round of reads, issued ahead
    PG_DEV void request_window_cells(int win_lx, int nx, int win_ly, int ny_full, PG_LANE_ARR_REF(int, cells0, 4)) const {
        const int ncell = nx * ny_full;
        const uint32_t ny_inv = (uint32_t)(((1u << 20) + (uint32_t)ny_full - 1u) / (uint32_t)ny_full);
        PG_R_LANES(l) {
            for (int q = 0; q < 4; q++) {
                const int cidx = q * 64 + l;
                const int cc = cidx < ncell ? cidx : 0;
                const int cx = (int)(((uint32_t)cc * ny_inv) >> 20);
                PG_LA(cells0, q, l) = get_obj(win_lx + cx, win_ly + (cc - cx * ny_full));
            }
        }
    }
    PG_DEV bool build_pull_tables(int win_lx, int nx, int win_ly, int ny_full, uint64_t &colseam, uint64_t &rowseam, uint64_t &rowany, bool &multi, int &nfill_out, PG_LANE_ARR_REF(const int, cells0, 4)) {
        nfill_out = 0;
        rowany = ~0ull;
        const int ref_w = d.assets->ref_w, ref_h = d.assets->ref_h;
        uint32_t *present = fb;  // scratch: the band buffer is idle during set-up
        uint32_t *span = fb + 64;
        PG_R_LANES(l) {
            present[l] = 0;
            lds->ci[0][l] = lds->ci[1][l] = lds->ri[0][l] = lds->ri[1][l] = 0;
            if constexpr (RenderLds::SIZE_CLASSES)
                for (int k = 0; k < 3; k++)
                    for (int sl = 0; sl < 2; sl++) {
                        lds->srcx[k][sl][l] = 0xffu;
                        lds->srcyw[k][sl][l] = 0xffffu;
                    }
        }
        PG_SYNC();
        // pixel spans of the cell columns (lanes 0..31) and rows (lanes 32..63): the rect alone decides them.  First of all (round 6): a cell in
        // a column or row without a pixel draws nothing whatever it holds, so a game that says so (GRID_RARELY_ON_SCREEN) does not look at it --
        // fruitbot's window is the 20 columns of its world plus the out-of-bounds wall columns either side of the screen, dodgeball's is a world
        // of SPACE -- and a frame without a grid cell on screen skips the rest of this set-up and the grid pass (fruitbot +11.5 %, dodgeball +2 %,
        // profiles/r06_call37_ab.txt; unconditionally the check cost the games with real grids 0.2-0.7 %)
        PG_R_LANES(l) {
            const bool col = l < 32;
            const int idx = col ? l : l - 32;
            uint32_t sp = 0;
            if (idx < (col ? nx : ny_full)) {
                const RectD r = col ? get_screen_rect((float)(win_lx + idx), 1.0f, 1, 1, RENDER_EPS) : get_screen_rect(0.0f, (float)(win_ly + idx + 1), 1, 1, RENDER_EPS);
                const double pos = col ? r.x : r.y, len = col ? r.w : r.h;
                int t1 = q_round(pos), t2 = q_round(pos + len);
                if (t1 < 0) t1 = 0;
                if (t2 > (col ? RES_W : RES_H)) t2 = col ? RES_W : RES_H;
                if (t2 > t1) sp = (uint32_t)t1 | ((uint32_t)(t2 - t1) << 8);
            }
            span[l] = sp;
        }
        PG_SYNC();
        // cell -> grid object type (kept in cellimg until the classes are known)
        const int ncell = nx * ny_full;
        const uint32_t ny_inv = (uint32_t)(((1u << 20) + (uint32_t)ny_full - 1u) / (uint32_t)ny_full);
        bool ok = true;
        PG_LANE_VAR(uint32_t, any_fill_l);
        PG_R_LANES(l) { PG_LV(any_fill_l, l) = 0; }
        for (int base4 = 0; base4 < ncell; base4 += 256) {
            PG_LANE_ARR(int, types, 4);
            PG_R_LANES(l) {  // the grid reads of four chunks in flight together (the first round was requested ahead)
                for (int q = 0; q < 4; q++) {
                    const int cidx = base4 + q * 64 + l;
                    const int cc = cidx < ncell ? cidx : 0;
                    const int cx = (int)(((uint32_t)cc * ny_inv) >> 20);
                    PG_LA(types, q, l) = base4 == 0 ? PG_LA(cells0, q, l) : get_obj(win_lx + cx, win_ly + (cc - cx * ny_full));
                }
            }
            for (int q = 0; q < 4 && base4 + q * 64 < ncell; q++) {
                const uint64_t bad = PG_BALLOT(l, ({
                                                   const int cidx = base4 + q * 64 + l;
                                                   bool b = false;
                                                   if (cidx < ncell) {
                                                       bool on_screen = true;
                                                       if constexpr (GameGridRarelyOnScreen<Game>::value) {
                                                           const int cx_ = (int)(((uint32_t)cidx * ny_inv) >> 20);
                                                           on_screen = span[cx_] != 0 && span[32 + cidx - cx_ * ny_full] != 0;
                                                       }
                                                       const int type = on_screen ? PG_LA(types, q, l) : (int)SPACE;
                                                       uint8_t v = CELL8_NONE;
                                                       bool is_fill = false;
                                                       if constexpr (GameHasGridFills<Game>::value) is_fill = on_screen && Game::is_grid_fill(*this, type);
                                                       if (is_fill) {
                                                           PG_LV(any_fill_l, l) = 1;
                                                           v = CELL8_FILL;
                                                       } else if (type >= 0 && type < 64) {
                                                           const uint32_t tv = typeany[type];
                                                           if (tv == TYPE_SLOW) b = true;
                                                           else if (tv != CELL_NONE) {
                                                               v = (uint8_t)type;
                                                               present[type] = 1;  // several lanes may store the same 1
                                                           }
                                                       } else if (type != INVALID_OBJ && type != SPACE) {
                                                           b = true;
                                                       }
                                                       lds->cellimg[cidx] = v;
                                                   }
                                                   b;
                                               }));
                ok = ok && bad == 0;
            }
        }
        if (!ok) return false;
        PG_SYNC();
        if (PG_BALLOT(l, present[l] != 0 || PG_LV(any_fill_l, l) != 0) == 0) {  // no cell with an image or a fill on screen: an empty pull form
            colseam = rowseam = rowany = 0;
            multi = false;
            return true;
        }
        // size classes of the types on screen
        PG_LANE_VAR(uint32_t, key);
        PG_LANE_VAR(uint32_t, cls);
        const uint32_t key0 = ((uint32_t)ref_w << 16) | (uint32_t)ref_h;
        PG_R_LANES(l) {
            PG_LV(key, l) = present[l] ? typesz[l] : key0;
            PG_LV(cls, l) = 0;
        }
        uint32_t ckey[4] = {key0, key0, key0, key0};
        int ncls = 1;
        uint64_t todo = PG_BALLOT(l, PG_LV(key, l) != key0);
        while (todo) {
            const int leader = pg_ctz64(todo);
            const uint32_t kk = PG_READLANE(key, leader);
            if (ncls >= 4 || (kk >> 16) > 255u || (kk & 0xffffu) > 255u) return false;
            const uint64_t same = PG_BALLOT(l, PG_LV(key, l) == kk);
            PG_R_LANES(l) {
                if ((same >> l) & 1ull) PG_LV(cls, l) = (uint32_t)ncls;
            }
            ckey[ncls++] = kk;
            todo &= ~same;
        }
        multi = ncls > 1;
        if (multi && !RenderLds::SIZE_CLASSES) return false;  // (a second image size in a single-size game's window: the per-cell path)
        PG_R_LANES(l) {
            const uint32_t tv = typeany[l];
            if (tv != CELL_NONE && tv != TYPE_SLOW) typeany[l] = tv | (PG_LV(cls, l) << 27);
        }
        PG_SYNC();
        // bit r: cell row r of the window holds a cell with an image.  Cells are x-major (index = column * ny_full + row): every lane marks
        // the rows of its cells, one OR over the wave folds them (round 6; a scalar fold of each 64-cell chunk's ballot, column by column, was
        // ~400 scalar instructions a frame)
        uint32_t cellrows = 0;
        {
            PG_LANE_VAR(uint32_t, rowbits);
            PG_R_LANES(l) { PG_LV(rowbits, l) = 0; }
            for (int base = 0; base < ncell; base += 64) {
                PG_R_LANES(l) {
                    const int cidx = base + l;
                    if (cidx < ncell) {
                        const uint8_t t = lds->cellimg[cidx];
                        const int cx = (int)(((uint32_t)cidx * ny_inv) >> 20);
                        if (t != CELL8_NONE && t != CELL8_FILL) PG_LV(rowbits, l) |= 1u << (cidx - cx * ny_full);  // (the cell keeps its type id: pull_fetch goes through typeany)
                    }
                }
            }
            cellrows = PG_WAVE_OR(rowbits);
        }
        if constexpr (GameHasGridFills<Game>::value) {
            // Solid-colour cells (chaser's orbs) become fill commands kept behind the cell table.  They may run after the
            // image cells only if no neighbouring cell's rect reaches the pixels they paint (their own cell draws
            // nothing else): then no draw order between them and anything else in the grid pass is observable.
            int nfill = 0;
            const int fill_cap = 256;  // RenderLdsT::fillcmd (more solid-colour cells on screen: the per-cell path draws the frame)
            for (int base = 0; base < ncell; base += 64) {
                PG_LANE_VAR(uint32_t, fg);
                PG_LANE_VAR(uint32_t, fcol);
                const uint64_t conflict = PG_BALLOT(l, ({
                                                        bool bad = false;
                                                        const int cidx = base + l;
                                                        PG_LV(fg, l) = 0;
                                                        PG_LV(fcol, l) = 0;
                                                        if (cidx < ncell && lds->cellimg[cidx] == CELL8_FILL) {
                                                            const int cx = (int)(((uint32_t)cidx * ny_inv) >> 20);
                                                            const int cy = cidx - cx * ny_full;
                                                            const int x = win_lx + cx, y = win_ly + cy;
                                                            const RectD cell = get_screen_rect((float)x, (float)(y + 1), 1, 1, RENDER_EPS);
                                                            RectD fr;
                                                            uint32_t color = 0, g = 0, sc = 0, au = 0;
                                                            Game::grid_fill(*this, get_obj(x, y), cell, fr, color);
                                                            cmd_fill_rect(fr, color, g, sc, au);
                                                            if (g != 0) {
                                                                const int x1 = (int)(g & 0x7fu), y1 = (int)((g >> 7) & 0x7fu), x2 = x1 + (int)((g >> 14) & 0x7fu), y2 = y1 + (int)((g >> 21) & 0x7fu);
                                                                for (int dlt = -1; dlt <= 1; dlt += 2) {
                                                                    const uint32_t sx = (cx + dlt >= 0 && cx + dlt < nx) ? span[cx + dlt] : 0u;
                                                                    const uint32_t sy = (cy + dlt >= 0 && cy + dlt < ny_full) ? span[32 + cy + dlt] : 0u;
                                                                    const int ax1 = (int)(sx & 0xffu), ax2 = ax1 + (int)(sx >> 8), ay1 = (int)(sy & 0xffu), ay2 = ay1 + (int)(sy >> 8);
                                                                    bad = bad || (ax2 > ax1 && ax1 < x2 && ax2 > x1) || (ay2 > ay1 && ay1 < y2 && ay2 > y1);
                                                                }
                                                            }
                                                            PG_LV(fg, l) = g;
                                                            PG_LV(fcol, l) = sc;
                                                            lds->cellimg[cidx] = CELL8_NONE;
                                                        }
                                                        bad;
                                                    }));
                if (conflict) return false;
                const uint64_t vis = PG_BALLOT(l, PG_LV(fg, l) != 0);
                const int cnt = pg_popc64(vis);
                if (nfill + cnt > fill_cap) return false;
                PG_R_LANES(l) {
                    if ((vis >> l) & 1ull) {
                        const int slot = nfill + pg_popc64(vis & pg_mask_lt(l));
                        lds->fillcmd[2 * slot] = PG_LV(fg, l);
                        lds->fillcmd[2 * slot + 1] = PG_LV(fcol, l);
                    }
                }
                nfill += cnt;
            }
            nfill_out = nfill;
        }
        PG_LANE_VAR(uint32_t, over);
        PG_R_LANES(l) { PG_LV(over, l) = 0; }
        for (int k = 0; k < ncls; k++) {
            const int cw = (int)(ckey[k] >> 16), ch = (int)(ckey[k] & 0xffffu);
            PG_R_LANES(l) {
                const bool col = l < 32;
                const int idx = col ? l : l - 32;
                const uint32_t sp = span[l];
                const int t1 = (int)(sp & 0xffu), n0 = (int)(sp >> 8);
                if (n0 > 0) {
                    const RectD r = col ? get_screen_rect((float)(win_lx + idx), 1.0f, 1, 1, RENDER_EPS) : get_screen_rect(0.0f, (float)(win_ly + idx + 1), 1, 1, RENDER_EPS);
                    const double pos = col ? r.x : r.y, len = col ? r.w : r.h;
                    const int src_len = col ? cw : ch;
                    const int step = (int)(65536 / (len / (double)src_len));  // as cmd_image, one axis
                    const uint32_t b = (uint32_t)((int)pg_ceil((t1 + 0.5 - pos) * step) - 1);
                    int n = n0;
                    const int end = (int)((b + (uint32_t)step * (uint32_t)(n - 1)) >> 16);
                    if (end < 0 || end >= src_len) --n;
                    const uint32_t s1 = idx >= 1 ? span[l - 1] : 0u, s2 = idx >= 2 ? span[l - 2] : 0u;
                    for (int j = 0; j < n0; j++) {
                        const int p = t1 + j;
                        // cells before this one in draw order that also cover the pixel take the lower slot
                        const int slot = ((p >= (int)(s1 & 0xffu) && p < (int)((s1 & 0xffu) + (s1 >> 8))) ? 1 : 0) + ((p >= (int)(s2 & 0xffu) && p < (int)((s2 & 0xffu) + (s2 >> 8))) ? 1 : 0);
                        if (slot >= 2) {
                            PG_LV(over, l) = 1;
                            continue;
                        }
                        const bool sv = j < n;
                        const uint32_t sc = (b + (uint32_t)j * (uint32_t)step) >> 16;
                        if (k == 0) {
                            const uint32_t e = (1u << 31) | (sv ? (1u << 30) : 0u) | ((uint32_t)idx << 12) | (sv ? (sc & 0xfffu) : 0u);
                            if (col) lds->ci[slot][p] = e;
                            else lds->ri[slot][p] = e;
                        } else if constexpr (RenderLds::SIZE_CLASSES) {
                            if (col) lds->srcx[k - 1][slot][p] = (uint8_t)(sv ? sc : 0xffu);
                            else lds->srcyw[k - 1][slot][p] = (uint16_t)(sv ? sc * (uint32_t)cw : 0xffffu);
                        }
                    }
                }
            }
        }
        PG_SYNC();
        if (PG_BALLOT(l, PG_LV(over, l) != 0) != 0) return false;
        colseam = PG_BALLOT(l, (lds->ci[1][l] >> 31) != 0);
        rowseam = PG_BALLOT(l, (lds->ri[1][l] >> 31) != 0);
        {
            // screen rows that some cell with an image reaches: a cell row with nothing to draw (sky) costs its pixels nothing
            rowany = PG_BALLOT(l, ({
                                   const uint32_t cr = cellrows, e0 = lds->ri[0][l], e1 = lds->ri[1][l];
                                   ((e0 >> 31) != 0 && ((cr >> ((e0 >> 12) & 0x1fu)) & 1u) != 0) || ((e1 >> 31) != 0 && ((cr >> ((e1 >> 12) & 0x1fu)) & 1u) != 0);
                               }));
        }
        PG_R_LANES(l) {
            if ((colseam >> l) & 1ull) lds->seamcols[pg_popc64(colseam & pg_mask_lt(l))] = (uint8_t)l;
        }
        PG_SYNC();
        return true;
    }
    // texel of the cell under one pixel for the column slot sc / row slot sr, branch-free: lanes without a cell fetch
    // atlas word 0 and report no hit (per-lane `if`s around memory operations cost exec-mask juggling on the CU's
    // single scalar unit).  MULTI: the frame has cells of more than one image size.
    // (REG: the tables of a register-table rasterizer -- called from lane sections every lane runs, `l` the calling lane)
    template <bool REG>
    PG_DEV uint32_t tab_typeany(int l, uint32_t type) const {
        if constexpr (REG) return PG_SHFL(tr_ty, l, (int)type);
        else return typeany[type];
    }
    template <bool REG>
    PG_DEV uint32_t tab_ci1(int l, int x) const {
        if constexpr (REG) return PG_SHFL(tr_ci1, l, x);
        else return lds->ci[1][x];
    }
    template <bool REG>
    PG_DEV uint32_t tab_ri(int l, int slot, int y) const {
        if constexpr (REG) return slot ? PG_SHFL(tr_ri1, l, y) : PG_SHFL(tr_ri0, l, y);
        else return lds->ri[slot][y];
    }
    template <bool MULTI, bool REG = false>
    PG_DEV bool pull_fetch(uint32_t ce, uint32_t re, int sc, int sr, int x, int y, int ny_full, int ref_w, uint32_t &tex, bool &opaque, int lane = 0) const {
        const uint32_t both = ce & re;
        const bool covered = (both >> 31) != 0;
        const uint32_t ct = lds->cellimg[((ce >> 12) & 0x1fu) * (uint32_t)ny_full + ((re >> 12) & 0x1fu)];
        const uint32_t tv_ = tab_typeany<REG>(lane, ct & 63u);  // (read unconditionally: a load inside a ?: arm becomes a branch around it)
        const uint32_t cell = ct < 64u ? tv_ : CELL_NONE;
        opaque = (cell >> 31) != 0;
        const bool v0 = ((both >> 30) & 1u) != 0;
        uint32_t rel = (re & 0xfffu) * (uint32_t)ref_w + (ce & 0xfffu);
        bool hit = covered && cell != CELL_NONE;
        if constexpr (MULTI && RenderLds::SIZE_CLASSES) {
            const uint32_t k = (cell >> 27) & 3u;  // CELL_NONE reads class 3: in bounds, never a hit
            const uint32_t kk = k ? k - 1u : 0u;
            const uint32_t sx1 = lds->srcx[kk][sc][x], sy1 = lds->srcyw[kk][sr][y];
            hit = hit && (k ? (sx1 != 0xffu && sy1 != 0xffffu) : v0);
            rel = k ? sy1 + sx1 : rel;
        } else {
            hit = hit && v0;
        }
        tex = d.pixels[hit ? (cell & 0x7ffffffu) + rel : 0u];
        return hit;
    }
    // Stages (c0, r0) and (c0, r1) of the pull form -- every pixel's first covering cell column, its one or two covering cell rows -- for a
    // frame whose cell images share one size, ROW by row with the cell lookup hoisted: the band's 16 screen rows in order, each a scalar
    // decode of its row entry (covering cell row, source row); the lanes (lane = screen column) look their cell up only where the cell row
    // changes (~5 times a band), fetch one texel per row at (scalar row base) + (lane offset), all 16 fetches in flight, and store (opaque
    // images: ground, walls, crates) or blend them after one wait.  Row slot 1 -- the rows a second cell row covers as well -- follows for
    // the few rows that have one.  (Round 6.  The per-pixel lookup chain of rounds 2-5 -- two LDS reads and ~25 vector instructions per
    // pixel row -- was 42 % of the kernel's vector instructions, profiles/r06_valu_by_phase.txt; two cell-row-major forms tried first spent
    // what they saved on scalar run bookkeeping and on one dependent round trip per cell row, profiles/r06_call3_ab.txt, r06_raster_ablation.txt.)
    // MULTI: the frame shows cell images of several sizes (maze: 128 x 128 sand + 27 x 27 cheese; miner: three sizes): a lane's cell then
    // has a size class k, its source column comes from that class's column table (looked up with the cell, once per cell row) and its source
    // row from the class's row table (one LDS read per row and lane instead of the scalar decode); Qt may have dropped a class's last
    // sample of a row or column, per class.  Measured slower than the per-pixel form for these frames (draw_tiles_pull), so it is the
    // A/B variant, not the default.
    template <bool MULTI, bool REG = false>
    PG_DEV void rows_pass(int ny_full, int ref_w, uint32_t band_any, uint32_t band_seam) {
        static_assert(BAND_ROWS <= 16, "the band's row entries live in lanes 0..15 (slot 0) and 16..31 (slot 1)");
        // (plain lane sections: the empty asm a PG_R_LANES section takes its lane id through makes the compiler wait for every texel in
        // flight before the next row's section, i.e. one fetch at a time)
        PG_LANE_VAR(uint32_t, riv);
        PG_LANE_VAR(uint32_t, ce);
        PG_FOR_LANES(l) {
            if constexpr (REG) {
                const uint32_t e0 = PG_SHFL(tr_ri0, l, row0 + (l & (BAND_ROWS - 1))), e1 = PG_SHFL(tr_ri1, l, row0 + (l & (BAND_ROWS - 1)));
                PG_LV(riv, l) = ((l >> 4) & 1) ? e1 : e0;
                PG_LV(ce, l) = PG_LV(tr_ci0, l);
            } else {
                PG_LV(riv, l) = lds->ri[(l >> 4) & 1][row0 + (l & (BAND_ROWS - 1))];
                PG_LV(ce, l) = lds->ci[0][l];
            }
        }
        _Pragma("nounroll") for (int sr = 0; sr < 2; sr++) {  // (one copy of the 16-row body: unrolled, the two copies' scalars spilled into each other)
            const uint32_t rows = sr ? band_seam : band_any;
            if (rows == 0) continue;
            PG_LANE_ARR(uint32_t, tex, BAND_ROWS);  // (defined up front: a register the allocator reuses as a temporary is a wait on a fetch in flight)
            PG_LANE_VAR(uint32_t, hm);     // bit j: this lane draws row j
            PG_LANE_VAR(uint32_t, cboff);  // byte offset of the lane's cell image of the current cell row + its source column
            PG_LANE_VAR(uint32_t, hbit);   // 1: the lane has a cell with an image in the current cell row
            PG_LANE_VAR(uint32_t, kcls);   // MULTI: the size class of that image
            PG_FOR_LANES(l) {
                PG_LV(hm, l) = 0;
                PG_LV(cboff, l) = 0;
                PG_LV(hbit, l) = 0;
                PG_LV(kcls, l) = 0;
                for (int j = 0; j < BAND_ROWS; j++) PG_LA(tex, j, l) = 0;
            }
            int prev_cy = -1;
            bool translucent = false;  // some cell row of the band shows a translucent image: its rows are composited, not stored
            uint32_t drawn = 0;        // rows with a fetch in flight
            _Pragma("unroll") for (int j = 0; j < BAND_ROWS; j++) {
                if (((rows >> j) & 1u) == 0) continue;
                const uint32_t e = PG_READLANE(riv, sr * 16 + j);
                if (MULTI ? (e >> 31) == 0 : (e >> 30) != 3u) continue;  // not covered in this slot (one size: or Qt dropped the row's sample)
                const int cy = (int)((e >> 12) & 0x1fu);
                if (cy != prev_cy) {
                    prev_cy = cy;
                    PG_LANE_VAR(uint32_t, tl);
                    PG_FOR_LANES(l) {
                        const uint32_t c = PG_LV(ce, l);
                        const uint32_t ct = lds->cellimg[((c >> 12) & 0x1fu) * (uint32_t)ny_full + (uint32_t)cy];
                        const uint32_t tv = tab_typeany<REG>(l, ct & 63u);
                        const uint32_t cell = ct < 64u ? tv : CELL_NONE;
                        bool h = (c >> 31) != 0 && cell != CELL_NONE;
                        uint32_t col = c & 0xfffu, k = 0;
                        if constexpr (MULTI && RenderLds::SIZE_CLASSES) {
                            k = (cell >> 27) & 3u;  // (CELL_NONE reads class 3: in bounds, never a hit)
                            const uint32_t sx1 = lds->srcx[k ? k - 1u : 0u][0][l];
                            h = h && (k ? sx1 != 0xffu : ((c >> 30) & 1u) != 0);
                            col = k ? sx1 : col;
                        } else {
                            h = h && ((c >> 30) & 1u) != 0;  // Qt kept the column's sample
                        }
                        PG_LV(cboff, l) = h ? ((cell & 0x7ffffffu) + col) << 2 : 0u;  // (a lane without a cell fetches a word of the atlas' first row and draws nothing)
                        PG_LV(hbit, l) = h ? 1u : 0u;
                        PG_LV(kcls, l) = k;
                        PG_LV(tl, l) = (h && (cell >> 31) == 0) ? 1u : 0u;
                    }
                    if (PG_BALLOT(l, PG_LV(tl, l) != 0) != 0) translucent = true;
                }
                if constexpr (MULTI && RenderLds::SIZE_CLASSES) {
                    const uint32_t off0 = (e & 0xfffu) * (uint32_t)ref_w;  // class 0: wave-uniform source row
                    const bool ok0 = ((e >> 30) & 1u) != 0;
                    const char *base = reinterpret_cast<const char *>(d.pixels);
                    PG_FOR_LANES(l) {
                        const uint32_t k = PG_LV(kcls, l);
                        const uint32_t syw = lds->srcyw[k ? k - 1u : 0u][sr][row0 + j];
                        const bool h = PG_LV(hbit, l) != 0 && (k ? syw != 0xffffu : ok0);
                        const uint32_t boff = h ? PG_LV(cboff, l) + ((k ? syw : off0) << 2) : 0u;
                        PG_LA(tex, j, l) = *reinterpret_cast<const uint32_t *>(base + boff);
                        PG_LV(hm, l) |= (h ? 1u : 0u) << j;
                    }
                } else {
                    const char *rowp = reinterpret_cast<const char *>(d.pixels + (e & 0xfffu) * (uint32_t)ref_w);  // wave-uniform
                    PG_FOR_LANES(l) {
                        PG_LA(tex, j, l) = *reinterpret_cast<const uint32_t *>(rowp + PG_LV(cboff, l));  // scalar base + 32-bit lane offset
                        PG_LV(hm, l) |= PG_LV(hbit, l) << j;
                    }
                }
                drawn |= 1u << j;
            }
            if (drawn == 0) continue;
            dma_join();  // (the band's background rows, requested before these texels)
            if (!translucent) {
                PG_FOR_LANES(l) {
                    _Pragma("unroll") for (int j = 0; j < BAND_ROWS; j++)
                        if ((PG_LV(hm, l) >> j) & 1u) fb[j * RES_W + l] = PG_LA(tex, j, l);  // this lane owns the column
                }
            } else {
                _Pragma("unroll") for (int j = 0; j < BAND_ROWS; j++) {
                    if (((drawn >> j) & 1u) == 0) continue;
                    PG_FOR_LANES(l) {
                        uint32_t *dp = &fb[j * RES_W + l];
                        *dp = blend(((PG_LV(hm, l) >> j) & 1u) ? PG_LA(tex, j, l) : 0u, *dp, 256, 255u);  // (a transparent texel leaves the pixel as it is: BYTE_MUL(dst, 255) == dst)
                    }
                }
            }
            PG_SYNC();
        }
    }
    // the same two stages pixel by pixel, for a frame with cell images of several sizes (maze, miner): a band of fetches in flight
    template <bool MULTI>
    PG_DEV void rows_pass_by_pixel(int ny_full, int ref_w, uint32_t band_any, uint32_t band_seam) {
        // stage 1: (c0, r0), lane = screen column
        for (int yb = row0; yb < row1; yb += WIDE_ROWS) {
            if (((band_any >> (yb - row0)) & ((1u << WIDE_ROWS) - 1u)) == 0) continue;
            PG_R_LANES(l) {
                const uint32_t ce = lds->ci[0][l];
                uint32_t tex[WIDE_ROWS];
                bool hit[WIDE_ROWS], opq[WIDE_ROWS];
                _Pragma("unroll") for (int j = 0; j < WIDE_ROWS; j++) {
                    opq[j] = false;
                    tex[j] = 0;
                    hit[j] = pull_fetch<MULTI>(ce, lds->ri[0][yb + j], 0, 0, l, yb + j, ny_full, ref_w, tex[j], opq[j]);
                }
                dma_join();  // (the band's background rows, requested before these texels)
                _Pragma("unroll") for (int j = 0; j < WIDE_ROWS; j++) {
                    uint32_t *dp = &fb[(yb + j - row0) * RES_W + l];  // this lane owns the pixel
                    const uint32_t old = *dp;
                    const uint32_t over = opq[j] ? tex[j] : blend(tex[j], old, 256, 255u);
                    *dp = hit[j] ? over : old;
                }
            }
            PG_SYNC();
        }
        // stage 2: (c0, r1) on the doubly covered rows only, four of them per round; stage 4 walks the same rows
        for (uint32_t m = band_seam; m != 0;) {
            int ys[4], cnt = 0;
            _Pragma("unroll") for (int q = 0; q < 4; q++) {
                ys[q] = row0;
                if (m != 0) {
                    ys[q] = row0 + pg_ctz64((uint64_t)m);
                    m &= m - 1u;
                    cnt = q + 1;
                }
            }
            PG_R_LANES(l) {
                const uint32_t ce = lds->ci[0][l];
                uint32_t tex[4];
                bool hit[4], opq[4];
                _Pragma("unroll") for (int q = 0; q < 4; q++) {
                    opq[q] = false;
                    tex[q] = 0;
                    hit[q] = false;
                    if (q < cnt) hit[q] = pull_fetch<MULTI>(ce, lds->ri[1][ys[q]], 0, 1, l, ys[q], ny_full, ref_w, tex[q], opq[q]);
                }
                dma_join();
                _Pragma("unroll") for (int q = 0; q < 4; q++) {
                    if (q < cnt) {
                        uint32_t *dp = &fb[(ys[q] - row0) * RES_W + l];
                        const uint32_t old = *dp;
                        const uint32_t over = opq[q] ? tex[q] : blend(tex[q], old, 256, 255u);
                        *dp = hit[q] ? over : old;
                    }
                }
            }
            PG_SYNC();
        }
    }
    template <bool MULTI, int NJ, bool REG = false>
    PG_DEV void seam_cols_round(int base, int npx, uint32_t inv, int nseam, int ny_full, int ref_w) {
        PG_R_LANES(l) {
            uint32_t tex[NJ];
            int fbi[NJ];
            bool opq[NJ];
            _Pragma("unroll") for (int j = 0; j < NJ; j++) {
                const int p = base + j * 64 + l;
                const bool in = p < npx;
                const int pc = in ? p : 0;
                const int yl = (int)(((uint32_t)pc * inv) >> 20);
                const int x = (int)lds->seamcols[pc - yl * nseam];
                const bool hit = pull_fetch<MULTI, REG>(tab_ci1<REG>(l, x), tab_ri<REG>(l, 0, row0 + yl), 1, 0, x, row0 + yl, ny_full, ref_w, tex[j], opq[j], l) && in;
                fbi[j] = hit ? yl * RES_W + x : BAND_ROWS * RES_W + l;  // masked-off lanes use the dump row
            }
            dma_join();
            _Pragma("unroll") for (int j = 0; j < NJ; j++) {
                const uint32_t old = fb[fbi[j]];
                fb[fbi[j]] = opq[j] ? tex[j] : blend(tex[j], old, 256, 255u);
            }
        }
        PG_SYNC();
    }
    template <bool MULTI, bool REG = false>
    PG_DEV void draw_tiles_pull(int ny_full, uint64_t colseam, uint64_t rowseam, uint64_t rowany, int ref_w) {
        static_assert(!(MULTI && REG), "register tables: frames with one cell image size");
        const int nseam = pg_popc64(colseam);
        const uint32_t band_any = (uint32_t)((rowany >> row0) & ((1ull << BAND_ROWS) - 1ull));
        if (band_any == 0) return;  // no cell with an image reaches these rows (sky)
        const uint32_t band_seam = (uint32_t)((rowseam >> row0) & ((1ull << BAND_ROWS) - 1ull)) & band_any;
        // stages 1 and 2: (c0, r0), and (c0, r1) on the doubly covered rows
        // (frames with several cell image sizes keep the per-pixel form: the row-major one with per-lane row tables measured 3-10 % SLOWER on
        // maze, miner, climber, jumper, caveflyer, profiles/r06_call17_ab16.txt; PROCGEN_AMD_DEBUG & 2097152 selects it for the A/B)
        if constexpr (REG) {
            rows_pass<false, true>(ny_full, ref_w, band_any, band_seam);
        } else {
            if (MULTI && !PG_FDBG(d, 2097152)) rows_pass_by_pixel<MULTI>(ny_full, ref_w, band_any, band_seam);
            else rows_pass<MULTI>(ny_full, ref_w, band_any, band_seam);
        }
        if (nseam == 0) return;
        // stage 3: (c1, r0): the doubly covered columns x the band's rows, laid out linearly over the lanes, one, two or four pixels per lane
        // and round (coinrun shows two or three such columns, 48 pixels a band: a four-deep round spent three quarters of its instructions
        // on lanes without a pixel, profiles/r06_raster_ablation.txt)
        {
            const uint32_t inv = (uint32_t)(((1u << 20) + (uint32_t)nseam - 1u) / (uint32_t)nseam);
            const int npx = nseam * BAND_ROWS;
            if (npx <= 64) seam_cols_round<MULTI, 1, REG>(0, npx, inv, nseam, ny_full, ref_w);
            else if (npx <= 128) seam_cols_round<MULTI, 2, REG>(0, npx, inv, nseam, ny_full, ref_w);
            else
                for (int base = 0; base < npx; base += 256) seam_cols_round<MULTI, 4, REG>(base, npx, inv, nseam, ny_full, ref_w);
        }
        // stage 4: (c1, r1): doubly covered columns x doubly covered rows; lane = seam column, four rows per round
        for (uint32_t m = band_seam; m != 0;) {
            int ys[4], cnt = 0;
            _Pragma("unroll") for (int q = 0; q < 4; q++) {
                ys[q] = row0;
                if (m != 0) {
                    ys[q] = row0 + pg_ctz64((uint64_t)m);
                    m &= m - 1u;
                    cnt = q + 1;
                }
            }
            PG_R_LANES(l) {
                const bool in = l < nseam;
                const int x = (int)lds->seamcols[in ? l : 0];
                const uint32_t ce = tab_ci1<REG>(l, x);
                uint32_t tex[4];
                int fbi[4];
                bool opq[4];
                _Pragma("unroll") for (int q = 0; q < 4; q++) {
                    opq[q] = false;
                    tex[q] = 0;
                    fbi[q] = BAND_ROWS * RES_W + l;
                    if (q < cnt) {
                        const bool hit = pull_fetch<MULTI, REG>(ce, tab_ri<REG>(l, 1, ys[q]), 1, 1, x, ys[q], ny_full, ref_w, tex[q], opq[q], l) && in;
                        fbi[q] = hit ? (ys[q] - row0) * RES_W + x : BAND_ROWS * RES_W + l;
                    }
                }
                dma_join();
                _Pragma("unroll") for (int q = 0; q < 4; q++) {
                    if (q < cnt) {
                        const uint32_t old = fb[fbi[q]];
                        fb[fbi[q]] = opq[q] ? tex[q] : blend(tex[q], old, 256, 255u);
                    }
                }
            }
            PG_SYNC();
        }
    }

    // ---- command execution ----------------------------------------------------------------------------------
    PG_DEV static DrawCmd unpack(uint32_t geom, uint32_t basex, uint32_t srcy0, uint32_t ix, uint32_t iy, uint32_t src, uint32_t aux, uint32_t e0 = 0, uint32_t e1 = 0) {
        DrawCmd c;
        c.e0 = e0;
        c.e1 = e1;
        c.tx1 = (int)(geom & 0x7fu);
        c.ty1 = (int)((geom >> 7) & 0x7fu);
        c.w = (int)((geom >> 14) & 0x7fu);
        c.h = (int)((geom >> 21) & 0x7fu);
        c.basex = basex;
        c.srcy0 = srcy0;
        c.ix = ix;
        c.iy = iy;
        c.src = src;
        c.aux = aux;
        return c;
    }
    PG_DEV static uint32_t blend(uint32_t sp, uint32_t dst, int io, uint32_t ca) {
        if (io != 256) sp = byte_mul(sp, ca);
        return sp + byte_mul(dst, 255u - (sp >> 24));
    }
    // one command of any size: every lane takes up to 8 pixels per round, all 8 texel fetches before the blends.
    // Wide commands (backgrounds) map lane -> column and walk 8 rows per round; narrower ones split a linear pixel
    // index with a reciprocal multiply.
    PG_DEV void exec_large(const DrawCmd &c) {
        if (cmd_fill(c.aux)) {
            RectD r = {(double)c.tx1, (double)c.ty1, (double)c.w, (double)c.h};
            exec_fill(r, c.src);
            return;
        }
        const uint32_t *src = (GEN && cmd_bgcanvas(c.aux)) ? d.gen_bg + (size_t)env * GEN_BG_WORDS : d.pixels + c.src;
        const int sw = cmd_src_w(c.aux);
        const bool mirrored = cmd_mirrored(c.aux);
        const bool opaque = cmd_opaque(c.aux);
        const int io = cmd_alpha(c.aux);
        const uint32_t ca = (uint32_t)((io * 255) >> 8);
        const int y0 = c.ty1 > row0 ? c.ty1 : row0;
        const int y1 = (c.ty1 + c.h) < row1 ? (c.ty1 + c.h) : row1;
        if (c.w > 32) {
            for (int yb = y0; yb < y1; yb += WIDE_ROWS) {  // a whole band of texel fetches in flight
                const int last = y1 - 1;  // rows past the end re-sample the last row into the dump row (no per-row branches)
                PG_R_LANES(l) {
                    const bool in = l < c.w;
                    const int lc = in ? l : 0;
                    uint32_t tex[WIDE_ROWS];
                    _Pragma("unroll") for (int j = 0; j < WIDE_ROWS; j++) {
                        const int y = (yb + j) < last ? (yb + j) : last;
                        int sxp, syp;
                        sample_xy(c, lc, y, sw, sxp, syp);
                        tex[j] = src[syp * sw + (mirrored ? (sw - 1 - sxp) : sxp)];
                    }
                    dma_join();
                    _Pragma("unroll") for (int j = 0; j < WIDE_ROWS; j++) {
                        const bool ok = in && (yb + j) <= last;
                        uint32_t *dp = &fb[ok ? ((yb + j - row0) * RES_W + c.tx1 + l) : (BAND_ROWS * RES_W + l)];
                        *dp = opaque ? tex[j] : blend(tex[j], *dp, io, ca);
                    }
                }
            }
        } else {
            const int npix = c.w * (y1 - y0);
            const uint32_t inv = (uint32_t)(((1u << 20) + (uint32_t)c.w - 1u) / (uint32_t)c.w);  // p / w == (p * inv) >> 20 for p < 4096
            for (int base = 0; base < npix; base += 512) {
                PG_R_LANES(l) {
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
                            int sxp, syp;
                            sample_xy(c, px, y, sw, sxp, syp);
                            tex[j] = src[syp * sw + (mirrored ? (sw - 1 - sxp) : sxp)];
                            fbi[j] = (y - row0) * RES_W + c.tx1 + px;
                        }
                    }
                    dma_join();
                    _Pragma("unroll") for (int j = 0; j < 8; j++) {
                        if (fbi[j] >= 0) fb[fbi[j]] = opaque ? tex[j] : blend(tex[j], fb[fbi[j]], io, ca);
                    }
                }
            }
        }
        PG_SYNC();
    }
    // The background of a band without a register in between: an opaque, unmirrored scaled image wider than half the frame (what
    // draw_background draws, BAG:979-1007) has lane = screen column, and row y of the band is 64 consecutive words of fb -- exactly the
    // shape of an LDS-DMA load (lane l's texel lands at row base + 4 l).  One instruction per row, no VALU work per pixel, and the
    // whole band's rows are in flight while the wave goes on to request the band's first cell / sprite texels: whoever reads or
    // writes fb next joins the copies (dma_join) after issuing its own fetches.
    PG_DEV bool bg_dma_ok(const DrawCmd &c) const {
        return !GEN && !PG_FDBG(d, 131072) && c.w > 32 && cmd_opaque(c.aux) && !cmd_mirrored(c.aux) && !cmd_fill(c.aux) && !cmd_rotated(c.aux) && !cmd_tiled(c.aux);
    }
    PG_DEV void exec_bg_dma(const DrawCmd &c) {
        const uint32_t *src = d.pixels + c.src;
        const int sw = cmd_src_w(c.aux);
        const int y0 = c.ty1 > row0 ? c.ty1 : row0;
        const int y1 = (c.ty1 + c.h) < row1 ? (c.ty1 + c.h) : row1;
        PG_R_LANES(l) {
            const bool in = l < c.w;
            const uint32_t *col = src + (int)((c.basex + (uint32_t)(in ? l : 0) * c.ix) >> 16);
            if (in) {  // (one exec mask for the band's rows, not one per row)
                uint32_t acc = c.srcy0 + (uint32_t)(y0 - c.ty1) * c.iy;
                for (int y = y0; y < y1; y++, acc += c.iy) {
                    uint32_t *row = &fb[(y - row0) * RES_W + c.tx1];
                    PG_DMA_DWORD(col + (int)(acc >> 16) * sw, row, l);
                }
            }
        }
    }
    PG_DEV void dma_join() { PG_DMA_JOIN(); }
    // QPainter::fillRect(QRectF, QColor) without antialiasing: [qRound(left), qRound(right)) x [qRound(top), qRound(bottom)),
    // normalized, opaque colour (used by the games' HUD overlays)
    PG_DEV void exec_fill(RectD r, uint32_t color) {
        int x1 = q_round(r.x), x2 = q_round(r.x + r.w), y1 = q_round(r.y), y2 = q_round(r.y + r.h);
        if (x2 < x1) { const int t = x1; x1 = x2; x2 = t; }
        if (y2 < y1) { const int t = y1; y1 = y2; y2 = t; }
        if (x1 < 0) x1 = 0;
        if (x2 > RES_W) x2 = RES_W;
        if (y1 < row0) y1 = row0;
        if (y2 > row1) y2 = row1;
        if (x1 >= x2 || y1 >= y2) return;
        dma_join();
        for (int y = y1; y < y2; y++) {
            PG_R_LANES(l) {
                if (l >= x1 && l < x2) fb[(y - row0) * RES_W + l] = color;
            }
        }
        PG_SYNC();
    }
    // one horizontal span blended onto the band (spans of the ellipse / line primitives; px premultiplied ARGB)
    PG_DEV void exec_span(int sx, int sy, int len, uint32_t px) {
        if (sy < row0 || sy >= row1 || len <= 0) return;
        dma_join();
        PG_R_LANES(l) {
            if (l >= sx && l < sx + len) {
                uint32_t *dp = &fb[(sy - row0) * RES_W + l];
                *dp = px + byte_mul(*dp, 255u - (px >> 24));
            }
        }
        PG_SYNC();
    }
    // a shape rasterised ahead of time into two 64-bit masks per frame row, words [row][brush lo, hi, pen lo, hi]: brush
    // pixels, then pen pixels over them, both opaque (the jumper compass on a non-integer rect, game_jumper.h host_tables)
    PG_DEV void exec_row_masks(const uint32_t *rows, int y_first, int y_end, uint32_t brush_px, uint32_t pen_px) {
        dma_join();
        for (int y = y_first > row0 ? y_first : row0; y < (y_end < row1 ? y_end : row1); y++) {
            PG_R_LANES(l) {
                const uint32_t *w = rows + y * 4;
                const uint32_t b = l < 32 ? w[0] >> l : w[1] >> (l - 32), p = l < 32 ? w[2] >> l : w[3] >> (l - 32);
                if ((b | p) & 1u) fb[(y - row0) * RES_W + l] = (p & 1u) ? pen_px : brush_px;
            }
        }
        PG_SYNC();
    }
    // QRasterPaintEngine::drawEllipse on an integer-aligned rect, no antialiasing: drawEllipse_midpoint_i +
    // drawEllipsePoints (Qt 5.9.7 qpaintengine_raster.cpp): brush spans, then outline spans of a pen of width <= 1.
    // Pinned with tests/tools/qt_compass_probe.py.
    PG_DEV void ellipse_points(int rx, int ry, int rw, int rh, bool pen, uint32_t pen_px, uint32_t brush_px, int px_, int py_, int length) {
        if (length == 0) return;
        const int midx = rx + (rw + 1) / 2, midy = ry + (rh + 1) / 2;
        const int x = px_ + midx, y = midy - py_;
        const int o0x = midx + midx - x - (length - 1) - (rw & 1);
        const int o0len = length < x - o0x ? length : x - o0x;
        const int o2y = midy + midy - y - (rh & 1);
        if (o0x + o0len < x) {
            const int f0x = o0x + o0len - 1, f0len = x - f0x > 0 ? x - f0x : 0;
            exec_span(f0x, y, f0len, brush_px);
            if (!(y >= o2y)) exec_span(f0x, o2y, f0len, brush_px);
        }
        if (pen) {
            exec_span(o0x, y, o0len, pen_px);
            exec_span(x, y, length, pen_px);
            if (!(y >= o2y)) {
                exec_span(o0x, o2y, o0len, pen_px);
                exec_span(x, o2y, length, pen_px);
            }
        }
    }
    PG_DEV void exec_ellipse(int rx, int ry, int rw, int rh, bool pen, uint32_t pen_px, uint32_t brush_px) {
        if (rw <= 0 || rh <= 0) return;
        if (ry > row1 || ry + rh + 1 < row0) return;  // outside this band
        const double a = rw / 2.0, b = rh / 2.0;
        double d = b * b - (a * a * b) + 0.25 * a * a;
        int x = 0, y = (rh + 1) / 2, startx = x;
        while (a * a * (2 * y - 1) > 2 * b * b * (x + 1)) {
            if (d < 0) {
                d += b * b * (2 * x + 3);
                ++x;
            } else {
                d += b * b * (2 * x + 3) + a * a * (-2 * y + 2);
                ellipse_points(rx, ry, rw, rh, pen, pen_px, brush_px, startx, y, x - startx + 1);
                startx = ++x;
                --y;
            }
        }
        ellipse_points(rx, ry, rw, rh, pen, pen_px, brush_px, startx, y, x - startx + 1);
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
            ellipse_points(rx, ry, rw, rh, pen, pen_px, brush_px, x, y, 1);
        }
    }
    // QCosmeticStroker::drawLine (Qt 5.9.7 qcosmeticstroker.cpp): solid width-0 pen, square caps, integer end points:
    // 26.6 end points, 16.16 minor-axis walker, half a pixel of cap at both ends.  Opaque colour.
    PG_DEV void exec_line(int X1, int Y1, int X2, int Y2, uint32_t px) {
        if (X1 == X2 && Y1 == Y2) {
            exec_span(X1, Y1, (X1 >= 0 && X1 < RES_W) ? 1 : 0, px);
            return;
        }
        int x1 = X1 * 64, y1 = Y1 * 64, x2 = X2 * 64, y2 = Y2 * 64;
        const int dx = x2 - x1 < 0 ? x1 - x2 : x2 - x1, dy = y2 - y1 < 0 ? y1 - y2 : y2 - y1;
        // Qt divides in 64 bits; end points within +-127 pixels keep delta * 65536 below 2^31, and a 32-bit divide costs a
        // fraction of the registers of the 64-bit expansion (this uniform code shares the kernel's allocation)
        if (dx >= (1 << 14) || dy >= (1 << 14)) {
            fail(PGE_UNSUPPORTED_DRAW);
            return;
        }
        if (dx < dy) {
            if (y1 > y2) { int t = y1; y1 = y2; y2 = t; t = x1; x1 = x2; x2 = t; }
            const int xinc = ((x2 - x1) * 65536) / (y2 - y1);
            int x = x1 * 1024;
            y1 -= 32; x -= xinc >> 1; y2 += 32;
            int y = (y1 + 32) >> 6;
            const int ys = (y2 + 32) >> 6, rnd = xinc > 0 ? 32 : 0;
            if (y != ys) {
                x += (int)(((long long)((y * 64) + rnd - y1) * xinc) >> 6);
                do {
                    const int xx = x >> 16;
                    exec_span(xx, y, (xx >= 0 && xx < RES_W) ? 1 : 0, px);
                    x += xinc;
                } while (++y < ys);
            }
        } else {
            if (!dx) return;
            if (x1 > x2) { int t = y1; y1 = y2; y2 = t; t = x1; x1 = x2; x2 = t; }
            const int yinc = ((y2 - y1) * 65536) / (x2 - x1);
            int y = y1 * 1024;
            x1 -= 32; y -= yinc >> 1; x2 += 32;
            int x = (x1 + 32) >> 6;
            const int xs = (x2 + 32) >> 6, rnd = yinc > 0 ? 32 : 0;
            if (x != xs) {
                y += (int)(((long long)((x * 64) + rnd - x1) * yinc) >> 6);
                do {
                    exec_span(x, y >> 16, (x >= 0 && x < RES_W) ? 1 : 0, px);
                    y += yinc;
                } while (++x < xs);
            }
        }
    }
    PG_DEV RectD get_abs_rect(float x, float y, float dx, float dy) const {  // BAG:803-805
        RectD r;
        r.x = (double)(x * G.unit);
        r.y = (double)(y * G.unit);
        r.w = (double)(dx * G.unit);
        r.h = (double)(dy * G.unit);
        return r;
    }
    // one rotated command (qt_transform_image_rasterize): the pixels of its bounding box are laid out linearly
    // over the lanes; a pixel is covered when its row falls into one of the three trapezoids and its column into
    // that row's span [x_l >> 16, x_r >> 16); its texel is the clamped inverse mapping of the pixel position.
    // a rotation record travels as 24 wave-uniform scalars passed by value: kept in an array or a struct, the ?: chains over
    // its fields turn into one load at a computed offset and the record lands in scratch memory
    template <int NJ>
    PG_DEV void exec_rotated_pass(const DrawCmd &c, int base, int npix, int y0, uint32_t inv, uint32_t u0, uint32_t v0, uint32_t dudx, uint32_t dudy, uint32_t dvdx, uint32_t dvdy, uint32_t tf0, uint32_t tt0, uint32_t txl0, uint32_t tdl0, uint32_t txr0, uint32_t tdr0, uint32_t tf1, uint32_t tt1, uint32_t txl1, uint32_t tdl1, uint32_t txr1, uint32_t tdr1, uint32_t tf2, uint32_t tt2, uint32_t txl2, uint32_t tdl2, uint32_t txr2, uint32_t tdr2) {
        const uint32_t *src = d.pixels + c.src;
        const int sw = cmd_src_w(c.aux), sh = (int)c.iy;
        const bool mirrored = cmd_mirrored(c.aux);
        const int io = cmd_alpha(c.aux);
        const uint32_t ca = (uint32_t)((io * 255) >> 8);
        PG_R_LANES(l) {
            uint32_t tex[NJ];
            int fbi[NJ];
            _Pragma("unroll") for (int j = 0; j < NJ; j++) {
                const int p = base + j * 64 + l;
                const int pc = p < npix ? p : 0;
                const int pyb = (int)(((uint32_t)pc * inv) >> 20);
                const int X = c.tx1 + (pc - pyb * c.w);
                const int Y = y0 + pyb;
                // the trapezoids cover disjoint row ranges
                const bool s0 = (uint32_t)Y >= tf0 && (uint32_t)Y < tt0;
                const bool s1 = (uint32_t)Y >= tf1 && (uint32_t)Y < tt1;
                const bool s2 = (uint32_t)Y >= tf2 && (uint32_t)Y < tt2;
                const bool in_rows = s0 || s1 || s2;
                const uint32_t k = (uint32_t)Y - (s0 ? tf0 : (s1 ? tf1 : tf2));
                int from_x = (int)((s0 ? txl0 : (s1 ? txl1 : txl2)) + k * (s0 ? tdl0 : (s1 ? tdl1 : tdl2))) >> 16;
                int to_x = (int)((s0 ? txr0 : (s1 ? txr1 : txr2)) + k * (s0 ? tdr0 : (s1 ? tdr1 : tdr2))) >> 16;
                if (from_x < 0) from_x = 0;
                if (to_x > RES_W) to_x = RES_W;
                const bool in = p < npix && in_rows && X >= from_x && X < to_x;
                int uu = (int)((uint32_t)X * dudx + (uint32_t)Y * dudy + u0) >> 16;
                int vv = (int)((uint32_t)X * dvdx + (uint32_t)Y * dvdy + v0) >> 16;
                uu = uu < 0 ? 0 : (uu > sw - 1 ? sw - 1 : uu);
                vv = vv < 0 ? 0 : (vv > sh - 1 ? sh - 1 : vv);
                tex[j] = src[in ? (vv * sw + (mirrored ? (sw - 1 - uu) : uu)) : 0];
                fbi[j] = in ? ((Y - row0) * RES_W + X) : (BAND_ROWS * RES_W + l);  // masked-off lanes use the dump row
            }
            dma_join();
            _Pragma("unroll") for (int j = 0; j < NJ; j++) { fb[fbi[j]] = blend(tex[j], fb[fbi[j]], io, ca); }
        }
    }
    // GEN: a turned sprite, BAG:897-906: p.translate(cx, cy); p.rotate(deg); p.drawImage(QRectF(-w/2, -h/2, w, h), sprite) on Qt's
    // generic route.  Everything is worked out again from the entity (wave-uniform doubles); lanes take the pixels of the
    // bounding box that fall into this band.
    PG_DEV void exec_generic_rotated(const DrawCmd &c) {
        const int i = (int)c.basex;
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
        ImgDesc imd;
        const int im = resolve_image(meta_image_type(mm), meta_image_theme(mm), 0.0f, 0.0f, r1, nullptr, &imd);
        if (im < 0) return;
        const bool mirrored = (mm & MF_REFLECTED) != 0;
        const int io = opacity_to_io(ef(EF_ALPHA, i));
        const uint32_t ca = (uint32_t)((io * 255) >> 8);
        const double cx = r1.x + r1.w / 2, cy = r1.y + r1.h / 2;
        const double a = (double)(ef(EF_ROTATION, i) * 180 / PG_PI);
        const RectD r = {-r1.w / 2, -r1.h / 2, r1.w, r1.h};
        if (!(r.w > 0 && r.h > 0)) return;
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
        const double m11 = cosa, m12 = sina, m21 = -sina, m22 = cosa;
        int type = 0;  // QTransform::type(): 4 rotate, 2 scale, 1 translate, 0 none
        if (!q_fuzzy_is_null(m12) || !q_fuzzy_is_null(m21)) type = 4;
        else if (!q_fuzzy_is_null(m11 - 1) || !q_fuzzy_is_null(m22 - 1)) type = 2;
        else if (!q_fuzzy_is_null(cx) || !q_fuzzy_is_null(cy)) type = 1;
        // sampling matrix (QSpanData::setupMatrix): copy = m; copy.translate(r.x, r.y); copy.scale(r.w / sw, r.h / sh)
        double c11 = m11, c12 = m12, c21 = m21, c22 = m22, cdx = cx, cdy = cy;
        int ctype = type;
        if (ctype == 0) { cdx = r.x; cdy = r.y; ctype = 1; }
        else if (ctype == 1) { cdx += r.x; cdy += r.y; }
        else if (ctype == 2) { cdx += r.x * c11; cdy += r.y * c22; }
        else { cdx += r.x * c11 + r.y * c21; cdy += r.y * c22 + r.x * c12; }
        const double scx = r.w / (double)imd.w, scy = r.h / (double)imd.h;
        if (ctype == 4) { c12 *= scx; c21 *= scy; }
        c11 *= scx;
        c22 *= scy;
        const double dlt = 1.0 / 65536;
        double i11, i12, i21, i22, idx, idy;
        if (ctype != 4) {
            const double p11 = 1.0 * c11, p22 = 1.0 * c22, p31 = dlt * c11 + cdx, p32 = dlt * c22 + cdy;
            i11 = 1. / p11; i22 = 1. / p22; i12 = 0; i21 = 0;
            idx = -p31 * i11; idy = -p32 * i22;
        } else {
            const double p11 = 1.0 * c11 + 0.0 * c21, p12 = 1.0 * c12 + 0.0 * c22, p21 = 0.0 * c11 + 1.0 * c21, p22 = 0.0 * c12 + 1.0 * c22;
            const double p31 = dlt * c11 + dlt * c21 + cdx, p32 = dlt * c12 + dlt * c22 + cdy;
            const double dtr = p11 * p22 - p12 * p21, dinv = 1.0 / dtr;
            i11 = p22 * dinv; i12 = -p12 * dinv; i21 = -p21 * dinv; i22 = p11 * dinv;
            idx = (p21 * p32 - p22 * p31) * dinv; idy = (p12 * p31 - p11 * p32) * dinv;
        }
        const int fdx = (int)(i11 * 65536.), fdy = (int)(i12 * 65536.);
        // coverage
        int kind, box[4] = {0, -1, 0, -1}, qx[4] = {0, 0, 0, 0}, qy[4] = {0, 0, 0, 0};
        if (type == 2) {  // fillRect_normalized(QRect(qRound of the mapped rect))
            double bx = m11 * r.x + cx, by = m22 * r.y + cy, ww = m11 * r.w, hh = m22 * r.h;
            if (ww < 0) { ww = -ww; bx -= ww; }
            if (hh < 0) { hh = -hh; by -= hh; }
            box[0] = q_round(bx); box[2] = q_round(by); box[1] = q_round(bx + ww) - 1; box[3] = q_round(by + hh) - 1;
            if (box[0] < 0) box[0] = 0;
            if (box[2] < 0) box[2] = 0;
            if (box[1] > RES_W - 1) box[1] = RES_W - 1;
            if (box[3] > RES_H - 1) box[3] = RES_H - 1;
            kind = (box[1] >= box[0] && box[3] >= box[2]) ? 1 : 0;
        } else {
            const double l = r.x, t = r.y, rr = r.x + r.w, b = r.y + r.h;
            double ax = (l + l) * 0.5, ay = (t + b) * 0.5, bx = (rr + rr) * 0.5, by = (t + b) * 0.5;
            if (type == 1) { ax += cx; ay += cy; bx += cx; by += cy; }
            else if (type == 4) {
                const double tax = m11 * ax + m21 * ay + cx, tay = m12 * ax + m22 * ay + cy, tbx = m11 * bx + m21 * by + cx, tby = m12 * bx + m22 * by + cy;
                ax = tax; ay = tay; bx = tbx; by = tby;
            }
            kind = rasterize_line(ax, ay, bx, by, r.h / r.w, box, qx, qy);
        }
        if (kind == 0) return;
        const uint32_t *src = d.pixels + imd.off;
        const int sw = (int)imd.w, sh = (int)imd.h;
        const int y0 = c.ty1 > row0 ? c.ty1 : row0;
        const int y1 = (c.ty1 + c.h) < row1 ? (c.ty1 + c.h) : row1;
        const int npix = c.w * (y1 - y0);
        const uint32_t inv = (uint32_t)(((1u << 20) + (uint32_t)c.w - 1u) / (uint32_t)c.w);
        for (int base = 0; base < npix; base += 64) {
            PG_R_LANES(l) {
                const int p = base + l;
                if (p < npix) {
                    const int pyb = (int)(((uint32_t)p * inv) >> 20);
                    const int X = c.tx1 + (p - pyb * c.w), Y = y0 + pyb;
                    int x1 = box[0], x2 = box[1];
                    bool in;
                    if (kind == 1) in = Y >= box[2] && Y <= box[3];
                    else in = polygon_row(qx, qy, Y, x1, x2);
                    in = in && X >= x1 && X <= x2;
                    if (in) {
                        const double ccx = (double)x1 + 0.5, ccy = (double)Y + 0.5;
                        const int fx = (int)((i21 * ccy + i11 * ccx + idx) * 65536.) + (X - x1) * fdx;
                        const int fy = (int)((i22 * ccy + i12 * ccx + idy) * 65536.) + (X - x1) * fdy;
                        int px = fx >> 16, py = fy >> 16;
                        px = px < 0 ? 0 : (px > sw - 1 ? sw - 1 : px);
                        py = py < 0 ? 0 : (py > sh - 1 ? sh - 1 : py);
                        uint32_t *dp = &fb[(Y - row0) * RES_W + X];
                        *dp = blend(src[py * sw + (mirrored ? sw - 1 - px : px)], *dp, io, ca);
                    }
                }
            }
        }
        PG_SYNC();
    }
    PG_DEV void exec_rotated(const DrawCmd &c) {
        if constexpr (GEN) {
            exec_generic_rotated(c);
            return;
        }
        if constexpr (GameUsesRotation<Game>::value) {
            const uint32_t *rp = &lds->rot[(c.basex & 63u) * ROT_WORDS];
#define PG_ROT(k) ((uint32_t)PG_UNIFORM_I(rp[k]))
            const uint32_t u0 = PG_ROT(0);
            const uint32_t v0 = PG_ROT(1);
            const uint32_t dudx = PG_ROT(2);
            const uint32_t dudy = PG_ROT(3);
            const uint32_t dvdx = PG_ROT(4);
            const uint32_t dvdy = PG_ROT(5);
            const uint32_t tf0 = PG_ROT(6);
            const uint32_t tt0 = PG_ROT(7);
            const uint32_t txl0 = PG_ROT(8);
            const uint32_t tdl0 = PG_ROT(9);
            const uint32_t txr0 = PG_ROT(10);
            const uint32_t tdr0 = PG_ROT(11);
            const uint32_t tf1 = PG_ROT(12);
            const uint32_t tt1 = PG_ROT(13);
            const uint32_t txl1 = PG_ROT(14);
            const uint32_t tdl1 = PG_ROT(15);
            const uint32_t txr1 = PG_ROT(16);
            const uint32_t tdr1 = PG_ROT(17);
            const uint32_t tf2 = PG_ROT(18);
            const uint32_t tt2 = PG_ROT(19);
            const uint32_t txl2 = PG_ROT(20);
            const uint32_t tdl2 = PG_ROT(21);
            const uint32_t txr2 = PG_ROT(22);
            const uint32_t tdr2 = PG_ROT(23);
#undef PG_ROT
            const int y0 = c.ty1 > row0 ? c.ty1 : row0;
            const int y1 = (c.ty1 + c.h) < row1 ? (c.ty1 + c.h) : row1;
            const int npix = c.w * (y1 - y0);
            const uint32_t inv = (uint32_t)(((1u << 20) + (uint32_t)c.w - 1u) / (uint32_t)c.w);
            // small turned sprites (bossfight's bullets) fit one or two pixels per lane: no point in issuing the
            // eight-deep fetch batch for them
            if (npix <= 128) {
                for (int base = 0; base < npix; base += 64) exec_rotated_pass<1>(c, base, npix, y0, inv, u0, v0, dudx, dudy, dvdx, dvdy, tf0, tt0, txl0, tdl0, txr0, tdr0, tf1, tt1, txl1, tdl1, txr1, tdr1, tf2, tt2, txl2, tdl2, txr2, tdr2);
            } else {
                for (int base = 0; base < npix; base += 512) exec_rotated_pass<8>(c, base, npix, y0, inv, u0, v0, dudx, dudy, dvdx, dvdy, tf0, tt0, txl0, tdl0, txr0, tdr0, tf1, tt1, txl1, tdl1, txr1, tdr1, tf2, tt2, txl2, tdl2, txr2, tdr2);
            }
            PG_SYNC();
        } else {
            (void)c;
        }
    }
    // up to 8 commands of at most 8x8 pixels each: lane = (row, column) of the footprint; the 8 texel fetches are
    // issued together, the blends then run command by command (two commands may touch the same pixel from
    // different lanes, so each blend is its own lane section)
    PG_DEV void exec_small_group(const DrawCmd (&c)[8], int count) {
        PG_LANE_ARR(uint32_t, tex, 8);
        PG_LANE_ARR(int, fbi, 8);
        PG_R_LANES(l) {
            const int lx = l & 7, ly = l >> 3;
            _Pragma("unroll") for (int g = 0; g < 8; g++) {
                PG_LA(fbi, g, l) = BAND_ROWS * RES_W + l;  // dump row
                PG_LA(tex, g, l) = 0;
                if (g < count) {  // wave-uniform
                    const int y = c[g].ty1 + ly;
                    const int sw = cmd_src_w(c[g].aux);
                    // where the lane's pixel of command g comes from: an upright sprite's 16.16 walk, or -- a turned sprite whose bounding
                    // box fits the footprint (bossfight's bullets, round 4) -- the clamped inverse mapping of the pixel, when its row
                    // falls into one of the command's three trapezoids and its column into that row's span (exec_rotated_pass; the
                    // record is read where it is used, so that eight such sprites, or a mix with upright ones, share one wait for texels)
                    bool in = lx < c[g].w && ly < c[g].h && y >= row0 && y < row1;
                    uint32_t addr;
                    int fi = (y - row0) * RES_W + c[g].tx1 + lx;
                    bool turned = false;
                    if constexpr (GameUsesRotation<Game>::value && !GEN) turned = cmd_rotated(c[g].aux);
                    if (turned) {
                        if constexpr (GameUsesRotation<Game>::value && !GEN) {
                            const uint32_t *rp = &lds->rot[(c[g].basex & 63u) * ROT_WORDS];
                            const int X = c[g].tx1 + lx, sh = (int)c[g].iy;
                            const uint32_t Y = (uint32_t)y;
                            const bool s0 = Y >= rp[6] && Y < rp[7], s1 = Y >= rp[12] && Y < rp[13], s2 = Y >= rp[18] && Y < rp[19];
                            const uint32_t *tp = rp + (s0 ? 6 : (s1 ? 12 : 18));
                            const uint32_t k = Y - tp[0];
                            int from_x = (int)(tp[2] + k * tp[3]) >> 16, to_x = (int)(tp[4] + k * tp[5]) >> 16;
                            if (from_x < 0) from_x = 0;
                            if (to_x > RES_W) to_x = RES_W;
                            in = in && (s0 || s1 || s2) && X >= from_x && X < to_x;
                            int uu = (int)((uint32_t)X * rp[2] + Y * rp[3] + rp[0]) >> 16;
                            int vv = (int)((uint32_t)X * rp[4] + Y * rp[5] + rp[1]) >> 16;
                            uu = uu < 0 ? 0 : (uu > sw - 1 ? sw - 1 : uu);
                            vv = vv < 0 ? 0 : (vv > sh - 1 ? sh - 1 : vv);
                            addr = c[g].src + (uint32_t)(vv * sw + (cmd_mirrored(c[g].aux) ? (sw - 1 - uu) : uu));
                        }
                    } else {
                        int sxp, syp;
                        sample_xy(c[g], lx, y, sw, sxp, syp);
                        addr = cmd_fill(c[g].aux) ? 0u : c[g].src + (uint32_t)(syp * sw + (cmd_mirrored(c[g].aux) ? (sw - 1 - sxp) : sxp));
                    }
                    PG_LA(tex, g, l) = cmd_fill(c[g].aux) ? c[g].src : d.pixels[in ? addr : 0u];  // branch-free: masked-off lanes fetch word 0
                    PG_LA(fbi, g, l) = in ? fi : (BAND_ROWS * RES_W + l);
                }
            }
        }
        dma_join();
        _Pragma("unroll") for (int g = 0; g < 8; g++) {
            if (g < count) {
                const int io = cmd_alpha(c[g].aux);
                const uint32_t ca = (uint32_t)((io * 255) >> 8);
                const bool opaque = cmd_opaque(c[g].aux);
                PG_R_LANES(l) {
                    const int fi = PG_LA(fbi, g, l);
                    fb[fi] = opaque ? PG_LA(tex, g, l) : blend(PG_LA(tex, g, l), fb[fi], io, ca);
                }
            }
        }
        PG_SYNC();
    }

    // one draw command per lane, produced by a set-up section and consumed by run_batch()
    struct CmdRegs : CmdExtra<GEN> {
        PG_LANE_VAR(uint32_t, geom);
        PG_LANE_VAR(uint32_t, basex);
        PG_LANE_VAR(uint32_t, srcy);
        PG_LANE_VAR(uint32_t, ix);
        PG_LANE_VAR(uint32_t, iy);
        PG_LANE_VAR(uint32_t, src);
        PG_LANE_VAR(uint32_t, aux);
    };
    PG_DEV static DrawCmd read_cmd(const CmdRegs &r, int k) {
        if constexpr (GEN)
            return unpack(PG_READLANE(r.geom, k), PG_READLANE(r.basex, k), PG_READLANE(r.srcy, k), PG_READLANE(r.ix, k), PG_READLANE(r.iy, k),
                          PG_READLANE(r.src, k), PG_READLANE(r.aux, k), PG_READLANE(r.e0, k), PG_READLANE(r.e1, k));
        else
            return unpack(PG_READLANE(r.geom, k), PG_READLANE(r.basex, k), PG_READLANE(r.srcy, k), PG_READLANE(r.ix, k), PG_READLANE(r.iy, k),
                          PG_READLANE(r.src, k), PG_READLANE(r.aux, k));
    }
    // lane l's command := nothing
    PG_DEV static void clear_cmd(CmdRegs &r, int l) {
        PG_LV(r.geom, l) = 0;
        PG_LV(r.basex, l) = PG_LV(r.srcy, l) = PG_LV(r.ix, l) = PG_LV(r.iy, l) = PG_LV(r.src, l) = PG_LV(r.aux, l) = 0;
        if constexpr (GEN) PG_LV(r.e0, l) = PG_LV(r.e1, l) = 0;
    }
    // lane l's command := p.drawImage(tr, image) with the untransformed painter
    PG_DEV void emit_image(CmdRegs &r, int l, const ImgDesc im, bool mirrored, RectD tr, float opacity) {
        if constexpr (GEN)
            cmd_image_generic(im, mirrored, tr, opacity, PG_LV(r.geom, l), PG_LV(r.basex, l), PG_LV(r.srcy, l), PG_LV(r.ix, l), PG_LV(r.iy, l), PG_LV(r.src, l), PG_LV(r.aux, l),
                              PG_LV(r.e0, l), PG_LV(r.e1, l));
        else
            cmd_image_fast(im, mirrored, tr, opacity, PG_LV(r.geom, l), PG_LV(r.basex, l), PG_LV(r.srcy, l), PG_LV(r.ix, l), PG_LV(r.iy, l), PG_LV(r.src, l), PG_LV(r.aux, l));
    }
    // executes the commands in lane order; runs of small commands go eight at a time
    // lane_mask2: a second selection that is drawn after the first one, each in lane order -- the entities of the next render_z
    // layer (nothing is painted between z = 0 and z = 1, BAG:957-958), so that groups of small sprites and their one wait for texels
    // span both layers
    template <bool NESTED = false>
    PG_DEV void run_batch(const CmdRegs &r, uint64_t lane_mask = ~0ull, uint64_t lane_mask2 = 0) {
        // commands that exist, are selected by the caller, and touch the rows of the current pass
        const uint64_t touch = PG_BALLOT(l, PG_LV(r.geom, l) != 0 && (int)((PG_LV(r.geom, l) >> 7) & 0x7fu) < row1 &&
                                                (int)(((PG_LV(r.geom, l) >> 7) & 0x7fu) + ((PG_LV(r.geom, l) >> 21) & 0x7fu)) > row0);
        uint64_t valid = touch & lane_mask, valid2 = touch & lane_mask2;
        if (valid == 0) {
            valid = valid2;
            valid2 = 0;
        }
        // (a turned sprite joins the groups of small ones when the game has rotation records and its bounding box fits; GEN draws them on Qt's generic route)
        const uint64_t small = PG_BALLOT(l, PG_LV(r.geom, l) != 0 && ((PG_LV(r.geom, l) >> 14) & 0x7fu) <= 8u && ((PG_LV(r.geom, l) >> 21) & 0x7fu) <= 8u &&
                                                (!cmd_rotated(PG_LV(r.aux, l)) || (GameUsesRotation<Game>::value && !GEN && !PG_FDBG(d, 524288))) && !cmd_tiled(PG_LV(r.aux, l)));
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
                        if (valid == 0) {  // the first selection is drawn: go on with the second
                            valid = valid2;
                            valid2 = 0;
                        }
                    }
                }
                exec_small_group(c, count);
            } else {
                valid &= valid - 1;
                const DrawCmd c = read_cmd(r, k);
                if (cmd_tiled(c.aux)) {
                    if constexpr (NESTED || !GameUsesTiledEntities<Game>::value) fail(PGE_ASSERT);
                    else exec_tiled(c);
                } else if (cmd_rotated(c.aux)) exec_rotated(c);
                else exec_large(c);
                if (valid == 0) {
                    valid = valid2;
                    valid2 = 0;
                }
            }
        }
    }

    // tile_image BAG:840-865 for one tiled entity: a row (tile_ratio > 0) or column (< 0) of equally sized tiles, each its
    // own drawImage with its own rounding.  Lanes set up 64 tiles at a time.  The entity's rect, image and flags come
    // from the record its set-up lane left in LDS (slot = c.srcy): a band pass pays no HBM round trips for them.
    PG_DEV static double words_to_double(uint32_t lo, uint32_t hi) { return __builtin_bit_cast(double, (uint64_t)lo | ((uint64_t)hi << 32)); }
    PG_DEV void exec_tiled(const DrawCmd &c) {
        const uint32_t *rp = &lds->rot[(c.srcy0 & 63u) * ROT_WORDS];
        const RectD rect = {words_to_double(rp[0], rp[1]), words_to_double(rp[2], rp[3]), words_to_double(rp[4], rp[5]), words_to_double(rp[6], rp[7])};
        ImgDesc imd;
        imd.off = rp[8];
        imd.w = (uint16_t)(rp[9] & 0xffffu);
        imd.h = (uint16_t)(rp[9] >> 16);
        imd.opaque = rp[10];
        const float alpha = __builtin_bit_cast(float, rp[11]);
        float tile_ratio = __builtin_bit_cast(float, rp[12]);
        const bool mirrored = (rp[13] & 1u) != 0;
        const bool vertical = tile_ratio < 0;
        if (vertical) tile_ratio = -1 * tile_ratio;
        int num_tiles = vertical ? (int)(rect.h / (rect.w * (double)tile_ratio)) : (int)(rect.w / (rect.h * (double)tile_ratio));
        if (num_tiles < 1) num_tiles = 1;
        const float tile_width = vertical ? (float)rect.w : (float)(rect.w / num_tiles);
        const float tile_height = vertical ? (float)(rect.h / num_tiles) : (float)rect.h;
        for (int base = 0; base < num_tiles; base += 64) {
            CmdRegs r;
            PG_R_LANES(l) {
                clear_cmd(r, l);
                const int t = base + l;
                if (t < num_tiles) {
                    RectD tr;
                    if (vertical) {
                        tr.x = rect.x;
                        tr.y = rect.y + (double)(tile_height * t);
                    } else {
                        tr.x = rect.x + (double)(tile_width * t);
                        tr.y = rect.y;
                    }
                    tr.w = (double)tile_width;
                    tr.h = (double)tile_height;
                    emit_image(r, l, imd, mirrored, tr, alpha);
                }
            }
            run_batch<true>(r);
        }
    }

    // What the set-up of the first 64 entity slots reads from memory, requested before the values are needed: the fields go out
    // together with the header (their addresses depend on the env alone), the image descriptors together with the frame's other
    // table lookups (render_env).  A dependent global access costs this kernel ~1.3 k wave cycles, whatever cache serves it.
    struct EntPre {
        PG_LANE_VAR(uint32_t, meta);
        PG_LANE_VAR(uint32_t, x);
        PG_LANE_VAR(uint32_t, y);
        PG_LANE_VAR(uint32_t, rx);
        PG_LANE_VAR(uint32_t, ry);
        PG_LANE_VAR(uint32_t, rot);
        PG_LANE_VAR(uint32_t, alpha);
        PG_LANE_VAR(ImgDesc, desc);
    };
    PG_DEV void request_entity_fields(EntPre &p) const {
        PG_R_LANES(l) {
            const uint32_t sl = (uint32_t)(l < ecap ? l : 0);
            PG_LV(p.meta, l) = ge[(uint32_t)(EF_META * ecap) + sl];
            PG_LV(p.x, l) = ge[(uint32_t)(EF_X * ecap) + sl];
            PG_LV(p.y, l) = ge[(uint32_t)(EF_Y * ecap) + sl];
            PG_LV(p.rx, l) = ge[(uint32_t)(EF_RX * ecap) + sl];
            PG_LV(p.ry, l) = ge[(uint32_t)(EF_RY * ecap) + sl];
            PG_LV(p.rot, l) = ge[(uint32_t)(EF_ROTATION * ecap) + sl];
            PG_LV(p.alpha, l) = ge[(uint32_t)(EF_ALPHA * ecap) + sl];
        }
    }
    // draw_entities BAG:1052-1066.  Commands of 64 entities are set up once (they do not depend on the layer) and
    // then executed per render_z layer through a lane mask.
    // rot_base >= 0: the chunk's turned entities take consecutive rotation records from rot_base on (compact_entities);
    // otherwise lane l uses record l.  Returns the number of records handed out.
    PG_DEV int setup_entities(int base, CmdRegs &r, uint64_t (&zmask)[3], int rot_base = -1) {
        EntPre none;
        return setup_entities_t<false>(base, r, zmask, rot_base, none);
    }
    static constexpr int ROT_POOL = GameRotPool<Game>::value;
    static constexpr bool ROT_POOLED = GameUsesRotation<Game>::value && ROT_POOL < 64;
    // ROT_POOLED: can this entity's turned sprite / row of tiles reach the rows [row0, row1)?  Conservative and cheap (float, nothing shared
    // with the exact rect arithmetic of the lane section: the first version, the exact rect's circumcircle in fp64, cost the kernels 25-30
    // VGPRs): the centre of get_object_rect's rect (BAG:799-817) and three times its half-extents plus three pixels -- a turned rect's
    // corners lie on the circumcircle (BAG:902-906), a row of tiles fills the rect (each tile rounded on its own), and no
    // adjusted_image_rect of a game moves or grows a rect beyond that.
    template <bool PRE>
    PG_DEV bool record_maybe_visible(const EntPre &pre, int i, int l) const {
        (void)l;
        const uint32_t mm = PRE ? PG_LV(pre.meta, l) : meta(i);
        const float x = PRE ? __builtin_bit_cast(float, PG_LV(pre.x, l)) : ex(i), y = PRE ? __builtin_bit_cast(float, PG_LV(pre.y, l)) : ey(i);
        const float rx = PRE ? __builtin_bit_cast(float, PG_LV(pre.rx, l)) : erx(i), ry = PRE ? __builtin_bit_cast(float, PG_LV(pre.ry, l)) : ery(i);
        const bool abs_coords = (mm & MF_ABS_COORDS) != 0;
        const float sc = abs_coords ? G.view_dim * G.unit : G.unit;
        const float cxs = abs_coords ? x * sc : x * sc - G.x_off;
        const float cys = abs_coords ? (y + 2 * ry) * sc : (G.view_dim - y) * sc + G.y_off;
        const float rad = 3 * ((rx < 0 ? -rx : rx) + (ry < 0 ? -ry : ry)) * sc + 3;
        return cxs + rad > 0 && cxs - rad < (float)RES_W && cys + rad > (float)row0 && cys - rad < (float)row1;
    }
    PG_DEV RectD object_rect(uint32_t mm, float x, float y, float rx, float ry) const {  // get_object_rect BAG:811-817
        RectD r1;
        if (mm & MF_ABS_COORDS) {
            const float vd = G.view_dim;
            r1.x = (double)((vd * (x - rx)) * G.unit);
            r1.y = (double)((vd * (y + ry)) * G.unit);
            r1.w = (double)((2 * vd * rx) * G.unit);
            r1.h = (double)((2 * vd * ry) * G.unit);
        } else {
            r1 = get_screen_rect(x - rx, y + ry, 2 * rx, 2 * ry, 0);
        }
        return r1;
    }
    // PRE: base == 0 and the slots' fields / image descriptors were requested ahead (EntPre)
    // ROT_POOLED only: `lanes` = the lanes of the chunk to set up (the others get no command); with `resume`, a window that needs more records
    // than the pool has left is cut in front of the first entity that does not fit and *resume = that lane (64: the window was set up whole)
    template <bool PRE>
    PG_DEV int setup_entities_t(int base, CmdRegs &r, uint64_t (&zmask)[3], int rot_base, const EntPre &pre, uint64_t lanes = ~0ull, int *resume = nullptr) {
        const int n = G.n_ents;
        for (int z = 0; z < 3; z++) zmask[z] = PG_BALLOT(l, (base + l) < n && meta_render_z(PRE ? PG_LV(pre.meta, l) : meta(base + l)) == z - 1);
        uint64_t rotmask = 0;  // entities that need an LDS record: turned sprites and tiled ones
        if constexpr (ROT_POOLED) {
            if (rot_base < 0) rot_base = 0;
            // (the test runs in a lane section of the renderer's kind, its lane id opaque to LICM, and the ballot reads a flag: inside the
            // ballot itself the same arithmetic cost the kernels ten more VGPRs)
            PG_LANE_VAR(uint32_t, needs);
            PG_R_LANES(l) {
                PG_LV(needs, l) = (((lanes >> l) & 1ull) && (base + l) < n && ((PRE ? __builtin_bit_cast(float, PG_LV(pre.rot, l)) : ef(EF_ROTATION, base + l)) != 0 || (GameUsesTiledEntities<Game>::value && Game::tile_aspect_ratio(*this, base + l) != 0)) &&
                                   record_maybe_visible<PRE>(pre, base + l, l))
                                      ? 1u
                                      : 0u;
            }
            rotmask = PG_BALLOT(l, PG_LV(needs, l) != 0);
            if (resume) *resume = 64;
            if (rot_base + pg_popc64(rotmask) > ROT_POOL) {
                if (!resume) return -1;
                uint64_t m = rotmask;
                for (int k = rot_base; k < ROT_POOL; k++) m &= m - 1;  // (the records that fit)
                const int cut = pg_ctz64(m);
                lanes &= pg_mask_lt(cut);
                rotmask &= pg_mask_lt(cut);
                *resume = cut;
#if defined(PGAMD_WAVE_EMU)
                pg_emu_counters()[7] += 1;  // windows cut for want of records
#endif
            }
        } else if constexpr (GameUsesRotation<Game>::value) {
            if (rot_base >= 0) {
                rotmask = PG_BALLOT(l, (base + l) < n && ((PRE ? __builtin_bit_cast(float, PG_LV(pre.rot, l)) : ef(EF_ROTATION, base + l)) != 0 || (GameUsesTiledEntities<Game>::value && Game::tile_aspect_ratio(*this, base + l) != 0)));
                if (rot_base + pg_popc64(rotmask) > 64) return -1;
            }
        }
        PG_R_LANES(l) {
            const int i = base + l;
            const int rot_slot = rot_base >= 0 ? rot_base + pg_popc64(rotmask & pg_mask_lt(l)) : l;
            clear_cmd(r, l);
            bool wanted = i < n && Game::should_draw_entity(*this, i);
            if constexpr (ROT_POOLED) {
                // outside the window; or a turned / tiled entity without a record: it cannot reach the rows being drawn
                wanted = wanted && ((lanes >> l) & 1ull);
                if (wanted && !((rotmask >> l) & 1ull) &&
                    ((PRE ? __builtin_bit_cast(float, PG_LV(pre.rot, l)) : ef(EF_ROTATION, i)) != 0 || (GameUsesTiledEntities<Game>::value && Game::tile_aspect_ratio(*this, i) != 0)))
                    wanted = false;
            }
            if (wanted) {
                const uint32_t mm = PRE ? PG_LV(pre.meta, l) : meta(i);
                const float x = PRE ? __builtin_bit_cast(float, PG_LV(pre.x, l)) : ex(i), y = PRE ? __builtin_bit_cast(float, PG_LV(pre.y, l)) : ey(i);
                const float rx = PRE ? __builtin_bit_cast(float, PG_LV(pre.rx, l)) : erx(i), ry = PRE ? __builtin_bit_cast(float, PG_LV(pre.ry, l)) : ery(i);
                const float e_alpha = PRE ? __builtin_bit_cast(float, PG_LV(pre.alpha, l)) : ef(EF_ALPHA, i);
                RectD r1 = object_rect(mm, x, y, rx, ry);
                const RectD r1_in = r1;
                uint32_t fc = 0;
                ImgDesc imd;
                const int im = resolve_image(meta_image_type(mm), meta_image_theme(mm), 0.0f, 0.0f, r1, &fc, &imd, PRE ? &PG_LV(pre.desc, l) : nullptr);
                const float rotation = PRE ? __builtin_bit_cast(float, PG_LV(pre.rot, l)) : ef(EF_ROTATION, i);
                const float tile_ratio = Game::tile_aspect_ratio(*this, i);
                if (im == IMG_FILL) cmd_fill_rect(r1_in, fc, PG_LV(r.geom, l), PG_LV(r.src, l), PG_LV(r.aux, l));
                if (im >= 0) {
                    if (rotation == 0 && tile_ratio != 0 && !GameUsesTiledEntities<Game>::value) {
                        fail(PGE_UNSUPPORTED_DRAW);
                    } else if (rotation == 0 && tile_ratio != 0) {
                        // bounding box of all tiles (each tile is rounded on its own: one pixel of slack either side)
                        int bx1 = q_round(r1.x) - 1, bx2 = q_round(r1.x + r1.w) + 1, by1 = q_round(r1.y) - 1, by2 = q_round(r1.y + r1.h) + 1;
                        if (bx1 < 0) bx1 = 0;
                        if (by1 < 0) by1 = 0;
                        if (bx2 > RES_W) bx2 = RES_W;
                        if (by2 > RES_H) by2 = RES_H;
                        if (bx2 > bx1 && by2 > by1) {
                            PG_LV(r.geom, l) = (uint32_t)bx1 | ((uint32_t)by1 << 7) | ((uint32_t)(bx2 - bx1) << 14) | ((uint32_t)(by2 - by1) << 21);
                            PG_LV(r.basex, l) = (uint32_t)i;
                            PG_LV(r.srcy, l) = (uint32_t)rot_slot;
                            PG_LV(r.aux, l) = 1u << 25;
                            if constexpr (GameUsesTiledEntities<Game>::value && GameUsesRotation<Game>::value) {  // what exec_tiled needs, see there
                                uint32_t *rp = &lds->rot[rot_slot * ROT_WORDS];
                                const double rv[4] = {r1.x, r1.y, r1.w, r1.h};
                                for (int k = 0; k < 4; k++) {
                                    const uint64_t bits = __builtin_bit_cast(uint64_t, rv[k]);
                                    rp[2 * k] = (uint32_t)bits;
                                    rp[2 * k + 1] = (uint32_t)(bits >> 32);
                                }
                                rp[8] = imd.off;
                                rp[9] = (uint32_t)imd.w | ((uint32_t)imd.h << 16);
                                rp[10] = imd.opaque;
                                rp[11] = __builtin_bit_cast(uint32_t, e_alpha);
                                rp[12] = __builtin_bit_cast(uint32_t, tile_ratio);
                                rp[13] = (mm & MF_REFLECTED) ? 1u : 0u;
                            }
                        }
                    } else if (rotation == 0) emit_image(r, l, imd, (mm & MF_REFLECTED) != 0, r1, e_alpha);
                    else if constexpr (GEN) {
                        // a turned generated sprite: coverage and sampling come from the painter matrix (exec_generic_rotated works
                        // them out again from the entity); here only a bounding box for the band test: the rect's circumcircle
                        const double ccx = r1.x + r1.w / 2, ccy = r1.y + r1.h / 2, rad = 0.5 * pg_sqrt(r1.w * r1.w + r1.h * r1.h) + 2;
                        int bx1 = (int)pg_floor(ccx - rad), bx2 = (int)pg_ceil(ccx + rad), by1 = (int)pg_floor(ccy - rad), by2 = (int)pg_ceil(ccy + rad);
                        if (bx1 < 0) bx1 = 0;
                        if (by1 < 0) by1 = 0;
                        if (bx2 > RES_W) bx2 = RES_W;
                        if (by2 > RES_H) by2 = RES_H;
                        if (bx2 > bx1 && by2 > by1) {
                            PG_LV(r.geom, l) = (uint32_t)bx1 | ((uint32_t)by1 << 7) | ((uint32_t)(bx2 - bx1) << 14) | ((uint32_t)(by2 - by1) << 21);
                            PG_LV(r.basex, l) = (uint32_t)i;
                            PG_LV(r.aux, l) = 1u << 15;
                        }
                    } else cmd_image_rotated(rot_slot, imd, (mm & MF_REFLECTED) != 0, r1, rotation, e_alpha, PG_LV(r.geom, l), PG_LV(r.basex, l), PG_LV(r.srcy, l), PG_LV(r.ix, l), PG_LV(r.iy, l), PG_LV(r.src, l), PG_LV(r.aux, l));
                }
            }
        }
        if constexpr (GameUsesRotation<Game>::value) PG_SYNC();  // the rotated commands' LDS records are read by every lane
        return pg_popc64(rotmask);
    }
    // More than 64 entities: most frames still show fewer than 128 of them (fruitbot keeps a whole level of
    // objects alive, a dozen on screen), so the commands that draw something are packed, in draw order, into two register
    // sets through the idle band buffer.  false: too many visible entities or turned sprites; draw_entities() then sets
    // every chunk up again for every band and layer.
    static constexpr int CMD_SETS = GameRenderCmdSets<Game>::value;
    static_assert(CMD_SETS == 1 || CMD_SETS == 2, "one or two register sets");
    PG_DEV bool compact_entities(CmdRegs (&er)[CMD_SETS], uint64_t (&ezm)[CMD_SETS][3]) {
        uint32_t *stage = fb;  // [8 (GEN: 10) words][64 * CMD_SETS slots]
        constexpr int SLOTS = 64 * CMD_SETS;
        static_assert(8 * SLOTS <= BAND_ROWS * RES_W + 64, "staging area");
        if constexpr (GEN && 10 * SLOTS > BAND_ROWS * RES_W + 64) return false;  // (two sets of ten-word commands do not fit: draw_entities() sets up per band)
        const int n = G.n_ents;
        int total = 0, rot_count = 0;
        for (int base = 0; base < n; base += 64) {
            CmdRegs r;
            uint64_t zm[3];
            const int nrot = setup_entities(base, r, zm, rot_count);
            if (nrot < 0) return false;
            rot_count += nrot;
            const uint64_t vis = PG_BALLOT(l, PG_LV(r.geom, l) != 0);
            const int cnt = pg_popc64(vis);
            if (total + cnt > SLOTS) return false;
            PG_R_LANES(l) {
                if ((vis >> l) & 1ull) {
                    const int slot = total + pg_popc64(vis & pg_mask_lt(l));
                    stage[0 * SLOTS + slot] = PG_LV(r.geom, l);
                    stage[1 * SLOTS + slot] = PG_LV(r.basex, l);
                    stage[2 * SLOTS + slot] = PG_LV(r.srcy, l);
                    stage[3 * SLOTS + slot] = PG_LV(r.ix, l);
                    stage[4 * SLOTS + slot] = PG_LV(r.iy, l);
                    stage[5 * SLOTS + slot] = PG_LV(r.src, l);
                    stage[6 * SLOTS + slot] = PG_LV(r.aux, l);
                    stage[7 * SLOTS + slot] = ((zm[0] >> l) & 1ull) ? 0u : (((zm[1] >> l) & 1ull) ? 1u : 2u);
                    if constexpr (GEN) {
                        stage[8 * SLOTS + slot] = PG_LV(r.e0, l);
                        stage[9 * SLOTS + slot] = PG_LV(r.e1, l);
                    }
                }
            }
            total += cnt;
        }
        PG_SYNC();
        _Pragma("unroll") for (int k = 0; k < CMD_SETS; k++) {
            PG_R_LANES(l) {
                const int idx = k * 64 + l;
                const bool in = idx < total;
                const int si = in ? idx : 0;
                PG_LV(er[k].geom, l) = in ? stage[0 * SLOTS + si] : 0u;
                PG_LV(er[k].basex, l) = in ? stage[1 * SLOTS + si] : 0u;
                PG_LV(er[k].srcy, l) = in ? stage[2 * SLOTS + si] : 0u;
                PG_LV(er[k].ix, l) = in ? stage[3 * SLOTS + si] : 0u;
                PG_LV(er[k].iy, l) = in ? stage[4 * SLOTS + si] : 0u;
                PG_LV(er[k].src, l) = in ? stage[5 * SLOTS + si] : 0u;
                PG_LV(er[k].aux, l) = in ? stage[6 * SLOTS + si] : 0u;
                if constexpr (GEN) {
                    PG_LV(er[k].e0, l) = in ? stage[8 * SLOTS + si] : 0u;
                    PG_LV(er[k].e1, l) = in ? stage[9 * SLOTS + si] : 0u;
                }
            }
            _Pragma("unroll") for (int z = 0; z < 3; z++) ezm[k][z] = PG_BALLOT(l, (k * 64 + l) < total && stage[7 * SLOTS + k * 64 + l] == (uint32_t)z);
        }
        PG_SYNC();
        return true;
    }
    PG_DEV void draw_entities(int render_z) {  // last resort: one set-up per band, layer and chunk
        const int n = G.n_ents;
        for (int base = 0; base < n; base += 64) {
            const uint64_t any = PG_BALLOT(l, (base + l) < n && meta_render_z(meta(base + l)) == render_z);
            if (!any) continue;
            CmdRegs r;
            uint64_t zmask[3];
            if constexpr (ROT_POOLED) {
                // windows of the chunk, in draw order, each with at most ROT_POOL turned / tiled entities that can reach this band
                EntPre none;
                int lo = 0;
                while (lo < 64) {
                    int next = 64;
                    setup_entities_t<false>(base, r, zmask, 0, none, lo ? ~pg_mask_lt(lo) : ~0ull, &next);
                    run_batch(r, zmask[render_z + 1]);
                    lo = next;
                }
            } else {
                setup_entities(base, r, zmask);
                run_batch(r, zmask[render_z + 1]);
            }
        }
    }

    // game_draw BAG:1009-1012 (draw_background BAG:979-1007 + draw_foreground BAG:921-970)
    // profiling aid (PROCGEN_AMD_DEBUG & 2048): wave cycles since the previous mark go to slot 16 + k of the table
    long long t_mark = 0;
    PG_DEV void phase(int k) {
#if !defined(PGAMD_WAVE_EMU)
        if (PG_FPHASES(d)) {
            const long long t = (long long)__builtin_readcyclecounter();
            if (PG_LANE_ID() == 0 && t_mark != 0) atomicAdd(d.phase_cycles + 32 * (env & 4095) + 16 + k, (unsigned long long)(t - t_mark));
            t_mark = (long long)__builtin_readcyclecounter();
        }
#else
        (void)k;
#endif
    }
    // the solid-colour cells of a pull-form frame (build_pull_tables left their commands in fillcmd)
    PG_DEV void draw_pull_fills(int nfill) {
        for (int base = 0; base < nfill; base += 64) {
            CmdRegs r;
            PG_R_LANES(l) {
                const bool in = base + l < nfill;
                const int si = 2 * (in ? base + l : 0);
                PG_LV(r.geom, l) = in ? lds->fillcmd[si] : 0u;
                PG_LV(r.src, l) = in ? lds->fillcmd[si + 1] : 0u;
                PG_LV(r.aux, l) = cmd_aux(1, false, true, 256) | (1u << 26);
                PG_LV(r.basex, l) = PG_LV(r.srcy, l) = PG_LV(r.ix, l) = PG_LV(r.iy, l) = 0;
                if constexpr (GEN) PG_LV(r.e0, l) = PG_LV(r.e1, l) = 0;
            }
            run_batch(r);
        }
    }
    PG_DEV void render_env() {
#if !defined(PGAMD_WAVE_EMU)
        if (PG_FPHASES(d)) t_mark = (long long)__builtin_readcyclecounter();
#endif
        // ---- requests, round 1: what depends on the env index alone -- the header and the fields of the first 64 entity slots
        EntPre epre;
        request_entity_fields(epre);
        {
            const auto *h = PG_SCALAR_PTR(EnvHdr, d.hdr + env);  // (written by this step's step kernels, read-only here: scalar loads)
#define PG_X(type, name) G.name = h->name;
            PG_HDR_FIELDS(PG_X)
#undef PG_X
        }
        opt = env_options(d.opt, G.opt_bits, G.opt_debug_mode);
        phase(6);
        // ---- frame-level set-up (rows [0, 64)) ------------------------------------------------------------------
        row0 = 0;
        row1 = RES_H;
        if PG_FDBG(d, 4) G.n_ents = 0;
        const bool force_chunks = PG_FDBG(d, 4096) != 0;  // test aid: every frame through draw_entities()
        bool one_chunk = G.n_ents <= 64 && !force_chunks;  // or: the visible ones fit the register sets
        int win_lx, win_hx, win_ly, win_hy;  // BAG:926-939
        if (Game::center_agent(opt)) {
            const float margin = (float)(G.visibility / 2.0 + 1);
            win_lx = (int)(G.center_x - margin);
            win_hx = (int)(G.center_x + margin);
            win_ly = (int)(G.center_y - margin);
            win_hy = (int)(G.center_y + margin);
        } else {
            win_lx = 0;
            win_hx = G.main_width - 1;
            win_ly = 0;
            win_hy = G.main_height - 1;
        }
        const int nx = win_hx - win_lx + 1;
        const int ny_full = win_hy - win_ly + 1;
        const int ref_w = d.assets->ref_w, ref_h = d.assets->ref_h;
        // (GEN: every cell is its own generic drawImage, set up lane-parallel per band: neither axis tables nor the pull form)
        const bool use_axes = !GEN && GameDrawsGrid<Game>::value && nx > 0 && ny_full > 0 && nx <= 32 && ny_full <= 32;
        const bool try_pull = GameDrawsGrid<Game>::value && !GEN && use_axes && nx * ny_full <= GamePullCells<Game>::value && !PG_FDBG(d, 1024);
        // ---- requests, round 2: every table lookup of the set-up, in flight together: the background image, the images of the
        // entities (lane = slot) and of the grid object types (lane = type), the first 256 window cells
        const ImgDesc bg_desc = (opt.use_backgrounds && !PG_FDBG(d, 1)) ? d.assets->bg_desc[G.background_index] : ImgDesc{IMG_NONE, 0, 0, 0};
        PG_LANE_VAR(ImgDesc, type_desc);
        PG_LANE_ARR(int, cells0, 4);
        PG_R_LANES(l) {
            PG_LV(epre.desc, l) = request_desc(meta_image_type(PG_LV(epre.meta, l)), meta_image_theme(PG_LV(epre.meta, l)));
            PG_LV(type_desc, l) = request_type_desc(l);
            for (int q = 0; q < 4; q++) PG_LA(cells0, q, l) = 0;
        }
        if (try_pull) request_window_cells(win_lx, nx, win_ly, ny_full, cells0);
        // wave-uniform background commands (one scaled image, or the visible tiles of a game's own background)
        // (one command unless the game tiles its background or draws its own: the slots are scalar registers held across the whole frame)
        constexpr int MAXBG = (GameTiledBackground<Game>::value || GameCustomBackground<Game>::value) ? 3 : 1;
        uint32_t bg_geom[MAXBG] = {0}, bg_basex[MAXBG] = {0}, bg_srcy[MAXBG] = {0}, bg_ix[MAXBG] = {0}, bg_iy[MAXBG] = {0}, bg_src[MAXBG] = {0}, bg_aux[MAXBG] = {0};
        int nbg = 0;
        bool bgt_many = false;  // tiled background with more tiles on screen than the register slots above
        double bgt_x = 0, bgt_y = 0, bgt_w = 0;
        float bgt_h = 0;
        int bgt_ia = 0, bgt_ib = 0;
        ImgDesc bgt_img = {0, 0, 0, 0};
        auto add_bg = [&](const ImgDesc &bgi, const RectD &rc) {
            uint32_t g, bx, sy, ix, iy, sr, au;
            cmd_image(bgi, false, rc, 1.0f, g, bx, sy, ix, iy, sr, au);  // RGB32 source: the scale path also with generated assets
            if (GEN) au |= 1u << 27;
            if (g != 0) {
                if (nbg >= MAXBG) {
                    fail(PGE_UNSUPPORTED_DRAW);
                    return;
                }
                bg_geom[nbg] = g; bg_basex[nbg] = bx; bg_srcy[nbg] = sy; bg_ix[nbg] = ix; bg_iy[nbg] = iy; bg_src[nbg] = sr; bg_aux[nbg] = au;
                nbg++;
            }
        };
        if constexpr (GameCustomBackground<Game>::value) {
            if (opt.use_backgrounds && !PG_FDBG(d, 1)) {
                RectD rects[4];
                const int nr = Game::background_rects(*this, rects);
                const ImgDesc bgi = bg_desc;
                _Pragma("unroll") for (int k = 0; k < 4; k++)
                    if (k < nr && rects[k].w > 0) add_bg(bgi, rects[k]);
            }
        } else if (opt.use_backgrounds && !PG_FDBG(d, 1)) {
            const RectD main_rect = get_screen_rect(0, (float)G.main_height, (float)G.main_width, (float)G.main_height, 0);
            const ImgDesc bgi = bg_desc;
            if (!GameTiledBackground<Game>::value && G.bg_tile_ratio < 0) fail(PGE_UNSUPPORTED_DRAW);  // (only fruitbot's constructor sets it)
            if (GameTiledBackground<Game>::value && G.bg_tile_ratio < 0) {  // tile_image(main_rect, bg_tile_ratio) BAG:842-853: a column of tiles; only the ones on screen
                const float tile_ratio = -1 * G.bg_tile_ratio;
                int num_tiles = (int)(main_rect.h / (main_rect.w * (double)tile_ratio));
                if (num_tiles < 1) num_tiles = 1;
                const float tile_height = (float)(main_rect.h / num_tiles);
                const float tile_width = (float)main_rect.w;
                const int i0 = (int)pg_floor(-main_rect.y / (double)tile_height) - 1;
                if constexpr (GameTiledBackground<Game>::value) {
                    int ia = i0 < 0 ? 0 : i0, ib = (int)pg_ceil(((double)RES_H - main_rect.y) / (double)tile_height) + 1;
                    if (ib > num_tiles) ib = num_tiles;
                    if (ib - ia > 4) {
                        // a tall world shown whole (fruitbot without center_agent): a dozen tiles on screen.  Their commands
                        // are set up lane-parallel in every band pass instead of living in registers across the passes.
                        bgt_many = true;
                        bgt_x = main_rect.x;
                        bgt_y = main_rect.y;
                        bgt_w = (double)tile_width;
                        bgt_h = tile_height;
                        bgt_ia = ia;
                        bgt_ib = ib;
                        bgt_img = bgi;
                    }
                }
                for (int i = i0; i < i0 + 4 && !bgt_many; i++) {
                    if (i < 0 || i >= num_tiles) continue;
                    const RectD tr = {main_rect.x, main_rect.y + (double)(tile_height * i), (double)tile_width, (double)tile_height};
                    add_bg(bgi, tr);
                }
            } else {
            const ImgDesc bim = bgi;
            const float bgw = (float)bim.w, bgh = (float)bim.h;
            const float bg_ar = bgw / bgh;
            const float world_ar = (float)(G.main_width * 1.0 / G.main_height);
            const float extra_w = bg_ar - world_ar;
            const float offset_x = G.bg_pct_x * extra_w;
            const RectD bg_rect = adjust_rect(main_rect, (double)(-offset_x), 0, (double)(bg_ar / world_ar), 1);
            add_bg(bgi, bg_rect);
            }
        }
        phase(7);
        int ix_ref = 0, iy_ref = 0;
        uint64_t colseam = 0, rowseam = 0, rowany = ~0ull;
        phase(9);
        build_type_table(type_desc);
        phase(10);
        bool pull = false, pull_multi = false;
        int pull_nfill = 0;
        if constexpr (GameDrawsGrid<Game>::value)
            pull = try_pull && build_pull_tables(win_lx, nx, win_ly, ny_full, colseam, rowseam, rowany, pull_multi, pull_nfill, cells0);

        if (use_axes && !pull) setup_tile_axes(win_lx, nx, win_ly, ny_full, ref_w, ref_h, ix_ref, iy_ref);  // only the per-cell path reads the axis tables

        // common case (<= 64 entities): their commands are built once and kept in registers for all passes
        CmdRegs er[CMD_SETS];
        uint64_t ezmask[CMD_SETS][3];
        _Pragma("unroll") for (int k = 0; k < CMD_SETS; k++) {
            ezmask[k][0] = ezmask[k][1] = ezmask[k][2] = 0;
            PG_R_LANES(l) { clear_cmd(er[k], l); }
        }
        if (one_chunk) {
            const int k = setup_entities_t<true>(0, er[0], ezmask[0], -1, epre);
            if (ROT_POOLED && k < 0) one_chunk = false;  // more turned / tiled sprites on screen than records: per band (draw_entities)
#if defined(PGAMD_WAVE_EMU)
            if (ROT_POOLED && k < 0) pg_emu_counters()[7] += 1000000;  // frames sent to the per-band path for want of records
#endif
        } else if (!force_chunks) one_chunk = compact_entities(er, ezmask);
        phase(8);
        phase(0);
        // ---- passes -------------------------------------------------------------------------------------------------
        for (int band = 0; band < NUM_BANDS; band++) {
            row0 = band * BAND_ROWS;
            row1 = row0 + BAND_ROWS;
            // a single background image that reaches every pixel of the band needs no black underneath (p.fillRect(rect, QColor(0,0,0)))
            const DrawCmd bc0 = unpack(bg_geom[0], bg_basex[0], bg_srcy[0], bg_ix[0], bg_iy[0], bg_src[0], bg_aux[0]);
            const bool bg_dma = nbg == 1 && !bgt_many && bg_dma_ok(bc0);
            const bool bg_full = bg_dma && bc0.tx1 == 0 && bc0.w == RES_W && bc0.ty1 <= row0 && bc0.ty1 + bc0.h >= row1;
            if (!bg_full) {
                for (int base = 0; base < BAND_ROWS * RES_W; base += 64) {
                    PG_R_LANES(l) { fb[base + l] = 0xff000000u; }  // p.fillRect(rect, QColor(0,0,0))
                }
            }
            PG_SYNC();
            if (bg_dma) {
                if (bc0.ty1 < row1 && bc0.ty1 + bc0.h > row0) exec_bg_dma(bc0);
            } else
            {
                // background commands of this pass: the frame's register slots, or -- with more tiles on screen than slots --
                // the tiles that can touch these rows, each set up here (wave-uniform) and fed to the same blit
                int bg_first = 0, bg_count = nbg;
                if constexpr (GameTiledBackground<Game>::value) {
                    if (bgt_many) {
                        bg_first = (int)pg_floor(((double)row0 - bgt_y) / (double)bgt_h) - 1;
                        int last = (int)pg_ceil(((double)row1 - bgt_y) / (double)bgt_h) + 1;
                        if (bg_first < bgt_ia) bg_first = bgt_ia;
                        if (last > bgt_ib) last = bgt_ib;
                        bg_count = last - bg_first;
                    }
                }
                for (int k = 0; k < bg_count; k++) {
                    uint32_t g = bg_geom[k < MAXBG ? k : 0], bx = bg_basex[k < MAXBG ? k : 0], sy = bg_srcy[k < MAXBG ? k : 0], ix = bg_ix[k < MAXBG ? k : 0],
                             iy = bg_iy[k < MAXBG ? k : 0], sr = bg_src[k < MAXBG ? k : 0], au = bg_aux[k < MAXBG ? k : 0];
                    if constexpr (GameTiledBackground<Game>::value) {
                        if (bgt_many) {
                            const RectD tr = {bgt_x, bgt_y + (double)(bgt_h * (bg_first + k)), bgt_w, (double)bgt_h};
                            cmd_image(bgt_img, false, tr, 1.0f, g, bx, sy, ix, iy, sr, au);
                            if (GEN) au |= 1u << 27;
                        }
                    }
                    const DrawCmd bc = unpack(g, bx, sy, ix, iy, sr, au);
                    if (g != 0 && bc.ty1 < row1 && bc.ty1 + bc.h > row0) {
                        // several background images (fruitbot's column of tiles, starpilot's own rects) go by LDS-DMA as well; two of them
                        // may share a pixel row (each is rounded on its own), so all but the last are joined before the next is issued
                        if (bg_dma_ok(bc)) {
                            exec_bg_dma(bc);
                            if (k + 1 < bg_count) dma_join();
                        } else {
                            exec_large(bc);
                        }
                    }
                }
            }
            phase(1);
            if (one_chunk) {
                // constant indices only: a loop the compiler declines to unroll would index the register sets through scratch memory
                if (ezmask[0][0]) run_batch(er[0], ezmask[0][0]);
                if constexpr (CMD_SETS > 1)
                    if (ezmask[CMD_SETS - 1][0]) run_batch(er[CMD_SETS - 1], ezmask[CMD_SETS - 1][0]);
            } else {
                draw_entities(-1);
            }
            // only cell rows whose (inflated) rect can reach these rows: screen y falls as cell y grows.  Conservative
            // by a full cell either side; cells outside the range draw nothing here, and dropping them keeps the
            // x-major order of the rest (BAG:941-955).
            int low_y = win_ly, ny = 0, ncell = 0;
            uint32_t ny_inv = 0;
            const int low_x = win_lx;
            if (GameDrawsGrid<Game>::value && !(pull || PG_FDBG(d, 2))) {  // (the pull form walks screen rows, not cells)
                int high_y = win_hy;
                const float inv_unit = 1.0f / G.unit;
                const int cy_hi = (int)pg_ceil((double)(G.view_dim - ((float)row0 - G.y_off) * inv_unit)) + 1;
                const int cy_lo = (int)pg_floor((double)(G.view_dim - ((float)row1 - G.y_off) * inv_unit)) - 2;
                if (cy_lo > low_y) low_y = cy_lo;
                if (cy_hi < high_y) high_y = cy_hi;
                ny = high_y - low_y + 1;
                ncell = (ny > 0 && nx > 0) ? nx * ny : 0;
                ny_inv = ny > 0 ? (uint32_t)(((1u << 20) + (uint32_t)ny - 1u) / (uint32_t)ny) : 0u;
                if (ncell > 4096) fail(PGE_ASSERT);
            }
            phase(2);
            if constexpr (GameDrawsGrid<Game>::value)
                if (pull && !PG_FDBG(d, 2)) {
                    if (pull_multi) draw_tiles_pull<true>(ny_full, colseam, rowseam, rowany, ref_w);
                    else draw_tiles_pull<false>(ny_full, colseam, rowseam, rowany, ref_w);
                    if constexpr (GameHasGridFills<Game>::value) draw_pull_fills(pull_nfill);
                }
            for (int base = 0; base < ncell; base += 64) {
                CmdRegs r;
                PG_R_LANES(l) {
                    const int cidx = base + l;
                    clear_cmd(r, l);
                    if (cidx < ncell) {
                        const int cx = (int)(((uint32_t)cidx * ny_inv) >> 20);  // cidx / ny (exact for cidx < 4096)
                        const int cy = cidx - cx * ny;
                        const int x = low_x + cx, y = low_y + cy;
                        const int type = get_obj(x, y);
                        bool is_fill = false;
                        if constexpr (GameHasGridFills<Game>::value) is_fill = Game::is_grid_fill(*this, type);
                        const uint32_t tv = (type >= 0 && type < 64) ? lds->typeimg[type] : TYPE_SLOW;
                        if (tv != TYPE_SLOW && use_axes) {
                            if (tv != CELL_NONE) {  // reference-size image on the unadjusted rect: geometry from the axis tables
                                const int ry = y - win_ly;
                                const uint32_t px = ax[cx], py = ax[32 + ry];
                                if ((px >> 16) && (py >> 16)) {
                                    const int ty1 = (int)(py & 0xffu), h = (int)((py >> 8) & 0xffu);
                                    if (!(ty1 >= row1 || ty1 + h <= row0)) {
                                        PG_LV(r.geom, l) = (px & 0xffu) | ((uint32_t)ty1 << 7) | (((px >> 8) & 0xffu) << 14) | ((uint32_t)h << 21);
                                        PG_LV(r.basex, l) = ax[64 + cx];
                                        PG_LV(r.srcy, l) = ax[64 + 32 + ry];
                                        PG_LV(r.ix, l) = (uint32_t)ix_ref;
                                        PG_LV(r.iy, l) = (uint32_t)iy_ref;
                                        PG_LV(r.src, l) = tv & 0x7fffffffu;
                                        PG_LV(r.aux, l) = cmd_aux(ref_w, false, (tv >> 31) != 0, 256);
                                    }
                                }
                            }
                        } else if (is_fill) {
                            if constexpr (GameHasGridFills<Game>::value) {  // draw_grid_obj override: p.fillRect(QRectF, QColor)
                                const RectD cell = get_screen_rect((float)x, (float)(y + 1), 1, 1, RENDER_EPS);
                                RectD fr;
                                uint32_t color;
                                Game::grid_fill(*this, type, cell, fr, color);
                                cmd_fill_rect(fr, color, PG_LV(r.geom, l), PG_LV(r.src, l), PG_LV(r.aux, l));
                            }
                        } else if (type != INVALID_OBJ && type != SPACE) {
                            const int theme = Game::theme_for_grid_obj(*this, type);
                            RectD r2 = get_screen_rect((float)x, (float)(y + 1), 1, 1, RENDER_EPS);
                            const RectD r2_in = r2;
                            uint32_t fc = 0;
                            ImgDesc imd;
                            const int im = resolve_image(type, theme, 0.0f, 0.0f, r2, &fc, &imd);
                            if (im == IMG_FILL) cmd_fill_rect(r2_in, fc, PG_LV(r.geom, l), PG_LV(r.src, l), PG_LV(r.aux, l));
                            if (im >= 0) {
                                const bool same_rect = r2.x == r2_in.x && r2.y == r2_in.y && r2.w == r2_in.w && r2.h == r2_in.h;
                                if (use_axes && same_rect && (int)imd.w == ref_w && (int)imd.h == ref_h) {
                                    const int ry = y - win_ly;  // row index in the frame-level axis table
                                    const uint32_t px = ax[cx], py = ax[32 + ry];
                                    if ((px >> 16) && (py >> 16)) {
                                        const int ty1 = (int)(py & 0xffu), h = (int)((py >> 8) & 0xffu);
                                        if (!(ty1 >= row1 || ty1 + h <= row0)) {
                                            PG_LV(r.geom, l) = (px & 0xffu) | ((uint32_t)ty1 << 7) | (((px >> 8) & 0xffu) << 14) | ((uint32_t)h << 21);
                                            PG_LV(r.basex, l) = ax[64 + cx];
                                            PG_LV(r.srcy, l) = ax[64 + 32 + ry];
                                            PG_LV(r.ix, l) = (uint32_t)ix_ref;
                                            PG_LV(r.iy, l) = (uint32_t)iy_ref;
                                            PG_LV(r.src, l) = imd.off;
                                            PG_LV(r.aux, l) = cmd_aux((int)imd.w, false, imd.opaque != 0, 256);
                                        }
                                    }
                                } else {
                                    emit_image(r, l, imd, false, r2, 1.0f);
                                }
                            }
                        }
                    }
                }
                run_batch(r);
            }
            phase(3);
            if (one_chunk) {
                if constexpr (CMD_SETS == 1) {
                    if (ezmask[0][1] | ezmask[0][2]) run_batch(er[0], ezmask[0][1], ezmask[0][2]);  // z = 0, then z = 1, in one pass
                } else {
                    if (ezmask[0][1]) run_batch(er[0], ezmask[0][1]);
                    if (ezmask[CMD_SETS - 1][1]) run_batch(er[CMD_SETS - 1], ezmask[CMD_SETS - 1][1]);
                    if (ezmask[0][2]) run_batch(er[0], ezmask[0][2]);
                    if (ezmask[CMD_SETS - 1][2]) run_batch(er[CMD_SETS - 1], ezmask[CMD_SETS - 1][2]);
                }
            } else {
                draw_entities(0);
                draw_entities(1);
            }
            if (G.has_useful_vel_info && opt.paint_vel_info) {  // BAG:960-969, to_shade reference src/qt-utils.h:21-28
                const float infodim = (float)(RES_H * .2);
                const int ag = G.agent;
                int s1 = (int)((float)(.5 * (double)evx(ag) / (double)G.maxspeed + .5) * 255);
                int s2 = (int)((float)(.5 * (double)evy(ag) / (double)G.max_jump + .5) * 255);
                s1 = s1 < 0 ? 0 : (s1 > 255 ? 255 : s1);
                s2 = s2 < 0 ? 0 : (s2 > 255 ? 255 : s2);
                const RectD d2 = {0, 0, (double)infodim, (double)infodim}, d3 = {(double)infodim, 0, (double)infodim, (double)infodim};
                exec_fill(d2, 0xff000000u | ((uint32_t)s1 << 16) | ((uint32_t)s1 << 8) | (uint32_t)s1);
                exec_fill(d3, 0xff000000u | ((uint32_t)s2 << 16) | ((uint32_t)s2 << 8) | (uint32_t)s2);
            }
            if constexpr (GameHasOverlay<Game>::value) Game::draw_overlay(*this);  // game_draw overrides that paint after the base frame
            PG_SYNC();
            phase(4);
            if (!PG_FDBG(d, 8)) store_band();
            PG_SYNC();
            phase(5);
        }
#if !defined(PGAMD_WAVE_EMU)
        if (PG_FPHASES(d) && PG_LANE_ID() == 0) atomicAdd(d.phase_cycles + 32 * (env & 4095) + 16 + 15, 1ull);
#else
        if (pg_emu_dma_outstanding() != 0) {  // (a band whose background copies nobody joined: store_band always does)
            fprintf(stderr, "render_env: %d LDS-DMA words still in flight at the end of the frame\n", pg_emu_dma_outstanding());
            abort();
        }
#endif
        if (G.error) {
#if defined(PGAMD_WAVE_EMU)
            pg_report_error(d, env, G.error, ERR_KIND_RENDER, 0, 0);
#else
            if (PG_LANE_ID() == 0) pg_report_error(d, env, G.error, ERR_KIND_RENDER, 0, 0);
#endif
        }
    }

    // The frame of a display-list game from its record (FrameRec, written by pg_prep.h's FramePrep::run): the rasterizer proper.  No header,
    // no options, no fp64: the record's scalars by scalar loads, one entity command per lane, the pull form's tables by LDS-DMA straight into
    // the arena, then the band passes of render_env for a frame whose commands fit one register set and whose grid is drawn in pull form
    // (every other frame is queued for render_env by the prep kernel).  Same painter's order: background, z = -1, grid cells, z = 0, z = 1
    // (BAG:979-1012, 921-970).
    // commands [SET set, SET set + SET) of a frame record, one per lane (z: the command's layer, render_z + 1; 255 = no command).  SET = 64,
    // the lanes of a register set; -DPG_CMD_SET_LANES=8 is the emulation tests' way to send ordinary frames down the several-sets path
#ifndef PG_CMD_SET_LANES
#define PG_CMD_SET_LANES 64
#endif
    static constexpr int CMD_SET = PG_CMD_SET_LANES;
    PG_DEV void load_cmd_set(const uint32_t *rec, int set, int ncmd, CmdRegs &r, PG_LANE_REF(uint32_t, z)) {
        typedef FrameRec<Game> Rec;
        PG_R_LANES(l) {
            const int k = set * CMD_SET + l;
            const bool in = l < CMD_SET && k < ncmd;
            const pg_u4 *c = reinterpret_cast<const pg_u4 *>(rec + Rec::CMD + Rec::CMD_WORDS * (in ? k : 0));
            const pg_u4 a = c[0], b = c[1];
            PG_LV(r.geom, l) = in ? a.x : 0u;
            PG_LV(r.basex, l) = in ? a.y : 0u;
            PG_LV(r.srcy, l) = in ? a.z : 0u;
            PG_LV(r.ix, l) = in ? a.w : 0u;
            PG_LV(r.iy, l) = in ? b.x : 0u;
            PG_LV(r.src, l) = in ? b.y : 0u;
            PG_LV(r.aux, l) = in ? b.z : 0u;
            if constexpr (GEN) PG_LV(r.e0, l) = PG_LV(r.e1, l) = 0;
            PG_LV(z, l) = in ? b.w : 255u;
        }
    }
    // layers zfirst..zlast of a record's commands, each layer in draw order over all register sets.  The first set lives in registers (er, ez)
    // for the whole frame, the others -- 1 frame in 10 000 has any -- are fetched again for every band and layer.  One routine for both so that
    // run_batch is inlined twice in the kernel, not seven times (seven copies doubled its size and its SGPR spills: +750 vector instructions a
    // frame of spill traffic, profiles/r06_final_sq_insts.csv vs r06_call8_11_ab.txt).
    // (nsets, not the record pointer and the command count: those two would stay live in scalar registers across the whole band loop for the
    // sake of this rare path -- and the grid pass in between pays for every live scalar with v_readlane spill traffic: +700 vector
    // instructions a frame, profiles/r06_call19_ablation.txt -- so the rare path works the pointer out again from the env)
    PG_DEV void draw_cmd_layers(int nsets, const CmdRegs &er, PG_LANE_REF(const uint32_t, ez), uint32_t zfirst, uint32_t zlast) {
        typedef FrameRec<Game> Rec;
        for (uint32_t layer = zfirst; layer <= zlast; layer++) {
            for (int set = 0; set < nsets; set++) {
                CmdRegs r;
                PG_LANE_VAR(uint32_t, z);
                const uint32_t *rec = nullptr;
                int ncmd = 0;
                if (set > 0) {
                    rec = d.frame_rec + (size_t)PG_UNIFORM_I(pg_opaque_i(env)) * Rec::WORDS;
                    ncmd = (int)(((uint32_t)PG_UNIFORM_I(rec[Rec::DIMS]) >> 8) & 0xffu);
                }
                if (set == 0) {
                    PG_R_LANES(l) {
                        PG_LV(r.geom, l) = PG_LV(er.geom, l);
                        PG_LV(r.basex, l) = PG_LV(er.basex, l);
                        PG_LV(r.srcy, l) = PG_LV(er.srcy, l);
                        PG_LV(r.ix, l) = PG_LV(er.ix, l);
                        PG_LV(r.iy, l) = PG_LV(er.iy, l);
                        PG_LV(r.src, l) = PG_LV(er.src, l);
                        PG_LV(r.aux, l) = PG_LV(er.aux, l);
                        if constexpr (GEN) PG_LV(r.e0, l) = PG_LV(r.e1, l) = 0;
                        PG_LV(z, l) = PG_LV(ez, l);
                    }
                } else {
                    load_cmd_set(rec, set, ncmd, r, z);
                }
                const uint64_t m = PG_BALLOT(l, PG_LV(z, l) == layer);
                if (m) run_batch(r, m);
            }
        }
    }
    PG_DEV void raster_env() {
        typedef FrameRec<Game> Rec;
        static_assert(!GEN, "generated assets draw through render_env");
        static_assert(Rec::HDR_WORDS <= 64, "the header's words live one per lane");
        const uint32_t *rec = d.frame_rec + (size_t)env * Rec::WORDS;
        // The header lives in ONE vector register, word k in lane k, and every use takes its word with a v_readlane where it is needed.
        // Held in scalar registers for the frame -- 16 words beside the band loop's own scalars -- the compiler spilled them and restored
        // whole tuples of them in every basic block of the grid pass: ~10 v_readlane per pixel row, +800 vector instructions a frame
        // (profiles/r06_call19_ablation.txt).  hdr_word(k) goes through an opaque copy per band, so the reads stay where they are written.
        PG_LANE_VAR(uint32_t, hv);
        PG_R_LANES(l) { PG_LV(hv, l) = rec[l < Rec::HDR_WORDS ? l : 0]; }
        const uint32_t flags = PG_READLANE(hv, Rec::FLAGS), dims = PG_READLANE(hv, Rec::DIMS);
        const int ncmd = (int)((dims >> 8) & 0xffu);
        const bool pull = (flags & Rec::F_PULL) != 0, multi = (flags & Rec::F_MULTI) != 0;
        G.error = 0;
        constexpr bool REG = GameRasterRegTabs<Game>::value;
        if constexpr (GameRasterBandOverTables<Game>::value) fb = &lds->ci[0][0];
        if constexpr (REG) {
            // register tables: the band buffer lies over ci / ri / typeany, which this kernel's arena does not hold as tables (RenderLdsT);
            // those come from the record one word per lane, the words in front of them (seam columns, cell types, fills) by LDS-DMA
            static_assert(GameRasterBandOverTables<Game>::value, "a register-table rasterizer's band buffer lies over the tables");
            PG_R_LANES(l) {
                const uint32_t *t = rec + Rec::TAB;
                PG_LV(tr_ci0, l) = pull ? t[Rec::TAB_CI + l] : 0u;
                PG_LV(tr_ci1, l) = pull ? t[Rec::TAB_CI + 64 + l] : 0u;
                PG_LV(tr_ri0, l) = pull ? t[Rec::TAB_RI + l] : 0u;
                PG_LV(tr_ri1, l) = pull ? t[Rec::TAB_RI + 64 + l] : 0u;
                PG_LV(tr_ty, l) = pull ? t[Rec::TAB_TYPEANY + l] : 0u;
            }
        }
        if constexpr (GameDrawsGrid<Game>::value) {
            if (pull) {
                uint32_t *tab = reinterpret_cast<uint32_t *>(lds) + Rec::LDS_TAB_WORD0;
                const int words = REG ? Rec::TAB_CI : multi ? Rec::TAB_WORDS : Rec::TAB_SINGLE_WORDS;
                for (int k0 = 0; k0 < words; k0 += 64) {
                    PG_R_LANES(l) {
                        if (k0 + l < words) PG_DMA_DWORD(rec + Rec::TAB + k0 + l, tab + k0, l);
                    }
                }
            }
        }
        CmdRegs er;
        PG_LANE_VAR(uint32_t, ez);
        load_cmd_set(rec, 0, ncmd, er, ez);  // the frame's first 64 commands stay in registers for all passes
        const int nsets = (ncmd + CMD_SET - 1) / CMD_SET > 0 ? (ncmd + CMD_SET - 1) / CMD_SET : 1;
        const bool any0 = PG_BALLOT(l, PG_LV(ez, l) == 0u) != 0 || nsets > 1, any12 = PG_BALLOT(l, PG_LV(ez, l) == 1u || PG_LV(ez, l) == 2u) != 0 || nsets > 1;
        dma_join();
        PG_SYNC();
        for (int band = 0; band < NUM_BANDS; band++) {
            row0 = band * BAND_ROWS;
            row1 = row0 + BAND_ROWS;
            PG_LANE_VAR(uint32_t, hb);  // this band's view of the header
            PG_R_LANES(l) { PG_LV(hb, l) = (uint32_t)pg_opaque_i((int)PG_LV(hv, l)); }
            {
                const uint32_t bg_geom = PG_READLANE(hb, Rec::BG);
                const DrawCmd bc0 = unpack(bg_geom, PG_READLANE(hb, Rec::BG + 1), PG_READLANE(hb, Rec::BG + 2), PG_READLANE(hb, Rec::BG + 3), PG_READLANE(hb, Rec::BG + 4),
                                           PG_READLANE(hb, Rec::BG + 5), PG_READLANE(hb, Rec::BG + 6));
                const bool bg_dma = bg_geom != 0 && bg_dma_ok(bc0);
                // a background image that reaches every pixel of the band needs no black underneath (p.fillRect(rect, QColor(0,0,0)))
                const bool bg_full = bg_dma && bc0.tx1 == 0 && bc0.w == RES_W && bc0.ty1 <= row0 && bc0.ty1 + bc0.h >= row1;
                if (!bg_full) {
                    for (int base = 0; base < BAND_ROWS * RES_W; base += 64) {
                        PG_R_LANES(l) { fb[base + l] = 0xff000000u; }
                    }
                }
                PG_SYNC();
                if (bg_geom != 0 && bc0.ty1 < row1 && bc0.ty1 + bc0.h > row0 && !PG_FDBG(d, 1)) {
                    if (bg_dma) exec_bg_dma(bc0);
                    else exec_large(bc0);
                }
            }
            if (any0 && !PG_FDBG(d, 4)) draw_cmd_layers(nsets, er, ez, 0u, 0u);
            if constexpr (GameDrawsGrid<Game>::value) {
                if (pull && !PG_FDBG(d, 2)) {
                    const uint32_t dm = PG_READLANE(hb, Rec::DIMS);
                    const int ny_full = (int)(dm & 0xffu), ref_w = (int)PG_READLANE(hb, Rec::REF_W);
                    const uint64_t colseam = (uint64_t)PG_READLANE(hb, Rec::COLSEAM) | ((uint64_t)PG_READLANE(hb, Rec::COLSEAM + 1) << 32);
                    const uint64_t rowseam = (uint64_t)PG_READLANE(hb, Rec::ROWSEAM) | ((uint64_t)PG_READLANE(hb, Rec::ROWSEAM + 1) << 32);
                    const uint64_t rowany = (uint64_t)PG_READLANE(hb, Rec::ROWANY) | ((uint64_t)PG_READLANE(hb, Rec::ROWANY + 1) << 32);
                    if constexpr (REG) {
                        draw_tiles_pull<false, true>(ny_full, colseam, rowseam, rowany, ref_w);  // (a frame with a second image size is not on the short path)
                    } else {
                        if (multi) draw_tiles_pull<true>(ny_full, colseam, rowseam, rowany, ref_w);
                        else draw_tiles_pull<false>(ny_full, colseam, rowseam, rowany, ref_w);
                    }
                    if constexpr (GameHasGridFills<Game>::value) draw_pull_fills((int)(dm >> 16));
                }
            }
            if (any12 && !PG_FDBG(d, 4)) draw_cmd_layers(nsets, er, ez, 1u, 2u);  // z = 0, then z = 1
            PG_SYNC();
            if (!PG_FDBG(d, 8)) store_band();
            else dma_join();
            PG_SYNC();
        }
#if defined(PGAMD_WAVE_EMU)
        if (pg_emu_dma_outstanding() != 0) {
            fprintf(stderr, "raster_env: %d LDS-DMA words still in flight at the end of the frame\n", pg_emu_dma_outstanding());
            abort();
        }
#endif
        if (G.error) {
#if defined(PGAMD_WAVE_EMU)
            pg_report_error(d, env, G.error, ERR_KIND_RENDER, 0, 0);
#else
            if (PG_LANE_ID() == 0) pg_report_error(d, env, G.error, ERR_KIND_RENDER, 0, 0);
#endif
        }
    }

    // bgr32_to_rgb888 + the ob write of Game::observe (reference src/game.cpp:8-23,159): 4 pixels -> 3 dwords per
    // lane; each wave-wide store instruction covers 768 contiguous bytes of the observation buffer.
    PG_DEV void store_band() {
        dma_join();
        uint32_t *out = reinterpret_cast<uint32_t *>(d.obs + (size_t)env * OBS_BYTES + (size_t)row0 * RES_W * 3);
        for (int base = 0; base < BAND_ROWS * RES_W; base += 256) {
            PG_R_LANES(l) {
                const uint32_t *p = &fb[base + 4 * l];
                const uint32_t p0 = p[0], p1 = p[1], p2 = p[2], p3 = p[3];
                // bytes: R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3   (pixel word = 0xffRRGGBB, i.e. bytes B G R A): one byte
                // permute per output dword
                uint32_t *o = out + (base / 4) * 3 + 3 * l;
                o[0] = pg_perm(p1, p0, 0x06000102u);
                o[1] = pg_perm(p2, p1, 0x05060001u);
                o[2] = pg_perm(p3, p2, 0x04050600u);
            }
        }
    }
};

}  // namespace pgamd
