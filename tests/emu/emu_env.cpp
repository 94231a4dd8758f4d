// tests/emu/emu_env.cpp -- TEST HARNESS ONLY.
// Builds the kernel sources (procgen_amd/csrc/pg_env.h + game policies) with PGAMD_WAVE_EMU, i.e. with wave.h's
// lane sections executed as 64-iteration host loops, so the wave-structured kernel logic can be compared with
// the oracle in a container that has no GPU.  Nothing here is linked into libenv.so.
#define PGAMD_WAVE_EMU 1
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "assets.h"
#include "games.h"
#include "pg_math.h"
#include "pg_assetgen.h"
#include "pg_bgpaint.h"
#include "pg_render.h"
#include "pg_prep.h"
#include "pg_human.h"
#include "host_state.h"
#include "state_io.h"

using namespace pgamd;

namespace pgamd {
template <class Game, int CAP>
struct EnvCap : Env<Game, CAP> {};
}  // namespace pgamd

struct EmuVec {
    int n, cap_small, cap_big;
    DevCtx d;
    std::vector<EnvHdr> hdr;
    std::vector<uint32_t> rng, ents;
    std::vector<uint8_t> grid, obs, first, plc;
    std::vector<int32_t> action, pls, ls;
    std::vector<float> rew;
    HostAssets assets;
    std::vector<uint32_t> game_tables;
    std::vector<uint32_t> gen_bg;  // use_generated_assets
    std::vector<int> bg_req;
    std::vector<uint32_t> frame_rec;  // display-list games (pg_prep.h)
    std::vector<int> slow_list;
    long long fast_frames = 0, slow_frames = 0;
    int use_small;
    int dev_error = 0;
    int game_id = -1;
    int kernel_id = -1;
    long long wave_steps = 0, split_resets = 0;  // env-steps taken, and how many of them went on to the reset kernel (SPLIT_RESET games)
};

// PG_EMU_POISON_LDS=1: every "workgroup" starts on an LDS arena full of pseudo-random words, as on a GPU that other processes' kernels
// share (LDS is not cleared between workgroups): a read of a word the kernel has not written shows up as a mismatch against the oracle
template <class T>
static void poison_lds(T *lds) {
    if (!getenv("PG_EMU_POISON_LDS")) return;
    static uint64_t x = 0x9e3779b97f4a7c15ull;
    uint32_t *w = reinterpret_cast<uint32_t *>(lds);
    for (size_t i = 0; i < sizeof(T) / 4; i++) {
        x ^= x << 13;
        x ^= x >> 7;
        x ^= x << 17;
        w[i] = (uint32_t)(x >> 16);
    }
}
template <class Game, int CAP>
static void run_env(EmuVec *v, int env, int mode) {
    static Lds<Game, CAP> lds;  // one "workgroup" at a time
    poison_lds(&lds);
    Env<Game, CAP> e(v->d, env, &lds);
    e.run(mode);
}
// a step kernel of a SPLIT_RESET game: no level generator, arena without scratch; an ended episode goes to "reset_list"
template <class Game, int CAP>
static void run_step_env(EmuVec *v, int env) {
    if constexpr (GameSplit<Game>::value) {
        static Lds<Game, CAP, false> lds;
        poison_lds(&lds);
        Env<Game, CAP, true> e(v->d, env, &lds);
        e.run(1);
        if (v->hdr[env].big == ROUTE_RESET) {
            v->split_resets++;
            run_env<Game, GameSplit<Game>::RESET_CAP>(v, env, 2);
        }
    } else {
        run_env<Game, CAP>(v, env, 1);
    }
}

template <class Game>
static void run_all(EmuVec *v, int mode) {
    for (int e = 0; e < v->n; e++) {  // "step kernels"
        int tier = v->use_small ? v->hdr[e].big : 2;
        if (mode == 1) {
            v->wave_steps++;
            if (tier == 0) run_step_env<Game, Game::ENT_CAP_T0>(v, e);
            else if (tier == 1) run_step_env<Game, Game::ENT_CAP_T1>(v, e);
            else run_step_env<Game, Game::ENT_CAP_T2>(v, e);
        } else {
            run_env<Game, GameSplit<Game>::RESET_CAP>(v, e, mode);  // "reset_grid" / the tier-0 grid in mode 0
        }
    }
    if (v->d.gen_bg) {  // "paint_backgrounds": the episodes that began this step
        static BgPaintLds blds;
        for (int e = 0; e < v->n; e++) {
            const int skip = v->bg_req[2 * e + 1];
            if (skip < 0) continue;
            int err = 0;
            paint_background(v->gen_bg.data() + (size_t)e * GEN_BG_WORDS, v->bg_req[2 * e], skip, &blds, &err);
            v->bg_req[2 * e + 1] = -1;
            if (err) v->dev_error |= PGE_ASSERT;
        }
    }
    static RenderLdsT<Game> rlds;
    if constexpr (GameDisplayList<Game>::value) {
        if (!v->d.gen_bg && !getenv("PG_EMU_NO_DISPLAY_LIST")) {  // "prep", "raster" (pg_prep.h)
            typedef FrameRec<Game> Rec;
            if (v->frame_rec.empty()) {
                v->frame_rec.assign((size_t)v->n * Rec::WORDS, 0xdeadbeefu);  // (a record word the prep wave does not write must not be read)
            }
            v->d.frame_rec = v->frame_rec.data();
            if (v->slow_list.empty()) v->slow_list.assign(v->n, -1);
            int slow_count = 0;
            for (int e0 = 0; e0 < v->n; e0 += PREP_ENVS) {
                poison_lds(&rlds);
                FramePrep<Game> p(v->d, &rlds, &slow_count, v->slow_list.data());
                p.run(e0, v->n - e0 < PREP_ENVS ? v->n - e0 : PREP_ENVS);
            }
            for (int e = 0; e < v->n; e++) {  // "raster"
                if (!(v->frame_rec[(size_t)e * Rec::WORDS + Rec::FLAGS] & Rec::F_FAST)) continue;
                poison_lds(&rlds);
                Renderer<Game> r(v->d, e, &rlds);
                r.raster_env();
                v->fast_frames++;
            }
            for (int k = 0; k < slow_count; k++) {  // "render_list"
                poison_lds(&rlds);
                Renderer<Game> r(v->d, v->slow_list[k], &rlds);
                r.render_env();
                v->slow_frames++;
            }
            return;
        }
    }
    for (int e = 0; e < v->n; e++) {  // "render kernel": one wave per env
        if (v->d.gen_bg) {
            poison_lds(&rlds);
            Renderer<Game, true> r(v->d, e, &rlds);
            r.render_env();
            continue;
        }
        poison_lds(&rlds);
        Renderer<Game> r(v->d, e, &rlds);
        r.render_env();
    }
}

extern "C" {

void *emu_make(const char *game, int num_envs, int rand_seed, int env_offset, int num_levels, int start_level, int distribution_mode,
               int center_agent, int use_backgrounds, int restrict_themes, int use_sequential_levels, int debug_mode, const char *resource_root,
               const char *atlas_path, int use_small, int use_monochrome_assets, int paint_vel_info, int use_generated_assets) {
    EmuVec *v = new EmuVec();
    v->n = num_envs;
    v->use_small = use_small;
    std::string err;
    const int game_id = game_id_from_name(game);
    const int gid = kernel_id_for(game_id, distribution_mode);  // the policy instantiation (caveflyer's memory mode has its own)
    bool known = false;
#define PG_X(Game) known = known || gid == Game::GAME_ID;
    PG_FOR_EACH_GAME(PG_X)
#undef PG_X
    if (!known) {
        fprintf(stderr, "emu: game %s not implemented\n", game);
        return nullptr;
    }
    v->game_id = game_id;
    v->kernel_id = gid;
    if (use_generated_assets) {
        bool (*block)(int) = nullptr;
#define PG_X(Game) \
    if (gid == Game::GAME_ID) block = GameBlockAsset<Game>::is;
        PG_FOR_EACH_GAME(PG_X)
#undef PG_X
        generate_game_assets(game, block, &v->assets);
    } else if (!load_game_assets(game_id, resource_root ? resource_root : "", atlas_path ? atlas_path : "", &v->assets, &err)) {
        fprintf(stderr, "emu: %s\n", err.c_str());
        return nullptr;
    }
    int ent_cap = 0, grid_bytes = 0;
#define PG_X(Game)                                                                            \
    if (gid == Game::GAME_ID) {                                                               \
        ent_cap = Game::ENT_CAP_T2;                                                           \
        grid_bytes = game_grid_bytes<Game>();                                                 \
    }
    PG_FOR_EACH_GAME(PG_X)
#undef PG_X
    v->hdr.resize(num_envs);
    v->rng.assign((size_t)num_envs * MT_SLOTS * MT_STRIDE, 0);
    v->ents.assign(ent_table_words(num_envs, ent_cap), 0);
    v->grid.assign((size_t)num_envs * grid_bytes, 0);
    v->obs.assign((size_t)num_envs * OBS_BYTES, 0);
    v->first.assign(num_envs, 0);
    v->plc.assign(num_envs, 0);
    v->action.assign(num_envs, 0);
    v->pls.assign(num_envs, 0);
    v->ls.assign(num_envs, 0);
    v->rew.assign(num_envs, 0);
#define PG_X(Game) \
    if (gid == Game::GAME_ID) init_env_state<Game>(num_envs, rand_seed, env_offset, 1, v->hdr.data(), v->rng.data());
    PG_FOR_EACH_GAME(PG_X)
#undef PG_X
    DevCtx &d = v->d;
    memset(&d, 0, sizeof(d));
    d.num_envs = num_envs;
    d.opt = GameOptions{};
    d.opt.use_backgrounds = use_backgrounds;
    d.opt.center_agent = center_agent;
    d.opt.restrict_themes = restrict_themes;
    d.opt.use_monochrome_assets = use_monochrome_assets;
    d.opt.paint_vel_info = paint_vel_info;
    d.opt.use_generated_assets = use_generated_assets;
    d.opt.distribution_mode = distribution_mode;
    d.opt.use_sequential_levels = use_sequential_levels;
    d.opt.debug_mode = debug_mode;
    if (const char *dbg = getenv("PROCGEN_AMD_DEBUG")) d.debug_flags = atoi(dbg);
    d.chunk_envs = (num_envs + TILE_ENVS - 1) / TILE_ENVS * TILE_ENVS;
    d.reset_chunk_envs = d.chunk_envs;
    d.reset_first = 0;
    level_seed_range(num_levels, start_level, &d.opt.level_seed_low, &d.opt.level_seed_high);
    for (auto &h : v->hdr) {
        h.level_seed_low = d.opt.level_seed_low;
        h.level_seed_high = d.opt.level_seed_high;
        h.opt_bits = env_option_bits(d.opt);
        h.opt_debug_mode = d.opt.debug_mode;
    }
    d.hdr = v->hdr.data();
    d.rng = v->rng.data();
    d.ents = v->ents.data();
    d.ent_cap = ent_cap;
    d.grid = v->grid.data();
    d.grid_bytes = grid_bytes;
    d.action = v->action.data();
    d.obs = v->obs.data();
    d.rew = v->rew.data();
    d.prev_level_seed = v->pls.data();
    d.level_seed = v->ls.data();
    d.first = v->first.data();
    d.prev_level_complete = v->plc.data();
    d.assets = &v->assets.table;
    d.pixels = v->assets.pixels.data();
    d.error = &v->dev_error;
    if (use_generated_assets) {
        v->gen_bg.assign((size_t)num_envs * GEN_BG_WORDS, 0u);
        v->bg_req.assign((size_t)num_envs * 2, -1);
        d.gen_bg = v->gen_bg.data();
        d.bg_req = v->bg_req.data();
    }
    v->game_tables.assign(2048, 0);
    int nw = 0;
#define PG_X(Game) \
    if (gid == Game::GAME_ID) nw = GameHostTables<Game>::build(d.opt, v->game_tables.data(), 2048);
    PG_FOR_EACH_GAME(PG_X)
#undef PG_X
    d.game_tables = nw > 0 ? v->game_tables.data() : nullptr;
    return v;
}

void emu_free(void *h) { delete (EmuVec *)h; }
static void run_game(EmuVec *v, int mode) {
#define PG_X(Game) \
    if (v->kernel_id == Game::GAME_ID) run_all<Game>(v, mode);
    PG_FOR_EACH_GAME(PG_X)
#undef PG_X
}
void emu_init(void *h) { run_game((EmuVec *)h, 0); }
void emu_step(void *h, const int32_t *actions) {
    EmuVec *v = (EmuVec *)h;
    memcpy(v->action.data(), actions, sizeof(int32_t) * v->n);
    run_game(v, 1);
}
void emu_observe(void *h, uint8_t *rgb, float *rew, uint8_t *first, int32_t *pls, uint8_t *plc, int32_t *ls) {
    EmuVec *v = (EmuVec *)h;
    memcpy(rgb, v->obs.data(), v->obs.size());
    memcpy(rew, v->rew.data(), 4 * v->n);
    memcpy(first, v->first.data(), v->n);
    memcpy(pls, v->pls.data(), 4 * v->n);
    memcpy(plc, v->plc.data(), v->n);
    memcpy(ls, v->ls.data(), 4 * v->n);
}
static void emu_snapshot(EmuVec *v, int env, EnvSnapshot *s) {
    const int cap = v->d.ent_cap;
    s->hdr = v->hdr[env];
    s->ent_cap = cap;
    s->ents.resize((size_t)EF_COUNT * cap);
    for (int k = 0; k < EF_COUNT * cap; k++) s->ents[k] = v->ents[ent_table_base(env, cap) + (size_t)k];
    s->rng.assign(v->rng.begin() + (size_t)env * MT_SLOTS * MT_STRIDE, v->rng.begin() + (size_t)env * MT_SLOTS * MT_STRIDE + 2 * MT_STRIDE);
    s->grid.assign(v->grid.begin() + (size_t)env * v->d.grid_bytes, v->grid.begin() + (size_t)(env + 1) * v->d.grid_bytes);
}
// the product's state_io.cpp on the emulated state: same wire format code as libenv.so's get_state / set_state
int emu_get_state(void *h, int env, char *data, int length) {
    EmuVec *v = (EmuVec *)h;
    EnvSnapshot s;
    emu_snapshot(v, env, &s);
    int written = 0;
    std::string err;
    if (!serialize_state(v->game_id, v->d.opt, env, s, data, length, &written, &err)) {
        fprintf(stderr, "emu_get_state: %s\n", err.c_str());
        return -1;
    }
    return written;
}
int emu_set_state(void *h, int env, const char *data, int length) {
    EmuVec *v = (EmuVec *)h;
    EnvSnapshot s;
    emu_snapshot(v, env, &s);
    std::string err;
    if (!deserialize_state(v->game_id, v->d.opt, &s, data, length, &err)) {
        fprintf(stderr, "emu_set_state: %s\n", err.c_str());
        return -1;
    }
    s.hdr.big = 2;  // the emulation picks the arena from this field alone; the largest arena is always safe
    const int cap = v->d.ent_cap;
    v->hdr[env] = s.hdr;
    for (int k = 0; k < EF_COUNT * cap; k++) v->ents[ent_table_base(env, cap) + (size_t)k] = s.ents[k];
    std::copy(s.rng.begin(), s.rng.end(), v->rng.begin() + (size_t)env * MT_SLOTS * MT_STRIDE);
    std::copy(s.grid.begin(), s.grid.end(), v->grid.begin() + (size_t)env * v->d.grid_bytes);
    v->rew[env] = s.hdr.reward;
    v->first[env] = (uint8_t)s.hdr.done;
    v->pls[env] = s.hdr.prev_level_seed;
    v->plc[env] = (uint8_t)s.hdr.level_complete;
    v->ls[env] = s.hdr.current_level_seed;
#define PG_X(Game)                            \
    if (v->kernel_id == Game::GAME_ID) {      \
        static RenderLdsT<Game> rlds;         \
        Renderer<Game> r(v->d, env, &rlds);   \
        r.render_env();                       \
    }
    PG_FOR_EACH_GAME(PG_X)
#undef PG_X
    return 0;
}
// pg_math.h restatements, exposed for tests/test_device_math.py
// the render_human frame of one env (pg_human.h): every band "workgroup" in turn; out = 512 x 512 x 3 bytes
int emu_render_human(void *h, int env, uint8_t *out) {
    EmuVec *v = (EmuVec *)h;
    std::vector<uint8_t> frames((size_t)v->n * HUMAN_BYTES);
    v->d.human = frames.data();
    static HumanLds hl;
    bool ok = false;
#define PG_X(Game)                                          \
    if (v->kernel_id == Game::GAME_ID) {                    \
        ok = true;                                          \
        for (int b = 0; b < HUMAN_BANDS; b++) {             \
            HumanRenderer<Game> r(v->d, env, &hl, b);       \
            r.render_band();                                \
        }                                                   \
    }
    PG_FOR_EACH_GAME(PG_X)
#undef PG_X
    memcpy(out, frames.data() + (size_t)env * HUMAN_BYTES, HUMAN_BYTES);
    v->d.human = nullptr;
    return ok ? 0 : -1;
}
void emu_atan2f_array(const float *y, const float *x, float *out, int n) {
    for (int i = 0; i < n; i++) out[i] = pg_atan2f(y[i], x[i]);
}
void emu_atan2d_array(const double *y, const double *x, double *out, int n) {
    for (int i = 0; i < n; i++) out[i] = pg_atan2_d(y[i], x[i]);
}
void emu_sincos_array(const double *x, double *s, double *c, int n) {
    for (int i = 0; i < n; i++) {
        s[i] = pg_sin_d(x[i]);
        c[i] = pg_cos_d(x[i]);
    }
}
// pg_qtpath.h (the header the product builds the jumper compass masks with) on a 64 x 64 canvas: 0 untouched, 1 brush, 2 pen
void emu_qt_path_ellipse(double x, double y, double w, double h, int pen, int brush, uint8_t *out) {
    Jumper::MaskSink m;
    memset(&m, 0, sizeof(m));
    int top, bot;
    if (brush) qtpath::fill_crossings(m, x, y, w, h, RES_W, RES_H, top, bot);
    if (pen) qtpath::stroke_ellipse(m, x, y, w, h, RES_W, RES_H);
    for (int yy = 0; yy < RES_H; yy++)
        for (int xx = 0; xx < RES_W; xx++) out[yy * RES_W + xx] = ((m.pen[yy] >> xx) & 1) ? 2 : (((m.brush[yy] >> xx) & 1) ? 1 : 0);
}
// pg_assetgen.h on the host: the sprite of an object type as libenv_make paints it, and a background from a generator seeded with `seed`
struct EmuMT {
    HostMT m;
    uint32_t u32() { return m.next(); }
};
void emu_generated_asset(const char *game, int type, uint32_t *out4096) {
    const int gid = game_id_from_name(game);
    bool block = false;
#define PG_X(Game) \
    if (gid == Game::GAME_ID) block = GameBlockAsset<Game>::is(type);
    PG_FOR_EACH_GAME(PG_X)
#undef PG_X
    EmuMT rng;
    rng.m.seed((int)(hash_str_uint32(game) + (uint32_t)type));
    int cnt[64], xa[64];
    assetgen::MemPainter mp{out4096, 64, 64, cnt, xa, 0u, 0u};
    assetgen::Gen<EmuMT, assetgen::MemPainter> gen{rng, mp};
    gen.generate_resource(64, 64, 0, 5, block);
}
void emu_generated_background(int seed, uint32_t *out250000) {
    EmuMT rng;
    rng.m.seed(seed);
    std::vector<int> cnt(500), xa(500);
    assetgen::MemPainter mp{out250000, 500, 500, cnt.data(), xa.data(), 0u, 0u};
    assetgen::Gen<EmuMT, assetgen::MemPainter> gen{rng, mp};
    gen.generate_resource(500, 500, 1, 50, true);
}
void emu_dump_background(void *h, int env, uint32_t *out) { memcpy(out, ((EmuVec *)h)->gen_bg.data() + (size_t)env * GEN_BG_WORDS, sizeof(uint32_t) * GEN_BG_WORDS); }
long long emu_counter(int k) { return pg_emu_counters()[k]; }
// display-list games: frames the rasterizer drew from their record / frames the full renderer drew (pg_prep.h)
long long emu_frame_count(void *h, int slow) { return slow ? ((EmuVec *)h)->slow_frames : ((EmuVec *)h)->fast_frames; }
void emu_path_counts(void *h, long long *out) {
    EmuVec *v = (EmuVec *)h;
    out[0] = 0;
    out[1] = 0;
    out[2] = v->wave_steps;
    out[3] = v->split_resets;
}
int emu_error(void *h, int env) { return ((EmuVec *)h)->hdr[env].error | ((EmuVec *)h)->dev_error; }
int emu_num_entities(void *h, int env) { return ((EmuVec *)h)->hdr[env].n_ents; }
int emu_is_big(void *h, int env) { return ((EmuVec *)h)->hdr[env].big; }
// entity dump in the reference's serialization order (31 words, reference src/entity.cpp:90-137)
void emu_dump_entities(void *h, int env, int32_t *out) {
    EmuVec *v = (EmuVec *)h;
    const int cap = v->d.ent_cap;
    const uint32_t *e = v->ents.data() + ent_table_base(env, cap);
    const int tile = 1;
    auto W = [&](int f, int i) { return (int32_t)e[(size_t)(f * cap + i) * tile]; };
    for (int i = 0; i < v->hdr[env].n_ents; i++) {
        int32_t *o = out + 31 * i;
        const uint32_t m = (uint32_t)W(EF_META, i);
        int k = 0;
        o[k++] = W(EF_X, i); o[k++] = W(EF_Y, i); o[k++] = W(EF_VX, i); o[k++] = W(EF_VY, i); o[k++] = W(EF_RX, i); o[k++] = W(EF_RY, i);
        o[k++] = meta_type(m); o[k++] = meta_image_type(m); o[k++] = meta_image_theme(m); o[k++] = meta_render_z(m);
        o[k++] = (m & MF_WILL_ERASE) != 0; o[k++] = (m & MF_COLLIDES) != 0;
        o[k++] = W(EF_COLLISION_MARGIN, i); o[k++] = W(EF_ROTATION, i); o[k++] = W(EF_VROT, i);
        o[k++] = (m & MF_REFLECTED) != 0; o[k++] = W(EF_FIRE_TIME, i); o[k++] = W(EF_SPAWN_TIME, i); o[k++] = W(EF_LIFE_TIME, i);
        o[k++] = W(EF_EXPIRE_TIME, i); o[k++] = (m & MF_ABS_COORDS) != 0;
        o[k++] = W(EF_FRICTION, i); o[k++] = (m & MF_SMART_STEP) != 0; o[k++] = (m & MF_AVOIDS) != 0; o[k++] = (m & MF_AUTO_ERASE) != 0;
        o[k++] = W(EF_ALPHA, i); o[k++] = W(EF_HEALTH, i); o[k++] = W(EF_THETA, i); o[k++] = W(EF_GROW_RATE, i); o[k++] = W(EF_ALPHA_DECAY, i);
        o[k++] = W(EF_CLIMBER_SPAWN_X, i);
    }
}
void emu_dump_grid(void *h, int env, int32_t *out, int *w, int *hh) {
    EmuVec *v = (EmuVec *)h;
    *w = v->hdr[env].main_width;
    *hh = v->hdr[env].main_height;
    const uint8_t *g = v->grid.data() + (size_t)env * v->d.grid_bytes;
    const bool wide = v->game_id == GAME_CHASER;  // u16 cells
    for (int i = 0; i < (*w) * (*hh); i++) out[i] = wide ? (int32_t) reinterpret_cast<const uint16_t *>(g)[i] : (int32_t)g[i];
}
}
