// pg_env.h -- the per-environment stepper: ONE wavefront (= one workgroup) advances ONE environment.
//
// This file is the device-side equivalent of the reference's Game / BasicAbstractGame runtime
// (reference src/game.cpp:93-165, src/basic-abstract-game.cpp ["BAG"], src/entity.cpp, src/randgen.cpp) and of
// the Qt raster calls it issues (QPainter::drawImage / fillRect on a 64x64 RGB32 image).  It is NOT a
// translation: entities live as an SoA table staged in LDS, order-dependent loops become
// ballot + highest-set-bit walks, level generation consumes a bit-exact MT19937 stream held in LDS, and the
// frame is produced by a two-phase rasterizer (lane-parallel draw-command setup in fp64, then 8x8 pixel
// blocks per command into an LDS framebuffer, then one coalesced RGB888 store).
//
// Written against wave.h's wave-structured execution model; game rules come from a policy class
// (game_coinrun.h, ...).  Floating point mirrors the reference expression by expression (float/double
// promotion order); build with -ffp-contract=off.
#pragma once
#include <type_traits>

#include "pg_assetgen.h"
#include "pg_defs.h"
#include "wave.h"

namespace pgamd {

constexpr float PG_PI = 3.14159265358979323846264338327950288f;  // reference src/cpp-utils.h:12
constexpr float MAXVTHETA = 15 * PG_PI / 180;                    // BAG:6
constexpr float MIXRATEROT = 0.5f;                               // BAG:7
constexpr float POS_EPS = -0.001f;                               // BAG:10
constexpr float RENDER_EPS = 0.02f;                              // BAG:14

// error codes stored in EnvHdr::error (host turns them into the reference's fatal()/fassert exit)
enum PgError : int { PGE_NONE = 0, PGE_ENT_OVERFLOW = 1, PGE_GRID_OOB = 2, PGE_ASSERT = 3, PGE_THEME = 4, PGE_UNSUPPORTED_DRAW = 5 };
// EnvHdr::error = code | source line of the fail() that raised it << 8 (the line is what a post-mortem needs: "code 3" alone names six
// call sites of the step kernels and a dozen in the game policies).  The handle-wide word DevCtx::error is the OR of the codes alone.
PG_DEV int pg_error_word(int code, int line) { return code | (line << 8); }
// called by lane 0 of an env's wave when the env carries an error: OR the code into the handle's sticky word and, for the first
// reporter (compare-and-swap on the record's word 0, pg_defs.h ERROR_INFO_OFFSET), leave who and where -- in the device record and in
// the handle's HOST-mapped copy of it, whose address the host left in the record's words 6 / 7: the host reads that copy after the
// stream join of the same libenv_observe, without a download (an error a render kernel raises ends the run before the caller sees the
// frame).  The emulation's error word stands alone.
PG_DEV void pg_report_error(const DevCtx &d, int env, int error_word, int kind, int n_ents, int agent) {
#if defined(PGAMD_WAVE_EMU)
    (void)env; (void)kind; (void)n_ents; (void)agent;
    if (d.error) *d.error |= (error_word & 0xff);
#else
    atomicOr(d.error, error_word & 0xff);
    int *w = d.error + ERROR_INFO_OFFSET;
    if (atomicCAS(w, 0, env + 1) == 0) {
        w[1] = error_word;
        w[2] = kind;
        w[3] = n_ents;
        w[4] = agent;
        volatile int *h = reinterpret_cast<volatile int *>(((unsigned long long)(unsigned)w[7] << 32) | (unsigned long long)(unsigned)w[6]);
        if (h) {
            h[1] = env + 1;
            h[2] = error_word;
            h[3] = kind;
            h[4] = n_ents;
            h[5] = agent;
            __threadfence_system();
            h[0] = error_word & 0xff;
        }
    }
#endif
}

// ---- EF_META packing -------------------------------------------------------------------------------------
constexpr uint32_t M_TYPE_MASK = 0x3ffu;
constexpr int M_IMG_SHIFT = 10;   // 8 bits
constexpr int M_THEME_SHIFT = 18; // 4 bits
constexpr int M_Z_SHIFT = 22;     // 2 bits, render_z + 1
constexpr uint32_t MF_WILL_ERASE = 1u << 24;
constexpr uint32_t MF_COLLIDES = 1u << 25;
constexpr uint32_t MF_REFLECTED = 1u << 26;
constexpr uint32_t MF_ABS_COORDS = 1u << 27;
constexpr uint32_t MF_SMART_STEP = 1u << 28;
constexpr uint32_t MF_AVOIDS = 1u << 29;
constexpr uint32_t MF_AUTO_ERASE = 1u << 30;
constexpr uint32_t MF_PAR_DONE = 1u << 31;  // transient, inside step_entities only: basic_step_object already done in the parallel pass

PG_DEV uint32_t meta_make(int type, int image_type, int image_theme, int render_z, uint32_t flags) {
    return ((uint32_t)type & M_TYPE_MASK) | (((uint32_t)image_type & 0xffu) << M_IMG_SHIFT) | (((uint32_t)image_theme & 0xfu) << M_THEME_SHIFT) |
           (((uint32_t)(render_z + 1) & 3u) << M_Z_SHIFT) | flags;
}
PG_DEV int meta_type(uint32_t m) { return (int)(m & M_TYPE_MASK); }
PG_DEV int meta_image_type(uint32_t m) { return (int)((m >> M_IMG_SHIFT) & 0xffu); }
PG_DEV int meta_image_theme(uint32_t m) { return (int)((m >> M_THEME_SHIFT) & 0xfu); }
PG_DEV int meta_render_z(uint32_t m) { return (int)((m >> M_Z_SHIFT) & 3u) - 1; }

struct RectD {
    double x, y, w, h;
};

PG_DEV RectD adjust_rect(RectD b, double ax, double ay, double aw, double ah) {  // reference src/qt-utils.h:12-19
    RectD r;
    r.x = b.x + b.w * ax;
    r.y = b.y + b.h * ay;
    r.w = b.w * aw;
    r.h = b.h * ah;
    return r;
}

PG_DEV int q_round(double d) {  // Qt 5.9 qRound(double)
    return d >= 0.0 ? (int)(d + 0.5) : (int)(d - (double)((int)(d - 1)) + 0.5) + (int)(d - 1);
}
PG_DEV uint32_t byte_mul(uint32_t x, uint32_t a) {  // Qt BYTE_MUL (qdrawhelper_p.h)
    uint32_t t = (x & 0xff00ffu) * a;
    t = (t + ((t >> 8) & 0xff00ffu) + 0x800080u) >> 8;
    t &= 0xff00ffu;
    x = ((x >> 8) & 0xff00ffu) * a;
    x = (x + ((x >> 8) & 0xff00ffu) + 0x800080u);
    x &= 0xff00ff00u;
    return x | t;
}
PG_DEV double sign_d(double x) { return x > 0 ? +1 : (x == 0 ? 0 : -1); }  // reference src/cpp-utils.h:43-45
PG_DEV float clip_abs(float x, float y) {                                   // reference src/cpp-utils.h:47-53
    if (x > y) return y;
    if (x < -y) return -y;
    return x;
}

// ---- LDS arena of one workgroup ---------------------------------------------------------------------------
struct NoScratch {};

// a game may declare `typedef X Scratch;` for LDS it needs during level generation (e.g. MazeScratch)
template <class Game, class = void>
struct GameScratch {
    typedef NoScratch type;
};
template <class Game>
struct GameScratch<Game, decltype((void)sizeof(typename Game::Scratch))> {
    typedef typename Game::Scratch type;
};

// a game may declare `static constexpr int AUX_WORDS` for per-env HBM storage behind its grid cells (the slab
// d.grid + env * d.grid_bytes holds the padded cells, then AUX_WORDS 32-bit words; not staged in LDS)
template <class Game, class = void>
struct GameAux {
    static constexpr int WORDS = 0;
};
template <class Game>
struct GameAux<Game, decltype((void)Game::AUX_WORDS)> {
    static constexpr int WORDS = Game::AUX_WORDS;
};
template <class Game>
constexpr int game_cell_bytes() {
    return (int)((sizeof(typename Game::cell_t) * Game::MAX_CELLS + 15) & ~(size_t)15);
}
template <class Game>
constexpr int game_grid_bytes() {
    return game_cell_bytes<Game>() + GameAux<Game>::WORDS * 4;
}

// A game whose level generator needs much more LDS than its steps (scratch for maze / room generation, or an entity table
// that only a reset fills) declares SPLIT_RESET = true and RESET_CAP: its step kernels are compiled without the generator
// (Env NO_RESET: no scratch in their arena, smaller entity tiers), an episode that ends is queued, and the reset kernel
// (reset_list, arena RESET_CAP + scratch) generates the next level behind the chunk's step kernel.
template <class Game, class = void>
struct GameSplit {
    static constexpr bool value = false;
    static constexpr int RESET_CAP = Game::ENT_CAP_T0;
};
template <class Game>
struct GameSplit<Game, decltype((void)Game::SPLIT_RESET)> {
    static constexpr bool value = Game::SPLIT_RESET;
    static constexpr int RESET_CAP = Game::RESET_CAP;
};

// Constant tables a game wants in HBM next to the sprite atlas (DevCtx::game_tables): a game declares HOST_TABLE_WORDS and
// host_tables(options, out, max_words) -> words written; built once per handle on the host from the options alone
// (jumper: the row masks of its compass ellipse).
template <class Game, class = void>
struct GameHostTables {
    static int build(const GameOptions &, uint32_t *, int) { return 0; }
};
template <class Game>
struct GameHostTables<Game, decltype((void)Game::HOST_TABLE_WORDS)> {
    static int build(const GameOptions &o, uint32_t *out, int max_words) { return Game::host_tables(o, out, max_words); }
};

// use_block_asset (BAG:404-406 and the games' overrides): the object types whose generated asset is a rect texture
// (use_generated_assets); every other type gets a shape on transparent ground
template <class Game, class = void>
struct GameBlockAsset {
    static bool is(int) { return false; }
};
template <class Game>
struct GameBlockAsset<Game, decltype((void)&Game::use_block_asset)> {
    static bool is(int t) { return Game::use_block_asset(t); }
};

// A game declares PAR_SMART = true and par_smart_type_ok(type) for the smart_step entity types (a) whose basic_step_object
// has no side effect beyond the object itself when no entity can block or reflect it, and (b) that no smart entity's
// sub_step scan can ever hit (may_interact(any smart type, type) is false): step_entities then steps all such objects of an
// env side by side instead of one after the other (Env::step_entities).  (c) The pass decides "nothing can block or reflect
// this object" (bso_scan_axes == 0) BEFORE anyone has moved, whereas the reference reaches object i after the entities above
// it have stepped: the reach test therefore allows for the target's own travel (|v|, plus the snapping / reflection slack
// when the target is a smart entity).  Blockers are never displaced by a push (push_obj moves the MOVER out of the blocker,
// BAG:240-268), and entities are only spawned after step_entities, so a target's motion within the step is bounded by that.
// Per game: coinrun (crates, ENEMY_BARRIER), climber, ninja, dodgeball, chaser, caveflyer -- every blocking / reflecting
// target of their parallel types has zero velocity; the widened reach is what keeps a future game with moving blockers exact.
template <class Game, class = void>
struct GameParSmart {
    static constexpr bool value = false;
};
template <class Game>
struct GameParSmart<Game, decltype((void)Game::PAR_SMART)> {
    static constexpr bool value = Game::PAR_SMART;
};

// a game whose is_blocked has a side effect on the moving entity (ninja's throwing stars stick to walls) declares
// HAS_BLOCK_HOOK and on_grid_block(e, obj): called when the corner probes of sub_step found a blocking cell
template <class Game, class = void>
struct GameHasBlockHook {
    static constexpr bool value = false;
};
template <class Game>
struct GameHasBlockHook<Game, decltype((void)Game::HAS_BLOCK_HOOK)> {
    static constexpr bool value = Game::HAS_BLOCK_HOOK;
};

// BLOCKED_ENTS_HAS_EFFECT and is_blocked_ents_peek(e, src, target, h): the game's is_blocked_ents writes game state (coinrun's
// is_on_crate); the peek is the same predicate without the write (Env::push_fixed_point asks before it knows the reference
// would have asked)
template <class Game, class = void>
struct GameBlockedEntsEffect {
    static constexpr bool value = false;
};
template <class Game>
struct GameBlockedEntsEffect<Game, decltype((void)Game::BLOCKED_ENTS_HAS_EFFECT)> {
    static constexpr bool value = Game::BLOCKED_ENTS_HAS_EFFECT;
};

#define PG_SYNC_E() PG_SYNC()  // memory ordering between two lane sections of the wave that owns this env

// WITH_SCRATCH = false: the arena of a step kernel that never generates a level (games with SPLIT_RESET, see GameSplit)
template <class Game, int CAP, bool WITH_SCRATCH = true>
struct Lds {
    uint32_t ent[EF_COUNT * CAP];
    uint32_t tmp[128];  // lane scratch; simple_choose keeps up to 128 picks here
    alignas(16) typename Game::cell_t grid[(Game::MAX_CELLS + 15) & ~15];
    typename std::conditional<WITH_SCRATCH, typename GameScratch<Game>::type, NoScratch>::type scratch;
#if defined(PG_LDS_PAD)
    uint8_t pad[PG_LDS_PAD];  // occupancy experiments only
#endif
};

// One wavefront advances this env: state staged in the workgroup's LDS arena (`s`), lanes cover entity slots / cells (wave.h's
// lane sections and ballots).
// NO_RESET: a step kernel of a SPLIT_RESET game -- it stops where the episode ends and queues the env for the reset kernel.
template <class Game, int CAP, bool NO_RESET_MODE = false>
struct Env {
    using cell_t = typename Game::cell_t;
    static constexpr int CAPACITY = CAP;
    static constexpr bool NO_RESET = NO_RESET_MODE;
    typedef Lds<Game, CAP, !NO_RESET_MODE> LdsT;
    const DevCtx &d;
    GameOptions opt;  // this env's options: the handle's, with the per-env ones of its header (pg_defs.h env_options), bound by load_env
    const int env;
    LdsT *s;
    EnvHdr G;
    // rand_gen bookkeeping: where the live 624-word state is (HBM home or LDS scratch)
    uint32_t *rg_home;
    uint32_t *rg_cur;
    bool rg_in_lds;
    // 64 state words of rand_gen ahead of the next draw, lane l = word rc_base + l of buffer rc_buf (rand_u32)
    PG_LANE_VAR(uint32_t, rc_words);
    const uint32_t *rc_buf = nullptr;
    int rc_base = 0;
    int rg_twists = 0;  // twists of rand_gen since rand_seed (draws since the reseed = (rg_twists - 1) * 624 + rand_idx)

    // profiling aid (PROCGEN_AMD_DEBUG & 2048): wave cycles spent since the previous mark are charged to phase k
    long long t_mark = 0, t_start = 0;
    bool needs_reset = false;  // NO_RESET: this step ended the episode
    PG_DEV void phase(int k) {
#if !defined(PGAMD_WAVE_EMU)
        if (PG_PHASES(d)) {
            const long long t = (long long)__builtin_readcyclecounter();
            if (PG_LANE_ID() == 0 && t_mark != 0) atomicAdd(d.phase_cycles + k + 32 * (env & 4095), (unsigned long long)(t - t_mark));
            if (PG_TRACE(d) && PG_LANE_ID() == 0 && t_mark != 0) atomicAdd(d.wave_trace + (size_t)env * 32 + 8 + k, (unsigned long long)(t - t_mark));  // this env, this step
            t_mark = (long long)__builtin_readcyclecounter();
        }
#else
        (void)k;
#endif
    }

    // profiling aid (PROCGEN_AMD_DEBUG & 8192): per-env counters of this step in the trace record (slots 24..31)
    PG_DEV void trace_add(int k, unsigned long long v) {
#if !defined(PGAMD_WAVE_EMU)
        if (PG_TRACE(d) && PG_PHASES(d) && PG_LANE_ID() == 0) d.wave_trace[(size_t)env * 32 + 24 + k] += v;
#else
        (void)k; (void)v;
#endif
    }
    // profiling aid for level generators (PROCGEN_AMD_DEBUG = 2048 + 16: the render kernel is off and its counter slots are free):
    // wave cycles since the previous mark / phase go to slot 16 + j; printed at libenv_close as "reset marks"
    PG_DEV void mark(int j) {
        if PG_DBG(d, 16) phase(16 + j);
    }
    PG_DEV Env(const DevCtx &d_, int env_, LdsT *s_) : d(d_), env(env_), s(s_) {
        rg_home = d.rng + (size_t)env * MT_SLOTS * MT_STRIDE;
        rg_cur = rg_home;
        rg_in_lds = false;
    }

    // ======================================================================================================
    // entity table accessors (LDS SoA)
    PG_DEV cell_t &cell(int idx) { return s->grid[idx]; }
    PG_DEV float &ef(int field, int i) { return reinterpret_cast<float *>(s->ent)[field * CAP + i]; }
    PG_DEV int &ei(int field, int i) { return reinterpret_cast<int *>(s->ent)[field * CAP + i]; }
    PG_DEV uint32_t &meta(int i) { return s->ent[EF_META * CAP + i]; }
    PG_DEV float &ex(int i) { return ef(EF_X, i); }
    PG_DEV float &ey(int i) { return ef(EF_Y, i); }
    PG_DEV float &evx(int i) { return ef(EF_VX, i); }
    PG_DEV float &evy(int i) { return ef(EF_VY, i); }
    PG_DEV float &erx(int i) { return ef(EF_RX, i); }
    PG_DEV float &ery(int i) { return ef(EF_RY, i); }
    PG_DEV int etype(int i) { return meta_type(meta(i)); }
    PG_DEV bool eflag(int i, uint32_t f) { return (meta(i) & f) != 0; }
    PG_DEV void set_flag(int i, uint32_t f, bool v) { meta(i) = v ? (meta(i) | f) : (meta(i) & ~f); }
    PG_DEV void set_image_type(int i, int t) { meta(i) = (meta(i) & ~(0xffu << M_IMG_SHIFT)) | (((uint32_t)t & 0xffu) << M_IMG_SHIFT); }
    PG_DEV void set_image_theme(int i, int t) { meta(i) = (meta(i) & ~(0xfu << M_THEME_SHIFT)) | (((uint32_t)t & 0xfu) << M_THEME_SHIFT); }
    PG_DEV void set_render_z(int i, int z) { meta(i) = (meta(i) & ~(3u << M_Z_SHIFT)) | (((uint32_t)(z + 1) & 3u) << M_Z_SHIFT); }

    PG_DEV void fail(int code, int line = __builtin_LINE()) {
        if (G.error == 0) G.error = pg_error_word(code, line);
    }
    // the game's aux words of this env in HBM (see GameAux)
    PG_DEV uint32_t *aux() { return reinterpret_cast<uint32_t *>(d.grid + (size_t)env * d.grid_bytes + game_cell_bytes<Game>()); }

    // Entity::Entity(x,y,vx,vy,rx,ry,type): reference src/entity.cpp:11-51
    PG_DEV void ent_init(int i, float x, float y, float vx, float vy, float rx, float ry, int type) {
        ex(i) = x; ey(i) = y; evx(i) = vx; evy(i) = vy; erx(i) = rx; ery(i) = ry;
        meta(i) = meta_make(type, type, 0, 0, MF_AUTO_ERASE);
        ei(EF_FIRE_TIME, i) = -1;
        ei(EF_SPAWN_TIME, i) = -1;
        ei(EF_LIFE_TIME, i) = 0;
        ei(EF_EXPIRE_TIME, i) = type == EXPLOSION ? 4 : -1;
        ef(EF_COLLISION_MARGIN, i) = 0.0f;
        ef(EF_ROTATION, i) = 0.0f;
        ef(EF_VROT, i) = 0.0f;
        ef(EF_FRICTION, i) = 1.0f;
        ef(EF_ALPHA, i) = 1.0f;
        ef(EF_HEALTH, i) = 1.0f;
        ef(EF_THETA, i) = -100.0f;
        ef(EF_GROW_RATE, i) = type == EXPLOSION ? 1.4f : (type == TRAIL ? 1.05f : 1.0f);
        ef(EF_ALPHA_DECAY, i) = type == TRAIL ? 0.8f : 1.0f;
        ef(EF_CLIMBER_SPAWN_X, i) = 0.0f;
    }

    // entities.push_back(new Entity(...)) from wave-uniform code: BAG:566-576
    PG_DEV int add_entity_rxy(float x, float y, float vx, float vy, float rx, float ry, int type) {
        int i = G.n_ents;
        if (i >= CAP - 1) {
            fail(PGE_ENT_OVERFLOW);
            return CAP - 2;
        }
        ent_init(i, x, y, vx, vy, rx, ry, type);
        G.n_ents = i + 1;
        return i;
    }
    PG_DEV int add_entity(float x, float y, float vx, float vy, float r, int type) { return add_entity_rxy(x, y, vx, vy, r, r, type); }

    // Entity::step: reference src/entity.cpp:57-82 (callable from a lane section or from uniform code)
    PG_DEV void ent_step(int i) {
        uint32_t m = meta(i) & ~MF_PAR_DONE;
        if (!(m & MF_SMART_STEP)) {
            ex(i) += evx(i);
            ey(i) += evy(i);
        }
        ef(EF_ROTATION, i) += ef(EF_VROT, i);
        float fr = ef(EF_FRICTION, i);
        evx(i) *= fr;
        evy(i) *= fr;
        int lt = ei(EF_LIFE_TIME, i) + 1;
        ei(EF_LIFE_TIME, i) = lt;
        int et = ei(EF_EXPIRE_TIME, i);
        if (et > 0 && lt > et) m |= MF_WILL_ERASE;
        if (meta_type(m) == EXPLOSION) {
            int it = meta_image_type(m);
            if (it < EXPLOSION5) m = (m & ~(0xffu << M_IMG_SHIFT)) | ((uint32_t)(it + 1) << M_IMG_SHIFT);
        }
        meta(i) = m;
        float gr = ef(EF_GROW_RATE, i);
        erx(i) *= gr;
        ery(i) *= gr;
        ef(EF_ALPHA, i) = ef(EF_ALPHA_DECAY, i) * ef(EF_ALPHA, i);
    }

    // ======================================================================================================
    // grid (staged in LDS): reference src/grid.h, BAG:125-131,167-223
    PG_DEV bool grid_contains(int x, int y) { return 0 <= y && y < G.main_height && 0 <= x && x < G.main_width; }
    PG_DEV int get_obj(int x, int y) {  // BAG:180-185
        if (!grid_contains(x, y)) return G.out_of_bounds_object;
        return (int)cell(y * G.main_width + x);
    }
    PG_DEV void set_obj(int x, int y, int v) {  // grid.h:54-57 (fassert on out-of-range)
        if (!grid_contains(x, y)) {
            fail(PGE_GRID_OOB);
            return;
        }
        cell(y * G.main_width + x) = (cell_t)v;
        G.grid_dirty = 1;
    }
    PG_DEV int get_obj_from_floats(float i, float j) {  // BAG:167-174
        if (i < 0) return G.out_of_bounds_object;
        if (j < 0) return G.out_of_bounds_object;
        return get_obj((int)pg_floorf(i), (int)pg_floorf(j));  // floor() of a float is the same number in float and double
    }
    // fill_elem BAG:125-131: rows are walked uniformly, lanes cover the columns
    PG_DEV void fill_elem(int x, int y, int dx, int dy, int elem) {
        if (dx <= 0 || dy <= 0) return;
        if (x < 0 || y < 0 || x + dx > G.main_width || y + dy > G.main_height) {
            fail(PGE_GRID_OOB);
            return;
        }
        const int w = G.main_width;
        for (int k = 0; k < dy; k++) {
            PG_FOR_LANES(l) {
                for (int j = l; j < dx; j += 64) s->grid[(y + k) * w + x + j] = (cell_t)elem;
            }
        }
        G.grid_dirty = 1;
        PG_SYNC_E();
    }

    // ======================================================================================================
    // MT19937 (std::mt19937 as used by RandGen: reference src/randgen.cpp:6-31,90-98; libstdc++ random.tcc)
    // scratch states live in HBM next to the generator's home (a reset is rare; LDS is kept for occupancy)
    PG_DEV uint32_t *mt_a() { return rg_home + 2 * MT_STRIDE; }
    PG_DEV uint32_t *mt_b() { return rg_home + 3 * MT_STRIDE; }

    // one twist of the whole state: src -> dst (distinct buffers, so lanes never read what others write)
    PG_DEV void mt_twist(const uint32_t *src, uint32_t *dst) {
        // k in [0,227): needs old[k], old[k+1], old[k+397]
        for (int base = 0; base < 227; base += 64) {
            PG_FOR_LANES(l) {
                int k = base + l;
                if (k < 227) {
                    uint32_t y = (src[k] & 0x80000000u) | (src[k + 1] & 0x7fffffffu);
                    dst[k] = src[k + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
                }
            }
        }
        PG_SYNC_E();
        // k in [227,454): new[k-227] was produced by the previous phase
        for (int base = 227; base < 454; base += 64) {
            PG_FOR_LANES(l) {
                int k = base + l;
                if (k < 454) {
                    uint32_t y = (src[k] & 0x80000000u) | (src[k + 1] & 0x7fffffffu);
                    dst[k] = dst[k - 227] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
                }
            }
        }
        PG_SYNC_E();
        for (int base = 454; base < 623; base += 64) {
            PG_FOR_LANES(l) {
                int k = base + l;
                if (k < 623) {
                    uint32_t y = (src[k] & 0x80000000u) | (src[k + 1] & 0x7fffffffu);
                    dst[k] = dst[k - 227] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
                }
            }
        }
        PG_SYNC_E();
        {
            uint32_t y = (src[623] & 0x80000000u) | (dst[0] & 0x7fffffffu);
            uint32_t v = dst[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            PG_FOR_LANES(l) {
                if (l == 0) dst[623] = v;
            }
        }
        PG_SYNC_E();
    }
    PG_DEV void mt_copy(const uint32_t *src, uint32_t *dst) {
        for (int base = 0; base < MT_N; base += 64) {
            PG_FOR_LANES(l) {
                int k = base + l;
                if (k < MT_N) dst[k] = src[k];
            }
        }
        PG_SYNC_E();
    }
    PG_DEV static uint32_t mt_temper(uint32_t z) {
        z ^= (z >> 11);
        z ^= (z << 7) & 0x9d2c5680u;
        z ^= (z << 15) & 0xefc60000u;
        z ^= (z >> 18);
        return z;
    }
    // rand_gen.seed(seed): state generated serially into LDS scratch A (the recurrence is a dependent chain)
    PG_DEV void rand_seed(int seed) {
        rc_buf = nullptr;  // (the scratch state is about to be rewritten)
        uint32_t *a = mt_a();
        uint32_t x = (uint32_t)seed;
        for (int i = 0; i < MT_N; i++) {
            if (i > 0) x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)i;
            PG_FOR_LANES(l) {
                if (l == 0) a[i] = x;
            }
        }
        PG_SYNC_E();
        rg_cur = a;
        rg_in_lds = true;
        G.rand_idx = MT_N;
        rg_twists = 0;
    }
    PG_DEV uint32_t rand_u32() {
        if (G.rand_idx >= MT_N) {
            uint32_t *dst = (rg_cur == mt_a()) ? mt_b() : mt_a();
            mt_twist(rg_cur, dst);
            rg_cur = dst;
            rg_in_lds = true;
            G.rand_idx = 0;
            rc_buf = nullptr;
            rg_twists++;
        }
        // The generator state lives in HBM (home or scratch): a draw used to be one dependent ~1 us round trip, and level
        // generators make hundreds to thousands of them (leaper's reset: 400 spawn rounds of up to nine draws, 1.5 ms of
        // a lone wave).  One coalesced load now brings the next 64 state words into a lane variable; draws read lanes.
        const int idx = G.rand_idx;
        if (rc_buf != rg_cur || idx < rc_base || idx >= rc_base + 64) {
            const uint32_t *src = rg_cur;
            PG_FOR_LANES(l) { PG_LV(rc_words, l) = idx + l < MT_N ? src[idx + l] : 0u; }
            rc_buf = rg_cur;
            rc_base = idx;
        }
        const uint32_t z = PG_READLANE(rc_words, idx - rc_base);
        G.rand_idx += 1;
        return mt_temper(z);
    }
    // The next `count` (<= 64) rand_gen draws WITHOUT consuming them: lane l < count gets the l-th (tempered); false when they
    // would cross a twist (the caller then draws one at a time).  rand_skip consumes draws seen this way.
    PG_DEV bool rand_peek_lanes(int count, PG_LANE_REF(uint32_t, out)) {
        const int idx = G.rand_idx;
        if (idx + count > MT_N) return false;
        if (rc_buf != rg_cur || idx < rc_base || idx + count > rc_base + 64) {
            const uint32_t *src = rg_cur;
            PG_FOR_LANES(l) { PG_LV(rc_words, l) = idx + l < MT_N ? src[idx + l] : 0u; }
            rc_buf = rg_cur;
            rc_base = idx;
        }
        const int shift = idx - rc_base;
        PG_FOR_LANES(l) { PG_LV(out, l) = l < count ? mt_temper(PG_SHFL(rc_words, l, shift + l)) : 0u; }
        return true;
    }
    PG_DEV void rand_skip(int count) { G.rand_idx += count; }

    // `count` (<= 64) consecutive rand_gen draws at once: lane l gets the l-th of them (tempered u32; lanes >= count get
    // 0).  Level generators that draw once per grid cell use this instead of 64 dependent trips to the generator state.
    PG_DEV void rand_u32_lanes(int count, PG_LANE_REF(uint32_t, out)) {
        int first = MT_N - G.rand_idx;  // draws still available before the next twist
        if (first > count) first = count;
        if (first < 0) first = 0;
        const uint32_t *cur0 = rg_cur;
        const int idx0 = G.rand_idx;
        PG_FOR_LANES(l) { PG_LV(out, l) = l < first ? mt_temper(cur0[idx0 + l]) : 0u; }
        G.rand_idx += first;
        if (first < count) {
            rc_buf = nullptr;
            uint32_t *dst = (rg_cur == mt_a()) ? mt_b() : mt_a();
            mt_twist(rg_cur, dst);
            rg_cur = dst;
            rg_in_lds = true;
            rg_twists++;
            const int rest = count - first;
            PG_FOR_LANES(l) {
                if (l >= first && l < count) PG_LV(out, l) = mt_temper(dst[l - first]);
            }
            G.rand_idx = rest;
        }
    }
    // write the live rand_gen state back to its HBM home (before the scratch is reused as framebuffer)
    PG_DEV void rand_flush() {
        if (rg_in_lds) {
            mt_copy(rg_cur, rg_home);
            rg_cur = rg_home;
            rg_in_lds = false;
        }
    }
    PG_DEV int randint(int low, int high) {  // randgen.cpp:6-11
        uint32_t x = rand_u32();
        uint32_t range = (uint32_t)(high - low);
        return (int)((uint32_t)low + (x % range));
    }
    struct EnvRng {  // rand_gen as pg_assetgen.h's Rng policy
        Env *e;
        PG_DEV uint32_t u32() { return e->rand_u32(); }
    };
    PG_DEV int randn(int high) { return (int)(rand_u32() % (uint32_t)high); }             // randgen.cpp:13-17
    PG_DEV float rand01() { return (float)((double)rand_u32() / 4294967296.0); }          // randgen.cpp:19-23
    // one draw from level_seed_rand_gen (state stays in HBM; a twist goes through scratch A)
    PG_DEV uint32_t level_seed_u32() {
        uint32_t *home = rg_home + MT_STRIDE;
        uint32_t z;
        if (G.lvl_rand_idx >= MT_N) {
            rc_buf = nullptr;  // (scratch A is used for the twist)
            mt_twist(home, mt_a());
            z = mt_a()[0];
            mt_copy(mt_a(), home);
            G.lvl_rand_idx = 1;
        } else {
            z = home[G.lvl_rand_idx];
            G.lvl_rand_idx += 1;
        }
        return mt_temper(z);
    }

    // ======================================================================================================
    // collision predicates: BAG:1068-1084,1126-1131,1145-1150
    PG_DEV bool has_collision_idx(int a, int b, float margin) {
        float tx = (erx(a) + erx(b)) + margin;
        float ty = (ery(a) + ery(b)) + margin;
        return (pg_fabsf(ex(a) - ex(b)) < tx) && (pg_fabsf(ey(a) - ey(b)) < ty);
    }
    PG_DEV bool is_out_of_bounds(int i) {
        float x = ex(i), y = ey(i), rx = erx(i), ry = ery(i);
        if (x + rx < 0) return true;
        if (y + ry < 0) return true;
        if (x - rx > G.main_width) return true;
        if (y - ry > G.main_height) return true;
        return false;
    }
    PG_DEV bool has_agent_collision(int i) {
        if (etype(i) == PLAYER) return false;
        return has_collision_idx(i, G.agent, ef(EF_COLLISION_MARGIN, i));
    }

    // ======================================================================================================
    // sub_step / push_obj: BAG:240-372.  The recursion (depth <= 5) is unrolled through the template depth.
    // ---- pruning the push recursion ----------------------------------------------------------------------------------
    // An object that touches k blocking entities at once (coinrun's agent on a pile of crates) is pushed out by each of them
    // at every level of the recursion: sub_step -> k x push_obj -> k x sub_step ... down to depth 5, ~k^5 calls, nearly all of
    // them asking for a displacement of zero from a position that no longer changes.  On a CPU that is microseconds; a lone
    // wave spent 1.3 M cycles on it (0.55 ms: 500 pushes in one env-step, 0.19 % of coinrun's env-steps -- and every step of
    // 65 536 envs waited for those, profiles/r03_wave_residency_*.txt).
    // The recursion only ever moves the object that is stepping (push_obj's target is the scanning object), and the hooks it
    // consults are functions of that object's x, y, vx, vy and of values that are constant while it steps (their only side
    // effect, coinrun's is_on_crate = 1, is idempotent).  So a nested sub_step is a function of (x, y, vx, vy, displacement,
    // depth), and one that ran without changing a bit of the object is a no-op: so is the same call again from the same state
    // at the same or a deeper level (induction over the remaining depth: it makes the same probes, finds the same hits, and
    // its own nested calls are the same calls one level further down).  Such calls are remembered (8 records in the upper
    // half of the lane scratch) and skipped; everything else runs as before.  PUSH_MEMO_OK: no hook writes entity words.
    static constexpr bool PUSH_MEMO_OK = !GameHasBlockHook<Game>::value;
    uint32_t push_chg = 0;  // times the recursion changed a bit of the stepping object
    int memo_n = 0;         // records in use (reset per object step)
    PG_DEV void obj_write(float &ref, float v) {
        if (__builtin_bit_cast(uint32_t, (float)ref) != __builtin_bit_cast(uint32_t, v)) {
            ref = v;
            push_chg++;
        }
    }
    // record r = s->tmp[64 + 8 r ..]: x, y, vx, vy, displacement (bit patterns), axis | depth << 1
    PG_DEV bool memo_hit(int obj, float disp, bool is_horizontal, int depth) {
        if (memo_n == 0) return false;
        const uint32_t bx = __builtin_bit_cast(uint32_t, (float)ex(obj)), by = __builtin_bit_cast(uint32_t, (float)ey(obj));
        const uint32_t bvx = __builtin_bit_cast(uint32_t, (float)evx(obj)), bvy = __builtin_bit_cast(uint32_t, (float)evy(obj));
        const uint32_t bd = __builtin_bit_cast(uint32_t, disp);
        const int n = memo_n < 8 ? memo_n : 8;
        return PG_BALLOT(l, ({
                             bool hit = false;
                             if (l < n) {
                                 const uint32_t *r = s->tmp + 64 + 8 * l;
                                 hit = r[0] == bx && r[1] == by && r[2] == bvx && r[3] == bvy && r[4] == bd && (int)(r[5] & 1u) == (is_horizontal ? 1 : 0) && (int)(r[5] >> 1) <= depth;
                             }
                             hit;
                         })) != 0;
    }
    PG_DEV void memo_insert(int obj, float disp, bool is_horizontal, int depth) {
        const uint32_t bx = __builtin_bit_cast(uint32_t, (float)ex(obj)), by = __builtin_bit_cast(uint32_t, (float)ey(obj));
        const uint32_t bvx = __builtin_bit_cast(uint32_t, (float)evx(obj)), bvy = __builtin_bit_cast(uint32_t, (float)evy(obj));
        uint32_t *r = s->tmp + 64 + 8 * (memo_n & 7);
        PG_FOR_LANES(l) {
            if (l == 0) {
                r[0] = bx; r[1] = by; r[2] = bvx; r[3] = bvy;
                r[4] = __builtin_bit_cast(uint32_t, disp);
                r[5] = (is_horizontal ? 1u : 0u) | ((uint32_t)depth << 1);
            }
        }
        memo_n++;
        PG_SYNC_E();
    }

    // The fixed point of the recursion, recognised before descending: inside sub_step(obj, a) at a depth >= 1 whose grid half
    // changed nothing, if every entity the scan would visit blocks `obj` and would push it by exactly `a` again (same axis, same
    // displacement bits -- in practice zero) and the velocity component a push clears is already +0, then every nested call is
    // this same call at the same state one level down; at depth 5 it makes no nested calls and changes nothing, so by induction
    // none of them changes anything at any depth.  The k^5 (or, for one hit, 5) nested calls are skipped; the hook's side
    // effects are applied in visiting order.  Returns true when the scan is settled that way (block2 = true).
    PG_DEV bool peek_blocked(int obj, int j, bool is_horizontal) {
        if constexpr (GameBlockedEntsEffect<Game>::value) return Game::is_blocked_ents_peek(*this, obj, j, is_horizontal);
        else return Game::is_blocked_ents(*this, obj, j, is_horizontal);
    }
    PG_DEV bool push_fixed_point(int obj, float _vx, float _vy, bool is_horizontal, int scan_axes, int otype, float orx, float ory) {
        const int n = ((scan_axes >> (is_horizontal ? 0 : 1)) & 1) ? G.n_ents : 0;
        if (n == 0) return false;
        if (__builtin_bit_cast(uint32_t, (float)(is_horizontal ? evx(obj) : evy(obj))) != 0u) return false;
        const float cx = ex(obj), cy = ey(obj);
        const uint32_t dispb = __builtin_bit_cast(uint32_t, is_horizontal ? _vx : _vy);
        bool any = false;
        for (int c = (n - 1) >> 6; c >= 0; c--) {
            // per lane: 0 = not visited, 1 = blocks and pushes by `a` again, 2 = anything else
            PG_LANE_VAR(int, code);
            PG_FOR_LANES(l) {
                const int idx = (c << 6) + l;
                int cd = 0;
                if (idx < n && idx != obj) {
                    const uint32_t mm = meta(idx);
                    if (!(mm & MF_WILL_ERASE) && Game::may_interact(*this, otype, meta_type(mm), is_horizontal)) {
                        const float tx = (orx + erx(idx)) + POS_EPS;
                        const float ty = (ory + ery(idx)) + POS_EPS;
                        if ((pg_fabsf(cx - ex(idx)) < tx) && (pg_fabsf(cy - ey(idx)) < ty)) {
                            cd = 2;
                            if (peek_blocked(obj, idx, is_horizontal)) {
                                // push_obj(src = idx, target = obj): BAG:240-250
                                const float rsum = is_horizontal ? (erx(idx) + orx) : (ery(idx) + ory);
                                float t;
                                if (is_horizontal) t = (float)((double)ex(idx) + sign_d((double)(cx - ex(idx))) * (double)rsum - (double)cx);
                                else t = (float)((double)ey(idx) + sign_d((double)(cy - ey(idx))) * (double)rsum - (double)cy);
                                const bool same_axis = is_horizontal ? (t != 0) : true;  // (a zero horizontal push is a vertical call)
                                if (same_axis && __builtin_bit_cast(uint32_t, t) == dispb) cd = 1;
                            }
                        }
                    }
                }
                PG_LV(code, l) = cd;
            }
            if (PG_BALLOT(l, PG_LV(code, l) == 2) != 0) return false;
            any = any || PG_BALLOT(l, PG_LV(code, l) == 1) != 0;
        }
        if (!any) return false;
        if constexpr (GameBlockedEntsEffect<Game>::value) {
            for (int c = (n - 1) >> 6; c >= 0; c--) {
                uint64_t m = PG_BALLOT(l, ({
                                           const int idx = (c << 6) + l;
                                           bool hit = false;
                                           if (idx < n && idx != obj) {
                                               const uint32_t mm = meta(idx);
                                               if (!(mm & MF_WILL_ERASE) && Game::may_interact(*this, otype, meta_type(mm), is_horizontal)) {
                                                   const float tx = (orx + erx(idx)) + POS_EPS;
                                                   const float ty = (ory + ery(idx)) + POS_EPS;
                                                   hit = (pg_fabsf(cx - ex(idx)) < tx) && (pg_fabsf(cy - ey(idx)) < ty);
                                               }
                                           }
                                           hit;
                                       }));
                while (m) {
                    const int jl = pg_highest(m);
                    m &= ~(1ull << jl);
                    (void)Game::is_blocked_ents(*this, obj, (c << 6) + jl, is_horizontal);
                }
            }
        }
#if defined(PGAMD_WAVE_EMU)
        pg_emu_counters()[6] += 1;  // scans settled as fixed points
#endif
        return true;
    }

    template <int DEPTH>
    PG_DEV void push_obj(int src, int target, bool is_horizontal, int scan_axes) {  // BAG:240-268
        float rsum = is_horizontal ? (erx(src) + erx(target)) : (ery(src) + ery(target));
        float delx = ex(target) - ex(src);
        float dely = ey(target) - ey(src);
        float t_vx = 0, t_vy = 0;
        if (is_horizontal) t_vx = (float)((double)ex(src) + sign_d((double)delx) * (double)rsum - (double)ex(target));
        else t_vy = (float)((double)ey(src) + sign_d((double)dely) * (double)rsum - (double)ey(target));
        if constexpr (DEPTH < 5) {
            if constexpr (PUSH_MEMO_OK) {
                // (sub_step's axis is `_vx != 0`: a zero horizontal displacement is a vertical call)
                const bool axis_h = t_vx != 0;
                const float disp = axis_h ? t_vx : t_vy;
                if (PG_DBG(d, 65536) || !memo_hit(target, disp, axis_h, DEPTH + 1)) {
                    const uint32_t c0 = push_chg;
                    sub_step<DEPTH + 1>(target, t_vx, t_vy, scan_axes);
                    if (push_chg == c0 && !PG_DBG(d, 65536)) memo_insert(target, disp, axis_h, DEPTH + 1);
#if defined(PGAMD_WAVE_EMU)
                    pg_emu_counters()[5] += 1;  // nested sub_steps run
#endif
                } else {
#if defined(PGAMD_WAVE_EMU)
                    pg_emu_counters()[6] += 1;  // ... skipped
#endif
                }
            } else {
                sub_step<DEPTH + 1>(target, t_vx, t_vy, scan_axes);
            }
        }
        if (is_horizontal) obj_write(evx(target), 0.0f);
        else obj_write(evy(target), 0.0f);
    }

    // scan_axes: bit 0 / bit 1 = some entity could block or reflect `obj` on a horizontal / vertical move
    // (computed once per object by basic_step_object); when clear, the entity scan of BAG:337-369 is a no-op.
    template <int DEPTH>
    PG_DEV bool sub_step(int obj, float _vx, float _vy, int scan_axes) {  // BAG:270-372
        if (eflag(obj, MF_WILL_ERASE)) return false;
        const uint32_t chg_in = push_chg;
        const int otype = etype(obj);
        const float orx = erx(obj), ory = ery(obj);
        float ny = ey(obj) + _vy;
        float nx = ex(obj) + _vx;
        const float margin = 0.98f;
        const bool is_horizontal = _vx != 0;
        bool block = false, reflect = false;
        {
            // the four corner probes of BAG:284-290 share two x and two y coordinates: floor each once
            const float mx = orx * margin, my = ory * margin;
            const float px[2] = {nx + mx * -1, nx + mx * 1};
            const float py[2] = {ny + my * -1, ny + my * 1};
            int cxi[2], cyi[2];
            bool xneg[2], yneg[2];
            for (int k = 0; k < 2; k++) {
                xneg[k] = px[k] < 0;
                yneg[k] = py[k] < 0;
                cxi[k] = (int)pg_floorf(px[k]);
                cyi[k] = (int)pg_floorf(py[k]);
            }
            for (int i = 0; i < 2; i++)
                for (int j = 0; j < 2; j++) {
                    const int type2 = (xneg[i] || yneg[j]) ? G.out_of_bounds_object : get_obj(cxi[i], cyi[j]);
                    block = block || Game::is_blocked(*this, otype, type2, is_horizontal);
                    reflect = reflect || Game::will_reflect(otype, type2);
                }
        }
        if constexpr (GameHasBlockHook<Game>::value) {
            if (block) Game::on_grid_block(*this, obj);
        }
        if (reflect) {
            if (is_horizontal) {
                float delta;
                if (_vx < 0) delta = (float)(pg_ceil((double)(nx - orx)) - (double)(nx - orx));
                else delta = (float)(pg_floor((double)(nx + orx)) - (double)(nx + orx));
                obj_write(evx(obj), -1 * evx(obj));
                nx = nx + 2 * delta;
            } else {
                float delta;
                if (_vy < 0) delta = (float)(pg_ceil((double)(ny - ory)) - (double)(ny - ory));
                else delta = (float)(pg_floor((double)(ny + ory)) - (double)(ny + ory));
                obj_write(evy(obj), -1 * evy(obj));
                ny = ny + 2 * delta;
            }
        } else if (block) {
            if (is_horizontal) {
                if (G.grid_step) nx = ex(obj);
                else nx = (float)(_vx > 0 ? (pg_floor((double)(nx + orx)) - (double)orx) : (pg_ceil((double)(nx - orx)) + (double)orx));
            } else {
                if (G.grid_step) ny = ey(obj);
                else ny = (float)(_vy > 0 ? (pg_floor((double)(ny + ory)) - (double)ory) : (pg_ceil((double)(ny - ory)) + (double)ory));
            }
        }
        obj_write(ex(obj), nx);
        obj_write(ey(obj), ny);
        PG_SYNC_E();

        if constexpr (PUSH_MEMO_OK && DEPTH >= 1 && DEPTH < 5) {
            if (push_chg == chg_in && !PG_DBG(d, 65536) && push_fixed_point(obj, _vx, _vy, is_horizontal, scan_axes, otype, orx, ory)) return true;
        }
        return entity_scan<DEPTH>(obj, _vx, _vy, is_horizontal, scan_axes, otype, orx, ory) || block;
    }

    // reverse scan of the entity list (BAG:337-369): broad phase = one ballot per 64 entities, hits are
    // visited from the highest index down; the ballot is re-evaluated only after a hit moved `obj`.  Works on the
    // LDS copy of `obj`; returns block2.
    template <int DEPTH>
    PG_DEV bool entity_scan(int obj, float _vx, float _vy, bool is_horizontal, int scan_axes, int otype, float orx, float ory) {
        bool block2 = false;
        const int n = ((scan_axes >> (is_horizontal ? 0 : 1)) & 1) ? G.n_ents : 0;
        for (int c = (n - 1) >> 6; c >= 0; c--) {
            int limit = 64;  // lanes >= limit of this chunk have been visited
            bool need_ballot = true;
            uint64_t m = 0;
            while (true) {
                if (need_ballot) {
                    const float cx = ex(obj), cy = ey(obj);
                    m = PG_BALLOT(l, ({
                                      const int idx = (c << 6) + l;
                                      bool hit = false;
                                      if (l < limit && idx < n && idx != obj) {
                                          const uint32_t mm = meta(idx);
                                          // only entities that can block or reflect `obj` need a visit (a hit with any
                                          // other entity has no effect in BAG:346-366)
                                          if (!(mm & MF_WILL_ERASE) && Game::may_interact(*this, otype, meta_type(mm), is_horizontal)) {
                                              const float tx = (orx + erx(idx)) + POS_EPS;
                                              const float ty = (ory + ery(idx)) + POS_EPS;
                                              hit = (pg_fabsf(cx - ex(idx)) < tx) && (pg_fabsf(cy - ey(idx)) < ty);
                                          }
                                      }
                                      hit;
                                  }));
                    need_ballot = false;
                }
                if (m == 0) break;
                const int jl = pg_highest(m);
                m &= ~(1ull << jl);
                limit = jl;
                const int j = (c << 6) + jl;
                bool curr_block = false;
                bool moved = false;
                if (Game::is_blocked_ents(*this, obj, j, is_horizontal)) {
                    curr_block = true;
                } else if (Game::will_reflect(otype, etype(j))) {
                    if (is_horizontal) {
                        float delx = ex(j) - ex(obj);
                        float rsum = erx(j) + orx;
                        obj_write(ex(obj), ex(obj) + (_vx > 0 ? -2 * (rsum - delx) : 2 * (rsum + delx)));
                        obj_write(evx(obj), -1 * evx(obj));
                    } else {
                        float dely = ey(j) - ey(obj);
                        float rsum = ery(j) + ory;
                        obj_write(ey(obj), ey(obj) + (_vy > 0 ? -2 * (rsum - dely) : 2 * (rsum + dely)));
                        obj_write(evy(obj), -1 * evy(obj));
                    }
                    moved = true;
                }
                if (curr_block) {
                    trace_add(2, 1);
                    push_obj<DEPTH>(j, obj, is_horizontal, scan_axes);
                    moved = true;
                }
                block2 = block2 || curr_block;
                if (moved) {
                    PG_SYNC_E();
                    need_ballot = true;
                }
            }
        }
        return block2;
    }

    // ---- top level of basic_step_object with the stepping entity held in registers --------------------------------
    // A smart_step entity makes >= 8 sub_steps per env-step and each one used to be a chain of dependent LDS reads and
    // writes of the entity's own fields; step_entities was two thirds of the step kernel's cycles
    // (profiles/r01_phase_cycles_coinrun.txt).  The common sub_step touches only the grid and -- through one ballot --
    // the other entities' boxes, so the entity's x, y, vx, vy, rx, ry live in registers for the whole object step and
    // go back to LDS only when an entity is actually hit (push / reflect recursion works on LDS) or a hook needs them.
    struct ObjRegs {
        float x, y, vx, vy, rx, ry;
        int type;
        // the four corner cells of the last probe per move direction and what they said: most sub_steps stay inside
        // the same cells, and neither the grid nor the hooks' answers change while one object steps
        int pc[2][4];
        int pneg[2];
        bool pvalid[2], pblock[2], preflect[2];
        bool eventful;  // the last sub_step reflected, was blocked or touched another entity
    };
    PG_DEV void obj_load(int obj, ObjRegs &R) {
        R.x = ex(obj); R.y = ey(obj); R.vx = evx(obj); R.vy = evy(obj); R.rx = erx(obj); R.ry = ery(obj);
        R.type = etype(obj);
        R.pvalid[0] = R.pvalid[1] = false;
        R.eventful = false;
    }
    PG_DEV void obj_flush(int obj, const ObjRegs &R) {
        ex(obj) = R.x; ey(obj) = R.y; evx(obj) = R.vx; evy(obj) = R.vy;
        PG_SYNC_E();
    }
    PG_DEV bool sub_step_top(int obj, ObjRegs &R, float _vx, float _vy, int scan_axes) {  // BAG:270-372, depth 0
        const int otype = R.type;
        const float orx = R.rx, ory = R.ry;
        float ny = R.y + _vy;
        float nx = R.x + _vx;
        const float margin = 0.98f;
        const bool is_horizontal = _vx != 0;
        bool block = false, reflect = false;
        {
            const float mx = orx * margin, my = ory * margin;
            const float px[2] = {nx + mx * -1, nx + mx * 1};
            const float py[2] = {ny + my * -1, ny + my * 1};
            int cxi[2], cyi[2];
            bool xneg[2], yneg[2];
            for (int k = 0; k < 2; k++) {
                xneg[k] = px[k] < 0;
                yneg[k] = py[k] < 0;
                cxi[k] = (int)pg_floorf(px[k]);
                cyi[k] = (int)pg_floorf(py[k]);
            }
            const int h = is_horizontal ? 1 : 0;
            const int neg = (xneg[0] ? 1 : 0) | (xneg[1] ? 2 : 0) | (yneg[0] ? 4 : 0) | (yneg[1] ? 8 : 0);
            if (R.pvalid[h] && R.pc[h][0] == cxi[0] && R.pc[h][1] == cxi[1] && R.pc[h][2] == cyi[0] && R.pc[h][3] == cyi[1] && R.pneg[h] == neg) {
                block = R.pblock[h];
                reflect = R.preflect[h];
            } else {
                for (int i = 0; i < 2; i++)
                    for (int j = 0; j < 2; j++) {
                        const int type2 = (xneg[i] || yneg[j]) ? G.out_of_bounds_object : get_obj(cxi[i], cyi[j]);
                        block = block || Game::is_blocked(*this, otype, type2, is_horizontal);
                        reflect = reflect || Game::will_reflect(otype, type2);
                    }
                R.pvalid[h] = true;
                R.pc[h][0] = cxi[0]; R.pc[h][1] = cxi[1]; R.pc[h][2] = cyi[0]; R.pc[h][3] = cyi[1];
                R.pneg[h] = neg;
                R.pblock[h] = block;
                R.preflect[h] = reflect;
            }
        }
        if constexpr (GameHasBlockHook<Game>::value) {
            if (block) {
                obj_flush(obj, R);
                Game::on_grid_block(*this, obj);
                PG_SYNC_E();
                R.vx = evx(obj);
                R.vy = evy(obj);
            }
        }
        if (reflect) {
            if (is_horizontal) {
                float delta;
                if (_vx < 0) delta = (float)(pg_ceil((double)(nx - orx)) - (double)(nx - orx));
                else delta = (float)(pg_floor((double)(nx + orx)) - (double)(nx + orx));
                R.vx = -1 * R.vx;
                nx = nx + 2 * delta;
            } else {
                float delta;
                if (_vy < 0) delta = (float)(pg_ceil((double)(ny - ory)) - (double)(ny - ory));
                else delta = (float)(pg_floor((double)(ny + ory)) - (double)(ny + ory));
                R.vy = -1 * R.vy;
                ny = ny + 2 * delta;
            }
        } else if (block) {
            if (is_horizontal) {
                if (G.grid_step) nx = R.x;
                else nx = (float)(_vx > 0 ? (pg_floor((double)(nx + orx)) - (double)orx) : (pg_ceil((double)(nx - orx)) + (double)orx));
            } else {
                if (G.grid_step) ny = R.y;
                else ny = (float)(_vy > 0 ? (pg_floor((double)(ny + ory)) - (double)ory) : (pg_ceil((double)(ny - ory)) + (double)ory));
            }
        }
        R.x = nx;
        R.y = ny;
        // does the entity scan find anything at all?  (the same broad phase as entity_scan, from the register copy)
        const int n = ((scan_axes >> (is_horizontal ? 0 : 1)) & 1) ? G.n_ents : 0;
        bool any_hit = false;
        for (int c = (n - 1) >> 6; c >= 0 && !any_hit; c--) {
            const uint64_t m = PG_BALLOT(l, ({
                                             const int idx = (c << 6) + l;
                                             bool hit = false;
                                             if (idx < n && idx != obj) {
                                                 const uint32_t mm = meta(idx);
                                                 if (!(mm & MF_WILL_ERASE) && Game::may_interact(*this, otype, meta_type(mm), is_horizontal)) {
                                                     const float tx = (orx + erx(idx)) + POS_EPS;
                                                     const float ty = (ory + ery(idx)) + POS_EPS;
                                                     hit = (pg_fabsf(nx - ex(idx)) < tx) && (pg_fabsf(ny - ey(idx)) < ty);
                                                 }
                                             }
                                             hit;
                                         }));
            any_hit = m != 0;
        }
        if (block || reflect) R.eventful = true;
        if (!any_hit) return block;
        R.eventful = true;
        obj_flush(obj, R);
#if !defined(PGAMD_WAVE_EMU)
        const long long t_scan0 = PG_TRACE(d) ? (long long)__builtin_readcyclecounter() : 0;
#endif
        const bool block2 = entity_scan<0>(obj, _vx, _vy, is_horizontal, scan_axes, otype, orx, ory);
        PG_SYNC_E();
#if !defined(PGAMD_WAVE_EMU)
        if (PG_TRACE(d)) {
            trace_add(1, 1);
            trace_add(3, (unsigned long long)((long long)__builtin_readcyclecounter() - t_scan0));
        }
#endif
        R.x = ex(obj); R.y = ey(obj); R.vx = evx(obj); R.vy = evy(obj);
        return block || block2;
    }

    // which axes of `obj` need the entity scan of sub_step at all this step: bit 0 / 1 = some entity that could block or reflect
    // it on a horizontal / vertical move lies within its reach (entity types do not change while it steps)
    PG_DEV int bso_scan_axes(int obj) {
        const int otype = etype(obj);
        const int n = G.n_ents;
        // Does any entity this object could interact with lie within its reach for this step (its own travel, < 1 cell of
        // block snapping, < 2 of a reflection)?  If not -- the usual case -- the per-sub_step scans cannot find anything.
        const float ox = ex(obj), oy = ey(obj), orx = erx(obj), ory = ery(obj);
        const float reach_x = pg_fabsf(evx(obj)) + 2.01f, reach_y = pg_fabsf(evy(obj)) + 2.01f;
        bool any = false;
        for (int c = 0; c < ((n + 63) >> 6) && !any; c++) {
            any = PG_BALLOT(l, ({
                                const int idx = (c << 6) + l;
                                bool near = false;
                                if (idx < n && idx != obj) {
                                    // (the target may itself move before `obj` meets it -- the reference steps higher indices
                                    // first, the parallel pass asks before anyone has moved: its own travel widens the reach)
                                    const uint32_t mt = meta(idx);
                                    const float tsl = (mt & MF_SMART_STEP) ? 2.01f : 0.0f;
                                    if ((pg_fabsf(ox - ex(idx)) < orx + erx(idx) + reach_x + (pg_fabsf(evx(idx)) + tsl)) && (pg_fabsf(oy - ey(idx)) < ory + ery(idx) + reach_y + (pg_fabsf(evy(idx)) + tsl))) {
                                        const int t = meta_type(mt);
                                        near = Game::may_interact(*this, otype, t, true) || Game::may_interact(*this, otype, t, false);
                                    }
                                }
                                near;
                            })) != 0;
        }
        if (!any) return 0;
        int scan_axes = 0;
        for (int c = 0; c < ((n + 63) >> 6) && scan_axes != 3; c++) {
            const uint64_t mh = PG_BALLOT(l, ((c << 6) + l) < n && ((c << 6) + l) != obj && Game::may_interact(*this, otype, etype((c << 6) + l), true));
            const uint64_t mv = PG_BALLOT(l, ((c << 6) + l) < n && ((c << 6) + l) != obj && Game::may_interact(*this, otype, etype((c << 6) + l), false));
            scan_axes |= (mh ? 1 : 0) | (mv ? 2 : 0);
        }
        return scan_axes;
    }

    // ---- free objects: sub_steps evaluated side by side ---------------------------------------------------------------
    // An object that no entity can block or reflect this step (scan axes 0) steps through the grid alone: sub_step k reads
    // the grid and the object's own position, nothing else.  Its 4-8 sub_steps were the longest dependent chain of the step
    // kernel (profiles/r03_phase_cycles_coinrun.txt: "bso: sub_steps").  They are now evaluated by 8 lanes at once, lane i
    // taking sub_step k0 + i from a PREDICTED start state -- per axis either "moves freely" (k0's position plus i times the
    // same float add the serial loop would make) or "stuck" (blocked and snapped back to where it was: a grounded agent's y
    // axis) -- and the longest prefix whose outcomes equal their successors' predicted starts is accepted, bit for bit what
    // the serial loop computes.  The first lane that disagrees has still done its own sub_step from a correct start, so a
    // round accepts at least one sub_step and re-predicts from what that one did; two rounds serve the common cases
    // (free flight: one).  Eight objects go side by side (lane = object * 8 + sub_step).
    struct FreeState {
        float x, y, vx, vy;
    };
    // the grid half of sub_step (BAG:270-334) for one axis move on a register copy (no entity scan: scan axes are 0)
    PG_DEV bool sub_step_grid(FreeState &R, int otype, float orx, float ory, float _vx, float _vy) {
        float ny = R.y + _vy;
        float nx = R.x + _vx;
        const float margin = 0.98f;
        const bool is_horizontal = _vx != 0;
        bool block = false, reflect = false;
        {
            const float mx = orx * margin, my = ory * margin;
            const float px[2] = {nx + mx * -1, nx + mx * 1};
            const float py[2] = {ny + my * -1, ny + my * 1};
            int cxi[2], cyi[2];
            bool xneg[2], yneg[2];
            for (int k = 0; k < 2; k++) {
                xneg[k] = px[k] < 0;
                yneg[k] = py[k] < 0;
                cxi[k] = (int)pg_floorf(px[k]);
                cyi[k] = (int)pg_floorf(py[k]);
            }
            for (int i = 0; i < 2; i++)
                for (int j = 0; j < 2; j++) {
                    const int type2 = (xneg[i] || yneg[j]) ? G.out_of_bounds_object : get_obj(cxi[i], cyi[j]);
                    block = block || Game::is_blocked(*this, otype, type2, is_horizontal);
                    reflect = reflect || Game::will_reflect(otype, type2);
                }
        }
        if (reflect) {
            if (is_horizontal) {
                float delta;
                if (_vx < 0) delta = (float)(pg_ceil((double)(nx - orx)) - (double)(nx - orx));
                else delta = (float)(pg_floor((double)(nx + orx)) - (double)(nx + orx));
                R.vx = -1 * R.vx;
                nx = nx + 2 * delta;
            } else {
                float delta;
                if (_vy < 0) delta = (float)(pg_ceil((double)(ny - ory)) - (double)(ny - ory));
                else delta = (float)(pg_floor((double)(ny + ory)) - (double)(ny + ory));
                R.vy = -1 * R.vy;
                ny = ny + 2 * delta;
            }
        } else if (block) {
            if (is_horizontal) nx = (float)(_vx > 0 ? (pg_floor((double)(nx + orx)) - (double)orx) : (pg_ceil((double)(nx - orx)) + (double)orx));
            else ny = (float)(_vy > 0 ? (pg_floor((double)(ny + ory)) - (double)ory) : (pg_ceil((double)(ny - ory)) + (double)ory));
        }
        R.x = nx;
        R.y = ny;
        return block;
    }
    static constexpr bool FREE_OBJECTS_OK = !GameHasBlockHook<Game>::value;  // (a block hook writes entity words from inside sub_step)
    // basic_step_object (BAG:593-656) for the objects s->tmp[first .. first + count), count <= 8, all alive, with scan axes 0,
    // in a game that is not grid_step
    PG_DEV void bso_free_objects(int first, int count) {
        PG_LANE_VAR(uint32_t, sx);  // true state of the lane's object at its sub_step k0 (bit patterns; the 8 lanes of an object agree)
        PG_LANE_VAR(uint32_t, sy);
        PG_LANE_VAR(uint32_t, svx);
        PG_LANE_VAR(uint32_t, svy);
        PG_LANE_VAR(uint32_t, ox);  // outcome of this lane's sub_step
        PG_LANE_VAR(uint32_t, oy);
        PG_LANE_VAR(uint32_t, ovx);
        PG_LANE_VAR(uint32_t, ovy);
        PG_LANE_VAR(int, oflags);  // 1 block_x, 2 block_y, 4 the accepted prefix ends here, 8 active, 16 / 32 x / y came out where it went in
        PG_LANE_VAR(int, k0);
        PG_LANE_VAR(int, nsub);
        PG_LANE_VAR(int, cntx);  // sub_steps that did not block x / y so far (vx_pct / vy_pct before the division)
        PG_LANE_VAR(int, cnty);
        PG_LANE_VAR(int, pred);  // 1: x predicted stuck, 2: y predicted stuck
        PG_FOR_LANES(l) {
            const int g = l >> 3;
            PG_LV(k0, l) = 0;
            PG_LV(nsub, l) = 0;
            PG_LV(cntx, l) = 0;
            PG_LV(cnty, l) = 0;
            PG_LV(pred, l) = 0;
            PG_LV(oflags, l) = 0;
            PG_LV(sx, l) = PG_LV(sy, l) = PG_LV(svx, l) = PG_LV(svy, l) = 0u;
            PG_LV(ox, l) = PG_LV(oy, l) = PG_LV(ovx, l) = PG_LV(ovy, l) = 0u;
            if (g < count) {
                const int obj = (int)s->tmp[first + g];
                const float vx = evx(obj), vy = evy(obj);
                int n = (int)(4 * pg_sqrt((double)(vx * vx + vy * vy)));  // double sqrt, see oracle note
                if (n < 4) n = 4;
                PG_LV(nsub, l) = n;
                PG_LV(sx, l) = __builtin_bit_cast(uint32_t, ex(obj));
                PG_LV(sy, l) = __builtin_bit_cast(uint32_t, ey(obj));
                PG_LV(svx, l) = __builtin_bit_cast(uint32_t, vx);
                PG_LV(svy, l) = __builtin_bit_cast(uint32_t, vy);
            }
        }
        while (true) {
            PG_FOR_LANES(l) {
                const int g = l >> 3, i = l & 7;
                int fl = 0;
                if (g < count && PG_LV(k0, l) + i < PG_LV(nsub, l)) {
                    const int obj = (int)s->tmp[first + g];
                    const int n = PG_LV(nsub, l);
                    const float pct = (float)(1.0 / n);
                    const float vx0 = evx(obj), vy0 = evy(obj);  // (untouched until the object is finished)
                    const float cmp = pg_fabsf(vx0) - pg_fabsf(vy0);
                    const int otype = etype(obj);
                    bool step_x_first = cmp == 0 ? (G.step_rand_int % 2 == 0) : (cmp > 0);
                    if (otype == PLAYER) {
                        if (G.action_vx != 0) step_x_first = true;
                        if (G.action_vy != 0) step_x_first = false;
                    }
                    const float orx = erx(obj), ory = ery(obj);
                    FreeState R;
                    R.x = __builtin_bit_cast(float, PG_LV(sx, l));
                    R.y = __builtin_bit_cast(float, PG_LV(sy, l));
                    R.vx = __builtin_bit_cast(float, PG_LV(svx, l));
                    R.vy = __builtin_bit_cast(float, PG_LV(svy, l));
                    const int pr = PG_LV(pred, l);
                    const float dvx = R.vx * pct, dvy = R.vy * pct;
                    for (int t = 0; t < i; t++) {  // predicted start of sub_step k0 + i
                        if (!(pr & 1)) R.x = R.x + dvx;
                        if (!(pr & 2)) R.y = R.y + dvy;
                    }
                    const float in_x = R.x, in_y = R.y;
                    const float next_x = (pr & 1) ? R.x : R.x + dvx, next_y = (pr & 2) ? R.y : R.y + dvy;  // ... and of its successor
                    for (int h = 0; h < 2; h++) {  // BAG:627-640
                        const bool xaxis = (h == 0) == step_x_first;
                        const float mvx = xaxis ? R.vx * pct : 0.0f;
                        const float mvy = xaxis ? 0.0f : R.vy * pct;
                        if (sub_step_grid(R, otype, orx, ory, mvx, mvy)) fl |= xaxis ? 1 : 2;
                    }
                    const uint32_t bx = __builtin_bit_cast(uint32_t, R.x), by = __builtin_bit_cast(uint32_t, R.y);
                    const uint32_t bvx = __builtin_bit_cast(uint32_t, R.vx), bvy = __builtin_bit_cast(uint32_t, R.vy);
                    const bool as_predicted = bx == __builtin_bit_cast(uint32_t, next_x) && by == __builtin_bit_cast(uint32_t, next_y) && bvx == PG_LV(svx, l) && bvy == PG_LV(svy, l);
                    if (!as_predicted || PG_LV(k0, l) + i == n - 1 || (fl & 3) == 3) fl |= 4;
                    fl |= 8;
                    if (bx == __builtin_bit_cast(uint32_t, in_x)) fl |= 16;
                    if (by == __builtin_bit_cast(uint32_t, in_y)) fl |= 32;
                    PG_LV(ox, l) = bx;
                    PG_LV(oy, l) = by;
                    PG_LV(ovx, l) = bvx;
                    PG_LV(ovy, l) = bvy;
                }
                PG_LV(oflags, l) = fl;
            }
            const uint64_t m_act = PG_BALLOT(l, (PG_LV(oflags, l) & 8) != 0);
            if (m_act == 0) break;
#if defined(PGAMD_WAVE_EMU)
            pg_emu_counters()[1] += 1;                  // rounds
            pg_emu_counters()[2] += pg_popc64(m_act);   // sub_steps evaluated (accepted or not)
#endif
            const uint64_t m_stop = PG_BALLOT(l, (PG_LV(oflags, l) & 4) != 0);
            const uint64_t m_nbx = PG_BALLOT(l, (PG_LV(oflags, l) & 9) == 8);
            const uint64_t m_nby = PG_BALLOT(l, (PG_LV(oflags, l) & 10) == 8);
            PG_FOR_LANES(l) {
                const int g = l >> 3;
                const uint32_t act = (uint32_t)(m_act >> (g * 8)) & 0xffu;
                if (act) {
                    const uint32_t stopm = (uint32_t)(m_stop >> (g * 8)) & 0xffu;  // (never 0: the object's last sub_step stops)
                    const int j = pg_ctz64((uint64_t)stopm);
                    const uint32_t pre = (2u << j) - 1u;
                    PG_LV(cntx, l) += pg_popc64((uint64_t)((uint32_t)(m_nbx >> (g * 8)) & pre));
                    PG_LV(cnty, l) += pg_popc64((uint64_t)((uint32_t)(m_nby >> (g * 8)) & pre));
                    const int src = g * 8 + j;
                    const int fj = PG_SHFL(oflags, l, src);
                    const uint32_t nx = PG_SHFL(ox, l, src), ny = PG_SHFL(oy, l, src), nvx = PG_SHFL(ovx, l, src), nvy = PG_SHFL(ovy, l, src);
                    PG_LV(sx, l) = nx;
                    PG_LV(sy, l) = ny;
                    PG_LV(svx, l) = nvx;
                    PG_LV(svy, l) = nvy;
                    PG_LV(pred, l) = (fj >> 4) & 3;
                    PG_LV(k0, l) = (fj & 3) == 3 ? PG_LV(nsub, l) : PG_LV(k0, l) + j + 1;  // both axes blocked: the reference loop breaks
                }
            }
        }
#if defined(PGAMD_WAVE_EMU)
        pg_emu_counters()[3] += 1;      // calls
        pg_emu_counters()[4] += count;  // objects
#endif
        PG_FOR_LANES(l) {
            const int g = l >> 3;
            if (g < count && (l & 7) == 0) {
                const int obj = (int)s->tmp[first + g];
                const int n = PG_LV(nsub, l);
                float vx_pct = (float)PG_LV(cntx, l), vy_pct = (float)PG_LV(cnty, l);
                vx_pct = vx_pct / n;
                vy_pct = vy_pct / n;
                ex(obj) = __builtin_bit_cast(float, PG_LV(sx, l));
                ey(obj) = __builtin_bit_cast(float, PG_LV(sy, l));
                evx(obj) = __builtin_bit_cast(float, PG_LV(svx, l)) * vx_pct;
                evy(obj) = __builtin_bit_cast(float, PG_LV(svy, l)) * vy_pct;
            }
        }
        PG_SYNC_E();
    }

    PG_DEV void basic_step_object(int obj) {  // BAG:593-656
        if (eflag(obj, MF_WILL_ERASE)) return;
        const int scan_axes = bso_scan_axes(obj);
        if constexpr (FREE_OBJECTS_OK) {
            if (scan_axes == 0 && !G.grid_step && !PG_DBG(d, 32768)) {
                s->tmp[0] = (uint32_t)obj;
                PG_SYNC_E();
                bso_free_objects(0, 1);
                return;
            }
        }
        bso_core<true>(obj, scan_axes);
    }

    // basic_step_object once the scan axes are known.  With scan_axes == 0 it reads and writes nothing but `obj`'s own
    // words, the grid and scalars of G: wave = env kernels then run it for several objects at once, one lane per object
    // (WAVE_UNIFORM = false: no profiling marks, which assume a uniform call).
    template <bool WAVE_UNIFORM>
    PG_DEV void bso_core(int obj, int scan_axes) {
        if constexpr (WAVE_UNIFORM) memo_n = 0;  // (push recursion records belong to one object's step)
        int num_sub_steps;
        {
            const float vx = evx(obj), vy = evy(obj);
            if (G.grid_step) {
                num_sub_steps = 1;
            } else {
                num_sub_steps = (int)(4 * pg_sqrt((double)(vx * vx + vy * vy)));  // double sqrt, see oracle note
                if (num_sub_steps < 4) num_sub_steps = 4;
            }
        }
        const float pct = (float)(1.0 / num_sub_steps);
        const float cmp = pg_fabsf(evx(obj)) - pg_fabsf(evy(obj));
        bool step_x_first = cmp == 0 ? (G.step_rand_int % 2 == 0) : (cmp > 0);
        if (etype(obj) == PLAYER) {
            if (G.action_vx != 0) step_x_first = true;
            if (G.action_vy != 0) step_x_first = false;
        }
        ObjRegs R;
        obj_load(obj, R);
        if constexpr (WAVE_UNIFORM) phase(9);
        float vx_pct = 0, vy_pct = 0;
        for (int st = 0; st < num_sub_steps; st++) {
            bool block_x = false, block_y = false;
            R.eventful = false;
            trace_add(0, 1);
            for (int h = 0; h < 2; h++) {  // one call site for sub_step_top
                const bool xaxis = (h == 0) == step_x_first;
                const float dvx = xaxis ? R.vx * pct : 0.0f;
                const float dvy = xaxis ? 0.0f : R.vy * pct;
                const bool b = sub_step_top(obj, R, dvx, dvy, scan_axes);
                if (xaxis) block_x = b;
                else block_y = b;
            }
            if (!block_x) vx_pct += 1;
            if (!block_y) vy_pct += 1;
            if (block_x && block_y) break;
            if (!R.eventful && R.vx * pct == 0 && R.vy * pct == 0) {
                // an object at rest that nothing blocked, reflected or touched: the remaining sub_steps see the same state
                vx_pct += (float)(num_sub_steps - 1 - st);
                vy_pct += (float)(num_sub_steps - 1 - st);
                break;
            }
        }
        vx_pct = vx_pct / num_sub_steps;
        vy_pct = vy_pct / num_sub_steps;
        R.vx *= vx_pct;
        R.vy *= vy_pct;
        obj_flush(obj, R);
        if constexpr (WAVE_UNIFORM) phase(10);
    }

    // step_entities BAG:1086-1098: reverse order; runs of non-smart entities are stepped lane-parallel,
    // smart_step entities (agent, walkers) serially in their list position.
    PG_DEV void step_entities() {
        const int n0 = G.n_ents;
        if constexpr (GameParSmart<Game>::value) {
            // Parallel pass.  A smart entity that no entity can block or reflect this step (scan axes 0) steps through the
            // grid alone: its basic_step_object reads and writes its own words only, so it commutes with every other
            // entity's step, and the wave's idle lanes can take one such object each -- an env with ten walking enemies
            // pays for one object step instead of ten.  The objects done here are marked and then ride with the plain
            // entities (Entity::step) in the ordered loop below, which keeps only the objects that do interact.
            int ns = 0;
            for (int c = 0; c < ((n0 + 63) >> 6); c++) ns += pg_popc64(PG_BALLOT(l, ((c << 6) + l) < n0 && (meta((c << 6) + l) & MF_SMART_STEP) != 0));
            if (ns >= 2 && !PG_DBG(d, 32768)) {  // (PROCGEN_AMD_DEBUG & 32768: A/B switch, every object in the ordered loop)
                int np = 0;
                for (int c = 0; c < ((n0 + 63) >> 6) && np < 64; c++) {
                    uint64_t m = PG_BALLOT(l, ({
                                               const int idx = (c << 6) + l;
                                               bool ok = false;
                                               if (idx < n0) {
                                                   const uint32_t mm = meta(idx);
                                                   ok = (mm & MF_SMART_STEP) != 0 && !(mm & MF_WILL_ERASE) && Game::par_smart_type_ok(meta_type(mm));
                                               }
                                               ok;
                                           }));
                    while (m && np < 64) {
                        const int obj = (c << 6) + pg_ctz64(m);
                        m &= m - 1;
                        if (bso_scan_axes(obj) == 0) {
                            s->tmp[np] = (uint32_t)obj;
                            np++;
                        }
                    }
                }
                PG_SYNC_E();
                if (np >= 2) {
#if defined(PGAMD_WAVE_EMU)
                    pg_emu_counters()[0] += np;
#endif
                    bool done = false;
                    if constexpr (FREE_OBJECTS_OK) {
                        if (!G.grid_step) {  // eight objects at a time, their sub_steps side by side
                            for (int b = 0; b < np; b += 8) bso_free_objects(b, np - b < 8 ? np - b : 8);
                            PG_FOR_LANES(l) {
                                if (l < np) meta((int)s->tmp[l]) |= MF_PAR_DONE;
                            }
                            done = true;
                        }
                    }
                    if (!done) {  // one lane per object
                        PG_FOR_LANES(l) {
                            if (l < np) {
                                const int obj = (int)s->tmp[l];
                                bso_core<false>(obj, 0);
                                meta(obj) |= MF_PAR_DONE;
                            }
                        }
                    }
                    PG_SYNC_E();
                }
            }
            phase(9);
        }
        int hi = n0;  // entities [hi, n0) are done
        while (hi > 0) {
            // highest smart_step index below hi (that the parallel pass has not done)
            int sidx = -1;
            for (int c = (hi - 1) >> 6; c >= 0 && sidx < 0; c--) {
                uint64_t m = PG_BALLOT(l, ({
                                           const int idx = (c << 6) + l;
                                           idx < hi && (meta(idx) & (MF_SMART_STEP | MF_PAR_DONE)) == MF_SMART_STEP;
                                       }));
                if (m) sidx = (c << 6) + pg_highest(m);
            }
            const int lo = sidx + 1;  // [lo, hi) are non-smart
            for (int base = lo & ~63; base < hi; base += 64) {
                PG_FOR_LANES(l) {
                    const int idx = base + l;
                    if (idx >= lo && idx < hi) ent_step(idx);
                }
            }
            PG_SYNC_E();
            phase(11);
            if (sidx < 0) break;
            basic_step_object(sidx);
            ent_step(sidx);
            PG_SYNC_E();
            phase(12);
            hi = sidx;
        }
    }

    // Entity::step for every entity and nothing else: step_entities for a list whose smart_step entities' basic_step_object
    // is known to change nothing (see game_leaper.h game_reset: the agent waits at rest while the lanes fill up)
    PG_DEV void step_entities_all_plain() {
        const int n = G.n_ents;
        for (int base = 0; base < n; base += 64) {
            PG_FOR_LANES(l) {
                if (base + l < n) ent_step(base + l);
            }
        }
        PG_SYNC_E();
    }

    PG_DEV void check_grid_collisions(int ent) {  // BAG:145-165
        float ax = ex(ent), ay = ey(ent), arx = erx(ent), ary = ery(ent);
        int min_x = (int)(ax - (arx + POS_EPS));
        int max_x = (int)(ax + (arx + POS_EPS));
        int min_y = (int)(ay - (ary + POS_EPS));
        int max_y = (int)(ay + (ary + POS_EPS));
        for (int x = min_x; x <= max_x; x++)
            for (int y = min_y; y <= max_y; y++) {
                int grid_type = get_obj_from_floats((float)x, (float)y);
                if (grid_type != SPACE) Game::handle_grid_collision(*this, ent, grid_type, x, y);
            }
    }

    // collision pass BAG:719-741.  Entities that need any work (agent overlap, collides_with_entities,
    // smart_step) are found by ballot and visited from the highest index down; predicates are re-evaluated
    // at visit time, and the ballot is refreshed after every handler (handlers may change geometry).
    PG_DEV void collision_pass() {
        int limit = G.n_ents;
        while (limit > 0) {
            const int n = G.n_ents;
            int i = -1;
            for (int c = (limit - 1) >> 6; c >= 0 && i < 0; c--) {
                const int ag = G.agent;
                const float agx = ex(ag), agy = ey(ag), agrx = erx(ag), agry = ery(ag);
                uint64_t m = PG_BALLOT(l, ({
                                           const int idx = (c << 6) + l;
                                           bool w = false;
                                           if (idx < limit && idx < n) {
                                               const uint32_t mm = meta(idx);
                                               w = (mm & (MF_COLLIDES | MF_SMART_STEP)) != 0;
                                               if (!w && meta_type(mm) != PLAYER) {
                                                   const float cm = ef(EF_COLLISION_MARGIN, idx);
                                                   const float tx = (erx(idx) + agrx) + cm;
                                                   const float ty = (ery(idx) + agry) + cm;
                                                   w = (pg_fabsf(ex(idx) - agx) < tx) && (pg_fabsf(ey(idx) - agy) < ty);
                                               }
                                           }
                                           w;
                                       }));
                if (m) i = (c << 6) + pg_highest(m);
            }
            if (i < 0) break;
            if (has_agent_collision(i)) Game::handle_agent_collision(*this, i);
            if (eflag(i, MF_COLLIDES)) {
                if constexpr (Game::USES_ENTITY_COLLISIONS) {
                    // reverse j loop of BAG:727-735: overlaps found by ballot (handlers do not move entities), visited
                    // from the highest index down; will_erase is re-read at visit time (handlers set it); entities a
                    // handler appends are not part of this loop (its bound is the size at entry)
                    const int nj = G.n_ents;
                    const float cm = ef(EF_COLLISION_MARGIN, i);
                    const float ix = ex(i), iy = ey(i), irx = erx(i), iry = ery(i);
                    for (int cj = (nj - 1) >> 6; cj >= 0; cj--) {
                        uint64_t hits = PG_BALLOT(l, ({
                                                      const int j = (cj << 6) + l;
                                                      bool h = false;
                                                      if (j < nj && j != i) {
                                                          const float tx = (irx + erx(j)) + cm;
                                                          const float ty = (iry + ery(j)) + cm;
                                                          h = (pg_fabsf(ix - ex(j)) < tx) && (pg_fabsf(iy - ey(j)) < ty);
                                                      }
                                                      h;
                                                  }));
                        while (hits) {
                            const int j = (cj << 6) + pg_highest(hits);
                            hits &= ~(1ull << (j & 63));
                            if (!eflag(i, MF_WILL_ERASE) && !eflag(j, MF_WILL_ERASE)) {
                                Game::handle_collision(*this, i, j);
                                PG_SYNC_E();
                            }
                        }
                    }
                }
            }
            if (eflag(i, MF_SMART_STEP)) check_grid_collisions(i);
            PG_SYNC_E();
            limit = i;
        }
    }

    // erase_if_needed BAG:748-756: stable compaction of the SoA table, field by field through LDS scratch.
    PG_DEV void erase_if_needed() {
        const int n = G.n_ents;
        int kept = 0;
        int new_agent = G.agent;
        bool agent_erased = false;
        const int nchunks = (n + 63) >> 6;
        for (int c = 0; c < nchunks; c++) {
            const uint64_t valid = PG_BALLOT(l, ((c << 6) + l) < n);
            const uint64_t keep = PG_BALLOT(l, ({
                                                const int idx = (c << 6) + l;
                                                bool k = false;
                                                if (idx < n) {
                                                    const uint32_t mm = meta(idx);
                                                    k = !((mm & MF_WILL_ERASE) || ((mm & MF_AUTO_ERASE) && is_out_of_bounds(idx)));
                                                }
                                                k;
                                            }));
            const int base = c << 6;
            if (G.agent >= base && G.agent < base + 64) {
                const int al = G.agent - base;
                if (keep & (1ull << al)) new_agent = kept + pg_popc64(keep & pg_mask_lt(al));
                else agent_erased = true;
            }
            if (agent_erased && G.agent >= base && G.agent < base + 64) {
                // keep the detached agent readable in the reserved last slot (shared_ptr semantics, BAG:788-792)
                const int src_i = G.agent;
                PG_FOR_LANES(l) {
                    if (l < EF_COUNT) s->tmp[l] = s->ent[l * CAP + src_i];
                }
                PG_SYNC_E();
                PG_FOR_LANES(l) {
                    if (l < EF_COUNT) s->ent[l * CAP + (CAP - 1)] = s->tmp[l];
                }
                PG_SYNC_E();
                new_agent = CAP - 1;
            }
            if (keep != valid || kept != base) {
                // the chunk's 21 fields travel through registers in one round -- all reads, one fence, all writes (a slot only moves
                // down) -- instead of field by field through a staging row (42 fences; coinrun erases an expired trail in most steps)
                PG_LANE_ARR(uint32_t, v, EF_COUNT);
                PG_FOR_LANES(l) {
                    _Pragma("unroll") for (int f = 0; f < EF_COUNT; f++) PG_LA(v, f, l) = s->ent[f * CAP + base + l];
                }
                PG_SYNC_E();
                PG_FOR_LANES(l) {
                    if (keep & (1ull << l)) {
                        const int dst = kept + pg_popc64(keep & pg_mask_lt(l));
                        _Pragma("unroll") for (int f = 0; f < EF_COUNT; f++) s->ent[f * CAP + dst] = PG_LA(v, f, l);
                    }
                }
                PG_SYNC_E();
            }
            kept += pg_popc64(keep);
        }
        G.n_ents = kept;
        G.agent = new_agent;
    }

    // BasicAbstractGame::game_step BAG:686-746
    PG_DEV void bag_game_step() {
        G.step_rand_int = randint(0, 1000000);
        G.move_action = G.action % 9;
        G.special_action = 0;
        if (G.action >= 9) {
            G.special_action = G.action - 8;
            G.move_action = 4;
        }
        if (G.move_action != 4) G.last_move_action = G.move_action;
        G.action_vrot = 0;
        G.action_vx = 0;
        G.action_vy = 0;
        Game::set_action_xy(*this, G.move_action);
        const int ag = G.agent;
        if (G.grid_step) {
            evx(ag) = G.action_vx;
            evy(ag) = G.action_vy;
        } else {
            Game::update_agent_velocity(*this);
            float vrot = MIXRATEROT * ef(EF_VROT, ag);
            vrot += MIXRATEROT * MAXVTHETA * G.action_vrot;
            ef(EF_VROT, ag) = vrot;
        }
        PG_SYNC_E();
        phase(1);
        if (!PG_DBG(d, 64)) step_entities();
        phase(2);
        if (!PG_DBG(d, 128)) collision_pass();
        phase(3);
        if (!PG_DBG(d, 256)) erase_if_needed();
        phase(4);
        G.done = G.done || is_out_of_bounds(G.agent);
    }
    // default BAG::update_agent_velocity BAG:669-684 (games may override)
    PG_DEV void bag_update_agent_velocity(float v_scale) {
        const int ag = G.agent;
        float vx = (1 - G.mixrate) * evx(ag);
        float vy = (1 - G.mixrate) * evy(ag);
        vx += G.mixrate * G.maxspeed * G.action_vx * v_scale;
        vy += G.mixrate * G.maxspeed * G.action_vy * v_scale;
        evx(ag) = (float)(.9 * vx);
        evy(ag) = (float)(.9 * vy);
    }

    // BasicAbstractGame::game_reset BAG:758-797
    PG_DEV void bag_game_reset() {
        Game::choose_world_dim(*this);
        if (!(G.main_width > 0 && G.main_height > 0)) fail(PGE_ASSERT);
        G.bg_pct_x = rand01();
        G.background_index = randn(d.assets->n_bg);
        if (opt.use_generated_assets) {
            // BAG:769-773: AssetGen bggen(&rand_gen) paints this episode's background.  Here only its draws are made (the
            // generator without a painter); the background kernel re-seeds from the level seed, skips the draws made so far
            // and paints (pg_bgpaint.h)
            const int drawn = rg_twists > 0 ? (rg_twists - 1) * MT_N + G.rand_idx : 0;
            PG_FOR_LANES(l) {
                if (l == 0) {
                    d.bg_req[2 * env] = G.current_level_seed;
                    d.bg_req[2 * env + 1] = drawn;
                }
            }
            EnvRng er{this};
            assetgen::NoPainter np;
            assetgen::Gen<EnvRng, assetgen::NoPainter> gen{er, np};
            gen.generate_resource(500, 500, 1, 50, true);
        }
        G.n_ents = 0;
        float ax, ay;
        const float a_r = 0.4f;
        if (G.random_agent_start) {
            ax = rand01() * (G.main_width - 2 * a_r) + a_r;
            ay = rand01() * (G.main_height - 2 * a_r) + a_r;
        } else {
            ax = a_r;
            ay = a_r;
        }
        const int ag = add_entity(ax, ay, 0, 0, a_r, PLAYER);
        G.agent = ag;
        set_flag(ag, MF_SMART_STEP, true);
        set_render_z(ag, 1);
        PG_SYNC_E();
        erase_if_needed();
        fill_elem(0, 0, G.main_width, G.main_height, SPACE);
    }
    PG_DEV void choose_random_theme(int i) {  // BAG:1038-1041
        const int nt = d.assets->type_num_themes[meta_image_type(meta(i))];
        if (nt <= 0) {
            fail(PGE_THEME);
            return;
        }
        set_image_theme(i, randn(nt));
    }

    // asset_aspect_ratios[img_idx] comes from the image of the MASKED theme (initialize_asset_if_necessary BAG:82-86,114):
    // with restrict_themes every theme of a type has the aspect ratio of theme 0
    PG_DEV int aspect_theme(uint32_t mm) const {
        return (opt.restrict_themes && !Game::should_preserve_type_themes(meta_image_type(mm))) ? 0 : meta_image_theme(mm);
    }
    PG_DEV void match_aspect_ratio(int i) {  // BAG:1014-1023 (match_width), aspect ratio BAG:114
        const uint32_t mm = meta(i);
        const int img = (int)d.assets->type_theme_img[meta_image_type(mm)][aspect_theme(mm)];
        if (img < 0) {
            fail(PGE_THEME);
            return;
        }
        const ImgDesc im = d.assets->img[img];
        const float aspect = (float)((double)im.w * 1.0 / (double)im.h);
        ery(i) = erx(i) / aspect;
    }

    PG_DEV void fit_aspect_ratio(int i) {  // BAG:1025-1036
        const uint32_t mm = meta(i);
        const int img = (int)d.assets->type_theme_img[meta_image_type(mm)][aspect_theme(mm)];
        if (img < 0) {
            fail(PGE_THEME);
            return;
        }
        const ImgDesc im = d.assets->img[img];
        const float ar = (float)((double)im.w * 1.0 / (double)im.h);
        if (ar > 1) ery(i) = erx(i) / ar;
        else erx(i) = ery(i) * ar;
    }
    PG_DEV float rand_pos(float r, float min, float max) {  // BAG:1100-1108
        if (!(min <= max)) fail(PGE_ASSERT);
        if (max - min <= 2 * r) return (max + min) / 2;
        const float range = max - min;
        return (range - 2 * r) * rand01() + r + min;
    }
    // has_any_collision BAG:1114-1124 for the entity in table slot i (it need not be in the list yet)
    PG_DEV bool has_any_collision(int i, float margin) {
        const int n = G.n_ents;
        const float x = ex(i), y = ey(i), rx = erx(i), ry = ery(i);
        for (int c = 0; c < ((n + 63) >> 6); c++) {
            const uint64_t m = PG_BALLOT(l, ({
                                             const int idx = (c << 6) + l;
                                             bool hit = false;
                                             if (idx < n && !(meta(idx) & MF_AVOIDS)) {
                                                 const float tx = (rx + erx(idx)) + margin;
                                                 const float ty = (ry + ery(idx)) + margin;
                                                 hit = (pg_fabsf(x - ex(idx)) < tx) && (pg_fabsf(y - ey(idx)) < ty);
                                             }
                                             hit;
                                         }));
            if (m) return true;
        }
        return false;
    }
    PG_DEV void reposition(int i, float x, float y, float w, float h, bool check_collisions) {  // BAG:541-560
        const float rx = erx(i), ry = ery(i);
        ex(i) = rand_pos(rx, x, x + w);
        ey(i) = rand_pos(ry, y, y + h);
        PG_SYNC_E();
        int count = 0;
        while ((has_agent_collision(i) || (check_collisions && has_any_collision(i, 0))) && (count < 100)) {
            ex(i) = rand_pos(rx, x, x + w);
            ey(i) = rand_pos(ry, y, y + h);
            PG_SYNC_E();
            count++;
        }
    }
    // k-th grid cell (ascending index) whose value satisfies pred (-1 if there are fewer), and their count
    template <class Pred>
    PG_DEV int nth_cell(int k, Pred pred) {
        const int nc = G.main_width * G.main_height;
        for (int base = 0; base < nc; base += 64) {
            uint64_t m = PG_BALLOT(l, (base + l) < nc && pred((int)s->grid[base + l]));
            const int c = pg_popc64(m);
            if (k < c) {
                for (int q = 0; q < k; q++) m &= m - 1;
                return base + pg_ctz64(m);
            }
            k -= c;
        }
        return -1;
    }
    template <class Pred>
    PG_DEV int count_cells(Pred pred) {
        const int nc = G.main_width * G.main_height;
        int n = 0;
        for (int base = 0; base < nc; base += 64) n += pg_popc64(PG_BALLOT(l, (base + l) < nc && pred((int)s->grid[base + l])));
        return n;
    }
    // RandGen::simple_choose (reference src/randgen.cpp:71-88): k <= 128 distinct draws below n, kept in s->tmp[0..k)
    PG_DEV void simple_choose(int n, int k) {
        if (!(k <= n) || k > 128) {
            fail(PGE_ASSERT);
            return;
        }
        for (int i = 0; i < k; i++) {
            int next = randn(n);
            while ((PG_BALLOT(l, l < i && (int)s->tmp[l] == next) | PG_BALLOT(l, 64 + l < i && (int)s->tmp[64 + l] == next)) != 0) next = randn(n);
            s->tmp[i] = (uint32_t)next;
            PG_SYNC_E();
        }
    }

    PG_DEV bool agent_has_collision() {  // BAG:521-529
        const int n = G.n_ents;
        for (int c = 0; c < ((n + 63) >> 6); c++)
            if (PG_BALLOT(l, ((c << 6) + l) < n && has_agent_collision((c << 6) + l))) return true;
        return false;
    }
    PG_DEV void reposition_agent() {  // BAG:531-539
        const int ag = G.agent;
        int count = 0;
        do {
            ex(ag) = rand01() * (G.main_width - 2 * erx(ag)) + erx(ag);
            ey(ag) = rand01() * (G.main_height - 2 * ery(ag)) + ery(ag);
            PG_SYNC_E();
            count++;
        } while (agent_has_collision() && (count < 100));
    }
    // spawn_entity_rxy BAG:511-519: the entity is placed before it joins the list
    PG_DEV int spawn_entity_rxy(float rx, float ry, int type, float x, float y, float w, float h, bool check_collisions = true) {
        const int i = G.n_ents;
        if (i >= CAP - 1) {
            fail(PGE_ENT_OVERFLOW);
            return CAP - 2;
        }
        ent_init(i, 0, 0, 0, 0, rx, ry, type);
        PG_SYNC_E();
        reposition(i, x, y, w, h, check_collisions);
        G.n_ents = i + 1;
        return i;
    }
    PG_DEV int spawn_entity(float r, int type, float x, float y, float w, float h, bool check_collisions = true) {
        return spawn_entity_rxy(r, r, type, x, y, w, h, check_collisions);
    }
    PG_DEV void spawn_entities(int num, float r, int type, float x, float y, float w, float h) {  // BAG:585-589
        for (int k = 0; k < num; k++) spawn_entity(r, type, x, y, w, h);
    }

    PG_DEV void match_aspect_ratio_h(int i) {  // BAG:1014-1023 (match_width = false)
        const uint32_t mm = meta(i);
        const int img = (int)d.assets->type_theme_img[meta_image_type(mm)][aspect_theme(mm)];
        if (img < 0) {
            fail(PGE_THEME);
            return;
        }
        const ImgDesc im = d.assets->img[img];
        const float aspect = (float)((double)im.w * 1.0 / (double)im.h);
        erx(i) = ery(i) * aspect;
    }

    // Game::reset reference src/game.cpp:93-118
    PG_DEV void game_reset_full() {
        if (G.episodes_remaining == 0) {
            if (opt.use_sequential_levels && G.level_complete) {
                G.current_level_seed = (int32_t)((uint32_t)G.current_level_seed + 997u);
            } else {
                const uint32_t x = level_seed_u32();
                const uint32_t range = (uint32_t)(G.level_seed_high - G.level_seed_low);
                // range 0 (only a restored state can carry it, reference src/game.cpp:247-248): the reference's `x % range` ends the process
                if (range == 0) fail(PGE_ASSERT);
                G.current_level_seed = (int)((uint32_t)G.level_seed_low + (range ? x % range : 0u));
            }
            G.episodes_remaining = 1;
        } else {
            G.reward = 0;
            G.done = 0;
            G.level_complete = 0;
        }
        rand_seed(G.current_level_seed);
        Game::game_reset(*this);
        G.cur_time = 0;
        G.total_reward = 0;
        G.episodes_remaining -= 1;
        G.action = G.default_action;
    }

    // Game::step reference src/game.cpp:120-155 (observe() = render(), done by the caller)
    PG_DEV void game_step_full() {
        G.cur_time += 1;
        bool will_force_reset = false;
        if (G.action == -1) {
            G.action = G.default_action;
            will_force_reset = true;
        }
        G.reward = 0;
        G.done = 0;
        G.level_complete = 0;
        Game::game_step(*this);
        phase(5);
        G.done = G.done || will_force_reset || (G.cur_time >= G.timeout);
        G.total_reward += G.reward;
        if (G.reward != 0) {
            G.last_reward_timer = 10;
            G.last_reward = G.reward;
        }
        G.prev_level_seed = G.current_level_seed;
        // (what follows in Game::step -- the reset of a finished episode -- is finish_step's; a NO_RESET step kernel stops
        // here and a finished episode goes to the reset kernel, run(2))
        if constexpr (NO_RESET) needs_reset = G.done != 0;
    }
    // the rest of Game::step once game_step has run (reference src/game.cpp:144-155); initial = the reset + first
    // observation libenv_set_buffers asks for (reference src/vecgame.cpp:346-357).  One call site of the level
    // generator per kernel: the step kernels are sensitive to their code size (instruction cache).
    PG_DEV void finish_step(bool initial) {
        if constexpr (!NO_RESET) {
            if (initial || G.done) {
                game_reset_full();
                phase(6);
#if !defined(PGAMD_WAVE_EMU)
                if (PG_PHASES(d) && PG_LANE_ID() == 0) atomicAdd(d.phase_cycles + 15 + 32 * (env & 4095), 1ull);
#endif
            }
        }
        if (initial) {
            G.initial_reset_complete = 1;
        } else {
            if (opt.use_sequential_levels && G.level_complete) G.done = 0;
            G.episode_done = G.done;
        }
    }

    // ======================================================================================================
    // Camera scalars consumed by the render kernel (pg_render.h) and serialized by get_state: BAG:819-838
    PG_DEV void prepare_for_drawing(float rect_height) {  // BAG:819-838
        G.center_x = (float)(G.main_width * .5);
        G.center_y = (float)(G.main_height * .5);
        if (Game::center_agent(opt)) {
            Game::choose_center(*this, G.center_x, G.center_y);
        } else {
            G.visibility = (float)(G.main_width > G.main_height ? G.main_width : G.main_height);
            if (G.visibility < G.min_visibility) G.visibility = G.min_visibility;
        }
        const float raw_unit = 64 / G.visibility;
        G.unit = (float)((double)raw_unit * ((double)rect_height / 64.0));
        G.view_dim = (float)(64.0 / (double)raw_unit);
        G.x_off = G.unit * (G.center_x - G.view_dim / 2);
        G.y_off = G.unit * (G.center_y - G.view_dim / 2);
    }

    // Game::observe minus the frame (reference src/game.cpp:160-164)
    PG_DEV void store_outputs() {
        PG_FOR_LANES(l) {
            if (l == 0) {
                d.rew[env] = G.reward;
                d.first[env] = (uint8_t)G.done;
                d.prev_level_seed[env] = G.prev_level_seed;
                d.prev_level_complete[env] = (uint8_t)G.level_complete;
                d.level_seed[env] = G.current_level_seed;
            }
        }
    }

    // ======================================================================================================
    // HBM <-> LDS staging of one env
    // Requests first, uses after: the grid slab and the first SPEC entity slots are asked for before the header is read (their
    // addresses depend on the env alone; most envs hold fewer than 16 entities -- coinrun's median is 7 -- and a speculative slot
    // costs no extra sector: a field's 16 words are one 64-byte line), so a step starts with one round trip to memory, not three.
    static constexpr int SPEC = 16;
    PG_DEV void load_env(bool with_entities = true) {
        const uint32_t *ge = d.ents + ent_table_base(env, d.ent_cap);
        const uint32_t fstride = (uint32_t)d.ent_cap;  // words between two fields of one slot
        constexpr int NV = (int)((sizeof(cell_t) * Game::MAX_CELLS + 15) / 16);  // whole grid slab, 16 B per lane per access
        constexpr int NVL = (NV + 63) / 64;
        const pg_u4 *gg = reinterpret_cast<const pg_u4 *>(d.grid + (size_t)env * d.grid_bytes);
        PG_LANE_ARR(uint32_t, ev, EF_COUNT);
        PG_LANE_ARR(uint32_t, gvx, NVL);  // (four scalar arrays: an array of 16-byte structs stays in scratch memory)
        PG_LANE_ARR(uint32_t, gvy, NVL);
        PG_LANE_ARR(uint32_t, gvz, NVL);
        PG_LANE_ARR(uint32_t, gvw, NVL);
        PG_FOR_LANES(l) {
            _Pragma("unroll") for (int f = 0; f < EF_COUNT; f++) PG_LA(ev, f, l) = 0;
            if (with_entities && l < SPEC) {
                const uint32_t *gp = ge + (uint32_t)l;
                _Pragma("unroll") for (int f = 0; f < EF_COUNT; f++) PG_LA(ev, f, l) = gp[f * fstride];
            }
            _Pragma("unroll") for (int k = 0; k < NVL; k++) {
                const int i = k * 64 + l;
                const pg_u4 q = gg[i < NV ? i : 0];
                PG_LA(gvx, k, l) = q.x;
                PG_LA(gvy, k, l) = q.y;
                PG_LA(gvz, k, l) = q.z;
                PG_LA(gvw, k, l) = q.w;
            }
        }
        {
            const EnvHdr *h = d.hdr + env;  // wave-uniform address
#define PG_X(type, name) G.name = h->name;
            PG_HDR_FIELDS(PG_X)
#undef PG_X
        }
        opt = env_options(d.opt, G.opt_bits, G.opt_debug_mode);
        const int n = with_entities ? G.n_ents : 0;
        PG_FOR_LANES(l) {
            if (l < SPEC && l < n) {
                _Pragma("unroll") for (int f = 0; f < EF_COUNT; f++) s->ent[f * CAP + l] = PG_LA(ev, f, l);
            }
        }
        for (int base = 0; base < n; base += 64) {
            PG_FOR_LANES(l) {
                if (base + l < n && base + l >= SPEC) {
                    uint32_t v[EF_COUNT];  // all field loads in flight before the first LDS store (the loops must stay unrolled:
                                           // rolled, every load waits for the previous one)
                    const uint32_t *gp = ge + (uint32_t)(base + l);
                    _Pragma("unroll") for (int f = 0; f < EF_COUNT; f++) v[f] = gp[f * fstride];
                    _Pragma("unroll") for (int f = 0; f < EF_COUNT; f++) s->ent[f * CAP + base + l] = v[f];
                }
            }
        }
        {
            pg_u4 *lg = reinterpret_cast<pg_u4 *>(s->grid);
            PG_FOR_LANES(l) {
                _Pragma("unroll") for (int k = 0; k < NVL; k++) {
                    const int i = k * 64 + l;
                    if (i < NV) lg[i] = pg_u4{PG_LA(gvx, k, l), PG_LA(gvy, k, l), PG_LA(gvz, k, l), PG_LA(gvw, k, l)};
                }
            }
        }
        G.grid_dirty = 0;
        PG_SYNC_E();
    }
    PG_DEV void store_env() {
        const int n = G.n_ents;
        uint32_t *ge = d.ents + ent_table_base(env, d.ent_cap);
        if (n > d.ent_cap - 1) fail(PGE_ENT_OVERFLOW);
        const uint32_t fstride = (uint32_t)d.ent_cap;
        for (int base = 0; base < n; base += 64) {
            PG_FOR_LANES(l) {
                if (base + l < n && base + l < d.ent_cap) {
                    uint32_t *gp = ge + (uint32_t)(base + l);
                    _Pragma("unroll") for (int f = 0; f < EF_COUNT; f++) gp[f * fstride] = s->ent[f * CAP + base + l];
                }
            }
        }
        if (G.agent < 0 || G.agent >= n) fail(PGE_ASSERT);  // a detached agent never outlives the step (reset follows)
        if (G.grid_dirty) {
            constexpr int NV = (int)((sizeof(cell_t) * Game::MAX_CELLS + 15) / 16);
            pg_u4 *gg = reinterpret_cast<pg_u4 *>(d.grid + (size_t)env * d.grid_bytes);
            const pg_u4 *lg = reinterpret_cast<const pg_u4 *>(s->grid);
            for (int base = 0; base < NV; base += 64) {
                PG_FOR_LANES(l) {
                    if (base + l < NV) gg[base + l] = lg[base + l];
                }
            }
        }
        decide_route();
        publish_routing();
        {
            EnvHdr *h = d.hdr + env;
            PG_FOR_LANES(l) {
                if (l == 0) {
#define PG_X(type, name) h->name = G.name;
                    PG_HDR_FIELDS(PG_X)
#undef PG_X
                }
            }
        }
    }

    // which step kernel owns this env next step (EnvHdr::big -> route table): the smallest LDS arena its table fits
    PG_DEV void decide_route() {
        const int need = Game::slots_needed_next_step(*this);  // entity slots incl. growth of one step + the reserved one
        if (need > Game::ENT_CAP_T2) fail(PGE_ENT_OVERFLOW);
        G.big = need <= Game::ENT_CAP_T0 ? 0 : (need <= Game::ENT_CAP_T1 ? 1 : 2);
    }

    // hand this env (its episode just ended, its header is stored) to the reset kernel of its chunk
    PG_DEV void queue_reset() {
#if !defined(PGAMD_WAVE_EMU)
        if (PG_LANE_ID() == 0) {
            if (G.error) pg_report_error(d, env, G.error, CAP, G.n_ents, G.agent);
            const int c = d.reset_first > 0 ? (env >= d.reset_first ? 1 : 0) : env / d.reset_chunk_envs;
            const size_t base = d.reset_first > 0 ? (c ? (size_t)d.reset_first : 0) : (size_t)c * d.reset_chunk_envs;
            d.reset_list[base + atomicAdd(d.reset_count + c, 1)] = env;
        }
#endif
    }

    // tell the next step which kernel owns this env, and surface error codes to the host
    PG_DEV void publish_routing() {
#if defined(PGAMD_WAVE_EMU)
        if (G.error) pg_report_error(d, env, G.error, CAP, G.n_ents, G.agent);
#else
        if (PG_LANE_ID() == 0) {
            if (d.next_route) d.next_route[env] = (uint8_t)G.big;
            if (G.big == 1 || G.big == 2) {
                const int t = G.big, c = env / d.chunk_envs;
                const int slot = atomicAdd(d.next_big_count + c * NUM_TIERS + t, 1);
                d.next_big_list[(size_t)t * d.num_envs + (size_t)c * d.chunk_envs + slot] = env;
            }
            if (G.error) pg_report_error(d, env, G.error, CAP, G.n_ents, G.agent);
        }
#endif
    }

    // one libenv step (mode 1), the initial reset + first observation (mode 0), or the reset that finishes a step a NO_RESET
    // step kernel took up to the end of the episode (mode 2) of this env
    PG_DEV void run(int mode) {
#if !defined(PGAMD_WAVE_EMU)
        if (PG_PHASES(d)) t_mark = (long long)__builtin_readcyclecounter();
        if (PG_PHASES(d) && PG_TRACE(d) && PG_LANE_ID() < 24) d.wave_trace[(size_t)env * 32 + 8 + PG_LANE_ID()] = 0;
#endif
        load_env(mode != 2);  // a reset starts from an empty entity table (whose old size may exceed this arena)
        phase(0);
        if (mode == 1) G.action = d.action[env];  // reference src/vecgame.cpp:388
        if PG_DBG(d, 512) {
            // ablation: staging only
        } else {
            if (mode == 1) game_step_full();
            if constexpr (NO_RESET) {
                if (needs_reset) {  // the reset kernel behind this one takes over (run(2)): it needs the header only
                    G.big = ROUTE_RESET;
                    EnvHdr *h = d.hdr + env;
                    PG_FOR_LANES(l) {
                        if (l == 0) {
#define PG_X(type, name) h->name = G.name;
                            PG_HDR_FIELDS(PG_X)
#undef PG_X
                        }
                    }
                    queue_reset();
                    return;
                }
            }
            finish_step(mode == 0);
        }
        rand_flush();
        prepare_for_drawing((float)RES_H);  // draw_background + draw_foreground both call it (BAG:922,982)
        store_outputs();
        phase(7);
        store_env();
        phase(8);
#if !defined(PGAMD_WAVE_EMU)
        if (PG_PHASES(d) && mode != 0 && PG_LANE_ID() == 0) atomicAdd(d.phase_cycles + 14 + 32 * (env & 4095), 1ull);
#endif
    }
};

}  // namespace pgamd
