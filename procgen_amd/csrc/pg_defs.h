// pg_defs.h -- constants and HBM-resident data layout shared by the host driver and the kernels.
//
// References: object ids reference src/object-ids.h; resolution reference src/game.h:23-26;
// BasicAbstractGame constants reference src/basic-abstract-game.cpp:6-20 ("BAG").
#pragma once
#include <stddef.h>
#include <stdint.h>

// Profiling / ablation apparatus of the kernels (PROCGEN_AMD_DEBUG: phase cycle counters `& 2048`, the residency trace `& 8192`, the
// ablation bits of DevCtx::debug_flags): the kernels read it through these three macros, and -DPG_RELEASE turns them into constants, so a
// release build carries none of it (profiles/r06_release_ab.txt: the same-box A/B that decides which build __graft_entry__.build() ships).
#if defined(PG_RELEASE)
#define PG_RELEASE_STEP
#define PG_RELEASE_FRAME
#endif
// (the step kernels and the frame kernels take it separately -- PG_DBG / PG_PHASES / PG_TRACE are pg_env.h's, PG_FDBG / PG_FPHASES are
// pg_render.h's and pg_prep.h's -- because the A/B goes per kernel family: -DPG_RELEASE_STEP, -DPG_RELEASE_FRAME; -DPG_RELEASE = both)
#if defined(PG_RELEASE_STEP)
#define PG_DBG(d, bits) (false)
#define PG_PHASES(d) (false)
#define PG_TRACE(d) (false)
#else
#define PG_DBG(d, bits) (((d).debug_flags & (bits)) != 0)
#define PG_PHASES(d) ((d).phase_cycles != nullptr)
#define PG_TRACE(d) ((d).wave_trace != nullptr)
#endif
#if defined(PG_RELEASE_FRAME)
#define PG_FDBG(d, bits) (false)
#define PG_FPHASES(d) (false)
#else
#define PG_FDBG(d, bits) (((d).debug_flags & (bits)) != 0)
#define PG_FPHASES(d) ((d).phase_cycles != nullptr)
#endif

namespace pgamd {

constexpr int RES_W = 64;
constexpr int RES_H = 64;
constexpr int OBS_BYTES = RES_W * RES_H * 3;

constexpr int INVALID_OBJ = -1;
constexpr int PLAYER = 0;
constexpr int SPACE = 100;
constexpr int WALL_OBJ = 51;
constexpr int EXPLOSION = 54;
constexpr int EXPLOSION5 = 58;
constexpr int TRAIL = 59;
constexpr int USE_ASSET_THRESHOLD = 100;
constexpr int MAX_ASSETS = 100;
constexpr int MAX_IMAGE_THEMES = 10;

enum GameId : int {
    GAME_BIGFISH = 0, GAME_BOSSFIGHT, GAME_CAVEFLYER, GAME_CHASER, GAME_CLIMBER, GAME_COINRUN, GAME_DODGEBALL,
    GAME_FRUITBOT, GAME_HEIST, GAME_JUMPER, GAME_LEAPER, GAME_MAZE, GAME_MINER, GAME_NINJA, GAME_PLUNDER,
    GAME_STARPILOT, NUM_GAMES
};

// Kernel variants: the policy a handle is compiled against.  Normally the game id; caveflyer's memory mode (60x60 world)
// is its own instantiation with a larger LDS arena, so the default modes keep theirs small.
constexpr int KERNEL_CAVEFLYER_MEMORY = NUM_GAMES;
inline int kernel_id_for(int game_id, int distribution_mode) { return (game_id == GAME_CAVEFLYER && distribution_mode == 10) ? KERNEL_CAVEFLYER_MEMORY : game_id; }

enum DistributionMode : int { EasyMode = 0, HardMode = 1, ExtremeMode = 2, MemoryMode = 10 };
// which games accept the two optional modes (the fasserts of the reference's game constructors / game_reset: chaser.cpp, dodgeball.cpp,
// leaper.cpp, starpilot.cpp for extreme; caveflyer, dodgeball, heist, jumper, maze, miner for memory)
inline bool game_has_extreme_mode(int game_id) { return game_id == GAME_CHASER || game_id == GAME_DODGEBALL || game_id == GAME_LEAPER || game_id == GAME_STARPILOT; }
inline bool game_has_memory_mode(int game_id) {
    return game_id == GAME_CAVEFLYER || game_id == GAME_DODGEBALL || game_id == GAME_HEIST || game_id == GAME_JUMPER || game_id == GAME_MAZE || game_id == GAME_MINER;
}

#if defined(__HIPCC__)
#define PG_HOSTDEV_EARLY __host__ __device__
#else
#define PG_HOSTDEV_EARLY
#endif
// ---- options common to every env of a handle (reference src/game.h:45-60, src/vecgame.cpp:183-190) ----
struct GameOptions {
    int paint_vel_info, use_generated_assets, use_monochrome_assets, restrict_themes, use_backgrounds, center_agent;
    int debug_mode, distribution_mode, use_sequential_levels;
    int level_seed_low, level_seed_high;
};

// Options a state carries and the reference adopts per env on deserialize (reference src/game.cpp:233-246).  They live in every env's
// header (EnvHdr::opt_bits / opt_debug_mode) and override the handle's GameOptions when a kernel binds an env.  Round 5: the
// distribution_mode too (bits 8..15 hold mode + 1; 0 = the handle's): every mode a kernel instantiation serves is a run-time value to it,
// so a state saved under another mode of the same instantiation is adopted like the other options.  Only what selects the instantiation
// itself stays per handle: use_generated_assets, and caveflyer's memory mode (its own kernels and arenas, kernel_id_for).
enum EnvOptBit : int { EOB_PAINT_VEL_INFO = 1, EOB_MONOCHROME = 2, EOB_RESTRICT_THEMES = 4, EOB_BACKGROUNDS = 8, EOB_CENTER_AGENT = 16, EOB_SEQUENTIAL = 32 };
constexpr int EOB_MODE_SHIFT = 8;
PG_HOSTDEV_EARLY inline int env_option_bits(const GameOptions &o) {
    return (o.paint_vel_info ? EOB_PAINT_VEL_INFO : 0) | (o.use_monochrome_assets ? EOB_MONOCHROME : 0) | (o.restrict_themes ? EOB_RESTRICT_THEMES : 0) |
           (o.use_backgrounds ? EOB_BACKGROUNDS : 0) | (o.center_agent ? EOB_CENTER_AGENT : 0) | (o.use_sequential_levels ? EOB_SEQUENTIAL : 0) |
           (((o.distribution_mode + 1) & 0xff) << EOB_MODE_SHIFT);
}
PG_HOSTDEV_EARLY inline GameOptions env_options(const GameOptions &handle, int bits, int debug_mode) {
    GameOptions o = handle;
    o.paint_vel_info = (bits & EOB_PAINT_VEL_INFO) != 0;
    o.use_monochrome_assets = (bits & EOB_MONOCHROME) != 0;
    o.restrict_themes = (bits & EOB_RESTRICT_THEMES) != 0;
    o.use_backgrounds = (bits & EOB_BACKGROUNDS) != 0;
    o.center_agent = (bits & EOB_CENTER_AGENT) != 0;
    o.use_sequential_levels = (bits & EOB_SEQUENTIAL) != 0;
    o.debug_mode = debug_mode;
    const int m = (bits >> EOB_MODE_SHIFT) & 0xff;
    if (m) o.distribution_mode = m - 1;
    return o;
}

// ---- per-env scalar state: one record per env in HBM (Game + BasicAbstractGame + game scalars) ----
// Field names follow the reference members (reference src/game.h:62-111, src/basic-abstract-game.h:110-160).
// Every field is one 32-bit word; the list macro lets the kernels move the record HBM <-> registers field by
// field (the record stays in registers while an env steps).
#define PG_HDR_FIELDS(X)                                                                                          \
    /* Game */                                                                                                    \
    X(float, reward) X(int, done) X(int, level_complete) X(int, action) X(int, timeout)                           \
    X(int, current_level_seed) X(int, prev_level_seed) X(int, episodes_remaining) X(int, episode_done)            \
    X(int, last_reward_timer) X(float, last_reward) X(int, default_action) X(int, cur_time) X(int, grid_step)     \
    X(float, total_reward)                                                                                        \
    X(int, rand_idx)     /* position in rand_gen's 624-word state (std::mt19937 _M_p) */                          \
    X(int, lvl_rand_idx) /* same for level_seed_rand_gen */                                                       \
    X(int, initial_reset_complete)                                                                                \
    /* BasicAbstractGame */                                                                                       \
    X(int, n_ents) X(int, agent) /* index of the agent in the entity list */                                      \
    X(int, background_index) X(float, bg_tile_ratio) X(float, bg_pct_x)                                           \
    X(int, last_move_action) X(int, move_action) X(int, special_action)                                           \
    X(float, mixrate) X(float, maxspeed) X(float, max_jump)                                                       \
    X(float, action_vx) X(float, action_vy) X(float, action_vrot) X(float, center_x) X(float, center_y)           \
    X(int, random_agent_start) X(int, has_useful_vel_info) X(int, step_rand_int)                                  \
    X(int, main_width) X(int, main_height) X(int, out_of_bounds_object)                                           \
    X(float, unit) X(float, view_dim) X(float, x_off) X(float, y_off) X(float, visibility) X(float, min_visibility) \
    /* bookkeeping of this implementation */                                                                      \
    X(int, error)      /* first failed reference fassert / capacity overflow (0 = none) -> host fatal() */        \
    X(int, big)        /* arena tier (0,1,2) that must step this env next: smallest LDS entity table that fits */      \
    X(int, grid_dirty)                                                                                            \
    X(int, level_seed_low) X(int, level_seed_high) /* per env: set_state adopts the range a state was saved under (reference src/game.cpp:247-248) */ \
    X(int, opt_bits) X(int, opt_debug_mode) /* per env: the game options that select no kernel (env_option_bits); set_state adopts them (reference src/game.cpp:233-246) */ \
    /* game-specific scalars (meaning defined by the game policy, e.g. game_coinrun.h) */                         \
    X(int, gsi0) X(int, gsi1) X(int, gsi2) X(int, gsi3) X(int, gsi4) X(int, gsi5) X(int, gsi6) X(int, gsi7)       \
    X(float, gsf0) X(float, gsf1) X(float, gsf2) X(float, gsf3) X(float, gsf4) X(float, gsf5) X(float, gsf6) X(float, gsf7)

struct EnvHdr {
#define PG_X(type, name) type name;
    PG_HDR_FIELDS(PG_X)
#undef PG_X
};
constexpr int ENV_HDR_WORDS = (int)(sizeof(EnvHdr) / 4);

constexpr int MT_N = 624;
constexpr int MT_STRIDE = 640;  // words per generator state in HBM (624 + pad, 128-B multiple)
constexpr int MT_SLOTS = 4;     // per env: rand_gen, level_seed_rand_gen, two scratch states (seed / twist during a reset)
constexpr int NUM_TIERS = 3;    // LDS entity-arena sizes of the step kernel (Game::ENT_CAP_T0 < T1 < T2)

// ---- entity table: SoA [field][slot], one table per env in HBM, staged in LDS while an env steps ----
// The 31 members of the reference Entity (reference src/entity.h:9-48) with the 4 small ints and 7 bools
// packed into EF_META.
enum EntField : int {
    EF_X = 0, EF_Y, EF_VX, EF_VY, EF_RX, EF_RY,
    EF_META,  // type:10 | image_type:8 | image_theme:4 | (render_z+1):2 | flags:7  (see pg_ents.h)
    EF_FIRE_TIME, EF_SPAWN_TIME, EF_LIFE_TIME, EF_EXPIRE_TIME,
    EF_COLLISION_MARGIN, EF_ROTATION, EF_VROT, EF_FRICTION,
    EF_ALPHA, EF_HEALTH, EF_THETA, EF_GROW_RATE, EF_ALPHA_DECAY, EF_CLIMBER_SPAWN_X,
    EF_COUNT
};

// One table per env in HBM, [field][slot] contiguous (lanes = slots when a kernel stages it).
constexpr int TILE_ENVS = 64;  // env chunks are cut at multiples of this
#if defined(__HIPCC__)
#define PG_HOSTDEV __host__ __device__
#else
#define PG_HOSTDEV
#endif
PG_HOSTDEV inline size_t ent_table_base(int env, int ent_cap) { return (size_t)env * EF_COUNT * (size_t)ent_cap; }  // word index of (field 0, slot 0)
inline size_t ent_table_words(int num_envs, int ent_cap) { return (size_t)num_envs * EF_COUNT * (size_t)ent_cap; }

// values of the route table (which step kernel owns an env this step): 0..2 = the kernel with LDS arena tier 0..2
constexpr int MAX_CHUNKS = 8;
constexpr int LIST_COUNTERS = MAX_CHUNKS * 3;  // [chunk][tier] (NUM_TIERS == 3)  // env chunks of one step (step of chunk c+1 overlaps the render of chunk c)
constexpr int ERROR_INFO_WORDS = 8;
constexpr int ERROR_INFO_OFFSET = MAX_CHUNKS * 3 + 1;  // words from DevCtx::error to the record: the second list-counter block lies between them (libenv_hip.cpp d_small)
constexpr int ERR_KIND_RENDER = 100000, ERR_KIND_HUMAN = 200000, ERR_KIND_BGPAINT = 300000;  // DevCtx::error_info[2] (a step / reset kernel reports its entity arena size)
constexpr int ROUTE_RESET = 4;  // EnvHdr::big only: a NO_RESET step kernel ended the episode, the reset kernel of the same step takes over

constexpr int GEN_BG_DIM = 500;  // use_generated_assets: the per-env background canvas (reference BAG:62)
constexpr int GEN_BG_WORDS = GEN_BG_DIM * GEN_BG_DIM;

// ---- sprite atlas in HBM ----
// the render_human info frame (reference src/game.h:26 RENDER_RES; pg_human.h)
constexpr int HUMAN_RES = 512;
constexpr size_t HUMAN_BYTES = (size_t)HUMAN_RES * HUMAN_RES * 3;

struct ImgDesc {
    uint32_t off;  // first pixel (0xAARRGGBB words) in the atlas blob
    uint16_t w, h;
    uint32_t opaque;  // 1: every pixel has alpha 255 (SourceOver degenerates to a copy)
};
constexpr int MAX_GAME_IMAGES = 256;
constexpr int MAX_BACKGROUNDS = 128;
struct GameAssetsDev {
    ImgDesc img[MAX_GAME_IMAGES];
    int16_t type_theme_img[MAX_ASSETS][MAX_IMAGE_THEMES];  // -1 = none
    uint8_t type_num_themes[MAX_ASSETS];
    int32_t n_bg;
    int16_t bg_img[MAX_BACKGROUNDS];
    int32_t ref_w, ref_h;  // most common sprite size of the game (renderer's separable tile geometry)
    // the same tables with the indirection resolved on the host (finish_asset_tables): one dependent load instead of two when a
    // kernel turns (type, theme) or a background index into an image.  off == IMG_NONE: no such image
    ImgDesc type_theme_desc[MAX_ASSETS][MAX_IMAGE_THEMES];
    ImgDesc bg_desc[MAX_BACKGROUNDS];
};
constexpr uint32_t IMG_NONE = 0xffffffffu;
inline void finish_asset_tables(GameAssetsDev &t) {
    for (int ty = 0; ty < MAX_ASSETS; ty++)
        for (int th = 0; th < MAX_IMAGE_THEMES; th++) {
            const int i = t.type_theme_img[ty][th];
            ImgDesc none{IMG_NONE, 0, 0, 0};
            t.type_theme_desc[ty][th] = i >= 0 ? t.img[i] : none;
        }
    for (int b = 0; b < MAX_BACKGROUNDS; b++) {
        ImgDesc none{IMG_NONE, 0, 0, 0};
        t.bg_desc[b] = b < t.n_bg ? t.img[t.bg_img[b]] : none;
    }
}

// ---- everything a kernel launch needs ----
struct DevCtx {
    int num_envs;
    GameOptions opt;
    // per-env state
    EnvHdr *hdr;          // [num_envs]
    uint32_t *rng;        // [num_envs][MT_SLOTS][MT_STRIDE]  (0: rand_gen, 1: level_seed_rand_gen, 2-3: scratch)
    uint32_t *ents;       // [num_envs][EF_COUNT][ent_cap]
    int ent_cap;          // slots per env in HBM
    uint8_t *grid;        // [num_envs][grid_bytes]
    int grid_bytes;
    // boundary buffers (device side)
    const int32_t *action;  // [num_envs]
    uint8_t *obs;           // [num_envs][64][64][3]
    uint8_t *human;         // [num_envs][512][512][3] the render_human info frame (pg_human.h); null unless the handle was made with render_human
    float *rew;             // [num_envs]
    int32_t *prev_level_seed, *level_seed;  // [num_envs]
    uint8_t *first, *prev_level_complete;   // [num_envs]
    // assets
    const GameAssetsDev *assets;
    const uint32_t *pixels;
    const uint32_t *game_tables;  // Game::host_tables' words (null when the game has none)
    // use_generated_assets (null otherwise): per-env 500 x 500 background canvases and the requests the reset path leaves
    // for the background kernel (pg_bgpaint.h): {level seed, rand_gen draws made since the reseed}, skip < 0 = nothing to paint
    uint32_t *gen_bg;  // [num_envs][GEN_BG_WORDS]
    int *bg_req;       // [num_envs][2]
    // routing between the arena tiers of the step kernel: envs whose entity table may outgrow tier 0's LDS arena
    // are listed for the tier-1 / tier-2 kernels of the NEXT step (double-buffered by step parity)
    // List (tier, list chunk c) starts at big_list[tier * num_envs + c * chunk_envs], its length is
    // big_count[c * NUM_TIERS + tier]; the envs [c * chunk_envs, (c + 1) * chunk_envs) form list chunk c.  (One list chunk
    // today: per-chunk lists on per-chunk streams measured slower, each short kernel ends with its slowest env.)  Tier-0
    // envs are not listed: a grid over the env range steps them and skips the envs routed elsewhere.
    const int *big_list;   // [NUM_TIERS][num_envs] env ids the list kernels handle this step
    const int *big_count;  // [MAX_CHUNKS][NUM_TIERS]
    int *next_big_list;    // filled during this step
    int *next_big_count;   // zero when the step starts: the render kernel of the step before cleared them (they were its big_count)
    int chunk_envs;        // envs per chunk (a multiple of TILE_ENVS)
    // tier that owns each env THIS step (written during the previous step, so it is stable while the step kernels of
    // the three tiers run concurrently: an env re-routed by a fast kernel is not picked up again by a slower one)
    const uint8_t *route;  // [num_envs]
    uint8_t *next_route;   // [num_envs] written by every env's store_env
    // envs a NO_RESET step kernel (SPLIT_RESET games) stepped into `done`: the reset kernel of the same step generates their next level
    int reset_chunk_envs;   // envs per reset chunk: env e's reset goes to list chunk e / reset_chunk_envs ...
    int reset_first;        // ... unless > 0: two uneven chunks, [0, reset_first) and the rest (list chunk 1 starts at reset_first)
    int *reset_list;        // [num_envs]: the envs of chunk c are appended from index c * reset_chunk_envs on
    int *reset_count;       // [MAX_CHUNKS] per env chunk, this step
    int *next_reset_count;  // [MAX_CHUNKS] the next step's counters (double-buffered by step parity), zeroed by this step's lane kernel
    int *error;            // [1] OR of the per-env error codes raised so far (0 = none; sticky: the host stops at the first one)
    // error + ERROR_INFO_OFFSET, [ERROR_INFO_WORDS] (device builds; no pointer of its own: a kernel argument is live for a whole kernel, and the
    // render kernels have no scalar register to spare): who raised the first one, claimed with a compare-and-swap on word 0: env + 1,
    // code | source line << 8, kernel kind (a step kernel's arena size; ERR_KIND_*), n_ents, agent; words 6 / 7: the device address of the
    // handle's host-mapped copy of the record (pg_env.h pg_report_error).  The host prints it with the env's header when it ends the run
    // (libenv_hip.cpp VecGame::report_device_error).
    // launch order of the render kernel (experiment, PROCGEN_AMD_RENDER_ORDER; null = identity): workgroup j of a chunk's launch draws env
    // render_order[env_base + j], a permutation of that chunk's env range sorted by background image
    const int *render_order;  // [num_envs]
    // display-list games (pg_prep.h; null otherwise): the frame records prep<Game> writes and raster<Game> draws, and the envs of a launch
    // chunk whose frame the rasterizer cannot draw (render_list<Game>: the full renderer): chunk c's entries start at slow_list[its first
    // env], their count is slow_count[step_parity * MAX_CHUNKS + c]; raster<Game> zeroes the other parity's counter for the next step
    uint32_t *frame_rec;  // [num_envs][FrameRec<Game>::WORDS]
    int *slow_list;       // [num_envs]
    int *slow_count;      // [2][MAX_CHUNKS], in the block of small outputs the host downloads every step
    int *slow_flag;       // host-mapped word: set by a prep wave that queued a frame; the host launches render_list only then (libenv_observe)
    int step_parity;
    int clear_lists;       // render kernel: zero big_count[] (nobody reads it any more this step; it is the next step's next_big_count)
    unsigned long long *wave_trace;    // [num_envs][32] PROCGEN_AMD_DEBUG & 8192: 100 MHz timestamps of the last step's workgroups: step start / end / kind+HW_ID, render start / end / HW_ID (null otherwise)
    unsigned long long *phase_cycles;  // [4096][32] PROCGEN_AMD_DEBUG & 2048: per-phase wave cycles of the step (0-15) and render (16-31) kernels (null otherwise)
    int debug_flags;       // PROCGEN_AMD_DEBUG: phase ablation bits for profiling only (0 in normal operation)
};

}  // namespace pgamd
