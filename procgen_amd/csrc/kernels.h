// kernels.h -- host-callable entry points of kernels.hip
#pragma once
#include <hip/hip_runtime_api.h>

#include "host_state.h"
#include "pg_defs.h"

namespace pgamd {
struct LaunchStreams {
    hipStream_t main;
    hipEvent_t fork;
    hipStream_t lane[2];      // env chunks alternate between these two streams (step of chunk c+1 overlaps render of chunk c)
    hipStream_t side[3];      // [0]: launch-order experiments (PROCGEN_AMD_ORDER): the tier-2 list kernel off the chunk streams
    hipEvent_t lane_done[2];
    hipEvent_t side_done[3];
    hipEvent_t step_done[MAX_CHUNKS];
    hipEvent_t outputs_done[MAX_CHUNKS];  // recorded on a chunk's stream when its envs' small outputs are final (null: not wanted)
    hipEvent_t frames_done[MAX_CHUNKS];   // null: not wanted.  Recorded on chunk c's stream behind its render kernel: the chunk's observations are final (their D2H landing may start)
    hipEvent_t render_t0[MAX_CHUNKS], render_t1[MAX_CHUNKS];  // null: off.  Recorded around chunk c's render launch on its stream (procgen_amd_kernel_timing: the dominant kernel's own duration)
    int first_pct;            // experiment: share of the first of two chunks in percent (0 = even)
    int order;                // launch-order variant (PROCGEN_AMD_ORDER, see launch_game)
    int chunks;               // 1 = everything on `main`
    int list_count[MAX_CHUNKS][NUM_TIERS];  // entries of the lists this step reads (the host knows them from the previous step's download): an empty list's kernel is not launched
};
int first_chunk_envs(int num_envs, int first_pct);  // envs of the first of two uneven chunks (whole tiles; 0 = even cut)
int chunk_envs_for(int num_envs, int chunks);  // envs per chunk: whole tiles; one chunk below 4096 envs
// what one per-game kernel object (kernels_game.hip) exports
struct GameEntry {
    int game_id;
    hipError_t (*launch)(const DevCtx &, int mode, const LaunchStreams &);
    hipError_t (*render_one)(const DevCtx &, int env, int count, hipStream_t);  // redraws envs [env, env + count) with the full renderer
    int cap_t0, cap_t1, cap_t2;  // entity slots of the three LDS arenas; cap_t2 is the HBM table size
    int grid_bytes;
    void (*init_state)(int num_envs, int rand_seed, int env_offset, int env_stride, EnvHdr *hdr, uint32_t *rng);
    int (*host_tables)(const GameOptions &opt, uint32_t *out, int max_words);  // GameHostTables<Game>::build (pg_env.h)
    bool (*use_block_asset)(int type);  // GameBlockAsset<Game>::is (pg_env.h)
    hipError_t (*render_human)(const DevCtx &, int env_base, int count, hipStream_t);  // the 512 x 512 info frames of envs [env_base, env_base + count) (pg_human.h)
    bool split_reset;  // GameSplit<Game>::value: ended episodes are finished by reset_list kernels behind the step kernels
    int frame_rec_words;  // display-list games (pg_prep.h): words of an env's frame record, FrameRec<Game>::WORDS; 0: the game renders with one kernel
    hipError_t (*render_slow)(const DevCtx &, int env_base, int count, int chunk, hipStream_t);  // display-list games: render_list<Game> over one chunk's slow list
};
constexpr int MAX_GAME_TABLE_WORDS = 2048;
// mode 0: initial reset + first observation of every env; mode 1: one step
hipError_t launch_step(int game_id, const DevCtx &d, int mode, const LaunchStreams &ls);
hipError_t launch_render_one(int game_id, const DevCtx &d, int env, hipStream_t stream, int count = 1);
hipError_t launch_render_human(int game_id, const DevCtx &d, int env_base, int count, hipStream_t stream);
bool game_supported(int game_id);
int game_tier_for(int game_id, int slots_needed);
void game_limits(int game_id, int *ent_cap_hbm, int *grid_bytes);
void game_init_state(int game_id, int num_envs, int rand_seed, int env_offset, int env_stride, EnvHdr *hdr, uint32_t *rng);
int game_host_tables(int game_id, const GameOptions &opt, uint32_t *out, int max_words);
bool (*game_use_block_asset(int game_id))(int);
bool game_split_reset(int game_id);
int game_frame_rec_words(int game_id);
hipError_t launch_render_slow(int game_id, const DevCtx &d, int env_base, int count, int chunk, hipStream_t stream);
hipError_t launch_paint_backgrounds(const DevCtx &d, int env_base, int count, hipStream_t stream);  // use_generated_assets (pg_bgpaint.h); no-op otherwise
// render kernel launch order of the envs [base, base + count) of one launch chunk by background image (kernels.hip); scratch: MAX_BACKGROUNDS ints
hipError_t launch_render_order(const DevCtx &d, int base, int count, int *scratch, int *order, hipStream_t stream);
// device math self-tests (kernels.hip)
hipError_t selftest_bigfish_radius(const float *d_in, float *d_out, int n);
hipError_t selftest_sincos(uint32_t first_bits, int n, double *d_sin, double *d_cos);
hipError_t selftest_sincos_scaled(const uint32_t *d_bits, int n, double scale, float *d_sin, float *d_cos);
}  // namespace pgamd
