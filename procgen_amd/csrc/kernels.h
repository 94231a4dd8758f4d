// kernels.h -- host-callable entry points of kernels.hip
#pragma once
#include <hip/hip_runtime_api.h>

#include "host_state.h"
#include "pg_defs.h"

namespace pgamd {
struct LaunchStreams {
    hipStream_t main, side;   // side: large-arena step kernel, concurrent with the small-arena one
    hipEvent_t fork, join;
    hipStream_t lane[2];      // env chunks alternate between these two streams (step of chunk c+1 overlaps render of chunk c)
    hipEvent_t lane_done[2];
    hipEvent_t tier2_done;    // the tier-2 list kernel runs on lane[1] ahead of that lane's chunk
    int chunks;               // 1 = everything on `main`
    int list_count[2] = {-1, -1};  // entries of the tier-1 / tier-2 lists this step reads, when the host knows them: an empty list's kernel is not launched
};
// what one per-game kernel object (kernels_game.hip) exports
struct GameEntry {
    int game_id;
    hipError_t (*launch)(const DevCtx &, int mode, const LaunchStreams &);
    hipError_t (*render_one)(const DevCtx &, int env, hipStream_t);
    int cap_t0, cap_t1, cap_t2;  // entity slots of the three LDS arenas; cap_t2 is the HBM table size
    int grid_bytes;
    bool has_lane;  // the game has a lane = env step path (mode-1 steps launch lane_step + reset_list instead of the tier-0 grid)
    void (*init_state)(int num_envs, int rand_seed, int env_offset, int env_stride, EnvHdr *hdr, uint32_t *rng);
};
// mode 0: initial reset + first observation of every env; mode 1: one step
hipError_t launch_step(int game_id, const DevCtx &d, int mode, const LaunchStreams &ls);
hipError_t launch_render_one(int game_id, const DevCtx &d, int env, hipStream_t stream);
bool game_supported(int game_id);
bool game_has_lane(int game_id);
int game_tier_for(int game_id, int slots_needed);
void game_limits(int game_id, int *ent_cap_hbm, int *grid_bytes);
void game_init_state(int game_id, int num_envs, int rand_seed, int env_offset, int env_stride, EnvHdr *hdr, uint32_t *rng);
}  // namespace pgamd
