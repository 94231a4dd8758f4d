// kernels.hip -- gfx950 kernels of the vectorized stepper.  One workgroup = one 64-lane wavefront = one env.
//
//   step_tier0<Game> : grid = num_envs; LDS arena for Game::ENT_CAP_T0 entities (10 KB -> 16 workgroups / CU); skips
//                      envs routed to a larger arena.
//   step_list<Game,CAP,T> : fixed grids that walk the lists of envs whose entity table may outgrow the smaller
//                      arenas (ENT_CAP_T1 / ENT_CAP_T2), on a side stream.
//   render<Game>     : grid = num_envs, one wave per env: four passes of 16 rows through a 4 KB LDS band, RGB888
//                      observation stores (pg_render.h).
// The step kernels run Env<Game,CAP>::run (pg_env.h): HBM -> LDS staging, Game::step / reset + level generation,
// state write-back.
#include <hip/hip_runtime.h>

#include "games.h"
#include "pg_render.h"
#include "kernels.h"

namespace pgamd {

template <class Game>
__global__ __launch_bounds__(64) void step_tier0(DevCtx d, int mode, int env_base) {
    __shared__ Lds<Game, Game::ENT_CAP_T0> lds;
    const int env = env_base + (int)blockIdx.x;
    if (mode != 0 && d.route[env] != 0) return;  // owned by a larger arena this step
    Env<Game, Game::ENT_CAP_T0> e(d, env, &lds);
    e.run(mode);
}

template <class Game, int CAP, int TIER>
__global__ __launch_bounds__(64) void step_list(DevCtx d, int mode) {
    __shared__ Lds<Game, CAP> lds;
    const int count = d.big_count[TIER - 1];
    const int *list = d.big_list + (size_t)(TIER - 1) * d.num_envs;
    for (int k = (int)blockIdx.x; k < count; k += (int)gridDim.x) {
        const int env = list[k];
        if (d.route[env] != TIER) continue;  // set_state moved this env to another tier after the list was built
        Env<Game, CAP> e(d, env, &lds);
        e.run(mode);
        __syncthreads();
    }
}

template <class Game>
__global__ __launch_bounds__(64) void render(DevCtx d, int env_base) {
    __shared__ RenderLdsT<Game> lds;
    Renderer<Game> r(d, env_base + (int)blockIdx.x, &lds);
    r.render_env();
}

// The two step kernels touch disjoint envs, so the (few, slow, low-occupancy) large-arena envs run on a side
// stream concurrently with the small-arena grid.  The env range is further cut into chunks that alternate between
// two streams: the latency-bound step kernel of one chunk shares the CUs with the issue-bound render kernel of the
// previous chunk instead of the two phases running back to back.
template <class Game>
static hipError_t launch_game(const DevCtx &d, int mode, const LaunchStreams &ls) {
#define PG_TRY(x)                          \
    do {                                   \
        hipError_t e_ = (x);               \
        if (e_ != hipSuccess) return e_;   \
    } while (0)
    PG_TRY(hipEventRecord(ls.fork, ls.main));
    if (mode != 0) {
        PG_TRY(hipStreamWaitEvent(ls.side, ls.fork, 0));
        const int g1 = d.num_envs < 8192 ? d.num_envs : 8192, g2 = d.num_envs < 2048 ? d.num_envs : 2048;
        // the two list kernels run on their own streams (lane[1] is otherwise idle when chunks == 1)
        PG_TRY(hipStreamWaitEvent(ls.lane[1], ls.fork, 0));
        hipLaunchKernelGGL((step_list<Game, Game::ENT_CAP_T1, 1>), dim3(g1), dim3(64), 0, ls.side, d, mode);
        hipLaunchKernelGGL((step_list<Game, Game::ENT_CAP_T2, 2>), dim3(g2), dim3(64), 0, ls.lane[1], d, mode);
        PG_TRY(hipEventRecord(ls.tier2_done, ls.lane[1]));
        PG_TRY(hipStreamWaitEvent(ls.side, ls.tier2_done, 0));
        PG_TRY(hipEventRecord(ls.join, ls.side));
    }
    const int nchunk = (ls.chunks > 1 && d.num_envs >= 4096) ? ls.chunks : 1;
    const int per = (d.num_envs + nchunk - 1) / nchunk;
    for (int c = 0; c < nchunk; c++) {
        const int base = c * per;
        const int count = (d.num_envs - base) < per ? (d.num_envs - base) : per;
        if (count <= 0) break;
        hipStream_t st = nchunk == 1 ? ls.main : ls.lane[c & 1];
        if (nchunk > 1 && c < 2) PG_TRY(hipStreamWaitEvent(st, ls.fork, 0));
        if (!(d.debug_flags & 32) || mode == 0) hipLaunchKernelGGL(step_tier0<Game>, dim3(count), dim3(64), 0, st, d, mode, base);
        if (mode != 0) PG_TRY(hipStreamWaitEvent(st, ls.join, 0));
        if (!(d.debug_flags & 16)) hipLaunchKernelGGL(render<Game>, dim3(count), dim3(64), 0, st, d, base);
    }
    if (nchunk > 1) {
        for (int k = 0; k < 2; k++) {
            PG_TRY(hipEventRecord(ls.lane_done[k], ls.lane[k]));
            PG_TRY(hipStreamWaitEvent(ls.main, ls.lane_done[k], 0));
        }
    }
#undef PG_TRY
    return hipGetLastError();
}

hipError_t launch_step(int game_id, const DevCtx &d, int mode, const LaunchStreams &ls) {
    switch (game_id) {
#define PG_X(Game) \
    case Game::GAME_ID: return launch_game<Game>(d, mode, ls);
        PG_FOR_EACH_GAME(PG_X)
#undef PG_X
        default: return hipErrorInvalidValue;
    }
}

// re-renders one env (after set_state)
hipError_t launch_render_one(int game_id, const DevCtx &d, int env, hipStream_t stream) {
    switch (game_id) {
#define PG_X(Game)                                                                       \
    case Game::GAME_ID:                                                                  \
        hipLaunchKernelGGL(render<Game>, dim3(1), dim3(64), 0, stream, d, env);         \
        return hipGetLastError();
        PG_FOR_EACH_GAME(PG_X)
#undef PG_X
        default: return hipErrorInvalidValue;
    }
}

bool game_supported(int game_id) {
    switch (game_id) {
#define PG_X(Game) \
    case Game::GAME_ID: return true;
        PG_FOR_EACH_GAME(PG_X)
#undef PG_X
        default: return false;
    }
}

int game_tier_for(int game_id, int slots_needed) {
    switch (game_id) {
#define PG_X(Game) \
    case Game::GAME_ID: return slots_needed <= Game::ENT_CAP_T0 ? 0 : (slots_needed <= Game::ENT_CAP_T1 ? 1 : 2);
        PG_FOR_EACH_GAME(PG_X)
#undef PG_X
        default: return 0;
    }
}

void game_limits(int game_id, int *ent_cap_hbm, int *grid_bytes) {
    *ent_cap_hbm = 0;
    *grid_bytes = 0;
    switch (game_id) {
#define PG_X(Game)                                                                        \
    case Game::GAME_ID:                                                                   \
        *ent_cap_hbm = Game::ENT_CAP_T2;                                                  \
        *grid_bytes = game_grid_bytes<Game>();                                            \
        break;
        PG_FOR_EACH_GAME(PG_X)
#undef PG_X
        default: break;
    }
}

void game_init_state(int game_id, int num_envs, int rand_seed, int env_offset, int env_stride, EnvHdr *hdr, uint32_t *rng) {
    switch (game_id) {
#define PG_X(Game) \
    case Game::GAME_ID: init_env_state<Game>(num_envs, rand_seed, env_offset, env_stride, hdr, rng); break;
        PG_FOR_EACH_GAME(PG_X)
#undef PG_X
        default: break;
    }
}

}  // namespace pgamd
