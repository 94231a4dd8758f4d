// kernels_game.hip -- gfx950 kernels of the vectorized stepper, compiled once per game (-DPG_GAME=<policy struct>):
// the sixteen instantiations build in parallel and land in one libenv.so; kernels.hip holds the table that dispatches on game id.  One workgroup = one 64-lane wavefront = one env.
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

// Occupancy hint of the render kernel (RENDER_MIN_WAVES in a policy); the default leaves the register allocation alone.
// Tried for coinrun (133 -> 128 VGPRs, a fourth wave per SIMD): +2..4 % steps/s, but the 104 B of spill per lane showed
// up as +54 % WRITE_SIZE, so no policy sets it.
template <class Game, class = void>
struct GameRenderMinWaves {
    static constexpr int value = 1;
};
template <class Game>
struct GameRenderMinWaves<Game, decltype((void)Game::RENDER_MIN_WAVES)> {
    static constexpr int value = Game::RENDER_MIN_WAVES;
};
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

// lane = env physics: one LANE per env, one wave per tile of 64 consecutive envs (the tile-interleaved entity table makes a
// wave's accesses to one slot contiguous); LDS holds the hot words of each env's first entity slots (LaneLds).  Takes the envs the route table gives it; everything wave-structured
// (level generation after an episode ends, generator twists) goes to the wave = env kernels through the lists.
template <class Game>
__global__ __launch_bounds__(64) void lane_step(DevCtx d, int chunk, int env_base, int env_end) {
    __shared__ LaneLds<typename Game::cell_t> cache;
    // One wave per workgroup and a long chain of dependent operations: when it shares a SIMD with the issue-hungry waves of
    // another chunk's render kernel it should win the arbitration, the others fill its stalls.
    if (!(d.debug_flags & 8192)) __builtin_amdgcn_s_setprio(3);
    if (blockIdx.x == 0 && threadIdx.x == 0) d.next_reset_count[chunk] = 0;
    const int env = env_base + (int)blockIdx.x * TILE_ENVS + (int)threadIdx.x;
    if (env >= env_end) return;
    if (d.route[env] != ROUTE_LANE) return;
    Env<Game, Game::ENT_CAP_T2, true> e(d, env, nullptr);
    e.lcache = (PG_LDS_PTR(uint32_t))(cache.c + threadIdx.x);
    e.lwin = (PG_LDS_PTR(typename Game::cell_t))(cache.win + threadIdx.x);
    e.lcand = (PG_LDS_PTR(uint32_t))(cache.cand + threadIdx.x);
    e.has_lds = true;
    e.run_lane(chunk, env_base);
}

// the episodes the lane kernel of this chunk ended: reset + level generation, outputs and state write-back (Env::run mode 2)
template <class Game>
__global__ __launch_bounds__(64) void reset_list(DevCtx d, int chunk, int env_base) {
    __shared__ Lds<Game, Game::ENT_CAP_T0> lds;
    const int count = d.reset_count[chunk];
    for (int k = (int)blockIdx.x; k < count; k += (int)gridDim.x) {
        Env<Game, Game::ENT_CAP_T0> e(d, d.reset_list[env_base + k], &lds);
        e.run(2);
        __syncthreads();
    }
}

template <class Game>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GameRenderMinWaves<PG_GAME>::value))) void render(DevCtx d, int env_base) {
    __shared__ RenderLdsT<Game> lds;
    Renderer<Game> r(d, env_base + (int)blockIdx.x, &lds);
    r.render_env();
}

// The two step kernels touch disjoint envs, so the (few, slow, low-occupancy) large-arena envs run on a side
// stream concurrently with the small-arena grid.  The env range is further cut into chunks that alternate between
// two streams: the latency-bound step kernel of one chunk shares the CUs with the issue-bound render kernel of the
// previous chunk instead of the two phases running back to back.
// the step work of the envs [base, base + count) on one stream: the tier-0 grid, or (games with a lane = env path) the lane
// kernel and the reset kernel behind it
template <class Game>
static void launch_chunk_step(const DevCtx &d, int mode, int chunk, int base, int count, hipStream_t st) {
    if constexpr (GameLane<Game>::value) {
        if (mode != 0) {
            if (d.debug_flags & 32) return;
            hipLaunchKernelGGL(lane_step<Game>, dim3((count + TILE_ENVS - 1) / TILE_ENVS), dim3(64), 0, st, d, chunk, base, base + count);
            const int rg = count < 512 ? count : 512;
            hipLaunchKernelGGL(reset_list<Game>, dim3(rg), dim3(64), 0, st, d, chunk, base);
            return;
        }
    }
    if (!(d.debug_flags & 32) || mode == 0) hipLaunchKernelGGL(step_tier0<Game>, dim3(count), dim3(64), 0, st, d, mode, base);
}

template <class Game>
static hipError_t launch_game(const DevCtx &d, int mode, const LaunchStreams &ls) {
#define PG_TRY(x)                          \
    do {                                   \
        hipError_t e_ = (x);               \
        if (e_ != hipSuccess) return e_;   \
    } while (0)
    if (d.num_envs < 4096) {
        // Small handles (and the 16 parts of a joint handle, each with its own streams): the kernels are far too short
        // for tier / chunk concurrency to matter, while every cross-stream event costs tens of microseconds and the
        // runtime multiplexes all streams of the process onto a few hardware queues.  Everything goes down one stream.
        if (mode != 0) {
            const int g1 = d.num_envs < 8192 ? d.num_envs : 8192, g2 = d.num_envs < 2048 ? d.num_envs : 2048;
            if (ls.list_count[0] != 0) hipLaunchKernelGGL((step_list<Game, Game::ENT_CAP_T1, 1>), dim3(g1), dim3(64), 0, ls.main, d, mode);
            if (ls.list_count[1] != 0) hipLaunchKernelGGL((step_list<Game, Game::ENT_CAP_T2, 2>), dim3(g2), dim3(64), 0, ls.main, d, mode);
        }
        launch_chunk_step<Game>(d, mode, 0, 0, d.num_envs, ls.main);
        if (!(d.debug_flags & 16)) hipLaunchKernelGGL(render<Game>, dim3(d.num_envs), dim3(64), 0, ls.main, d, 0);
        return hipGetLastError();
    }
    PG_TRY(hipEventRecord(ls.fork, ls.main));
    if (mode != 0) {
        PG_TRY(hipStreamWaitEvent(ls.side, ls.fork, 0));
        const int g1 = d.num_envs < 8192 ? d.num_envs : 8192, g2 = d.num_envs < 2048 ? d.num_envs : 2048;
        // the two list kernels run on their own streams (lane[1] is otherwise idle when chunks == 1)
        PG_TRY(hipStreamWaitEvent(ls.lane[1], ls.fork, 0));
        if (ls.list_count[0] != 0) hipLaunchKernelGGL((step_list<Game, Game::ENT_CAP_T1, 1>), dim3(g1), dim3(64), 0, ls.side, d, mode);
        if (ls.list_count[1] != 0) hipLaunchKernelGGL((step_list<Game, Game::ENT_CAP_T2, 2>), dim3(g2), dim3(64), 0, ls.lane[1], d, mode);
        PG_TRY(hipEventRecord(ls.tier2_done, ls.lane[1]));
        PG_TRY(hipStreamWaitEvent(ls.side, ls.tier2_done, 0));
        PG_TRY(hipEventRecord(ls.join, ls.side));
    }
    const int nchunk = (ls.chunks > 1 && d.num_envs >= 4096) ? ls.chunks : 1;
    const int per = ((d.num_envs + nchunk - 1) / nchunk + TILE_ENVS - 1) / TILE_ENVS * TILE_ENVS;  // whole tiles
    for (int c = 0; c < nchunk; c++) {
        const int base = c * per;
        const int count = (d.num_envs - base) < per ? (d.num_envs - base) : per;
        if (count <= 0) break;
        hipStream_t st = nchunk == 1 ? ls.main : ls.lane[c & 1];
        if (nchunk > 1 && c < 2) PG_TRY(hipStreamWaitEvent(st, ls.fork, 0));
        launch_chunk_step<Game>(d, mode, c, base, count, st);
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


template <class Game>
static hipError_t render_one(const DevCtx &d, int env, hipStream_t stream) {  // re-renders one env (after set_state)
    hipLaunchKernelGGL(render<Game>, dim3(1), dim3(64), 0, stream, d, env);
    return hipGetLastError();
}

#define PG_CAT2(a, b) a##b
#define PG_CAT(a, b) PG_CAT2(a, b)
// a host function (a namespace-scope const table would also be emitted for the device, where the launchers do not exist)
const GameEntry *PG_CAT(game_entry_, PG_GAME)() {
    static const GameEntry e = {
        PG_GAME::GAME_ID,    launch_game<PG_GAME>,       render_one<PG_GAME>,     PG_GAME::ENT_CAP_T0, PG_GAME::ENT_CAP_T1,
        PG_GAME::ENT_CAP_T2, game_grid_bytes<PG_GAME>(), GameLane<PG_GAME>::value, init_env_state<PG_GAME>,
    };
    return &e;
}

}  // namespace pgamd
