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
#include "pg_prep.h"
#include "pg_human.h"
#include "kernels.h"

namespace pgamd {

// profiling aid (PROCGEN_AMD_DEBUG & 8192): when was this env's workgroup resident, and where.  Per env 32 words:
// [0] step start, [1] step end (100 MHz clock), [2] kernel kind << 32 | HW_ID; [3] first; [4], [5], [6] the same for the render kernel; [8..23] this step's phase cycles (with & 2048)
__device__ inline void trace_wave(const DevCtx &d, int env, int base, bool end, int kind) {
    if (PG_TRACE(d) && threadIdx.x == 0) {
        unsigned long long *t = d.wave_trace + (size_t)env * 32 + base;
        if (end) {
            t[1] = wall_clock64();
            if (base == 0) t[3] = d.first[env];
        } else {
            unsigned hw;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            t[0] = wall_clock64();
            t[2] = ((unsigned long long)kind << 32) | hw;
        }
    }
}

// Occupancy hint of the render kernel (RENDER_MIN_WAVES in a policy); the default leaves the register allocation alone.
// Round 1 tried it for coinrun (133 -> 128 VGPRs): the 104 B of spill per lane showed up as +54 % WRITE_SIZE.  Round 4: with the
// renderer's lane ids opaque to LICM (pg_render.h) and its LDS tables overlaid (8068 B), coinrun's kernel fits 96 VGPRs without
// scratch, i.e. five waves per SIMD: +3 % steps/s on the same box (tools/gpu/r4_occ.sh), so CoinRun sets 5.
#ifndef PG_RENDER_TRACE
#define PG_RENDER_TRACE 0
#endif
template <class Game, class = void>
struct GameRenderMinWaves {
    static constexpr int value = 1;  // no hint: the register allocation is left alone
};
template <class Game>
struct GameRenderMinWaves<Game, decltype((void)Game::RENDER_MIN_WAVES)> {
    static constexpr int value = Game::RENDER_MIN_WAVES;
};
// the same for the raster kernel of a display-list game (RASTER_MIN_WAVES; -DPG_RASTER_WAVES=n is the A/B's override)
template <class Game, class = void>
struct GameRasterMinWaves {
    static constexpr int value = 1;
};
template <class Game>
struct GameRasterMinWaves<Game, decltype((void)Game::RASTER_MIN_WAVES)> {
    static constexpr int value = Game::RASTER_MIN_WAVES;
};
#ifndef PG_RASTER_WAVES
#define PG_RASTER_WAVES GameRasterMinWaves<PG_GAME>::value
#endif
// games with SPLIT_RESET: their step kernels carry no level generator (Env NO_RESET, arena without scratch)
template <class Game, int CAP>
using StepEnv = Env<Game, CAP, GameSplit<Game>::value>;

template <class Game>
__global__ __launch_bounds__(64) void step_tier0(DevCtx d, int mode, int env_base) {
    __shared__ typename StepEnv<Game, Game::ENT_CAP_T0>::LdsT lds;
    if (GameSplit<Game>::value && blockIdx.x == 0 && threadIdx.x == 0) d.next_reset_count[d.reset_first > 0 ? (env_base >= d.reset_first ? 1 : 0) : env_base / d.reset_chunk_envs] = 0;
    // (a resident grid walking the chunk with a stride was tried: the loop form costs 60 VGPRs -- 121 -> 181, two waves per SIMD
    // instead of four -- and the dispatcher keeps the arenas full anyway, tools/gpu/micro/residency.hip)
    const int env = env_base + (int)blockIdx.x;
    if (mode != 0 && d.route[env] != 0) return;  // owned by a larger arena this step
    trace_wave(d, env, 0, false, 0);
    StepEnv<Game, Game::ENT_CAP_T0> e(d, env, &lds);
    e.run(mode);
    trace_wave(d, env, 0, true, 0);
}

// SPLIT_RESET games: the initial reset + first observation of every env (mode 0) needs the level generator's arena
template <class Game>
__global__ __launch_bounds__(64) void reset_grid(DevCtx d, int env_base) {
    __shared__ Lds<Game, GameSplit<Game>::RESET_CAP> lds;
    Env<Game, GameSplit<Game>::RESET_CAP> e(d, env_base + (int)blockIdx.x, &lds);
    e.run(0);
}

template <class Game, int CAP, int TIER>
__global__ __launch_bounds__(64) void step_list(DevCtx d, int mode, int chunk) {
    __shared__ typename StepEnv<Game, CAP>::LdsT lds;
    const int count = d.big_count[chunk * NUM_TIERS + TIER];
    const int *list = d.big_list + (size_t)TIER * d.num_envs + (size_t)chunk * d.chunk_envs;
    for (int k = (int)blockIdx.x; k < count; k += (int)gridDim.x) {
        const int env = list[k];
        if (d.route[env] != TIER) continue;  // set_state moved this env to another tier after the list was built
        trace_wave(d, env, 0, false, TIER);
        StepEnv<Game, CAP> e(d, env, &lds);
        e.run(mode);
        __syncthreads();
        trace_wave(d, env, 0, true, TIER);
    }
}

// SPLIT_RESET games: the episodes this chunk's step kernels ended: reset + level generation, outputs and state write-back (Env::run mode 2)
template <class Game>
__global__ __launch_bounds__(64) void reset_list(DevCtx d, int chunk, int env_base) {
    __shared__ Lds<Game, GameSplit<Game>::RESET_CAP> lds;
    const int count = d.reset_count[chunk];
    for (int k = (int)blockIdx.x; k < count; k += (int)gridDim.x) {
        Env<Game, GameSplit<Game>::RESET_CAP> e(d, d.reset_list[env_base + k], &lds);
        e.run(2);
        __syncthreads();
    }
}

// GEN: the handle runs with use_generated_assets (sprites on Qt's generic span route, per-env background canvases; pg_render.h)
template <class Game, bool GEN>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GEN ? 1 : GameRenderMinWaves<PG_GAME>::value))) void render(DevCtx d, int env_base) {
    __shared__ RenderLdsT<Game> lds;
    if (d.clear_lists && blockIdx.x == 0 && threadIdx.x < LIST_COUNTERS) const_cast<int *>(d.big_count)[threadIdx.x] = 0;
    // (the residency trace of the render kernel is a build option, -DPG_RENDER_TRACE=1: its two calls cost coinrun's renderer the
    // three registers that separate 96 VGPRs from a spill at five waves per SIMD)
    const int slot = env_base + (int)blockIdx.x;
    const int env = d.render_order ? __builtin_amdgcn_readfirstlane(d.render_order[slot]) : slot;  // (kept in a scalar register: every address of the frame derives from it)
    if (PG_RENDER_TRACE) trace_wave(d, env, 4, false, 8);
    Renderer<Game, GEN> r(d, env, &lds);
    r.render_env();
    if (PG_RENDER_TRACE) trace_wave(d, env, 4, true, 8);
}
// ---- display-list games (pg_prep.h): prep -> raster -> render_list ----
#ifndef PG_PREP_WAVES
#define PG_PREP_WAVES 1
#endif
template <class Game>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PG_PREP_WAVES))) void prep(DevCtx d, int env_base, int count, int chunk) {
    // the render arena without its band buffer (but for the words the table builders use as scratch): 2.9 KB instead of 6.2 KB for coinrun,
    // i.e. eight waves of this latency-bound kernel per SIMD instead of five
    __shared__ __attribute__((aligned(16))) uint32_t arena[offsetof(RenderLdsT<Game>, fb) / 4 + RenderLdsT<Game>::PREP_FB_WORDS];
    RenderLdsT<Game> *lds = reinterpret_cast<RenderLdsT<Game> *>(arena);
    const int env0 = env_base + (int)blockIdx.x * PREP_ENVS;
    const int left = env_base + count - env0;
    FramePrep<Game> p(d, lds, d.slow_count + d.step_parity * MAX_CHUNKS + chunk, d.slow_list + env_base);
    p.run(env0, left < PREP_ENVS ? left : PREP_ENVS);
}
// (no occupancy hint and no full renderer in here: 59 VGPRs, so the arena alone bounds the waves per SIMD -- six for coinrun against five
// with render_env inside, +3.4 % steps/s, profiles/r06_call11_ab.txt)
template <class Game>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PG_RASTER_WAVES))) void raster(DevCtx d, int env_base, int chunk) {
    // (the arena up to the end of the band buffer: typeimg, behind it, is the per-cell path's.  The band buffer of a register-table rasterizer,
    // and of a game without a grid, starts where the column / row / type tables would lie, RenderLdsT: coinrun 4688 bytes, bigfish 4432 =
    // four LDS granules of 1280 bytes, eight waves per SIMD)
    constexpr size_t ARENA_BYTES = GameRasterBandOverTables<Game>::value ? offsetof(RenderLdsT<Game>, ci) + sizeof(RenderLdsT<Game>::fb) : offsetof(RenderLdsT<Game>, typeimg);
    __shared__ __attribute__((aligned(16))) uint32_t arena[ARENA_BYTES / 4];
    RenderLdsT<Game> &lds = *reinterpret_cast<RenderLdsT<Game> *>(arena);
    const int slot = env_base + (int)blockIdx.x;
    const int env = d.render_order ? __builtin_amdgcn_readfirstlane(d.render_order[slot]) : slot;
    // (the record is read before this kernel's first store: the wave-uniform loads of its header are then scalar loads, not vector loads
    // followed by a v_readfirstlane each)
    if ((d.frame_rec[(size_t)env * FrameRec<Game>::WORDS + FrameRec<Game>::FLAGS] & FrameRec<Game>::F_FAST) != 0) {  // (else: on the chunk's slow list)
        Renderer<Game, false> r(d, env, &lds);
        r.raster_env();
    }
    if (d.clear_lists && blockIdx.x == 0 && threadIdx.x < LIST_COUNTERS) const_cast<int *>(d.big_count)[threadIdx.x] = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) d.slow_count[(d.step_parity ^ 1) * MAX_CHUNKS + chunk] = 0;  // the next step's counter of this chunk
}
// the frames of a chunk the rasterizer's short path cannot draw: the full renderer, a small grid walking the chunk's slow list
template <class Game>
__global__ __launch_bounds__(64) void render_list(DevCtx d, int env_base, int chunk) {
    __shared__ RenderLdsT<Game> lds;
    const int count = d.slow_count[d.step_parity * MAX_CHUNKS + chunk];
    for (int k = (int)blockIdx.x; k < count; k += (int)gridDim.x) {
        Renderer<Game, false> r(d, __builtin_amdgcn_readfirstlane(d.slow_list[env_base + k]), &lds);
        r.render_env();
        __syncthreads();
    }
}
// clear_lists: this is the step's render of env 0 (every list kernel of the step is done when a render kernel starts); a
// hipMemsetAsync per step instead was two fill kernels, each tens of microseconds on a busy device with many handles
// t0 / t1 (null: off): events recorded around the launch of the frame kernel proper -- render, or raster for a display-list game -- on its
// stream (procgen_amd_kernel_timing: the dominant kernel's own duration)
template <class Game>
static void launch_render(const DevCtx &d0, int env_base, int count, hipStream_t st, bool clear_lists = false, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr, int chunk = 0) {
    DevCtx d = d0;
    d.clear_lists = clear_lists ? 1 : 0;
    if constexpr (GameDisplayList<Game>::value) {
        if (d.frame_rec && !d.gen_bg) {
            hipLaunchKernelGGL(prep<Game>, dim3((count + PREP_ENVS - 1) / PREP_ENVS), dim3(64), 0, st, d, env_base, count, chunk);
            if (t0) (void)hipEventRecord(t0, st);
            hipLaunchKernelGGL(raster<Game>, dim3(count), dim3(64), 0, st, d, env_base, chunk);
            if (t1) (void)hipEventRecord(t1, st);
            return;  // (frames the short path cannot draw: render_slow, launched by libenv_observe when a prep wave raised the flag)
        }
    }
    if (t0) (void)hipEventRecord(t0, st);
    if (d.gen_bg) hipLaunchKernelGGL((render<Game, true>), dim3(count), dim3(64), 0, st, d, env_base);
    else hipLaunchKernelGGL((render<Game, false>), dim3(count), dim3(64), 0, st, d, env_base);
    if (t1) (void)hipEventRecord(t1, st);
}

// One step of a handle.  The envs are stepped by up to three kinds of kernel that touch disjoint envs (route table):
//   * step_tier0 grids over env chunks -- chunk c on stream lane[c & 1], followed by the chunk's render kernel, so the
//     latency-bound step work of one chunk shares the CUs with the issue-bound render kernel of its neighbour;
//   * the tier-1 and tier-2 list kernels (larger LDS arenas).
// A chunk's render kernel waits for the list kernels (their envs lie in every chunk).
template <class Game>
static hipError_t launch_game(const DevCtx &d, int mode, const LaunchStreams &ls) {
#define PG_TRY(x)                          \
    do {                                   \
        hipError_t e_ = (x);               \
        if (e_ != hipSuccess) return e_;   \
    } while (0)
    const int *cnt = ls.list_count[0];
    const bool t1 = mode != 0 && cnt[1] != 0, t2 = mode != 0 && cnt[2] != 0;
    const int g1 = cnt[1] < 0 || cnt[1] > 8192 ? (d.num_envs < 8192 ? d.num_envs : 8192) : cnt[1];
    const int g2 = cnt[2] < 0 || cnt[2] > 2048 ? (d.num_envs < 2048 ? d.num_envs : 2048) : cnt[2];
    const int rg = d.num_envs < 1024 ? d.num_envs : 1024;
    if (d.num_envs < 4096) {
        // Small handles (and the parts of a joint handle, each with its own stream): the kernels are far too short for
        // concurrency to matter, while every cross-stream event costs tens of microseconds and the runtime multiplexes
        // all streams of the process onto a few hardware queues.  Everything goes down one stream.
        if (t1) hipLaunchKernelGGL((step_list<Game, Game::ENT_CAP_T1, 1>), dim3(g1), dim3(64), 0, ls.main, d, mode, 0);
        if (t2) hipLaunchKernelGGL((step_list<Game, Game::ENT_CAP_T2, 2>), dim3(g2), dim3(64), 0, ls.main, d, mode, 0);
        if constexpr (GameSplit<Game>::value) {
            if (mode == 0) {
                hipLaunchKernelGGL(reset_grid<Game>, dim3(d.num_envs), dim3(64), 0, ls.main, d, 0);
            } else {
                hipLaunchKernelGGL(step_tier0<Game>, dim3(d.num_envs), dim3(64), 0, ls.main, d, mode, 0);
                hipLaunchKernelGGL(reset_list<Game>, dim3(rg), dim3(64), 0, ls.main, d, 0, 0);
            }
        } else {
            if (!(d.debug_flags & 32) || mode == 0) hipLaunchKernelGGL(step_tier0<Game>, dim3(d.num_envs), dim3(64), 0, ls.main, d, mode, 0);
        }
        PG_TRY(launch_paint_backgrounds(d, 0, d.num_envs, ls.main));
        if (!(d.debug_flags & 16)) launch_render<Game>(d, 0, d.num_envs, ls.main, true, ls.render_t0[0], ls.render_t1[0]);
        return hipGetLastError();
    }
    // At most four streams: the runtime deals a process's streams round-robin onto four hardware queues, and two streams on one
    // queue run their kernels one after the other.
    //   main    : tier-1 list          lane[1] : tier-2 list, then chunk 1, 3, ...
    //                                  lane[0] : chunk 0, 2, ...
    // order 4: chunk 0 -- the step's longest chain -- runs on the MAIN stream, in order behind the upload of the actions, and the tier-1
    // list takes lane[0]: the first step_tier0 grid then starts with the step instead of a cross-queue event (~20-30 us) later
    const bool c0_main = ls.order == 4;
    hipStream_t s_t1 = c0_main ? ls.lane[0] : ls.main;
    PG_TRY(hipEventRecord(ls.fork, ls.main));
    PG_TRY(hipStreamWaitEvent(ls.lane[0], ls.fork, 0));
    PG_TRY(hipStreamWaitEvent(ls.lane[1], ls.fork, 0));
    const int nchunk = ls.chunks > 1 ? (ls.chunks < MAX_CHUNKS ? ls.chunks : MAX_CHUNKS) : 1;
    const int per = ((d.num_envs + nchunk - 1) / nchunk + TILE_ENVS - 1) / TILE_ENVS * TILE_ENVS;
    // Two chunks are cut unevenly (PROCGEN_AMD_FIRST_PCT, default 75 / 25; games without split resets): with equal chunks the two
    // streams run in lockstep -- both step kernels, then both render kernels -- and the step / render overlap the chunks exist
    // for hardly happens.  Measured 25 / 75 or 75 / 25 against 50 / 50: starpilot +9 %, maze +3-6 %, bigfish +4 %, coinrun +1 %
    // (coinrun only with the large chunk first: its tier-2 list kernel already delays the second stream).  Three or four
    // chunks, even or uneven (50/30/20, 40/30/20/10, ...), measured 15-18 % slower than two for coinrun and starpilot.
    const int first = (nchunk == 2 && ls.first_pct > 0) ? first_chunk_envs(d.num_envs, ls.first_pct) : 0;  // (== DevCtx::reset_first)
    // (submitting chunk 0's step grid ahead of the list kernels -- the device idles until the first large grid is in its queue -- measured
    // 0 for coinrun and -1.7 % for bigfish, profiles/r06_call32_order16.txt: not kept)
    if (t1) {
        hipLaunchKernelGGL((step_list<Game, Game::ENT_CAP_T1, 1>), dim3(g1), dim3(64), 0, s_t1, d, mode, 0);
        PG_TRY(hipEventRecord(ls.side_done[0], s_t1));
    }
    // launch order experiments (PROCGEN_AMD_ORDER): 0 = tier-2 list ahead of chunk 1 on its stream (which also delays that
    // chunk's step kernel: an accidental pipeline); 1 / 3 = tier-2 list on the side stream and chunk c + 1's step kernel
    // explicitly behind chunk c's; 2 = side stream, no chaining
    const bool t2_side = ls.order != 0 && ls.side[0] != nullptr;
    const bool chain = ls.order == 1 || ls.order == 3;
    if (t2) {
        hipStream_t s2 = t2_side ? ls.side[0] : ls.lane[1];
        if (t2_side) PG_TRY(hipStreamWaitEvent(s2, ls.fork, 0));
        hipLaunchKernelGGL((step_list<Game, Game::ENT_CAP_T2, 2>), dim3(g2), dim3(64), 0, s2, d, mode, 0);
        PG_TRY(hipEventRecord(ls.side_done[1], s2));
    }
    for (int c = 0; c < nchunk; c++) {
        const int base = first > 0 ? (c == 0 ? 0 : first) : c * per;
        const int count = first > 0 ? (c == 0 ? first : d.num_envs - first) : ((d.num_envs - base) < per ? (d.num_envs - base) : per);
        if (count <= 0) break;
        hipStream_t st = (c0_main && (c & 1) == 0) ? ls.main : ls.lane[c & 1];
        if (chain && c > 0 && mode != 0) PG_TRY(hipStreamWaitEvent(st, ls.step_done[c - 1], 0));
        if (GameSplit<Game>::value && mode == 0) {
            if constexpr (GameSplit<Game>::value) hipLaunchKernelGGL(reset_grid<Game>, dim3(count), dim3(64), 0, st, d, base);
        } else if (!(d.debug_flags & 32) || mode == 0) {
            hipLaunchKernelGGL(step_tier0<Game>, dim3(count), dim3(64), 0, st, d, mode, base);
        }
        if (chain && mode != 0) PG_TRY(hipEventRecord(ls.step_done[c], st));
        if (t1) PG_TRY(hipStreamWaitEvent(st, ls.side_done[0], 0));
        if (t2 && ((c & 1) == 0 || t2_side)) PG_TRY(hipStreamWaitEvent(st, ls.side_done[1], 0));
        if constexpr (GameSplit<Game>::value) {
            // the episodes this chunk's step kernel (and the list kernels, for its envs) ended: next level, outputs, routing
            if (mode != 0) hipLaunchKernelGGL(reset_list<Game>, dim3(count < 1024 ? count : 1024), dim3(64), 0, st, d, c, base);
        }
        PG_TRY(launch_paint_backgrounds(d, base, count, st));  // (after the list kernels: their envs lie in every chunk)
        // rew / first / info of this chunk's envs, and the list counters, are final here (step, list and reset kernels done): what the
        // early download of the step's small outputs waits for (libenv_hip.cpp VecGame::launch)
        if (ls.outputs_done[c]) PG_TRY(hipEventRecord(ls.outputs_done[c], st));
        if (!(d.debug_flags & 16)) launch_render<Game>(d, base, count, st, c == 0, ls.render_t0[c], ls.render_t1[c], c);
        if (ls.frames_done[c]) PG_TRY(hipEventRecord(ls.frames_done[c], st));
    }
    for (int k = 0; k < 2; k++) {
        PG_TRY(hipEventRecord(ls.lane_done[k], ls.lane[k]));
        PG_TRY(hipStreamWaitEvent(ls.main, ls.lane_done[k], 0));
    }
#undef PG_TRY
    return hipGetLastError();
}

// render_human (pg_human.h): one wave per (env, band of 32 rows) of the 512 x 512 antialiased info frame.  Not a hot path: the
// reference draws these frames serially on the Python thread (src/vecgame.cpp:367-375); handles made without render_human never launch it.
template <class Game>
__global__ __launch_bounds__(64) void render_human(DevCtx d, int env_base) {
    __shared__ HumanLds lds;
    HumanRenderer<Game> r(d, env_base + (int)(blockIdx.x / HUMAN_BANDS), &lds, (int)(blockIdx.x % HUMAN_BANDS));
    r.render_band();
}
template <class Game>
static hipError_t launch_human(const DevCtx &d, int env_base, int count, hipStream_t stream) {
    hipLaunchKernelGGL(render_human<Game>, dim3((unsigned)count * HUMAN_BANDS), dim3(64), 0, stream, d, env_base);
    return hipGetLastError();
}

template <class Game>
static hipError_t render_one(const DevCtx &d, int env, int count, hipStream_t stream) {  // re-renders envs [env, env + count) (after set_state / set_states)
    DevCtx d1 = d;
    d1.render_order = nullptr;  // (the envs themselves, not slots of a chunk launch)
    d1.frame_rec = nullptr;     // (and the full renderer draws them)
    launch_render<Game>(d1, env, count, stream);
    return hipGetLastError();
}

// display-list games: the frames of chunk `chunk` = envs [env_base, env_base + count) that its prep kernel queued for the full renderer
template <class Game>
static hipError_t render_slow(const DevCtx &d, int env_base, int count, int chunk, hipStream_t stream) {
    if constexpr (GameDisplayList<Game>::value) {
        DevCtx d1 = d;
        d1.render_order = nullptr;
        hipLaunchKernelGGL(render_list<Game>, dim3(count < 4096 ? count : 4096), dim3(64), 0, stream, d1, env_base, chunk);
    }
    return hipGetLastError();
}

#define PG_CAT2(a, b) a##b
#define PG_CAT(a, b) PG_CAT2(a, b)
// a host function (a namespace-scope const table would also be emitted for the device, where the launchers do not exist)
const GameEntry *PG_CAT(game_entry_, PG_GAME)() {
    static const GameEntry e = {
        PG_GAME::GAME_ID,    launch_game<PG_GAME>,       render_one<PG_GAME>,     PG_GAME::ENT_CAP_T0, PG_GAME::ENT_CAP_T1,
        PG_GAME::ENT_CAP_T2, game_grid_bytes<PG_GAME>(), init_env_state<PG_GAME>,
        GameHostTables<PG_GAME>::build,
        GameBlockAsset<PG_GAME>::is,
        launch_human<PG_GAME>,
        GameSplit<PG_GAME>::value,
        GameDisplayList<PG_GAME>::value ? FrameRec<PG_GAME>::WORDS : 0,
        render_slow<PG_GAME>,
    };
    return &e;
}

}  // namespace pgamd
