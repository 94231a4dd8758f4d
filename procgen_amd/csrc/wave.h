// wave.h -- the execution model the kernels are written against: ONE 64-lane wavefront == ONE workgroup
// == ONE environment.
//
// Kernel code is "wave-structured": control flow is wave-uniform (every lane takes the same branches on
// the same values), and lane parallelism is expressed only through
//   PG_FOR_LANES(l) { ... }      the body runs once per lane, `l` = lane id (0..63)
//   PG_BALLOT(l, pred)           64-bit mask of lanes whose predicate holds
//   PG_SYNC()                    orders LDS/global traffic between two lane sections
//   PG_LANE_VAR(T, v) / PG_LV(v, l) / PG_READLANE(v, k)
//                                a per-lane value that outlives a lane section, and its read from uniform
//                                code (v_readlane on the GPU)
// Rule: inside one PG_FOR_LANES section a lane may only read locations that no other lane writes in the
// same section; cross-lane hand-offs go through LDS with a PG_SYNC() between the sections.
//
// On the GPU (hipcc, gfx950) PG_FOR_LANES is just "this lane", PG_BALLOT is v_cmp -> SGPR pair, and
// PG_SYNC is a workgroup-scope memory fence for the wave (no s_barrier: waves of a multi-wave workgroup,
// e.g. the 4 band-waves of the render kernel, never wait for each other).  tests/emu builds the same sources with
// PGAMD_WAVE_EMU, where a lane section is a 64-iteration loop on the host: a debugging harness for the
// kernel logic (this container has no GPU); it is never part of libenv.so.
#pragma once
#include <stdint.h>

#if defined(PGAMD_WAVE_EMU)

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define PG_DEV inline
// the lane a lane section is currently executing (-1: wave-uniform code); lets the emulation give
// PG_UNIFORM_I its GPU meaning (the value of the first lane) so that a readfirstlane on a lane-varying value is
// caught by the CPU tests instead of only on the device
inline int &pg_emu_lane() {
    static thread_local int lane = -1;
    return lane;
}
// event counters the harness reads back (which code paths a test really took); slot 0: objects stepped by the parallel pass,
// 1-4: bso_free_objects rounds / sub_steps evaluated / calls / objects; 5-6: nested sub_steps run / skipped; 7: -DPG_ROT_POOL builds of the
// renderer: windows cut for want of rotation records + 1e6 x frames sent to the per-band path
inline long long *pg_emu_counters() {
    static long long c[8] = {0};
    return c;
}
struct PgEmuLaneScope {
    int saved;
    PgEmuLaneScope() : saved(pg_emu_lane()) {}
    ~PgEmuLaneScope() { pg_emu_lane() = saved; }
};
#define PG_FOR_LANES(l) \
    if (PgEmuLaneScope pg_scope_{}; true) \
        for (int l = 0; l < 64 && ((pg_emu_lane() = l), true); ++l)
#define PG_FOR_LANES_NOHOIST(l) PG_FOR_LANES(l)
// global -> LDS copy of one dword per lane without a register in between (global_load_lds_dword on the GPU): lane l's word lands at
// lds_base[l].  The copy is asynchronous there: pg_dma_join() before the LDS words are read or overwritten.
// The emulation models the asynchrony at its worst: the words are read at once but LAND at the join, in issue order.  A reader of the
// LDS words that forgot the join sees the old contents, a writer that forgot it is overwritten by the late copy -- either way the CPU
// goldens differ, instead of only a GPU run being able to tell (round-4 advisor finding).  pg_emu_dma_outstanding() lets a kernel's
// end assert that nothing is left in flight.
struct PgEmuDmaWord {
    uint32_t *dst;
    uint32_t val;
};
inline PgEmuDmaWord *pg_emu_dma_queue(int **count) {
    static thread_local PgEmuDmaWord q[64 * 64];
    static thread_local int n = 0;
    *count = &n;
    return q;
}
inline void pg_emu_dma_push(uint32_t *dst, uint32_t val) {
    int *n;
    PgEmuDmaWord *q = pg_emu_dma_queue(&n);
    if (*n >= 64 * 64) {
        fprintf(stderr, "wave emulation: more than 4096 LDS-DMA words in flight\n");
        abort();
    }
    q[*n].dst = dst;
    q[*n].val = val;
    ++*n;
}
inline void pg_emu_dma_flush() {
    int *n;
    PgEmuDmaWord *q = pg_emu_dma_queue(&n);
    for (int i = 0; i < *n; i++) *q[i].dst = q[i].val;
    *n = 0;
}
inline int pg_emu_dma_outstanding() {
    int *n;
    (void)pg_emu_dma_queue(&n);
    return *n;
}
#define PG_DMA_DWORD(gptr, lds_base, l) pg_emu_dma_push(&(lds_base)[l], *(gptr))
#define PG_DMA_JOIN() pg_emu_dma_flush()
#define PG_BALLOT(l, pred)                              \
    ({                                                  \
        uint64_t m_ = 0;                                \
        PgEmuLaneScope pg_bscope_{};                    \
        for (int l = 0; l < 64; ++l) {                  \
            pg_emu_lane() = l;                          \
            if (pred) m_ |= (1ull << l);                \
        }                                               \
        m_;                                             \
    })
#define PG_SYNC() ((void)0)
#define PG_UNIFORM_I(x)                                 \
    ({                                                  \
        static thread_local int pg_first_;              \
        const int v_ = (int)(x);                        \
        if (pg_emu_lane() <= 0) pg_first_ = v_;         \
        pg_emu_lane() > 0 ? pg_first_ : v_;             \
    })
#define PG_LANE_VAR(T, v) T v[64]
#define PG_LANE_REF(T, v) T (&v)[64]
#define PG_LV(v, l) v[l]
#define PG_READLANE(v, k) v[k]
#define PG_LANE_ARR(T, v, N) T v[N][64]
#define PG_LA(v, j, l) v[j][l]
#define PG_LANE_ARR_REF(T, v, N) T (&v)[N][64]
#define PG_SHFL(v, l, src) v[(src) & 63]  // inside a lane section: the value lane `src` holds (v is only read in that section)
// wave-uniform code: the OR of a 32-bit lane variable over the 64 lanes
#define PG_WAVE_OR(v)                              \
    ({                                             \
        uint32_t r_ = 0;                           \
        for (int l_ = 0; l_ < 64; ++l_) r_ |= (uint32_t)v[l_]; \
        r_;                                        \
    })
// atomic OR on a 32-bit word shared by the wave's lanes, returns the old value (the emulation's lanes run one after the other)
PG_DEV uint32_t pg_atomic_or(uint32_t *p, uint32_t v) {
    const uint32_t o = *p;
    *p = o | v;
    return o;
}
PG_DEV int pg_popc64(uint64_t m) { return __builtin_popcountll(m); }
PG_DEV int pg_clz64(uint64_t m) { return m ? __builtin_clzll(m) : 64; }
PG_DEV int pg_ctz64(uint64_t m) { return m ? __builtin_ctzll(m) : 64; }
PG_DEV uint64_t pg_brev64(uint64_t m) {
    uint64_t r = 0;
    for (int i = 0; i < 64; i++) r |= ((m >> i) & 1ull) << (63 - i);
    return r;
}
PG_DEV double pg_sqrt(double x) { return sqrt(x); }
PG_DEV double pg_floor(double x) { return floor(x); }
PG_DEV float pg_floorf(float x) { return floorf(x); }
PG_DEV double pg_ceil(double x) { return ceil(x); }
PG_DEV float pg_fabsf(float x) { return fabsf(x); }
PG_DEV float pg_roundf(float x) { return roundf(x); }
PG_DEV double pg_fabs(double x) { return fabs(x); }
PG_DEV double pg_pow(double x, double y) { return pow(x, y); }  // glibc, as the reference
// v_perm_b32: byte i of the result is byte sel[i] of the 8-byte value (hi << 32 | lo) (selectors 0-7 only)
PG_DEV uint32_t pg_perm(uint32_t hi, uint32_t lo, uint32_t sel) {
    const uint64_t v = ((uint64_t)hi << 32) | lo;
    uint32_t r = 0;
    for (int i = 0; i < 4; i++) r |= (uint32_t)((v >> (8 * ((sel >> (8 * i)) & 7u))) & 0xffu) << (8 * i);
    return r;
}

#else

#include <hip/hip_runtime.h>
#define PG_DEV __device__ __forceinline__
#define PG_LANE_ID() ((int)(threadIdx.x & 63u))
// PG_FOR_LANES_NOHOIST: the lane id of the section is taken through an empty asm, so what the section derives from it (LDS /
// global addresses, row and column indices) cannot be hoisted out of the loops around the section.  Left to itself LLVM's LICM
// moves those few-instruction values to the top of the kernel and keeps them in VGPRs for its whole length -- ~60 of the render
// kernel's 128 registers (tools/asm/vgpr_liveness.py).  That is the right trade while the register count stays below an
// occupancy step (measured, round 4: taking every section's lane id this way cut the render kernel from 138 to 111 VGPRs and
// step_list from 208 to 154, both without reaching the next step, and cost 3-8 % steps/s in re-computed addresses), so only
// sections that run rarely, and whose hoisted values would push a kernel over a step, use this form.
__device__ __forceinline__ int pg_lane_opaque() {
    int l = (int)(threadIdx.x & 63u);
    __asm__ volatile("" : "+v"(l));
    __builtin_assume(l >= 0 && l < 64);
    return l;
}
#define PG_FOR_LANES(l) for (int l = PG_LANE_ID(), pg_once_ = 1; pg_once_; pg_once_ = 0)
#define PG_FOR_LANES_NOHOIST(l) for (int l = pg_lane_opaque(), pg_once_ = 1; pg_once_; pg_once_ = 0)
// LDS-DMA: the lane's dword goes from global memory to lds_base + 4 * lane (the LDS address is wave-uniform, M0) without passing
// through a VGPR; completion is counted by vmcnt like any other load
#define PG_DMA_DWORD(gptr, lds_base, l) \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr), (__attribute__((address_space(3))) void *)(lds_base), 4, 0, 0)
// (the builtin, not an asm string: the compiler's own wait-count bookkeeping sees it and knows that nothing fetched before it is still in
// flight -- behind an asm it assumed the opposite and put a full vmcnt(0) in front of later, unrelated register writes.  0x0F70 = vmcnt(0),
// expcnt and lgkmcnt untouched, in gfx9's encoding)
#define PG_DMA_JOIN()                          \
    do {                                       \
        __builtin_amdgcn_s_waitcnt(0x0F70);    \
        __asm__ volatile("" ::: "memory");     \
    } while (0)
// nothing is scheduled across this point: keeps the loads of the next unrolled iteration from being hoisted over this one's arithmetic
// (and their results from piling up in registers)
#define PG_BALLOT(l, pred)                              \
    ({                                                  \
        const int l = PG_LANE_ID();                     \
        (void)l;                                        \
        (uint64_t)__ballot((pred) ? 1 : 0);             \
    })
#define PG_SYNC()                                               \
    do {                                                        \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  \
        __builtin_amdgcn_wave_barrier();                        \
    } while (0)
#define PG_LANE_VAR(T, v) T v
#define PG_LANE_REF(T, v) T &v  // a lane variable passed by reference
#define PG_LV(v, l) v
#define PG_LANE_ARR(T, v, N) T v[N]
#define PG_LA(v, j, l) v[j]
#define PG_LANE_ARR_REF(T, v, N) T (&v)[N]
// 32-bit integer lane values only (a float would be value-converted, not bit-copied)
#define PG_READLANE(v, k)                                                                                      \
    ({                                                                                                         \
        static_assert(__is_integral(decltype(v)) && sizeof(v) == 4, "PG_READLANE takes a 32-bit integer");    \
        (decltype(v))__builtin_amdgcn_readlane((int)(v), (k));                                                 \
    })
#define PG_SHFL(v, l, src) ((decltype(v))__shfl((int)(v), (src), 64))  // ds_bpermute: the value lane `src` holds
// the OR of a 32-bit lane variable over the wave: a six-step butterfly, result in a scalar register
__device__ __forceinline__ uint32_t pg_wave_or(uint32_t v) {
    _Pragma("unroll") for (int o = 32; o >= 1; o >>= 1) v |= (uint32_t)__shfl_xor((int)v, o, 64);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
#define PG_WAVE_OR(v) pg_wave_or((uint32_t)(v))
PG_DEV uint32_t pg_atomic_or(uint32_t *p, uint32_t v) { return atomicOr(p, v); }
// value known to be wave-uniform: move it to an SGPR so branches on it are scalar branches
#define PG_UNIFORM_I(x) __builtin_amdgcn_readfirstlane((int)(x))
PG_DEV int pg_popc64(uint64_t m) { return __popcll(m); }
PG_DEV int pg_clz64(uint64_t m) { return m ? __clzll((long long)m) : 64; }
PG_DEV int pg_ctz64(uint64_t m) { return m ? (__ffsll((long long)m) - 1) : 64; }
PG_DEV uint64_t pg_brev64(uint64_t m) { return __brevll(m); }
PG_DEV double pg_sqrt(double x) { return __builtin_sqrt(x); }   // IEEE correctly rounded (no fast-math)
PG_DEV double pg_floor(double x) { return __builtin_floor(x); }
PG_DEV float pg_floorf(float x) { return __builtin_floorf(x); }
PG_DEV double pg_ceil(double x) { return __builtin_ceil(x); }
PG_DEV float pg_fabsf(float x) { return __builtin_fabsf(x); }
PG_DEV float pg_roundf(float x) { return __builtin_roundf(x); }  // half away from zero, as C roundf
PG_DEV double pg_fabs(double x) { return __builtin_fabs(x); }
// ROCm device libm (OCML), < 1 ulp in double; the caller narrows the result to float (bit-exactness notes); sin / cos / atan2 are restated in pg_math.h
PG_DEV double pg_pow(double x, double y) { return pow(x, y); }
PG_DEV uint32_t pg_perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }

#endif

// A wave-uniform pointer to data no kernel in flight writes (an env's header inside the render kernels, the asset tables): loads through it are
// scalar loads (constant address space), whatever stores the kernel has made before -- the compiler otherwise falls back to a vector load
// plus a v_readfirstlane per word as soon as any store precedes the load (69 header words per frame, round 6).
#if defined(PGAMD_WAVE_EMU)
#define PG_SCALAR_PTR(T, p) (static_cast<const T *>(p))
#else
#define PG_SCALAR_PTR(T, p) ((const __attribute__((address_space(4))) T *)(uintptr_t)(p))
#endif

// A value the optimizer must take as it is.  Choosing one of several adjacent struct fields with a ?: chain otherwise
// becomes ONE load at a computed offset, which pins the whole per-env state struct in scratch memory instead of registers.
#if defined(PGAMD_WAVE_EMU)
PG_DEV float pg_opaque_f(float v) { return v; }
PG_DEV int pg_opaque_i(int v) { return v; }
#else
PG_DEV float pg_opaque_f(float v) {
    __asm__ volatile("" : "+v"(v));
    return v;
}
PG_DEV int pg_opaque_i(int v) {
    __asm__ volatile("" : "+v"(v));
    return v;
}
#endif

struct alignas(16) pg_u4 {  // 16-byte move unit for staging copies
    uint32_t x, y, z, w;
};

// mask of lanes strictly below / at-or-below lane l
PG_DEV uint64_t pg_mask_lt(int l) { return l >= 64 ? ~0ull : ((1ull << l) - 1ull); }
// index of the highest set bit (-1 if none)
PG_DEV int pg_highest(uint64_t m) { return 63 - pg_clz64(m); }
