// pg_mazegen.h -- MazeGen::generate_maze / place_objects (reference src/mazegen.cpp:112-187,292-306) on one wave.
//
// The reference runs Kruskal over std::set cell sets and erases the chosen wall from a std::vector.  Here the sets
// are one label per cell in LDS (they only carry connectivity; a union is a lane-parallel relabel), and the wall
// vector is an immutable array plus a 512-bit "alive" mask: "walls[n]" of the shrinking vector is the n-th alive
// wall, found with popcounts -- no element shifting.  The RNG stream (one randn(walls.size()) per wall) is the spec.
#pragma once
#include "pg_env.h"

namespace pgamd {

constexpr int MAZE_OFFSET = 1;  // reference src/mazegen.h:14

// LDS the generator works in, sized by the largest maze_dim the game asks for (MAXDIM) -- the arena of a step kernel that resets in place
// carries it, and 1280 bytes more or less of arena are a workgroup more or less per CU (profiles/r06_lds_granule.txt): chaser (<= 19) 3056
// bytes, heist (<= 23, with doors) 5488, maze (<= 31) 7952; one size for all (8328) held chaser's step kernel at 12 workgroups per CU
// instead of 16, maze's and heist's at 14.  DOORS: generate_maze_with_doors reuses the arrays as flag sets and cell lists over the
// (maze_dim + 2)^2 cells of the bordered grid: label -> s0 | s1 | curr, free_cells -> next, walls -> the cell list.
template <int MAXDIM, bool DOORS>
struct MazeScratchT {
    static constexpr int MAX_DIM = MAXDIM;
    static constexpr bool HAS_DOORS = DOORS;
    static constexpr int CELLS = MAXDIM * MAXDIM, ACELLS = (MAXDIM + 2) * (MAXDIM + 2);
    static constexpr int FLAG_STRIDE = (ACELLS + 15) & ~15;                          // bytes of one flag set
    static constexpr int NWALLS = 2 * ((MAXDIM - 1) / 2) * ((MAXDIM + 1) / 2);       // generate_maze's wall list (mazegen.cpp:140-154)
    static_assert(MAXDIM % 2 == 1 && NWALLS <= 512, "the alive mask of generate_maze has 512 bits");
    static constexpr int pg_max(int a, int b) { return a > b ? a : b; }
    uint16_t label[(pg_max(CELLS, DOORS ? 3 * FLAG_STRIDE / 2 : 0) + 1) & ~1];      // cell_sets_idxs (cell = maze_dim * y + x)
    uint16_t free_cells[(pg_max(CELLS, DOORS ? FLAG_STRIDE / 2 : 0) + 1) & ~1];     // 0xffff = taken (-1 in the reference)
    uint32_t walls[pg_max(NWALLS, DOORS ? (ACELLS + 1) / 2 : 0)];                    // x1 | y1<<5 | x2<<10 | y2<<15
    uint16_t mgrid[ACELLS + 3];  // MazeGen::grid (index y * array_dim + x); door / key ids exceed 255
};
typedef MazeScratchT<31, false> MazeScratch;  // maze_dim <= 31 (maze's memory mode), array_dim = maze_dim + 2
constexpr int MG_EXIT_OBJ = 52, MG_AGENT_OBJ = 53, MG_DOOR_OBJ = 200, MG_KEY_OBJ = 300;  // reference src/object-ids.h

template <class E, class S = MazeScratch>
struct MazeGenDev {
    E &e;
    S &m;
    int maze_dim, array_dim, num_free_cells;

    // (a maze_dim beyond the scratch's would write past its arrays: the games' choose_world_dim bound it, the emulation checks it)
    PG_DEV MazeGenDev(E &e_, S &m_, int maze_dim_)
        : e(e_), m(m_), maze_dim(maze_dim_ > S::MAX_DIM ? S::MAX_DIM : maze_dim_), array_dim(maze_dim + 2), num_free_cells(0) {
        if (maze_dim_ > S::MAX_DIM) e.fail(PGE_ASSERT);
    }

    PG_DEV int grid_at(int x, int y) const { return (int)m.mgrid[y * array_dim + x]; }

    PG_DEV void set_free_cell(int x, int y) {  // mazegen.cpp:26-34 (membership in free_cell_set == the cell already being SPACE)
        const int gi = (y + MAZE_OFFSET) * array_dim + x + MAZE_OFFSET;
        const bool was_free = m.mgrid[gi] == (uint16_t)SPACE;
        const int cell = maze_dim * y + x;
        PG_FOR_LANES(l) {
            if (l == 0) {
                m.mgrid[gi] = (uint16_t)SPACE;
                if (!was_free) m.free_cells[num_free_cells] = (uint16_t)cell;
            }
        }
        if (!was_free) num_free_cells += 1;
        PG_SYNC();
    }

    PG_DEV void generate_maze() {  // mazegen.cpp:112-187
        const int md = maze_dim, ad = array_dim;
        for (int base = 0; base < ad * ad; base += 64) {
            PG_FOR_LANES(l) {
                if (base + l < ad * ad) m.mgrid[base + l] = (uint16_t)WALL_OBJ;
            }
        }
        for (int base = 0; base < md * md; base += 64) {
            PG_FOR_LANES(l) {
                if (base + l < md * md) m.label[base + l] = (uint16_t)(base + l);
            }
        }
        PG_SYNC();
        PG_FOR_LANES(l) {
            if (l == 0) m.mgrid[MAZE_OFFSET * ad + MAZE_OFFSET] = 0;
        }
        num_free_cells = 0;
        // wall list in the reference's construction order (mazegen.cpp:140-154)
        const int nh = ((md - 1) / 2) * ((md + 1) / 2);  // i odd in [1, md-2], j even
        int nw = 0;
        {
            const int nj = (md + 1) / 2;
            for (int base = 0; base < nh; base += 64) {
                PG_FOR_LANES(l) {
                    const int k = base + l;
                    if (k < nh) {
                        const int i = 1 + 2 * (k / nj), j = 2 * (k % nj);
                        m.walls[k] = (uint32_t)(i - 1) | ((uint32_t)j << 5) | ((uint32_t)(i + 1) << 10) | ((uint32_t)j << 15);
                    }
                }
            }
            const int nj2 = (md - 1) / 2;  // j odd in [1, md-2]
            const int nv = ((md + 1) / 2) * nj2;
            for (int base = 0; base < nv; base += 64) {
                PG_FOR_LANES(l) {
                    const int k = base + l;
                    if (k < nv) {
                        const int i = 2 * (k / nj2), j = 1 + 2 * (k % nj2);
                        m.walls[nh + k] = (uint32_t)i | ((uint32_t)(j - 1) << 5) | ((uint32_t)i << 10) | ((uint32_t)(j + 1) << 15);
                    }
                }
            }
            nw = nh + nv;
        }
        PG_SYNC();
        uint64_t alive[8];
        for (int w = 0; w < 8; w++) {
            const int lo = w * 64;
            alive[w] = nw >= lo + 64 ? ~0ull : (nw > lo ? ((1ull << (nw - lo)) - 1ull) : 0ull);
        }
        // a lone wave pays for every instruction of this loop in full, so it is kept short: the draws come 64 at a time with
        // their moduli taken in the lanes, and only the live words of the mask are looked at
        PG_LANE_VAR(int, draws);
        for (int remaining = nw; remaining > 0; remaining--) {
            const int done = nw - remaining;
            if ((done & 63) == 0) {
                PG_LANE_VAR(uint32_t, u);
                const int cnt = remaining < 64 ? remaining : 64;
                e.rand_u32_lanes(cnt, u);
                PG_FOR_LANES(l) { PG_LV(draws, l) = l < cnt ? (int)(PG_LV(u, l) % (uint32_t)(remaining - l)) : 0; }  // randn(walls.size())
            }
            int n = PG_READLANE(draws, done & 63);
            // n-th alive wall in original order == walls[n] of the reference's shrinking vector: word by popcounts, bit by a
            // six-step select
            uint64_t word = 0;
            int wsel = 0;
            bool found = false;
            _Pragma("unroll") for (int w = 0; w < 8; w++) {
                if (!found && w * 64 < nw) {
                    const int c = pg_popc64(alive[w]);
                    if (n < c) {
                        word = alive[w];
                        wsel = w;
                        found = true;
                    } else {
                        n -= c;
                    }
                }
            }
            int pos = 0;
            _Pragma("unroll") for (int width = 32; width >= 1; width >>= 1) {
                const int c = pg_popc64((word >> pos) & ((1ull << width) - 1ull));
                if (n >= c) {
                    n -= c;
                    pos += width;
                }
            }
            _Pragma("unroll") for (int w = 0; w < 8; w++) {
                if (w == wsel) alive[w] &= ~(1ull << pos);
            }
            const uint32_t wall = (uint32_t)PG_UNIFORM_I(m.walls[wsel * 64 + pos]);
            const int x1 = (int)(wall & 31u), y1 = (int)((wall >> 5) & 31u), x2 = (int)((wall >> 10) & 31u), y2 = (int)((wall >> 15) & 31u);
            const int x0 = (x1 + x2) / 2, y0 = (y1 + y2) / 2;
            const int center = md * y0 + x0;
            const int g1 = (y1 + MAZE_OFFSET) * ad + x1 + MAZE_OFFSET, g0 = (y0 + MAZE_OFFSET) * ad + x0 + MAZE_OFFSET, g2 = (y2 + MAZE_OFFSET) * ad + x2 + MAZE_OFFSET;
            // one LDS trip for everything the decision needs
            const int s0_idx = (int)m.label[md * y1 + x1];
            const int s1_idx = (int)m.label[md * y2 + x2];
            const int o1 = (int)m.mgrid[g1], o2 = (int)m.mgrid[g2];
            const int s0u = PG_UNIFORM_I(s0_idx), s1u = PG_UNIFORM_I(s1_idx);
            // (the wall's own cell is still a wall: every wall comes up once, and nothing else opens it)
            const bool can_remove = s0u != s1u;
            if (can_remove) {
                // set_free_cell(x1, y1), (x0, y0), (x2, y2) mazegen.cpp:26-34 in one go (the centre is a wall, so it is new)
                const bool new1 = PG_UNIFORM_I(o1) != SPACE, new2 = PG_UNIFORM_I(o2) != SPACE;
                const int nf = num_free_cells;
                PG_FOR_LANES(l) {
                    if (l == 0) {
                        m.mgrid[g1] = (uint16_t)SPACE;
                        m.mgrid[g0] = (uint16_t)SPACE;
                        m.mgrid[g2] = (uint16_t)SPACE;
                        int k = nf;
                        if (new1) m.free_cells[k++] = (uint16_t)(md * y1 + x1);
                        m.free_cells[k++] = (uint16_t)center;
                        if (new2) m.free_cells[k++] = (uint16_t)(md * y2 + x2);
                    }
                }
                num_free_cells = nf + 1 + (new1 ? 1 : 0) + (new2 ? 1 : 0);
                for (int base = 0; base < md * md; base += 64) {
                    PG_FOR_LANES(l) {
                        const int i = base + l;
                        if (i < md * md && ((int)m.label[i] == s0u || i == center)) m.label[i] = (uint16_t)s1u;
                    }
                }
                PG_SYNC();
            }
        }
    }

    // ---- variants that post-process the spanning tree (mazegen.cpp:36-111,189-290) ------------------------------------
    // The reference's std::set<int> objects are membership flags here (ascending cell order == the set's iteration
    // order); the scratch arrays of generate_maze are free by then and are reused: label -> s0 | s1 | curr flags,
    // free_cells -> next flags, walls -> cell lists.  Serial wave-uniform code: a reset-time cost of a few thousand
    // LDS operations.
    PG_DEV uint8_t *flags_s0() { return reinterpret_cast<uint8_t *>(m.label); }
    PG_DEV uint8_t *flags_s1() { return reinterpret_cast<uint8_t *>(m.label) + S::FLAG_STRIDE; }
    PG_DEV uint8_t *flags_curr() { return reinterpret_cast<uint8_t *>(m.label) + 2 * S::FLAG_STRIDE; }
    PG_DEV uint8_t *flags_next() { return reinterpret_cast<uint8_t *>(m.free_cells); }
    PG_DEV uint16_t *cell_list() { return reinterpret_cast<uint16_t *>(m.walls); }

    PG_DEV int get_obj(int idx) const {  // mazegen.cpp:36-46
        const int x = idx % array_dim, y = idx / array_dim;
        if (x <= 0 || x >= array_dim - 1) return INVALID_OBJ;
        if (y <= 0 || y >= array_dim - 1) return INVALID_OBJ;
        return (int)m.mgrid[idx];
    }
    PG_DEV int get_obj_u(int idx) const { return PG_UNIFORM_I(get_obj(idx)); }  // wave-uniform idx only
    // get_neighbors mazegen.cpp:48-66 for an interior cell (wave-uniform idx): order (-1,0) (0,-1) (0,1) (1,0)
    PG_DEV int get_neighbors(int idx, int type, int (&out)[4]) const {
        const int cand[4] = {idx - 1, idx - array_dim, idx + array_dim, idx + 1};
        int n = 0;
        for (int k = 0; k < 4; k++)
            if (get_obj_u(cand[k]) == type) out[n++] = cand[k];
        return n;
    }
    PG_DEV int count_neighbors(int idx, int type) const {
        return (get_obj(idx - 1) == type) + (get_obj(idx - array_dim) == type) + (get_obj(idx + array_dim) == type) + (get_obj(idx + 1) == type);
    }
    PG_DEV void clear_flags(uint8_t *f, int nc) {
        for (int base = 0; base < nc; base += 64) {
            PG_FOR_LANES(l) {
                if (base + l < nc) f[base + l] = 0;
            }
        }
        PG_SYNC();
    }
    PG_DEV void or_flags(uint8_t *dst, const uint8_t *src, int nc) {
        for (int base = 0; base < nc; base += 64) {
            PG_FOR_LANES(l) {
                if (base + l < nc) dst[base + l] = dst[base + l] | src[base + l];
            }
        }
        PG_SYNC();
    }
    PG_DEV int expand_to_type(int type) {  // mazegen.cpp:68-99 on (s0, s1)
        const int nc = array_dim * array_dim;
        uint8_t *s0 = flags_s0(), *s1 = flags_s1(), *curr = flags_curr(), *next = flags_next();
        for (int base = 0; base < nc; base += 64) {
            PG_FOR_LANES(l) {
                if (base + l < nc) curr[base + l] = s0[base + l];
            }
        }
        PG_SYNC();
        for (;;) {
            bool any = false;
            for (int base = 0; base < nc; base += 64) any = any || PG_BALLOT(l, (base + l) < nc && curr[base + l] != 0) != 0;
            if (!any) break;
            clear_flags(next, nc);
            for (int base = 0; base < nc; base += 64) {
                uint64_t todo = PG_BALLOT(l, (base + l) < nc && curr[base + l] != 0);
                while (todo) {  // ascending order
                    const int elem = base + pg_ctz64(todo);
                    todo &= todo - 1;
                    int adj[4];
                    const int na = get_neighbors(elem, SPACE, adj);
                    for (int k = 0; k < 4; k++) {
                        if (k < na) {
                            const int j = adj[k];
                            if (!PG_UNIFORM_I(s0[j]) && !PG_UNIFORM_I(s1[j])) {
                                next[j] = 1;
                                s1[j] = 1;
                            }
                        }
                    }
                    int tgt[4];
                    if (get_neighbors(elem, type, tgt) > 0) return tgt[0];
                }
            }
            PG_SYNC();
            for (int base = 0; base < nc; base += 64) {
                PG_FOR_LANES(l) {
                    if (base + l < nc) curr[base + l] = next[base + l];
                }
            }
            PG_SYNC();
        }
        return -1;
    }
    // collects the cells with flag / predicate into cell_list() in ascending order; returns the count
    template <class Pred>
    PG_DEV int collect_cells(Pred pred) {
        const int nc = array_dim * array_dim;
        uint16_t *list = cell_list();
        int n = 0;
        for (int base = 0; base < nc; base += 64) {
            const uint64_t mask = PG_BALLOT(l, (base + l) < nc && pred(base + l));
            PG_FOR_LANES(l) {
                if ((mask >> l) & 1ull) list[n + pg_popc64(mask & pg_mask_lt(l))] = (uint16_t)(base + l);
            }
            n += pg_popc64(mask);
        }
        PG_SYNC();
        return n;
    }

    // mazegen.cpp:189-209: sequential (each opened wall changes later neighbour counts, and an opened cell further on is
    // itself visited).  The dead ends of a 64-cell chunk are found with one ballot over the grid as it stands; after every
    // opened wall the rest of the chunk is balloted again, so each visit sees what the reference's cell-by-cell walk sees.
    PG_DEV void generate_maze_no_dead_ends() {
        generate_maze();
        e.mark(10);
        const int nc = array_dim * array_dim;
        for (int base = 0; base < nc; base += 64) {
            int from = 0;
            for (;;) {
                const uint64_t dead = PG_BALLOT(l, l >= from && (base + l) < nc && get_obj(base + l) == SPACE && count_neighbors(base + l, SPACE) == 1 &&
                                                       count_neighbors(base + l, WALL_OBJ) > 0);
                if (!dead) break;
                const int lane = pg_ctz64(dead);
                const int i = base + lane;
                int wl[4];
                const int nw = get_neighbors(i, WALL_OBJ, wl);
                const int n = e.randn(nw);
                const int cell = n == 0 ? wl[0] : (n == 1 ? wl[1] : (n == 2 ? wl[2] : wl[3]));
                PG_FOR_LANES(l) {
                    if (l == 0) m.mgrid[cell] = (uint16_t)SPACE;
                }
                PG_SYNC();
                from = lane + 1;
                if (from >= 64) break;
            }
        }
    }

    PG_DEV void generate_maze_with_doors(int num_doors) {  // mazegen.cpp:211-290
        static_assert(S::HAS_DOORS, "the scratch is sized for the flag sets and the cell list");
        generate_maze();
        const int nc = array_dim * array_dim;
        uint16_t *list = cell_list();
        // forks: SPACE cells with more than two SPACE neighbours
        int nforks = collect_cells([&](int i) { return get_obj(i) == SPACE && count_neighbors(i, SPACE) > 2; });
        // RandGen::choose_n (reference src/randgen.cpp:49-69): chosen cells become doors right away (order is irrelevant)
        if (num_doors > nforks) {
            for (int i = 0; i < nforks; i++) m.mgrid[PG_UNIFORM_I(list[i])] = (uint16_t)MG_DOOR_OBJ;
            num_doors = nforks;
        } else {
            int nrem = nforks;
            for (int c = 0; c < num_doors; c++) {
                const int idx = e.randn(nrem);
                const int cell = PG_UNIFORM_I(list[idx]);
                PG_SYNC();
                {   // rem_elems.erase(begin + idx): shift the tail down by one (all reads before the writes)
                    PG_LANE_ARR(uint16_t, v, 10);
                    PG_FOR_LANES(l) {
                        for (int q = 0; q < 10; q++) {
                            const int k = idx + l + 64 * q;
                            PG_LA(v, q, l) = k < nrem - 1 ? list[k + 1] : (uint16_t)0;
                        }
                    }
                    PG_SYNC();
                    PG_FOR_LANES(l) {
                        for (int q = 0; q < 10; q++) {
                            const int k = idx + l + 64 * q;
                            if (k < nrem - 1) list[k] = PG_LA(v, q, l);
                        }
                    }
                }
                PG_SYNC();
                nrem--;
                m.mgrid[cell] = (uint16_t)MG_DOOR_OBJ;
            }
        }
        PG_SYNC();
        int agent_cell;
        {
            const int ns = collect_cells([&](int i) { return get_obj(i) == SPACE; });
            if (ns <= 0) {
                e.fail(PGE_ASSERT);
                return;
            }
            do {
                agent_cell = PG_UNIFORM_I(list[e.randn(ns)]);
            } while (count_neighbors(agent_cell, MG_DOOR_OBJ) > 0);
            m.mgrid[agent_cell] = (uint16_t)MG_AGENT_OBJ;
        }
        PG_SYNC();
        uint8_t *s0 = flags_s0(), *s1 = flags_s1();
        clear_flags(s0, nc);
        s0[agent_cell] = 1;
        PG_SYNC();
        for (int door_num = 0; door_num < num_doors + 1; door_num++) {
            clear_flags(s1, nc);
            int found_door = -1;
            if (door_num < num_doors) {
                found_door = expand_to_type(MG_DOOR_OBJ);
                if (found_door < 0) {
                    e.fail(PGE_ASSERT);
                    return;
                }
                m.mgrid[found_door] = (uint16_t)(MG_DOOR_OBJ + door_num + 1);
                PG_SYNC();
                or_flags(s0, s1, nc);
            }
            expand_to_type(-999);
            PG_SYNC();
            const int ns = collect_cells([&](int i) { return s1[i] != 0; });
            if (ns <= 0) {
                e.fail(PGE_ASSERT);
                return;
            }
            const int key_cell = PG_UNIFORM_I(list[e.randn(ns)]);
            m.mgrid[key_cell] = (uint16_t)(door_num == num_doors ? MG_EXIT_OBJ : (MG_KEY_OBJ + door_num + 1));
            PG_SYNC();
            or_flags(s0, s1, nc);
            if (found_door >= 0) s0[found_door] = 1;
            PG_SYNC();
        }
    }

    PG_DEV void place_objects(int start_obj, int num_objs) {  // mazegen.cpp:292-306
        for (int j = 0; j < num_objs; j++) {
            int k = e.randn(num_free_cells);
            while (m.free_cells[k] == 0xffffu || m.free_cells[k] == 0) k = e.randn(num_free_cells);
            const int coin_cell = (int)m.free_cells[k];
            PG_FOR_LANES(l) {
                if (l == 0) {
                    m.free_cells[k] = 0xffffu;
                    m.mgrid[(coin_cell / maze_dim + MAZE_OFFSET) * array_dim + coin_cell % maze_dim + MAZE_OFFSET] = (uint16_t)(start_obj + j);
                }
            }
            PG_SYNC();
        }
    }
};

}  // namespace pgamd
