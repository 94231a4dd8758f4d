// pg_roomgen.h -- RoomGenerator (reference src/roomgen.cpp) on one wave, over the env's grid in LDS.
//
// The cellular-automaton update and expand_room do not depend on a visiting order and run lane-parallel; build_room
// (what lands in a room depends on the start cell only, but the quirk that the start cell joins only through a
// neighbour is kept by walking the same queue) and find_path (the path IS the visiting order) walk their queues in
// wave-uniform code.  std::set<int> objects are byte flags (ascending cell index == the set's iteration order).
#pragma once
#include "pg_env.h"

namespace pgamd {

template <int CELLS>
struct RoomScratch {
    uint8_t f0[CELLS], f1[CELLS], f2[CELLS], f3[CELLS];  // flag / cell buffers
    uint16_t queue[CELLS + 64];                           // BFS queue / find_path's `expanded`
    uint16_t parents[CELLS + 64];                         // find_path's `parents` (0xffff = -1)
};

template <class E, int CELLS>
struct RoomGenDev {
    E &e;
    RoomScratch<CELLS> &m;
    PG_DEV RoomGenDev(E &e_, RoomScratch<CELLS> &m_) : e(e_), m(m_) {}

    PG_DEV int ncells() const { return e.G.main_width * e.G.main_height; }
    PG_DEV int to_grid_idx(int x, int y) const { return (0 <= y && y < e.G.main_height && 0 <= x && x < e.G.main_width) ? y * e.G.main_width + x : -2; }
    PG_DEV void clear(uint8_t *f) {
        const int n = ncells();
        for (int base = 0; base < n; base += 64) {
            PG_FOR_LANES(l) {
                if (base + l < n) f[base + l] = 0;
            }
        }
        PG_SYNC();
    }
    PG_DEV void copy(uint8_t *dst, const uint8_t *src) {
        const int n = ncells();
        for (int base = 0; base < n; base += 64) {
            PG_FOR_LANES(l) {
                if (base + l < n) dst[base + l] = src[base + l];
            }
        }
        PG_SYNC();
    }

    // update roomgen.cpp:22-37 (count_neighbors :3-20: the 3x3 block including the cell, out-of-range = out_of_bounds_object)
    PG_DEV void update() {
        const int n = ncells(), w = e.G.main_width, h = e.G.main_height;
        const int oob_is_wall = e.G.out_of_bounds_object == WALL_OBJ ? 1 : 0;
        for (int base = 0; base < n; base += 64) {
            PG_FOR_LANES(l) {
                const int idx = base + l;
                if (idx < n) {
                    const int x = idx % w, y = idx / w;
                    int cnt = 0;
                    for (int i = -1; i <= 1; i++)
                        for (int j = -1; j <= 1; j++) {
                            const int xx = x + i, yy = y + j;
                            if (xx < 0 || xx >= w || yy < 0 || yy >= h) cnt += oob_is_wall;
                            else cnt += (int)e.s->grid[yy * w + xx] == WALL_OBJ;
                        }
                    m.f0[idx] = (uint8_t)(cnt >= 5 ? WALL_OBJ : SPACE);
                }
            }
        }
        PG_SYNC();
        for (int base = 0; base < n; base += 64) {
            PG_FOR_LANES(l) {
                if (base + l < n) e.s->grid[base + l] = (typename E::cell_t)m.f0[base + l];
            }
        }
        PG_SYNC();
    }

    // build_room roomgen.cpp:39-70 into `room` (flags, set by OR: cells already flagged are not entered); returns the number of
    // cells inserted.  What lands in a room and how many insertions happen do not depend on the visiting order (the quirk
    // that the start cell joins only through a neighbour is kept: it is not flagged up front), so the queue is worked off
    // 64 cells at a time, one lane per cell, neighbours claimed with an atomic OR on the flag's word -- a reset's critical
    // path is the serial instruction count of one wave, and the one-cell-at-a-time walk was a third of jumper's.
    PG_DEV int build_room(int idx, uint8_t *room) {
        const int w = e.G.main_width;
        if ((int)e.s->grid[idx] != SPACE) return 0;
        int head = 0, tail = 1, size = 0;
        PG_FOR_LANES(l) {
            if (l == 0) m.queue[0] = (uint16_t)idx;
        }
        PG_SYNC();
        while (head < tail) {
            const int cnt = (tail - head) < 64 ? (tail - head) : 64;
            PG_LANE_VAR(uint32_t, won);  // bit k: this lane's cell claimed its k-th neighbour
            PG_FOR_LANES(l) {
                uint32_t wbits = 0;
                if (l < cnt) {
                    const int curr = (int)m.queue[head + l];
                    const int x = curr % w, y = curr / w;
                    const int nb[4] = {to_grid_idx(x - 1, y), to_grid_idx(x, y - 1), to_grid_idx(x, y + 1), to_grid_idx(x + 1, y)};
                    for (int k = 0; k < 4; k++) {
                        const int nx = nb[k];
                        if (nx >= 0 && (int)e.s->grid[nx] == SPACE) {
                            const uintptr_t addr = (uintptr_t)(room + nx);
                            const uint32_t bit = 1u << (8 * (int)(addr & 3));
                            if (!(pg_atomic_or((uint32_t *)(addr & ~(uintptr_t)3), bit) & bit)) wbits |= 1u << k;
                        }
                    }
                }
                PG_LV(won, l) = wbits;
            }
            PG_SYNC();
            int t = tail;
            for (int k = 0; k < 4; k++) {
                const uint64_t mk = PG_BALLOT(l, (PG_LV(won, l) >> k) & 1u);
                PG_FOR_LANES(l) {
                    if ((PG_LV(won, l) >> k) & 1u) {
                        const int curr = (int)m.queue[head + l];
                        const int x = curr % w, y = curr / w;
                        const int nx = k == 0 ? to_grid_idx(x - 1, y) : (k == 1 ? to_grid_idx(x, y - 1) : (k == 2 ? to_grid_idx(x, y + 1) : to_grid_idx(x + 1, y)));
                        m.queue[t + pg_popc64(mk & pg_mask_lt(l))] = (uint16_t)nx;
                    }
                }
                t += pg_popc64(mk);
            }
            size += t - tail;
            head += cnt;
            tail = t;
            PG_SYNC();
        }
        return size;
    }

    // find_best_room roomgen.cpp:128-148 -> best room flags in f2 (all_rooms f0; f1 is not used any more); returns its size.
    // The reference builds every room into a fresh set, merges it into all_rooms and keeps a copy of the largest so far; the
    // same rooms, sizes and first-largest choice come out of claiming every room directly in all_rooms (a start cell is
    // skipped iff some room holds it, as before) and building the winner once more, alone, at the end.
    PG_DEV int find_best_room() {
        const int n = ncells();
        clear(m.f0);
        clear(m.f2);
        int best_size = -1, best_start = -1;
        for (int base = 0; base < n; base += 64) {
            // cells of this chunk that are SPACE and in no room yet (rooms found while walking the chunk are re-checked)
            uint64_t cand = PG_BALLOT(l, (base + l) < n && (int)e.s->grid[base + l] == SPACE);
            while (cand) {
                const int i = base + pg_ctz64(cand);
                cand &= cand - 1;
                if (PG_UNIFORM_I(m.f0[i])) continue;
                const int sz = build_room(i, m.f0);
                if (sz > best_size) {
                    best_size = sz;
                    best_start = i;
                }
            }
        }
        if (best_start >= 0) build_room(best_start, m.f2);
        return best_size;
    }

    // find_path roomgen.cpp:72-126 -> path membership flags in `path`; returns the path length.
    // The reference pops one cell at a time and pushes its free neighbours (left, up, down, right); the path is the parent
    // chain of dst, so the queue order IS the result.  The same queue comes out of taking the next <= 64 queued cells at
    // once, one lane each: a cell wanted by several lanes goes to the lowest lane (the earlier pop), and the new entries
    // are appended in (lane, direction) order.  `covered` holds 0 = free, 1 = queued, lane + 2 = tentatively claimed in the
    // round under way.  (One cell per iteration was 0.3 ms of jumper's 1 ms level generation: a chain of dependent LDS trips.)
    PG_DEV int find_path(int src, int dst, uint8_t *path, uint8_t *covered) {
        const int w = e.G.main_width;
        clear(covered);
        clear(path);
        if ((int)e.s->grid[src] != SPACE) return 0;
        PG_FOR_LANES(l) {
            if (l == 0) {
                m.queue[0] = (uint16_t)src;
                m.parents[0] = 0xffffu;
            }
        }
        PG_SYNC();
        int head = 0, tail = 1, found_at = -1;
        while (head < tail && found_at < 0) {
            const int cnt = (tail - head) < 64 ? (tail - head) : 64;
            {
                const uint64_t at = PG_BALLOT(l, l < cnt && (int)m.queue[head + l] == dst);
                if (at) {  // dst is popped in this round: nothing queued from here on can be its ancestor
                    found_at = head + pg_ctz64(at);
                    break;
                }
            }
            if (tail + 4 * cnt > CELLS + 64) {
                e.fail(PGE_ASSERT);
                return 0;
            }
            PG_LANE_ARR(int, nb, 4);  // this lane's free neighbours (-1: none)
            PG_FOR_LANES(l) {
                for (int k = 0; k < 4; k++) PG_LA(nb, k, l) = -1;
                if (l < cnt) {
                    const int curr = (int)m.queue[head + l];
                    const int x = curr % w, y = curr / w;
                    const int c4[4] = {to_grid_idx(x - 1, y), to_grid_idx(x, y - 1), to_grid_idx(x, y + 1), to_grid_idx(x + 1, y)};
                    for (int k = 0; k < 4; k++) {
                        const int nx = c4[k];
                        if (nx >= 0 && covered[nx] == 0 && (int)e.s->grid[nx] == SPACE) PG_LA(nb, k, l) = nx;
                    }
                }
            }
            PG_SYNC();
            // lowest lane wins a contested cell: plain stores race, so lanes keep lowering the entry until nobody has to
            for (;;) {
                const uint64_t wrote = PG_BALLOT(l, ({
                                                     bool wr = false;
                                                     for (int k = 0; k < 4; k++) {
                                                         const int nx = PG_LA(nb, k, l);
                                                         if (nx >= 0) {
                                                             const int cur = (int)covered[nx];
                                                             if (cur == 0 || cur > l + 2) {
                                                                 covered[nx] = (uint8_t)(l + 2);
                                                                 wr = true;
                                                             }
                                                         }
                                                     }
                                                     wr;
                                                 }));
                PG_SYNC();
                if (!wrote) break;
            }
            PG_LANE_VAR(uint32_t, win);
            PG_FOR_LANES(l) {
                uint32_t wb = 0;
                for (int k = 0; k < 4; k++) {
                    const int nx = PG_LA(nb, k, l);
                    if (nx >= 0 && (int)covered[nx] == l + 2) wb |= 1u << k;
                }
                PG_LV(win, l) = wb;
            }
            PG_SYNC();
            uint64_t mk[4];
            for (int k = 0; k < 4; k++) mk[k] = PG_BALLOT(l, (PG_LV(win, l) >> k) & 1u);
            PG_FOR_LANES(l) {
                const uint64_t below = pg_mask_lt(l);
                int pos = tail + pg_popc64(mk[0] & below) + pg_popc64(mk[1] & below) + pg_popc64(mk[2] & below) + pg_popc64(mk[3] & below);
                for (int k = 0; k < 4; k++) {
                    if ((PG_LV(win, l) >> k) & 1u) {
                        const int nx = PG_LA(nb, k, l);
                        m.queue[pos] = (uint16_t)nx;
                        m.parents[pos] = (uint16_t)(head + l);
                        covered[nx] = 1;
                        pos++;
                    }
                }
            }
            tail += pg_popc64(mk[0]) + pg_popc64(mk[1]) + pg_popc64(mk[2]) + pg_popc64(mk[3]);
            head += cnt;
            PG_SYNC();
        }
        int len = 0;
        if (found_at >= 0) {
            int k = found_at;
            while (k != 0xffff) {
                const int cell = PG_UNIFORM_I(m.queue[k]);
                PG_FOR_LANES(l) {
                    if (l == 0) path[cell] = 1;
                }
                len++;
                k = PG_UNIFORM_I(m.parents[k]);
            }
        }
        PG_SYNC();
        return len;
    }

    // expand_room roomgen.cpp:150-182: `set` grows by n_loops rings of 8-connected SPACE cells (curr / next: scratch flags)
    PG_DEV void expand_room(uint8_t *set, int n_loops, uint8_t *curr, uint8_t *next) {
        const int n = ncells(), w = e.G.main_width, h = e.G.main_height;
        copy(curr, set);
        for (int loop = 0; loop < n_loops; loop++) {
            for (int base = 0; base < n; base += 64) {
                PG_FOR_LANES(l) {
                    const int idx = base + l;
                    if (idx < n) {
                        bool add = false;
                        if (!set[idx] && (int)e.s->grid[idx] == SPACE) {
                            const int x = idx % w, y = idx / w;
                            for (int i = -1; i <= 1; i++)
                                for (int j = -1; j <= 1; j++) {
                                    const int xx = x + i, yy = y + j;
                                    if ((i != 0 || j != 0) && xx >= 0 && xx < w && yy >= 0 && yy < h) {
                                        const int c = yy * w + xx;
                                        add = add || (curr[c] && (int)e.s->grid[c] == SPACE);
                                    }
                                }
                        }
                        next[idx] = add ? 1 : 0;
                    }
                }
            }
            PG_SYNC();
            for (int base = 0; base < n; base += 64) {
                PG_FOR_LANES(l) {
                    if (base + l < n) {
                        set[base + l] = set[base + l] | next[base + l];
                        curr[base + l] = next[base + l];
                    }
                }
            }
            PG_SYNC();
        }
    }
};

}  // namespace pgamd
