// pg_roomgen.h -- RoomGenerator (reference src/roomgen.cpp) on one wave, over the env's grid in LDS.
//
// The cellular-automaton update, the rooms (build_room / find_best_room) and expand_room do not depend on a visiting order
// and work on bit rows: the grid as one 64-bit row per lane, neighbours by shifts and the rows above / below.  find_path
// (the path IS the visiting order) keeps the reference's queue, 64 entries per round.  std::set<int> objects are byte
// flags (ascending cell index == the set's iteration order).
#pragma once
#include "pg_env.h"

namespace pgamd {

template <int CELLS>
struct RoomScratch {
    uint8_t f0[CELLS], f1[CELLS], f2[CELLS], f3[CELLS];  // flag / cell buffers
    uint16_t queue[CELLS + 64];                           // BFS queue / find_path's `expanded`
    alignas(8) uint16_t parents[CELLS + 64];              // find_path's `parents` (0xffff = -1); update_rows' bit rows
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

    // ---- bit rows: a cell set as one 64-bit row per lane (needs main_width, main_height <= 64) ------------------------------
    // `parents` holds the staging: two bit streams in cell order (one ballot per 64 cells) and the row exchange array, whose
    // entries 0 and h + 1 are the out-of-range rows.
    PG_DEV uint64_t *bit_stream(int k) { return reinterpret_cast<uint64_t *>(m.parents) + 60 * k; }
    PG_DEV uint64_t *bit_rows() { return reinterpret_cast<uint64_t *>(m.parents) + 120; }  // [h + 2]
    PG_DEV bool bit_rows_ok() {
        static_assert(sizeof(m.parents) >= (120 + 66) * 8, "bit rows do not fit");
        if (e.G.main_width > 64 || e.G.main_height > 64 || ncells() > 3648) {
            e.fail(PGE_ASSERT);
            return false;
        }
        return true;
    }
    PG_DEV uint64_t width_mask() const { return e.G.main_width >= 64 ? ~0ull : ((1ull << e.G.main_width) - 1ull); }
    // rows of the cells for which pa(idx) / pb(idx) hold
    template <class PA, class PB>
    PG_DEV void load_rows(PA pa, PB pb, PG_LANE_REF(uint64_t, ra), PG_LANE_REF(uint64_t, rb)) {
        const int n = ncells(), w = e.G.main_width, h = e.G.main_height;
        uint64_t *sa = bit_stream(0), *sb = bit_stream(1);
        int nch = 0;
        for (int base = 0; base < n; base += 64, nch++) {
            const uint64_t am = PG_BALLOT(l, (base + l) < n && pa(base + l));
            const uint64_t bm = PG_BALLOT(l, (base + l) < n && pb(base + l));
            PG_FOR_LANES(l) {
                if (l == 0) {
                    sa[nch] = am;
                    sb[nch] = bm;
                }
            }
        }
        PG_FOR_LANES(l) {
            if (l == 0) {
                sa[nch] = 0;
                sb[nch] = 0;
            }
        }
        PG_SYNC();
        const uint64_t wmask = width_mask();
        PG_FOR_LANES(l) {
            uint64_t a = 0, b = 0;
            if (l < h) {
                const int o = l * w, wd = o >> 6, sh = o & 63;
                a = sa[wd] >> sh;
                b = sb[wd] >> sh;
                if (sh) {
                    a |= sa[wd + 1] << (64 - sh);
                    b |= sb[wd + 1] << (64 - sh);
                }
            }
            PG_LV(ra, l) = a & wmask;
            PG_LV(rb, l) = b & wmask;
        }
        PG_SYNC();
    }
    // every lane's row into the exchange array, `edge` in the two out-of-range rows
    PG_DEV void publish_rows(PG_LANE_REF(uint64_t, r), uint64_t edge) {
        const int h = e.G.main_height;
        uint64_t *rows = bit_rows();
        PG_FOR_LANES(l) {
            if (l < h) rows[l + 1] = PG_LV(r, l);
            if (l == 0) {
                rows[0] = edge;
                rows[h + 1] = edge;
            }
        }
        PG_SYNC();
    }
    // f(idx, bit) for every cell, from the published rows
    template <class F>
    PG_DEV void for_cells_of_rows(F f) {
        const int n = ncells(), w = e.G.main_width;
        const uint64_t *rows = bit_rows();
        for (int base = 0; base < n; base += 64) {
            PG_FOR_LANES(l) {
                const int idx = base + l;
                if (idx < n) f(idx, (int)((rows[idx / w + 1] >> (idx % w)) & 1ull));
            }
        }
        PG_SYNC();
    }

    // update roomgen.cpp:22-37 (count_neighbors :3-20: the 3x3 block including the cell, out-of-range = out_of_bounds_object)
    // `iters` times; with keep_space, the flagged cells are put back to SPACE after every pass (caveflyer.cpp's
    // path-preserving smoothing).  The grid is two-valued here, so a pass is bit arithmetic on one row per lane: the 3x3
    // wall count is three bit-sliced 3-input adds per row triple and `>= 5` a few gates -- about 60 VALU operations for the
    // whole grid against 9 byte reads per cell.
    PG_DEV void update_rows(int iters, const uint8_t *keep_space) {
        if (!bit_rows_ok()) return;
        const int w = e.G.main_width, h = e.G.main_height;
        const uint64_t *rows = bit_rows();
        const bool oob_is_wall = e.G.out_of_bounds_object == WALL_OBJ;
        const uint64_t wmask = width_mask();
        const uint64_t oob_lo = oob_is_wall ? 1ull : 0ull, oob_hi = oob_is_wall ? (1ull << (w - 1)) : 0ull;
        PG_LANE_VAR(uint64_t, row);
        PG_LANE_VAR(uint64_t, kept);
        load_rows([&](int i) { return (int)e.s->grid[i] == WALL_OBJ; }, [&](int i) { return keep_space && keep_space[i] != 0; }, row, kept);
        for (int it = 0; it < iters; it++) {
            publish_rows(row, oob_is_wall ? wmask : 0ull);
            PG_FOR_LANES(l) {
                if (l < h) {
                    uint64_t s0[3], s1[3];
                    for (int r = 0; r < 3; r++) {
                        const uint64_t b = rows[l + r];
                        const uint64_t a = ((b << 1) & wmask) | oob_lo, c = (b >> 1) | oob_hi;
                        s0[r] = a ^ b ^ c;
                        s1[r] = (a & b) | (a & c) | (b & c);
                    }
                    const uint64_t l0 = s0[0] ^ s0[1] ^ s0[2];                                  // weight 1
                    const uint64_t l1 = (s0[0] & s0[1]) | (s0[0] & s0[2]) | (s0[1] & s0[2]);    // weight 2
                    const uint64_t t0 = s1[0] ^ s1[1] ^ s1[2];                                  // weight 2
                    const uint64_t t1 = (s1[0] & s1[1]) | (s1[0] & s1[2]) | (s1[1] & s1[2]);    // weight 4
                    const uint64_t u0 = t0 ^ l1, cy = t0 & l1;                                  // weight 2, carry
                    const uint64_t u1 = t1 ^ cy, u2 = t1 & cy;                                  // weight 4, 8
                    const uint64_t wall = u2 | (u1 & (u0 | l0));                                // count >= 5
                    PG_LV(row, l) = wall & wmask & ~PG_LV(kept, l);
                }
            }
            PG_SYNC();
        }
        publish_rows(row, 0ull);
        for_cells_of_rows([&](int idx, int bit) { e.s->grid[idx] = (typename E::cell_t)(bit ? WALL_OBJ : SPACE); });
    }

    // the cells of `seed` (a subset of `space`) and every cell of `space` reachable from them along the row: adding the seeds
    // to the row carries through each run of ones above a seed; the mirror image does the other side
    PG_DEV static uint64_t span_fill(uint64_t space, uint64_t seed) {
        const uint64_t up = (space & ~(space + seed)) | seed;
        const uint64_t rs = pg_brev64(space), rd = pg_brev64(seed);
        return up | pg_brev64((rs & ~(rs + rd)) | rd);
    }

    // find_best_room roomgen.cpp:128-148 (build_room :39-70) -> best room flags in f2; returns its size.
    // The reference walks every room with a queue from the first cell (in index order) no earlier room holds, merges it into
    // all_rooms and keeps a copy of the largest so far.  What lands in a room does not depend on the walk; its size is the
    // number of insertions, one per cell -- except that the start cell only joins through a neighbour, so a one-cell room
    // is empty and has size 0.  Here a room is flooded on bit rows: whole spans at a time along the rows, one row up and
    // down per pass, until a pass adds nothing.
    PG_DEV int find_best_room() {
        clear(m.f2);
        if (!bit_rows_ok()) return -1;
        const int h = e.G.main_height;
        const uint64_t *rows = bit_rows();
        PG_LANE_VAR(uint64_t, space);
        PG_LANE_VAR(uint64_t, none);
        PG_LANE_VAR(uint64_t, seen);  // all_rooms (one-cell rooms included: their start is never looked at again either way)
        PG_LANE_VAR(uint64_t, room);
        PG_LANE_VAR(uint64_t, best);
        load_rows([&](int i) { return (int)e.s->grid[i] == SPACE; }, [&](int) { return false; }, space, none);
        PG_FOR_LANES(l) {
            PG_LV(seen, l) = 0;
            PG_LV(best, l) = 0;
        }
        int best_size = -1;
        for (;;) {
            // first cell, in index order, that is SPACE and in no room yet
            const uint64_t open = PG_BALLOT(l, (PG_LV(space, l) & ~PG_LV(seen, l)) != 0);
            if (!open) break;
            const int y0 = pg_ctz64(open);
            PG_FOR_LANES(l) {
                const uint64_t rem = PG_LV(space, l) & ~PG_LV(seen, l);
                PG_LV(room, l) = l == y0 ? span_fill(PG_LV(space, l), rem & (~rem + 1ull)) : 0ull;
            }
            for (;;) {
                publish_rows(room, 0ull);
                const uint64_t grew = PG_BALLOT(l, ({
                                                    bool g = false;
                                                    if (l < h) {
                                                        const uint64_t have = PG_LV(room, l);
                                                        const uint64_t seed = (have | rows[l] | rows[l + 2]) & PG_LV(space, l);
                                                        if (seed & ~have) {
                                                            PG_LV(room, l) = span_fill(PG_LV(space, l), seed);
                                                            g = true;
                                                        }
                                                    }
                                                    g;
                                                }));
                PG_SYNC();
                if (!grew) break;
            }
            PG_LANE_VAR(int, cnt);
            PG_FOR_LANES(l) {
                PG_LV(cnt, l) = pg_popc64(PG_LV(room, l));
                PG_LV(seen, l) |= PG_LV(room, l);
            }
            int size = 0;
            for (int b = 0; b < 7; b++) size += pg_popc64(PG_BALLOT(l, (PG_LV(cnt, l) >> b) & 1)) << b;
            if (size < 2) size = 0;
            if (size > best_size) {
                best_size = size;
                PG_FOR_LANES(l) { PG_LV(best, l) = size ? PG_LV(room, l) : 0ull; }
            }
        }
        if (best_size > 0) {
            publish_rows(best, 0ull);
            for_cells_of_rows([&](int idx, int bit) { m.f2[idx] = (uint8_t)bit; });
        }
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

    // expand_room roomgen.cpp:150-182: `set` grows by n_loops rings of 8-connected SPACE cells.  On bit rows: a ring is the
    // 3x3 dilation of the last ring's SPACE cells, cut to SPACE cells outside the set.
    PG_DEV void expand_room(uint8_t *set, int n_loops) {
        if (!bit_rows_ok()) return;
        const int h = e.G.main_height;
        const uint64_t *rows = bit_rows();
        const uint64_t wmask = width_mask();
        PG_LANE_VAR(uint64_t, in_set);
        PG_LANE_VAR(uint64_t, space);
        PG_LANE_VAR(uint64_t, curr);
        load_rows([&](int i) { return set[i] != 0; }, [&](int i) { return (int)e.s->grid[i] == SPACE; }, in_set, space);
        PG_FOR_LANES(l) { PG_LV(curr, l) = PG_LV(in_set, l) & PG_LV(space, l); }
        for (int loop = 0; loop < n_loops; loop++) {
            publish_rows(curr, 0ull);
            PG_FOR_LANES(l) {
                if (l < h) {
                    const uint64_t up = rows[l], mid = rows[l + 1], dn = rows[l + 2];
                    const uint64_t v = up | dn;
                    const uint64_t ring = (v | (v << 1) | (v >> 1) | (mid << 1) | (mid >> 1)) & wmask & PG_LV(space, l) & ~PG_LV(in_set, l);
                    PG_LV(in_set, l) |= ring;
                    PG_LV(curr, l) = ring;
                }
            }
            PG_SYNC();
        }
        publish_rows(in_set, 0ull);
        for_cells_of_rows([&](int idx, int bit) { set[idx] = (uint8_t)bit; });
    }
};

}  // namespace pgamd
