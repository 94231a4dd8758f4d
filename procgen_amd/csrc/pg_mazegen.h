// pg_mazegen.h -- MazeGen::generate_maze / place_objects (reference src/mazegen.cpp:112-187,292-306) on one wave.
//
// The reference runs Kruskal over std::set cell sets and erases the chosen wall from a std::vector.  Here the sets
// are one label per cell in LDS (they only carry connectivity; a union is a lane-parallel relabel), and the wall
// vector is an immutable array plus a 512-bit "alive" mask: "walls[n]" of the shrinking vector is the n-th alive
// wall, found with popcounts -- no element shifting.  The RNG stream (one randn(walls.size()) per wall) is the spec.
#pragma once
#include "pg_env.h"

namespace pgamd {

constexpr int MG_MAX_DIM = 33;  // maze_dim <= 31 (memory mode), array_dim = maze_dim + 2
constexpr int MAZE_OFFSET = 1;  // reference src/mazegen.h:14

struct MazeScratch {
    uint16_t label[1024];       // cell_sets_idxs (cell = maze_dim * y + x)
    uint16_t free_cells[1024];  // 0xffff = taken (-1 in the reference)
    uint32_t walls[512];        // x1 | y1<<5 | x2<<10 | y2<<15
    uint8_t mgrid[MG_MAX_DIM * MG_MAX_DIM + 7];  // MazeGen::grid (index y * array_dim + x), values < 256
};

template <class E>
struct MazeGenDev {
    E &e;
    MazeScratch &m;
    int maze_dim, array_dim, num_free_cells;

    PG_DEV MazeGenDev(E &e_, MazeScratch &m_, int maze_dim_) : e(e_), m(m_), maze_dim(maze_dim_), array_dim(maze_dim_ + 2), num_free_cells(0) {}

    PG_DEV int grid_at(int x, int y) const { return (int)m.mgrid[y * array_dim + x]; }

    PG_DEV void set_free_cell(int x, int y) {  // mazegen.cpp:26-34 (membership in free_cell_set == the cell already being SPACE)
        const int gi = (y + MAZE_OFFSET) * array_dim + x + MAZE_OFFSET;
        const bool was_free = m.mgrid[gi] == (uint8_t)SPACE;
        const int cell = maze_dim * y + x;
        PG_FOR_LANES(l) {
            if (l == 0) {
                m.mgrid[gi] = (uint8_t)SPACE;
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
                if (base + l < ad * ad) m.mgrid[base + l] = (uint8_t)WALL_OBJ;
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
        for (int remaining = nw; remaining > 0; remaining--) {
            int n = e.randn(remaining);
            // n-th alive wall in original order == walls[n] of the reference's shrinking vector
            int widx = 0;
            _Pragma("unroll") for (int w = 0; w < 8; w++) {
                const int c = pg_popc64(alive[w]);
                if (n >= 0) {
                    if (n < c) {
                        uint64_t word = alive[w];
                        for (int k = 0; k < n; k++) word &= word - 1;
                        const int b = pg_ctz64(word);
                        widx = w * 64 + b;
                        alive[w] &= ~(1ull << b);
                        n = -1;
                    } else {
                        n -= c;
                    }
                }
            }
            const uint32_t wall = m.walls[widx];
            const int x1 = (int)(wall & 31u), y1 = (int)((wall >> 5) & 31u), x2 = (int)((wall >> 10) & 31u), y2 = (int)((wall >> 15) & 31u);
            const int s0_idx = (int)m.label[md * y1 + x1];
            const int s1_idx = (int)m.label[md * y2 + x2];
            const int x0 = (x1 + x2) / 2, y0 = (y1 + y2) / 2;
            const int center = md * y0 + x0;
            const bool can_remove = grid_at(x0 + MAZE_OFFSET, y0 + MAZE_OFFSET) == WALL_OBJ && s0_idx != s1_idx;
            if (can_remove) {
                set_free_cell(x1, y1);
                set_free_cell(x0, y0);
                set_free_cell(x2, y2);
                for (int base = 0; base < md * md; base += 64) {
                    PG_FOR_LANES(l) {
                        const int i = base + l;
                        if (i < md * md && ((int)m.label[i] == s0_idx || i == center)) m.label[i] = (uint16_t)s1_idx;
                    }
                }
                PG_SYNC();
            }
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
                    m.mgrid[(coin_cell / maze_dim + MAZE_OFFSET) * array_dim + coin_cell % maze_dim + MAZE_OFFSET] = (uint8_t)(start_obj + j);
                }
            }
            PG_SYNC();
        }
    }
};

}  // namespace pgamd
