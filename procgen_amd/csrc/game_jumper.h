// game_jumper.h -- Jumper rules as a policy for Env<> / Renderer<> (reference procgen/src/games/jumper.cpp).
// A double-jumping bunny in a cave: the level is a MazeGen maze blown up 3x, randomised, smoothed by RoomGenerator's
// cellular automaton, reduced to the widened path from the agent to the carrot, then decorated with spikes; a
// compass (ellipse + cosmetic line + distance bar, and a translucent shadow while double-jumping) is painted over the
// frame.  The compass rect depends on the options alone.  In hard / memory mode with center_agent it is integer aligned
// and Qt draws it with the midpoint algorithm (pg_render.h exec_ellipse); in easy mode and without center_agent it is
// not, and Qt takes its path route (pg_qtpath.h): that ellipse is rasterised once per handle on the host into two 64-bit
// row masks per frame row (brush, pen), which the renderer applies.
#pragma once
#include <string.h>

#include "pg_game_defaults.h"
#include "pg_math.h"
#include "pg_mazegen.h"
#include "pg_qtpath.h"
#include "pg_roomgen.h"

namespace pgamd {

struct JumperScratch {
    union {
        MazeScratch maze;                 // first the maze ...
        RoomScratch<45 * 45> room;        // ... then the room generator (the maze is dead by then)
    };
};

struct Jumper : BagDefaults<Jumper> {
    static constexpr int GAME_ID = GAME_JUMPER;
    static constexpr const char *NAME = "jumper";
    typedef JumperScratch Scratch;
    static constexpr int MAX_CELLS = 45 * 45;  // jumper.cpp:201-217 (memory mode)
    static constexpr bool HAS_OVERLAY = true;
    static constexpr int RENDER_MIN_WAVES = 4;  // 132 -> 128 VGPRs with 12 B of scratch (the phase-counter address): four render waves per SIMD measured +13 % on the same box (31.2 -> 35.3 M, profiles/r05_rot_pool_ab.txt)
    static constexpr bool HAS_HUMAN_OVERLAY = true;  // the compass under render_human: antialiased path draws (pg_human.h, pg_aapath.h)
    static constexpr int ENT_CAP_T0 = 64, ENT_CAP_T1 = 128, ENT_CAP_T2 = 256;  // agent, goal, spikes (~10-40), <= 8 trails
    // the level generator's scratch dominates the arena: step kernels without it, resets in the reset kernel (pg_env.h GameSplit)
    static constexpr bool SPLIT_RESET = true;
    static constexpr int RESET_CAP = ENT_CAP_T0;
    template <class E>
    PG_DEV static int slots_needed_next_step(E &e) { return e.G.n_ents + 1 + 1; }

    static constexpr int GOAL = 1, SPIKE = 2, CAVEWALL = 6, CAVEWALL_TOP = 7, PLAYER_JUMP = 9, PLAYER_LEFT1 = 10, PLAYER_LEFT2 = 11, PLAYER_RIGHT1 = 12,
                         PLAYER_RIGHT2 = 13;
    static constexpr int MAZE_SCALE = 3, JUMP_COOLDOWN = 3, NUM_WALL_THEMES = 4;

#define JP_JUMP_COUNT(G) (G).gsi0
#define JP_JUMP_DELTA(G) (G).gsi1
#define JP_JUMP_TIME(G) (G).gsi2
#define JP_HAS_SUPPORT(G) (G).gsi3
#define JP_FACING_RIGHT(G) (G).gsi4
#define JP_WALL_THEME(G) (G).gsi5
#define JP_COMPASS_DIM(G) (G).gsf0

    PG_DEV static bool is_wall(int t) { return t == CAVEWALL || t == CAVEWALL_TOP; }
    PG_HOSTDEV static bool use_block_asset(int t) { return t == CAVEWALL || t == CAVEWALL_TOP; }  // jumper.cpp:107-109

    static void construct(EnvHdr &G) { construct_defaults(G); }  // jumper.cpp:41-44
    // jumper.cpp:201-231: what the distribution mode fixes
    PG_HOSTDEV static float mode_visibility(int dm) { return dm == EasyMode ? 12.f : 16.f; }
    PG_HOSTDEV static float mode_compass_dim(int dm) { return dm == EasyMode ? 3.f : 2.f; }
    PG_HOSTDEV static int mode_world_dim(int dm) { return dm == HardMode ? 40 : (dm == MemoryMode ? 45 : 20); }
    template <class E>
    PG_DEV static void choose_world_dim(E &e) {  // jumper.cpp:201-217, preceded by game_reset's prologue :219-231
        EnvHdr &G = e.G;
        const int dm = e.opt.distribution_mode;
        G.visibility = mode_visibility(dm);
        JP_COMPASS_DIM(G) = mode_compass_dim(dm);
        if (dm == MemoryMode) G.timeout = 2000;
        const int wd = mode_world_dim(dm);
        G.main_width = wd;
        G.main_height = wd;
    }
    // compass_rect (jumper.cpp:138) as get_abs_rect (BAG:803-805) yields it after prepare_for_drawing (BAG:819-838)
    PG_HOSTDEV static void compass_rect(float unit, float view_dim, float cd, double r[4]) {
        const float cxf = (float)((double)(view_dim - cd) - .25), cyf = (float).25;
        r[0] = (double)(cxf * unit);
        r[1] = (double)(cyf * unit);
        r[2] = (double)(cd * unit);
        r[3] = (double)(cd * unit);
    }
    // game tables: [0..7] the compass rect (4 doubles) the masks were made for, then per frame row the brush mask and the
    // pen mask (bit x = column x), 64 x 2 x 64 bits.  Empty when the rect is integer aligned (midpoint route) or in memory mode.
    static constexpr int TABLE_WORDS = 8 + RES_H * 4, NUM_TABLES = 4, HOST_TABLE_WORDS = NUM_TABLES * TABLE_WORDS;  // one table per (easy / hard mode, center_agent setting): per-env options
    struct MaskSink {
        uint64_t brush[RES_H], pen[RES_H];
        int cnt[RES_H], xa[RES_H];
        void cross(int y, int x) {  // second crossing of a row: fill between the two (QScanConverter, odd-even)
            if (cnt[y]++ == 0) {
                xa[y] = x;
                return;
            }
            int x0 = xa[y] < x ? xa[y] : x, x1 = xa[y] < x ? x : xa[y];
            if (x0 < 0) x0 = 0;
            if (x1 > RES_W) x1 = RES_W;
            for (int c = x0; c < x1; c++) brush[y] |= 1ull << c;
        }
        void pixel(int x, int y) { pen[y] |= 1ull << x; }
    };
    static int host_tables(const GameOptions &o, uint32_t *out, int max_words) {
        if (max_words < HOST_TABLE_WORDS) return 0;
        memset(out, 0, HOST_TABLE_WORDS * sizeof(uint32_t));
        // both center_agent settings x the two modes that may draw the compass on a fractional rect (memory mode draws none): the handle's
        // own options and those of envs restored from states saved under the others (set_state adopts a state's options per env)
        int any = 0;
        for (int k = 0; k < NUM_TABLES; k++) {
            GameOptions v = o;
            v.center_agent = (k & 1) != 0;
            v.distribution_mode = (k & 2) ? HardMode : EasyMode;
            any |= host_table_one(v, out + k * TABLE_WORDS, TABLE_WORDS);
        }
        return any ? HOST_TABLE_WORDS : 0;
    }
    static int host_table_one(const GameOptions &o, uint32_t *out, int max_words) {
        if (o.distribution_mode == MemoryMode || max_words < TABLE_WORDS) return 0;
        // BAG::prepare_for_drawing for a 64-pixel frame: jumper keeps BAG's choose_center and min_visibility = 0
        const float world = (float)mode_world_dim(o.distribution_mode);
        const float visibility = o.center_agent ? mode_visibility(o.distribution_mode) : world;
        const float raw_unit = 64 / visibility;
        const float unit = (float)((double)raw_unit * (64.0 / 64.0));
        const float view_dim = (float)(64.0 / (double)raw_unit);
        double r[4];
        compass_rect(unit, view_dim, mode_compass_dim(o.distribution_mode), r);
        if (qtpath::is_integer_rect(r[0], r[1], r[2], r[3])) return 0;
        MaskSink *m = new MaskSink();
        memset(m, 0, sizeof(*m));
        int top, bot;
        qtpath::fill_crossings(*m, r[0], r[1], r[2], r[3], RES_W, RES_H, top, bot);
        qtpath::stroke_ellipse(*m, r[0], r[1], r[2], r[3], RES_W, RES_H);
        memcpy(out, r, sizeof(r));
        for (int y = 0; y < RES_H; y++) {
            out[8 + y * 4 + 0] = (uint32_t)m->brush[y];
            out[8 + y * 4 + 1] = (uint32_t)(m->brush[y] >> 32);
            out[8 + y * 4 + 2] = (uint32_t)m->pen[y];
            out[8 + y * 4 + 3] = (uint32_t)(m->pen[y] >> 32);
        }
        delete m;
        return TABLE_WORDS;
    }
    template <class E>
    PG_DEV static bool is_blocked(E &e, int src_type, int target, bool) {  // jumper.cpp:108-115
        return target == WALL_OBJ || target == e.G.out_of_bounds_object || (src_type == PLAYER && is_wall(target));
    }
    template <class E>
    PG_DEV static void handle_agent_collision(E &e, int obj) {  // jumper.cpp:82-92
        const int t = e.etype(obj);
        if (t == GOAL) {
            e.G.reward += 10.0f;
            e.G.level_complete = 1;
            e.G.done = 1;
        } else if (t == SPIKE) {
            e.G.done = 1;
        }
    }
    template <class E>
    PG_DEV static void update_agent_velocity(E &e) {  // jumper.cpp:94-100
        EnvHdr &G = e.G;
        const int ag = G.agent;
        const float v_scale = 1.0f;
        e.evx(ag) = (1 - G.mixrate) * e.evx(ag) + G.mixrate * G.maxspeed * G.action_vx * v_scale;
        if (G.action_vy != 0) e.evy(ag) = G.maxspeed * G.action_vy * 2;
    }
    template <class E>
    PG_DEV static void set_action_xy(E &e, int move_action) {  // jumper.cpp:398-430
        EnvHdr &G = e.G;
        G.action_vx = (float)(move_action / 3 - 1);
        G.action_vy = (float)((move_action % 3) - 1);
        if (G.action_vy < 0) G.action_vy = 0;
        if (G.action_vx > 0) JP_FACING_RIGHT(G) = 1;
        if (G.action_vx < 0) JP_FACING_RIGHT(G) = 0;
        const int ag = G.agent;
        const float ax = e.ex(ag), ay = e.ey(ag), arx = e.erx(ag), ary = e.ery(ag);
        const float by = (float)((double)ay - ((double)ary + .01));
        const int o1 = e.get_obj_from_floats((float)((double)ax - ((double)arx - .01)), by);
        const int o2 = e.get_obj_from_floats((float)((double)ax + ((double)arx - .01)), by);
        JP_JUMP_DELTA(G) = 0;
        const bool s1 = is_wall(o1) || o1 == G.out_of_bounds_object;
        const bool s2 = is_wall(o2) || o2 == G.out_of_bounds_object;
        JP_HAS_SUPPORT(G) = (s1 || s2) ? 1 : 0;
        if (JP_HAS_SUPPORT(G)) JP_JUMP_COUNT(G) = 2;
        if (G.action_vy == 1 && JP_JUMP_COUNT(G) > 0 && (G.cur_time - JP_JUMP_TIME(G) > JUMP_COOLDOWN)) {
            JP_JUMP_COUNT(G) -= 1;
            JP_JUMP_DELTA(G) = -1;
        } else {
            G.action_vy = 0;
        }
        if (G.action_vy > 0) JP_JUMP_TIME(G) = G.cur_time;
        G.action_vrot = 0;
    }

    // ---- level generation helpers (jumper.cpp:180-199) ----
    template <class E>
    PG_DEV static bool is_space_on_ground(E &e, int x, int y) {
        if (e.get_obj(x, y) != SPACE) return false;
        if (e.get_obj(x, y + 1) != SPACE) return false;
        const int below = e.get_obj(x, y - 1);
        return below == CAVEWALL || below == e.G.out_of_bounds_object;
    }
    template <class E>
    PG_DEV static bool is_left_wall(E &e, int x, int y) { return e.get_obj(x, y) == CAVEWALL && e.get_obj(x + 1, y) == SPACE; }
    template <class E>
    PG_DEV static bool is_right_wall(E &e, int x, int y) { return e.get_obj(x, y) == CAVEWALL && e.get_obj(x - 1, y) == SPACE; }
    // k-th cell index (ascending) satisfying an index predicate; -1 if fewer
    template <class E, class Pred>
    PG_DEV static int nth_index(E &e, int k, Pred pred, int *total = nullptr) {
        const int nc = e.G.main_width * e.G.main_height;
        int found = -1, seen = 0;
        for (int base = 0; base < nc; base += 64) {
            uint64_t m = PG_BALLOT(l, (base + l) < nc && pred(base + l));
            const int c = pg_popc64(m);
            if (found < 0 && k >= seen && k < seen + c) {
                uint64_t mm = m;
                for (int q = 0; q < k - seen; q++) mm &= mm - 1;
                found = base + pg_ctz64(mm);
                if (!total) return found;
            }
            seen += c;
        }
        if (total) *total = seen;
        return found;
    }

    template <class E>
    PG_DEV static void game_reset(E &e) {  // jumper.cpp:219-396
        e.bag_game_reset();
        EnvHdr &G = e.G;
        typedef typename E::cell_t cell_t;
        const int n = G.main_width * G.main_height, w = G.main_width, h = G.main_height;
        const int dm = e.opt.distribution_mode;
        G.out_of_bounds_object = WALL_OBJ;
        JP_WALL_THEME(G) = e.randn(NUM_WALL_THEMES);
        JP_JUMP_COUNT(G) = 0;
        JP_JUMP_DELTA(G) = 0;
        JP_JUMP_TIME(G) = 0;
        JP_HAS_SUPPORT(G) = 0;
        JP_FACING_RIGHT(G) = 1;
        const int maze_dim = w / MAZE_SCALE;
        PG_SYNC();
        e.mark(0);  // bag_game_reset, reseed
        {
            MazeGenDev<E> mg(e, e.s->scratch.maze, maze_dim);
            mg.generate_maze_no_dead_ends();
            e.mark(1);  // maze
            for (int base = 0; base < n; base += 64) {  // one draw per cell; walls of the 3x maze are solid with p = .8, corridors with p = .2
                PG_LANE_VAR(uint32_t, u);
                e.rand_u32_lanes((n - base) < 64 ? (n - base) : 64, u);
                PG_FOR_LANES(l) {
                    const int i = base + l;
                    if (i < n) {
                        const int obj = mg.grid_at((i % w) / MAZE_SCALE + 1, (i / w) / MAZE_SCALE + 1);
                        const float prob = obj == WALL_OBJ ? (float).8 : (float).2;
                        const float r01 = (float)((double)PG_LV(u, l) / 4294967296.0);
                        e.s->grid[i] = (cell_t)(r01 < prob ? WALL_OBJ : SPACE);
                    }
                }
            }
            PG_SYNC();
        }
        G.grid_dirty = 1;
        e.mark(2);  // 3x blow-up with per-cell draws
        RoomGenDev<E, 45 * 45> rg(e, e.s->scratch.room);
        auto &m = e.s->scratch.room;
        rg.update_rows(2, nullptr);
        e.fill_elem(0, 0, w, 1, CAVEWALL);  // border cells
        e.fill_elem(0, h - 1, w, 1, CAVEWALL);
        e.fill_elem(0, 0, 1, h, CAVEWALL);
        e.fill_elem(w - 1, 0, 1, h, CAVEWALL);
        e.mark(3);  // cellular automaton x 2, border
        const int best = rg.find_best_room();  // flags in f2
        e.mark(4);  // find_best_room
        if (best <= 0) {
            e.fail(PGE_ASSERT);
            return;
        }
        for (int base = 0; base < n; base += 64) {
            PG_FOR_LANES(l) {
                if (base + l < n) e.s->grid[base + l] = (cell_t)(m.f2[base + l] ? SPACE : CAVEWALL);
            }
        }
        PG_SYNC();
        const int nfree = e.count_cells([](int v) { return v == SPACE; });
        const int goal_cell = e.nth_cell(e.randn(nfree), [](int v) { return v == SPACE; });  // choose_one(free_cells)
        int ncand = 0;
        nth_index(e, -1, [&](int i) { return is_space_on_ground(e, i % w, i / w); }, &ncand);
        if (ncand <= 0) {
            e.fail(PGE_ASSERT);
            return;
        }
        const int agent_cell = nth_index(e, e.randn(ncand), [&](int i) { return is_space_on_ground(e, i % w, i / w); });
        e.mark(5);  // goal / agent cells
        rg.find_path(agent_cell, goal_cell, m.f3, m.f0);
        e.mark(6);  // find_path
        if (dm != MemoryMode) {  // should_prune
            rg.copy(m.f1, m.f3);
            rg.expand_room(m.f1, 4);
            for (int base = 0; base < n; base += 64) {
                PG_FOR_LANES(l) {
                    if (base + l < n) e.s->grid[base + l] = (cell_t)(m.f1[base + l] ? SPACE : CAVEWALL);
                }
            }
            PG_SYNC();
        }
        e.mark(7);  // expand_room
        e.add_entity((float)((goal_cell % w) + .5), (float)((goal_cell / w) + .5), 0, 0, (float).5, GOAL);  // spawn_entity_at_idx BAG:577-583
        const float spike_prob = dm == MemoryMode ? 0.0f : (float).2;
        // spikes: a placed spike can only disqualify later cells, so candidates are balloted per chunk and re-checked at the visit
        for (int base = 0; base < n; base += 64) {
            uint64_t cand = PG_BALLOT(l, ({
                                          const int i = base + l;
                                          (i < n) && is_space_on_ground(e, i % w, i / w) && is_space_on_ground(e, i % w - 1, i / w) && is_space_on_ground(e, i % w + 1, i / w);
                                      }));
            while (cand) {
                const int i = base + pg_ctz64(cand);
                cand &= cand - 1;
                const int x = i % w, y = i / w;
                if (is_space_on_ground(e, x, y) && is_space_on_ground(e, x - 1, y) && is_space_on_ground(e, x + 1, y)) {
                    if (e.rand01() < spike_prob) {
                        e.s->grid[i] = (cell_t)SPIKE;
                        PG_SYNC();
                    }
                }
            }
        }
        e.mark(8);  // spikes
        // long vertical walls are broken up; an opened cell can create new walls further on, so the chunk's candidates
        // are re-balloted after every change
        for (int base = 0; base < n; base += 64) {
            int from = 0;  // lanes below `from` are done
            for (;;) {
                const uint64_t lm = PG_BALLOT(l, ({
                                                  const int i = base + l;
                                                  (l >= from) && (i < n) && is_left_wall(e, i % w, i / w) && is_left_wall(e, i % w, i / w + 1) && is_left_wall(e, i % w, i / w + 2);
                                              }));
                const uint64_t rm = PG_BALLOT(l, ({
                                                  const int i = base + l;
                                                  (l >= from) && (i < n) && is_right_wall(e, i % w, i / w) && is_right_wall(e, i % w, i / w + 1) && is_right_wall(e, i % w, i / w + 2);
                                              }));
                if ((lm | rm) == 0) break;
                const int l0 = pg_ctz64(lm | rm);
                const int i = base + l0, x = i % w, y = i / w;
                if ((lm >> l0) & 1ull) {
                    e.set_obj(x, y + e.randn(3), SPACE);
                    PG_SYNC();
                }
                // the right-wall test of the same cell sees the grid after the left-wall change
                if (is_right_wall(e, x, y) && is_right_wall(e, x, y + 1) && is_right_wall(e, x, y + 2)) {
                    e.set_obj(x, y + e.randn(3), SPACE);
                    PG_SYNC();
                }
                from = l0 + 1;
                if (from >= 64) break;
            }
        }
        e.mark(9);  // vertical walls
        const int ag = G.agent;
        e.ex(ag) = (float)((agent_cell % w) + .5);
        e.ey(ag) = (agent_cell / w) + e.ery(ag);
        for (int base = 0; base < n; base += 64) {  // spike cells become entities, ascending index
            uint64_t sm = PG_BALLOT(l, (base + l) < n && (int)e.s->grid[base + l] == SPIKE);
            while (sm) {
                const int i = base + pg_ctz64(sm);
                sm &= sm - 1;
                e.s->grid[i] = (cell_t)SPACE;
                const float spike_ry = 0.4f, spike_rx = 0.23f;
                e.add_entity_rxy((float)((i % w) + .5), (i / w) + spike_ry, 0, 0, spike_rx, spike_ry, SPIKE);
            }
        }
        PG_SYNC();
        for (int base = 0; base < n; base += 64) {  // is_top_wall :189-191 (order-independent)
            PG_LANE_VAR(uint32_t, top);
            PG_FOR_LANES(l) {
                const int i = base + l;
                PG_LV(top, l) = (i < n && e.get_obj(i % w, i / w) == CAVEWALL && e.get_obj(i % w, i / w + 1) == SPACE) ? 1u : 0u;
            }
            PG_SYNC();
            PG_FOR_LANES(l) {
                if (PG_LV(top, l)) e.s->grid[base + l] = (cell_t)CAVEWALL_TOP;
            }
            PG_SYNC();
        }
        e.erx(ag) = 0.254f;
        e.ery(ag) = 0.4f;
        G.out_of_bounds_object = CAVEWALL;
        G.grid_dirty = 1;
        PG_SYNC();
    }

    template <class E>
    PG_DEV static void game_step(E &e) {  // jumper.cpp:432-449
        e.bag_game_step();
        EnvHdr &G = e.G;
        const int ag = G.agent;
        if (G.action_vx > 0) e.set_flag(ag, MF_REFLECTED, false);
        if (G.action_vx < 0) e.set_flag(ag, MF_REFLECTED, true);
        if (pg_fabs((double)e.evx(ag)) + pg_fabs((double)e.evy(ag)) > .05) {
            const int t = e.add_entity_rxy(e.ex(ag), (float)((double)e.ey(ag) - (double)e.ery(ag) * .5), 0, 0.01f, 0.3f, 0.2f, TRAIL);
            e.ei(EF_EXPIRE_TIME, t) = 8;
            e.ef(EF_ALPHA, t) = (float).5;
        }
        if (e.evy(ag) > -2) e.evy(ag) -= 0.15f;
        PG_SYNC();
    }

    template <class E>
    PG_DEV static int image_for_type(E &e, int type) {  // jumper.cpp:117-132
        if (type == PLAYER) {
            const EnvHdr &G = e.G;
            if ((double)pg_fabsf(e.evx(G.agent)) < .01 && G.action_vx == 0 && JP_HAS_SUPPORT(G)) return PLAYER;
            if (JP_FACING_RIGHT(G)) return (G.cur_time / 5 % 2 == 0 || !JP_HAS_SUPPORT(G)) ? PLAYER_RIGHT1 : PLAYER_RIGHT2;
            return (G.cur_time / 5 % 2 == 0 || !JP_HAS_SUPPORT(G)) ? PLAYER_LEFT1 : PLAYER_LEFT2;
        }
        return type < 0 ? -type : type;
    }
    template <class E>
    PG_DEV static int theme_for_grid_obj(E &e, int type) { return is_wall(type) ? JP_WALL_THEME(e.G) : 0; }  // jumper.cpp:102-107

    // draw_compass jumper.cpp:134-169 on the 512-pixel antialiased frame (render_human): the same calls, other Qt routes --
    // drawEllipse fills through the gray raster and strokes with the antialiased cosmetic stroker (pen width 1), the needle is a
    // rasterizeLine with a square cap (pen width 2 * compass_dim pixels), the bar an antialiased fillRect
    template <class R>
    PG_DEV static void draw_overlay_human(R &r) {
        const EnvHdr &G = r.G;
        if (r.opt.distribution_mode == MemoryMode) return;
        const int n = G.n_ents;
        int goal = -1;
        for (int c = 0; c < ((n + 63) >> 6) && goal < 0; c++) {
            const uint64_t m = PG_BALLOT(l, ((c << 6) + l) < n && r.etype((c << 6) + l) == GOAL);
            if (m) goal = (c << 6) + pg_highest(m);
        }
        if (goal < 0) {
            r.fail(PGE_ASSERT);
            return;
        }
        const int ag = G.agent;
        const float cd = JP_COMPASS_DIM(G);
        const float cxf = (float)((double)(G.view_dim - cd) - .25);
        double crv[4];
        compass_rect(G.unit, G.view_dim, cd, crv);
        const RectD cr_ = {crv[0], crv[1], crv[2], crv[3]};
        r.aa_fill_ellipse(cr_, 0xffa8a69eu);    // set_pen_brush_color(p, clock_color): brush ...
        r.aa_stroke_ellipse(cr_, 0xffa8a69eu);  // ... and a pen of width 1
        const float pen_thickness = (float)(R::FRAME_W / (256.0 / (double)cd));
        const int thickness = (int)pen_thickness;  // set_pen_brush_color(QPainter &, QColor, int thickness)
        const float cx = (float)(cr_.x + cr_.w / 2);
        const float cy = (float)(cr_.y + cr_.h / 2);
        const float cr = (float)(cr_.w / 2 * .95);
        const float theta = (float)pg_atan2_d((double)(r.ey(goal) - r.ey(ag)), (double)(r.ex(goal) - r.ex(ag)));  // get_theta BAG:233-238
        if (thickness < 2) {  // (a pen of width <= 1 would be the cosmetic stroker's line: never the case at 512 pixels, 2 * compass_dim >= 4)
            r.fail(PGE_UNSUPPORTED_DRAW);
            return;
        }
        r.aa_wide_line((int)cx, (int)cy, (int)((double)cx + (double)cr * pg_cos_d((double)theta)), (int)((double)cy - (double)cr * pg_sin_d((double)theta)), (double)thickness, 0xfffcba03u);
        const float ddx = r.ex(ag) - r.ex(goal), ddy = r.ey(ag) - r.ey(goal);
        const float dist = (float)pg_sqrt((double)(ddx * ddx + ddy * ddy));  // get_distance BAG:133-143
        const float dist_pct = (float)((double)dist / (G.main_width * pg_sqrt(2.0)));
        const float bar_thickness = cd / 8;
        r.exec_fill(r.get_abs_rect(cxf, (float)(.25 + (double)cd), cd * dist_pct, bar_thickness), 0xfffcba03u);
        if (JP_JUMP_DELTA(G) < 0 && !JP_HAS_SUPPORT(G)) {
            const RectD r1 = r.get_screen_rect(r.ex(ag) - r.erx(ag), r.ey(ag) + r.ery(ag), 2 * r.erx(ag), 2 * r.ery(ag), 0);
            const RectD sh = {(double)(int)r1.x, (double)(int)(r1.y + r1.h * (5.0 / 6)), (double)(int)r1.w, (double)(int)(r1.h / 3)};  // QRect(int, int, int, int)
            r.aa_fill_ellipse(sh, 0x78787878u);  // QColor(255, 255, 255, 120), premultiplied; Qt::NoPen
        }
    }

    // draw_compass jumper.cpp:134-169 (skipped in memory mode :171-178)
    template <class R>
    PG_DEV static void draw_overlay(R &r) {
        const EnvHdr &G = r.G;
        if (r.opt.distribution_mode == MemoryMode) return;
        const int n = G.n_ents;
        int goal = -1;
        for (int c = 0; c < ((n + 63) >> 6) && goal < 0; c++) {
            const uint64_t m = PG_BALLOT(l, ((c << 6) + l) < n && r.etype((c << 6) + l) == GOAL);
            if (m) goal = (c << 6) + pg_highest(m);
        }
        if (goal < 0) {
            r.fail(PGE_ASSERT);
            return;
        }
        const int ag = G.agent;
        const float cd = JP_COMPASS_DIM(G);
        const float cxf = (float)((double)(G.view_dim - cd) - .25);
        double crv[4];
        compass_rect(G.unit, G.view_dim, cd, crv);
        const RectD cr_ = {crv[0], crv[1], crv[2], crv[3]};
        if (qtpath::is_integer_rect(cr_.x, cr_.y, cr_.w, cr_.h)) {
            r.exec_ellipse((int)cr_.x, (int)cr_.y, (int)cr_.w, (int)cr_.h, true, 0xffa8a69eu, 0xffa8a69eu);
        } else {
            // Qt's path route: the handle's precomputed row masks -- valid for exactly this rect
            const uint32_t *t = r.d.game_tables;
            bool same = false;
            for (int k = 0; k < NUM_TABLES && t != nullptr && !same; k++) {  // the handle's tables (host_tables)
                const double *tr = (const double *)(r.d.game_tables + k * TABLE_WORDS);
                same = tr[0] == cr_.x && tr[1] == cr_.y && tr[2] == cr_.w && tr[3] == cr_.h && tr[2] != 0;
                if (same) t = r.d.game_tables + k * TABLE_WORDS;
            }
            if (!same) {
                r.fail(PGE_ASSERT);
                return;
            }
            r.exec_row_masks(t + 8, (int)cr_.y - 1, (int)(cr_.y + cr_.h) + 2, 0xffa8a69eu, 0xffa8a69eu);
        }
        const float cx = (float)(cr_.x + cr_.w / 2);
        const float cy = (float)(cr_.y + cr_.h / 2);
        const float cr = (float)(cr_.w / 2 * .95);
        const float theta = (float)pg_atan2_d((double)(r.ey(goal) - r.ey(ag)), (double)(r.ex(goal) - r.ex(ag)));  // get_theta BAG:233-238
        r.exec_line((int)cx, (int)cy, (int)((double)cx + (double)cr * pg_cos_d((double)theta)), (int)((double)cy - (double)cr * pg_sin_d((double)theta)), 0xfffcba03u);
        const float ddx = r.ex(ag) - r.ex(goal), ddy = r.ey(ag) - r.ey(goal);
        const float dist = (float)pg_sqrt((double)(ddx * ddx + ddy * ddy));  // get_distance BAG:133-143
        const float dist_pct = (float)((double)dist / (G.main_width * pg_sqrt(2.0)));
        const float bar_thickness = cd / 8;
        r.exec_fill(r.get_abs_rect(cxf, (float)(.25 + (double)cd), cd * dist_pct, bar_thickness), 0xfffcba03u);
        if (JP_JUMP_DELTA(G) < 0 && !JP_HAS_SUPPORT(G)) {
            const RectD r1 = r.get_screen_rect(r.ex(ag) - r.erx(ag), r.ey(ag) + r.ery(ag), 2 * r.erx(ag), 2 * r.ery(ag), 0);
            r.exec_ellipse((int)r1.x, (int)(r1.y + r1.h * (5.0 / 6)), (int)r1.w, (int)(r1.h / 3), false, 0u, 0x78787878u);  // QColor(255,255,255,120), premultiplied
        }
    }
};

}  // namespace pgamd
