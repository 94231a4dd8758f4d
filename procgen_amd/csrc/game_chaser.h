// game_chaser.h -- Chaser rules as a policy for Env<> / Renderer<> (reference procgen/src/games/chaser.cpp).
// Pac-Man on a 13x13 maze without dead ends: small orbs are grid cells with an id above 255 (u16 cells) drawn as
// solid green squares, large orbs and enemies are entities; enemies hatch from eggs and choose a direction at every
// junction from the Manhattan distance to the agent and the step's random integer.
#pragma once
#include "pg_game_defaults.h"
#include "pg_mazegen.h"

namespace pgamd {

struct Chaser : BagDefaults<Chaser> {
    static constexpr int GAME_ID = GAME_CHASER;
    static constexpr bool DISPLAY_LIST = true;  // frames are drawn prep -> raster (pg_prep.h)
    // pg_env.h GameParSmart: blocking / reflecting targets of this game are wall types only, never a smart entity's type,
    // and the hooks basic_step_object calls touch nothing but the moving object
    static constexpr bool PAR_SMART = true;
    PG_DEV static bool par_smart_type_ok(int t) { return t == PLAYER || t == ENEMY; }
    static constexpr const char *NAME = "chaser";
    typedef uint16_t cell_t;  // MARKER = 1001, ORB = 1002
    typedef MazeScratchT<19, false> Scratch;  // maze_dim 11 / 13 / 19 (chaser.cpp:141-156)
    static constexpr int MAX_CELLS = 19 * 19;  // chaser.cpp:137-160 (extreme mode)
    static constexpr bool HAS_GRID_FILLS = true;
    static constexpr int ENT_CAP_T0 = 24, ENT_CAP_T1 = 32, ENT_CAP_T2 = 48;  // agent + <= 5 large orbs + <= 5 eggs / enemies (+ hatching)
    template <class E>
    PG_DEV static int slots_needed_next_step(E &e) { return 2 * e.G.n_ents + 2; }

    static constexpr int LARGE_ORB = 2, ENEMY_WEAK = 3, ENEMY_EGG = 4, MAZE_WALL = 5, ENEMY = 6, ENEMY2 = 7, ENEMY3 = 8, MARKER = 1001, ORB = 1002;
    PG_HOSTDEV static bool use_block_asset(int t) { return t == MAZE_WALL; }  // chaser.cpp:74-76: generated as a rect texture (use_generated_assets)
    static constexpr float ORB_REWARD = 0.04f, ORB_DIM = 0.3f;
    static constexpr int EAT_TIMEOUT = 75, EGG_TIMEOUT = 50;
    static constexpr int INVALID_IDX = -2;

#define CH_EAT_TIME(G) (G).gsi0
#define CH_TOTAL_ORBS(G) (G).gsi1
#define CH_ORBS_COLLECTED(G) (G).gsi2
#define CH_MAZE_DIM(G) (G).gsi3
#define CH_TOTAL_ENEMIES(G) (G).gsi4
#define CH_NUM_FREE(G) (G).gsi5  // free_cells.size(): cells that are not walls

    PG_DEV static bool center_agent(const GameOptions &) { return false; }  // chaser.cpp:170
    PG_DEV static bool can_eat_enemies(const EnvHdr &G) { return G.cur_time - CH_EAT_TIME(G) < EAT_TIMEOUT; }

    static void construct(EnvHdr &G) {  // chaser.cpp:38-49
        construct_defaults(G);
        G.mixrate = 1;
        G.maxspeed = (float).5;
        G.has_useful_vel_info = 0;
    }
    template <class E>
    PG_DEV static void choose_world_dim(E &e) {  // chaser.cpp:132-135 (maze_dim is set first by game_reset :141-156)
        EnvHdr &G = e.G;
        const int dm = e.opt.distribution_mode;
        if (dm == EasyMode) { CH_MAZE_DIM(G) = 11; CH_TOTAL_ENEMIES(G) = 3; }
        else if (dm == HardMode) { CH_MAZE_DIM(G) = 13; CH_TOTAL_ENEMIES(G) = 3; }
        else if (dm == ExtremeMode) { CH_MAZE_DIM(G) = 19; CH_TOTAL_ENEMIES(G) = 5; }
        else e.fail(PGE_ASSERT);
        G.main_width = CH_MAZE_DIM(G);
        G.main_height = CH_MAZE_DIM(G);
    }
    template <class E>
    PG_DEV static bool is_blocked(E &e, int, int target, bool) {  // chaser.cpp:90-95
        return target == MAZE_WALL || target == WALL_OBJ || target == e.G.out_of_bounds_object;
    }
    template <class E>
    PG_DEV static void update_agent_velocity(E &e) {  // chaser.cpp:79-88
        EnvHdr &G = e.G;
        const int ag = G.agent;
        float vx = e.evx(ag), vy = e.evy(ag);
        if (G.action_vx != 0) vx = G.maxspeed * G.action_vx;
        if (G.action_vy != 0) vy = G.maxspeed * G.action_vy;
        e.evx(ag) = (float)(sign_d((double)vx) * (double)G.maxspeed);
        e.evy(ag) = (float)(sign_d((double)vy) * (double)G.maxspeed);
    }
    template <class E>
    PG_DEV static void handle_agent_collision(E &e, int obj) {  // chaser.cpp:121-135
        EnvHdr &G = e.G;
        const int t = e.etype(obj);
        if (t == LARGE_ORB) {
            CH_EAT_TIME(G) = G.cur_time;
            G.reward += ORB_REWARD;
            e.set_flag(obj, MF_WILL_ERASE, true);
        } else if (t == ENEMY) {
            if (can_eat_enemies(G)) e.set_flag(obj, MF_WILL_ERASE, true);
            else G.done = 1;
        }
    }

    template <class E>
    PG_DEV static void spawn_egg(E &e, int enemy_cell) {  // chaser.cpp:270-273
        const int md = CH_MAZE_DIM(e.G);
        const int egg = e.add_entity((float)((enemy_cell % md) + .5), (float)((enemy_cell / md) + .5), 0, 0, (float).5, ENEMY_EGG);
        e.ef(EF_HEALTH, egg) = (float)EGG_TIMEOUT;
    }
    // k-th cell (ascending grid index) whose value satisfies pred; -1 if there are fewer
    template <class E, class Pred>
    PG_DEV static int nth_cell(E &e, int k, Pred pred) {
        const int nc = e.G.main_width * e.G.main_height;
        for (int base = 0; base < nc; base += 64) {
            uint64_t m = PG_BALLOT(l, (base + l) < nc && pred((int)e.s->grid[base + l]));
            const int c = pg_popc64(m);
            if (k < c) {
                for (int q = 0; q < k; q++) m &= m - 1;
                return base + pg_ctz64(m);
            }
            k -= c;
        }
        return -1;
    }
    template <class E, class Pred>
    PG_DEV static int count_cells(E &e, Pred pred) {
        const int nc = e.G.main_width * e.G.main_height;
        int n = 0;
        for (int base = 0; base < nc; base += 64) n += pg_popc64(PG_BALLOT(l, (base + l) < nc && pred((int)e.s->grid[base + l])));
        return n;
    }
    // RandGen::simple_choose (reference src/randgen.cpp:71-88), k <= 6
    template <class E>
    PG_DEV static void simple_choose(E &e, int n, int k, int (&chosen)[6]) {
        if (!(k <= n) || k > 6) {
            e.fail(PGE_ASSERT);
            return;
        }
        for (int i = 0; i < k; i++) {
            int next = e.randn(n);
            bool dup;
            do {
                dup = false;
                for (int q = 0; q < 6; q++) dup = dup || (q < i && chosen[q] == next);
                if (dup) next = e.randn(n);
            } while (dup);
            chosen[i] = next;
        }
    }

    template <class E>
    PG_DEV static void game_reset(E &e) {  // chaser.cpp:137-264
        e.bag_game_reset();
        EnvHdr &G = e.G;
        const int dm = e.opt.distribution_mode;
        const int extra_orb_sign = dm == EasyMode ? 0 : (dm == HardMode ? -1 : 1);
        const int md = CH_MAZE_DIM(G);
        int ag = G.agent;
        e.erx(ag) = (float).5;
        e.ery(ag) = (float).5;
        CH_EAT_TIME(G) = -1 * EAT_TIMEOUT;
        PG_SYNC();
        // chaser.cpp:159-162: the generator is made ONCE per Game, at its first reset, with that reset's maze_dim -- the dimension of the
        // mode the handle was made with -- and kept.  An env restored from a state of another mode (round 5: the mode is adopted per env)
        // therefore generates mazes of the handle's dimension and copies the corner its own world shows; a smaller generator than the
        // world is the reference's `fassert(contains(x, y))` (grid.h:41).
        const int hm = e.d.opt.distribution_mode;
        const int md_gen = hm == EasyMode ? 11 : (hm == HardMode ? 13 : 19);
        if (md_gen < md) {
            e.fail(PGE_ASSERT);
            return;
        }
        MazeGenDev<E, Scratch> mg(e, e.s->scratch, md_gen);
        mg.generate_maze_no_dead_ends();
        const int extra_quad = e.randn(4);
        // grid <- maze (walls become MAZE_WALL); maze cell (i, j) is grid index j * md + i
        for (int j = 0; j < md; j++) {
            PG_FOR_LANES(l) {
                if (l < md) {
                    const int obj = mg.grid_at(l + MAZE_OFFSET, j + MAZE_OFFSET);
                    e.s->grid[j * md + l] = (cell_t)(obj == WALL_OBJ ? MAZE_WALL : obj);
                }
            }
        }
        G.grid_dirty = 1;
        PG_SYNC();
        // one or two large orbs per quadrant; a quadrant's cells are listed column by column (i outer, j inner)
        for (int q = 0; q < 4; q++) {
            const int num_orbs = 1 + (q == extra_quad ? extra_orb_sign : 0);
            const int qi = q >> 1, qj = q & 1;
            int nq = 0;
            for (int i = 0; i < md; i++) {
                if (((double)i >= md / 2.0 ? 1 : 0) != qi) continue;
                nq += pg_popc64(PG_BALLOT(l, l < md && (((double)l >= md / 2.0 ? 1 : 0) == qj) && mg.grid_at(i + MAZE_OFFSET, l + MAZE_OFFSET) == SPACE));
            }
            int sel[6] = {0, 0, 0, 0, 0, 0};
            simple_choose(e, nq, num_orbs, sel);
            for (int k = 0; k < num_orbs; k++) {
                int want = k == 0 ? sel[0] : sel[1];
                int cell = -1;
                for (int i = 0; i < md && cell < 0; i++) {
                    if (((double)i >= md / 2.0 ? 1 : 0) != qi) continue;
                    uint64_t m = PG_BALLOT(l, l < md && (((double)l >= md / 2.0 ? 1 : 0) == qj) && mg.grid_at(i + MAZE_OFFSET, l + MAZE_OFFSET) == SPACE);
                    const int c = pg_popc64(m);
                    if (want < c) {
                        for (int t = 0; t < want; t++) m &= m - 1;
                        cell = pg_ctz64(m) * md + i;
                    } else {
                        want -= c;
                    }
                }
                if (cell < 0) {
                    e.fail(PGE_ASSERT);
                    return;
                }
                e.add_entity((float)((cell % G.main_width) + .5), (float)((cell / G.main_width) + .5), 0, 0, 0.4f, LARGE_ORB);  // spawn_entity_at_idx BAG:577-583
                e.s->grid[cell] = (cell_t)MARKER;
                PG_SYNC();
            }
        }
        // agent start and enemy eggs among the remaining SPACE cells (get_cells_with_type: ascending index)
        const int nfree = count_cells(e, [](int v) { return v == SPACE; });
        int sel[6] = {0, 0, 0, 0, 0, 0};
        simple_choose(e, nfree, 1 + CH_TOTAL_ENEMIES(G), sel);
        int cells[6];
        for (int k = 0; k < 6; k++) cells[k] = k <= CH_TOTAL_ENEMIES(G) ? nth_cell(e, sel[k], [](int v) { return v == SPACE; }) : 0;
        ag = G.agent;
        e.ex(ag) = (float)((cells[0] % md) + .5);
        e.ey(ag) = (float)((cells[0] / md) + .5);
        for (int i = 0; i < CH_TOTAL_ENEMIES(G); i++) spawn_egg(e, cells[i + 1]);
        PG_SYNC();
        // every cell that was SPACE (including the start and the egg cells) gets an orb; the large orbs' markers turn into SPACE
        const int nc = md * md;
        for (int base = 0; base < nc; base += 64) {
            PG_FOR_LANES(l) {
                if (base + l < nc) {
                    const int v = (int)e.s->grid[base + l];
                    if (v == SPACE) e.s->grid[base + l] = (cell_t)ORB;
                    else if (v == MARKER) e.s->grid[base + l] = (cell_t)SPACE;
                }
            }
        }
        PG_SYNC();
        CH_TOTAL_ORBS(G) = nfree;
        CH_ORBS_COLLECTED(G) = 0;
        CH_NUM_FREE(G) = count_cells(e, [](int v) { return v != MAZE_WALL; });
        G.grid_dirty = 1;
    }

    PG_DEV static int to_grid_idx(const EnvHdr &G, int x, int y) {  // BAG:187-192
        if (!(0 <= y && y < G.main_height && 0 <= x && x < G.main_width)) return INVALID_IDX;
        return y * G.main_width + x;
    }

    template <class E>
    PG_DEV static void game_step(E &e) {  // chaser.cpp:301-390
        e.bag_game_step();
        EnvHdr &G = e.G;
        PG_SYNC();
        int num_enemies = 0;
        const bool can_eat = can_eat_enemies(G);
        const float default_enemy_speed = (float).5;
        const float vscale = can_eat ? (float)((double)default_enemy_speed * .5) : default_enemy_speed;
        const int mw = G.main_width;
        const int n0 = G.n_ents;
        for (int c = (n0 - 1) >> 6; c >= 0; c--) {
            uint64_t m = PG_BALLOT(l, ((c << 6) + l) < n0 && (e.etype((c << 6) + l) == ENEMY_EGG || e.etype((c << 6) + l) == ENEMY));
            while (m) {
                const int j = (c << 6) + pg_highest(m);
                m &= ~(1ull << (j & 63));
                num_enemies++;
                if (e.etype(j) == ENEMY_EGG) {
                    e.ef(EF_HEALTH, j) -= 1;
                    if (e.ef(EF_HEALTH, j) == 0) {
                        e.set_flag(j, MF_WILL_ERASE, true);
                        const int en = e.add_entity(e.ex(j), e.ey(j), 0, 0, (float).5, ENEMY);  // spawn_child BAG:225-231
                        e.set_flag(en, MF_SMART_STEP, true);
                    }
                } else {
                    const int ag = G.agent;
                    const float x = (float)((double)e.ex(j) - .5);
                    const float y = (float)((double)e.ey(j) - .5);
                    const int dist_scale = can_eat ? -1 : 1;
                    const int enemy_idx = to_grid_idx(G, (int)x, (int)y);
                    const int agent_idx = to_grid_idx(G, (int)e.ex(ag), (int)e.ey(ag));
                    const bool is_at_junction = (double)(pg_fabsf(x - pg_roundf(x)) + pg_fabsf(y - pg_roundf(y))) < .01;
                    const bool be_agressive = G.step_rand_int % 2 == 0;
                    const float evx = e.evx(j), evy = e.evy(j);
                    if ((evx == 0 && evy == 0) || is_at_junction) {
                        const int prev_idx = to_grid_idx(G, (int)((double)x - sign_d((double)evx)), (int)((double)y - sign_d((double)evy)));
                        const int ex_ = enemy_idx % mw, ey_ = enemy_idx / mw;
                        const int di[4] = {-1, 0, 0, 1}, dj[4] = {0, -1, 1, 0};  // get_adjacent chaser.cpp:280-299
                        int space_neighbors[4] = {0, 0, 0, 0};
                        int ns = 0;
                        int min_dist = 2 * mw;
                        for (int k = 0; k < 4; k++) {
                            const int adj = to_grid_idx(G, ex_ + di[k], ey_ + dj[k]);
                            if (adj == INVALID_IDX) continue;
                            if ((int)e.s->grid[adj] != MAZE_WALL && adj != prev_idx) {
                                const int dx = (adj % mw) - (agent_idx % mw), dy = (adj / mw) - (agent_idx / mw);
                                const int md_ = ((dx < 0 ? -dx : dx) + (dy < 0 ? -dy : dy)) * dist_scale;
                                if (be_agressive) {
                                    if (md_ < min_dist) {
                                        min_dist = md_;
                                        ns = 0;
                                        space_neighbors[ns++] = adj;
                                    } else if (md_ == min_dist) {
                                        space_neighbors[ns++] = adj;
                                    }
                                } else {
                                    space_neighbors[ns++] = adj;
                                }
                            }
                        }
                        if (ns == 0) {
                            e.fail(PGE_ASSERT);
                        } else {
                            const int pick = (int)((uint32_t)G.step_rand_int % (uint32_t)ns);
                            const int neighbor = pick == 0 ? space_neighbors[0] : (pick == 1 ? space_neighbors[1] : (pick == 2 ? space_neighbors[2] : space_neighbors[3]));
                            const int nx = neighbor % mw, ny = neighbor / mw;
                            e.evx(j) = (nx - x) * vscale;
                            e.evy(j) = (ny - y) * vscale;
                        }
                    }
                }
                PG_SYNC();
            }
        }
        if (num_enemies < CH_TOTAL_ENEMIES(G)) {
            const int selected_idx = (int)((uint32_t)G.step_rand_int % (uint32_t)CH_NUM_FREE(G));
            const int cell = nth_cell(e, selected_idx, [](int v) { return v != MAZE_WALL; });
            if (cell < 0) e.fail(PGE_ASSERT);
            else spawn_egg(e, cell);
        }
        const int ag = G.agent;
        const int agent_idx = (int)e.ey(ag) * G.main_width + (int)e.ex(ag);  // get_agent_index BAG:176-178
        if (agent_idx >= 0 && agent_idx < G.main_width * G.main_height && (int)e.s->grid[agent_idx] == ORB) {
            e.s->grid[agent_idx] = (cell_t)SPACE;
            G.grid_dirty = 1;
            G.reward += ORB_REWARD;
            CH_ORBS_COLLECTED(G) += 1;
        }
        if (CH_ORBS_COLLECTED(G) == CH_TOTAL_ORBS(G)) {
            G.reward += 10.0f;
            G.level_complete = 1;
            G.done = 1;
        }
        PG_SYNC();
    }

    template <class E>
    PG_DEV static int image_for_type(E &e, int type) {  // chaser.cpp:97-110
        if (type == ENEMY) {
            if (can_eat_enemies(e.G)) return ENEMY_WEAK;
            int rem = (e.G.cur_time / 2) % 4;
            if (rem == 3) rem = 1;
            return ENEMY + rem;
        }
        return type < 0 ? -type : type;
    }
    // draw_grid_obj override chaser.cpp:112-119: orbs are solid green squares inside their cell rect
    template <class R>
    PG_DEV static bool is_grid_fill(R &, int type) { return type == ORB; }
    template <class R>
    PG_DEV static void grid_fill(R &, int, const RectD &rect, RectD &out, uint32_t &color) {
        out.x = rect.x + rect.w * (double)(1 - ORB_DIM) / 2;
        out.y = rect.y + rect.h * (double)(1 - ORB_DIM) / 2;
        out.w = rect.w * (double)ORB_DIM;
        out.h = rect.h * (double)ORB_DIM;
        color = 0xff00ff00u;
    }
};

}  // namespace pgamd
