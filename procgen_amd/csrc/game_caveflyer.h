// game_caveflyer.h -- CaveFlyer rules as a policy for Env<> / Renderer<> (reference procgen/src/games/caveflyer.cpp).
// A ship with rotational control (thrust along its heading, vrot) in a cave grown by a cellular automaton
// (RoomGenerator): the largest room is kept, a BFS path from the agent to the goal is widened and re-eroded, then
// obstacles, targets and patrolling enemies are dropped on free cells.  Bullets and exhaust are rotated sprites.
// CaveFlyerT<40 * 40> serves the easy / hard worlds; the memory-mode 60x60 world is its own instantiation
// (CaveFlyerMemory, kernel id KERNEL_CAVEFLYER_MEMORY) so that the default modes keep the small LDS arena.
#pragma once
#include "pg_game_defaults.h"
#include "pg_math.h"
#include "pg_roomgen.h"

namespace pgamd {

template <int CELLS, int KERNEL_ID>
struct CaveFlyerT : BagDefaults<CaveFlyerT<CELLS, KERNEL_ID>> {
    static constexpr int GAME_ID = KERNEL_ID;
    // pg_env.h GameParSmart: blocking / reflecting targets of this game are wall types only, never a smart entity's type,
    // and the hooks basic_step_object calls touch nothing but the moving object
    static constexpr bool PAR_SMART = true;
    PG_DEV static bool par_smart_type_ok(int t) { return t == PLAYER || t == ENEMY; }
    static constexpr const char *NAME = "caveflyer";
    static constexpr int MAX_CELLS = CELLS;  // caveflyer.cpp:131-146
    typedef RoomScratch<MAX_CELLS> Scratch;
    static constexpr bool USES_ENTITY_COLLISIONS = true;
    static constexpr bool USES_ROTATION = true;
    // a level drops 3 * (free cells / 80) objects: <= 60 in the 40x40 world, <= 135 in the 60x60 one
    static constexpr int ENT_CAP_T0 = CELLS > 1600 ? 160 : 64, ENT_CAP_T1 = CELLS > 1600 ? 192 : 96, ENT_CAP_T2 = CELLS > 1600 ? 256 : 160;
    // the level generator's scratch dominates the arena: step kernels without it, resets in the reset kernel (pg_env.h GameSplit)
    static constexpr bool SPLIT_RESET = true;
    static constexpr int RESET_CAP = ENT_CAP_T0;
    // a bullet, an exhaust puff, one explosion per bullet (wall hits, collisions) and per target, the reserved slot
    template <class E>
    PG_DEV static int slots_needed_next_step(E &e) {
        const int n = e.G.n_ents;
        int nb = 0;
        for (int c = 0; c < ((n + 63) >> 6); c++)
            nb += pg_popc64(PG_BALLOT(l, ((c << 6) + l) < n && (e.etype((c << 6) + l) == PLAYER_BULLET || e.etype((c << 6) + l) == TARGET)));
        return n + 2 + 2 * nb + 2 + 1;
    }

    static constexpr int GOAL = 1, OBSTACLE = 2, TARGET = 3, PLAYER_BULLET = 4, ENEMY = 5, CAVEWALL = 8, EXHAUST = 9;
    PG_HOSTDEV static bool use_block_asset(int t) { return t == CAVEWALL; }  // caveflyer.cpp:81-83: generated as a rect texture (use_generated_assets)
    static constexpr int MARKER = 250;  // the reference's transient 1003 (never visible outside game_reset); any unused id does

    static void construct(EnvHdr &G) {  // caveflyer.cpp:27-30
        CaveFlyerT::construct_defaults(G);
        G.mixrate = 0.9f;
    }
    template <class E>
    PG_DEV static void choose_world_dim(E &e) {  // caveflyer.cpp:131-146
        const int dm = e.opt.distribution_mode;
        const int wd = dm == EasyMode ? 30 : (dm == HardMode ? 40 : (dm == MemoryMode ? 60 : 20));
        if (wd * wd > MAX_CELLS) e.fail(PGE_ASSERT);
        e.G.main_width = wd;
        e.G.main_height = wd;
    }
    template <class E>
    PG_DEV static bool is_blocked(E &e, int src_type, int target, bool) {  // caveflyer.cpp:87-94
        return target == WALL_OBJ || target == e.G.out_of_bounds_object || (src_type == PLAYER && target == CAVEWALL);
    }
    // will_reflect's out_of_bounds_object is CAVEWALL whenever entities move (set at the end of game_reset)
    PG_DEV static bool will_reflect(int src, int target) { return src == ENEMY && target == CAVEWALL; }  // caveflyer.cpp:127-129
    template <class E>
    PG_DEV static void handle_agent_collision(E &e, int obj) {  // caveflyer.cpp:55-69
        const int t = e.etype(obj);
        if (t == GOAL) {
            e.G.reward += 10.0f;
            e.G.level_complete = 1;
            e.G.done = 1;
        } else if (t == OBSTACLE || t == ENEMY || t == TARGET) {
            e.G.done = 1;
        }
    }
    template <class E>
    PG_DEV static void handle_collision(E &e, int src, int target) {  // caveflyer.cpp:96-125
        if (e.etype(target) != PLAYER_BULLET) return;
        const int st = e.etype(src);
        bool erase_bullet = false;
        if (st == TARGET) {
            e.ef(EF_HEALTH, src) -= 1;
            erase_bullet = true;
            if (e.ef(EF_HEALTH, src) <= 0 && !e.eflag(src, MF_WILL_ERASE)) {
                e.add_entity(e.ex(src), e.ey(src), 0, 0, (float)(.5 * (double)e.erx(src)), EXPLOSION);  // spawn_child BAG:225-231
                e.set_flag(src, MF_WILL_ERASE, true);
                e.G.reward += 3.0f;
            }
        } else if (st == OBSTACLE || st == ENEMY || st == GOAL) {
            erase_bullet = true;
        }
        if (erase_bullet && !e.eflag(target, MF_WILL_ERASE)) {
            e.set_flag(target, MF_WILL_ERASE, true);
            const int x = e.add_entity(e.ex(target), e.ey(target), 0, 0, (float)(.5 * (double)e.erx(target)), EXPLOSION);
            e.evx(x) = e.evx(src);
            e.evy(x) = e.evy(src);
        }
    }
    template <class E>
    PG_DEV static void set_action_xy(E &e, int move_action) {  // caveflyer.cpp:264-285
        EnvHdr &G = e.G;
        const int ag = G.agent;
        float acceleration = (float)(move_action % 3 - 1);
        if (acceleration < 0) acceleration *= 0.33f;
        const float theta = -1 * e.ef(EF_ROTATION, ag) + PG_PI / 2;
        const double ct = pg_cos_d((double)theta), st = pg_sin_d((double)theta);
        if (acceleration > 0) {
            const float arx = e.erx(ag), ary = e.ery(ag);
            const int x = e.add_entity((float)((double)e.ex(ag) - (double)arx * ct), (float)((double)e.ey(ag) - (double)ary * st), 0, 0, (float)(.5 * (double)arx), EXHAUST);
            e.ei(EF_EXPIRE_TIME, x) = 4;
            e.ef(EF_ROTATION, x) = -1 * theta - PG_PI / 2;
            e.ef(EF_GROW_RATE, x) = (float)1.25;
            e.ef(EF_ALPHA_DECAY, x) = 0.8f;
        }
        G.action_vy = (float)((double)acceleration * st);
        G.action_vx = (float)((double)acceleration * ct);
        G.action_vrot = (float)(move_action / 3 - 1);
    }
    template <class E>
    PG_DEV static void update_agent_velocity(E &e) {  // caveflyer.cpp:71-78 (+ BAG:502-504,681-684)
        EnvHdr &G = e.G;
        const int ag = G.agent;
        const float v_scale = 1.0f;
        float vx = (float)((double)e.evx(ag) + (double)(G.mixrate * G.maxspeed * G.action_vx * v_scale) * .2);
        float vy = (float)((double)e.evy(ag) + (double)(G.mixrate * G.maxspeed * G.action_vy * v_scale) * .2);
        e.evx(ag) = (float)(.9 * (double)vx);
        e.evy(ag) = (float)(.9 * (double)vy);
    }

    template <class E>
    PG_DEV static void game_reset(E &e) {  // caveflyer.cpp:148-262
        e.bag_game_reset();
        EnvHdr &G = e.G;
        typedef typename E::cell_t cell_t;
        const int n = G.main_width * G.main_height, w = G.main_width;
        if (n > MAX_CELLS) {
            e.fail(PGE_ASSERT);
            return;
        }
        G.out_of_bounds_object = WALL_OBJ;
        for (int base = 0; base < n; base += 64) {  // one draw per cell, in index order
            PG_LANE_VAR(uint32_t, u);
            e.rand_u32_lanes((n - base) < 64 ? (n - base) : 64, u);
            PG_FOR_LANES(l) {
                if (base + l < n) {
                    const float r01 = (float)((double)PG_LV(u, l) / 4294967296.0);
                    e.s->grid[base + l] = (cell_t)(((double)r01 < .5) ? WALL_OBJ : SPACE);
                }
            }
        }
        G.grid_dirty = 1;
        PG_SYNC();
        e.mark(0);  // bag_game_reset, random fill
        RoomGenDev<E, MAX_CELLS> rg(e, e.s->scratch);
        auto &m = e.s->scratch;
        rg.update_rows(4, nullptr);
        e.mark(1);  // cellular automaton x 4
        const int best = rg.find_best_room();  // flags in f2
        e.mark(2);  // find_best_room
        if (best <= 0) {
            e.fail(PGE_ASSERT);
            return;
        }
        for (int base = 0; base < n; base += 64) {
            PG_FOR_LANES(l) {
                if (base + l < n) e.s->grid[base + l] = (cell_t)(m.f2[base + l] ? SPACE : WALL_OBJ);
            }
        }
        PG_SYNC();
        const int nfree0 = e.count_cells([](int v) { return v == SPACE; });
        e.simple_choose(nfree0, 2);
        const int agent_cell = e.nth_cell((int)PG_UNIFORM_I(e.s->tmp[0]), [](int v) { return v == SPACE; });
        const int goal_cell = e.nth_cell((int)PG_UNIFORM_I(e.s->tmp[1]), [](int v) { return v == SPACE; });
        const int ag = G.agent;
        e.ex(ag) = (float)((agent_cell % w) + .5);
        e.ey(ag) = (float)((agent_cell / w) + .5);
        const int goal = e.add_entity((float)((goal_cell % w) + .5), (float)((goal_cell / w) + .5), 0, 0, (float).5, GOAL);  // spawn_entity_at_idx BAG:577-583
        e.set_flag(goal, MF_COLLIDES, true);
        PG_SYNC();
        // goal path (flags in f3, kept until the end), covered flags in f0
        e.mark(3);  // agent / goal cells
        rg.find_path(agent_cell, goal_cell, m.f3, m.f0);
        e.mark(4);  // find_path
        if (e.opt.distribution_mode != MemoryMode) {  // should_prune: keep the path widened by 4 rings
            rg.copy(m.f1, m.f3);
            rg.expand_room(m.f1, 4);
            for (int base = 0; base < n; base += 64) {
                PG_FOR_LANES(l) {
                    if (base + l < n) e.s->grid[base + l] = (cell_t)(m.f1[base + l] ? SPACE : WALL_OBJ);
                }
            }
            PG_SYNC();
        }
        e.mark(5);  // expand_room
        rg.update_rows(4, m.f3);  // four passes, the path put back to SPACE after each
        e.mark(6);  // smoothing
        for (int base = 0; base < n; base += 64) {  // path -> MARKER, remaining walls -> CAVEWALL
            PG_FOR_LANES(l) {
                if (base + l < n) {
                    if (m.f3[base + l]) e.s->grid[base + l] = (cell_t)MARKER;
                    else if ((int)e.s->grid[base + l] == WALL_OBJ) e.s->grid[base + l] = (cell_t)CAVEWALL;
                }
            }
        }
        PG_SYNC();
        e.mark(7);  // markers, cave walls
        const int nfree = e.count_cells([](int v) { return v == SPACE; });
        const int chunk_size = nfree / 80;
        const int num_objs = 3 * chunk_size;
        e.simple_choose(nfree, num_objs);
        for (int i = 0; i < num_objs; i++) {
            const int val = e.nth_cell((int)PG_UNIFORM_I(e.s->tmp[i]), [](int v) { return v == SPACE; });
            const float px = (float)((val % w) + .5), py = (float)((val / w) + .5);
            if (i < chunk_size) {
                const int o = e.add_entity(px, py, 0, 0, (float).5, OBSTACLE);
                e.set_flag(o, MF_COLLIDES, true);
            } else if (i < 2 * chunk_size) {
                const int t = e.add_entity(px, py, 0, 0, (float).5, TARGET);
                e.ef(EF_HEALTH, t) = 5;
                e.set_flag(t, MF_COLLIDES, true);
            } else {
                const int en = e.add_entity(px, py, 0, 0, (float).5, ENEMY);
                const double mag = .1 * (double)e.rand01() + .1;  // the product's left operand draws first (pinned by the oracle)
                const int sgn = e.randn(2) * 2 - 1;
                const float vel = (float)(mag * sgn);
                if ((double)e.rand01() < .5) e.evx(en) = vel;
                else e.evy(en) = vel;
                e.set_flag(en, MF_SMART_STEP, true);
                e.set_flag(en, MF_COLLIDES, true);
            }
        }
        PG_SYNC();
        for (int base = 0; base < n; base += 64) {
            PG_FOR_LANES(l) {
                if (base + l < n && (int)e.s->grid[base + l] == MARKER) e.s->grid[base + l] = (cell_t)SPACE;
            }
        }
        PG_SYNC();
        e.mark(8);  // objects
        G.out_of_bounds_object = CAVEWALL;
        G.visibility = e.opt.distribution_mode == EasyMode ? 10.0f : 16.0f;
        G.grid_dirty = 1;
    }

    template <class E>
    PG_DEV static void game_step(E &e) {  // caveflyer.cpp:287-324
        e.bag_game_step();
        EnvHdr &G = e.G;
        if (G.special_action == 1) {
            const int ag = G.agent;
            const float theta = -1 * e.ef(EF_ROTATION, ag) + PG_PI / 2;
            const float vx = (float)pg_cos_d((double)theta);
            const float vy = (float)pg_sin_d((double)theta);
            const int b = e.add_entity_rxy(e.ex(ag), e.ey(ag), vx, vy, 0.1f, 0.25f, PLAYER_BULLET);
            e.ei(EF_EXPIRE_TIME, b) = 10;
            e.ef(EF_ROTATION, b) = e.ef(EF_ROTATION, ag);
        }
        PG_SYNC();
        {
            const int n0 = G.n_ents;
            for (int base = 0; base < n0; base += 64) {  // enemies face their heading (lane-parallel, no appends)
                PG_FOR_LANES(l) {
                    const int i = base + l;
                    if (i < n0 && e.etype(i) == ENEMY) {
                        const float dx = e.evx(i), dy = e.evy(i);
                        if (dx != 0 || dy != 0) e.ef(EF_ROTATION, i) = -1 * pg_atan2f(dy, dx) + -1 * PG_PI / 2;
                    }
                }
            }
            PG_SYNC();
            for (int c = (n0 - 1) >> 6; c >= 0; c--) {  // bullets that touch a cave wall explode, highest index first
                uint64_t m = PG_BALLOT(l, ({
                                           const int i = (c << 6) + l;
                                           bool found = false;
                                           if (i < n0 && e.etype(i) == PLAYER_BULLET) {
                                               const float x = e.ex(i), y = e.ey(i), rx = e.erx(i), ry = e.ery(i);
                                               for (int a = 0; a < 2; a++)
                                                   for (int b = 0; b < 2; b++) found = found || e.get_obj_from_floats(x + rx * (2 * a - 1), y + ry * (2 * b - 1)) == CAVEWALL;
                                           }
                                           found;
                                       }));
                while (m) {
                    const int i = (c << 6) + pg_highest(m);
                    m &= ~(1ull << (i & 63));
                    e.set_flag(i, MF_WILL_ERASE, true);
                    e.add_entity(e.ex(i), e.ey(i), 0, 0, (float)(.5 * (double)e.erx(i)), EXPLOSION);
                    PG_SYNC();
                }
            }
        }
        e.erase_if_needed();
        PG_SYNC();
    }
};
using CaveFlyer = CaveFlyerT<40 * 40, GAME_CAVEFLYER>;
using CaveFlyerMemory = CaveFlyerT<60 * 60, KERNEL_CAVEFLYER_MEMORY>;

}  // namespace pgamd
