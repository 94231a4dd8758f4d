// game_starpilot.h -- StarPilot rules as a policy for Env<> / Renderer<> (reference procgen/src/games/starpilot.cpp).
// The projectile-heavy game of BASELINE configs[3]: a side-scrolling shooter on a 16x16 world without grid objects.
// A level is a pre-generated, time-sorted list of "spawners" (up to ~250 future entities) that enter the entity list
// when cur_time reaches their spawn_time; enemies fire aimed bullets (atan2f -> rotation), the player fires
// left/right, bullets destroy enemies through the entity-entity collision handler.
//
// The spawner list lives in the env's aux slab in HBM as a second entity table ([EF_COUNT][SPAWN_CAP] SoA, same
// field encoding as the live table, so get_state / set_state serialize it with the same code).  Level generation
// builds each spawner in a free slot of the LDS entity table, copies the record out, sorts a permutation of the
// spawn_time keys in LDS with libstdc++'s introsort (the order of equal keys is part of the state) and applies the
// permutation to the HBM table field by field.
#pragma once
#include "pg_env.h"
#include "pg_math.h"

namespace pgamd {

struct StarScratch {
    int key[256];       // spawn_time of spawner k (generation order)
    uint8_t perm[256];  // sorted position -> generation index
    int16_t stack[32][3];  // pending (first, last, depth_limit) ranges of the introsort
};

struct StarPilot {
    static constexpr int GAME_ID = GAME_STARPILOT;
    static constexpr const char *NAME = "starpilot";
    typedef uint8_t cell_t;
    typedef StarScratch Scratch;
    static constexpr int MAX_CELLS = 16 * 16;  // starpilot.cpp:51-52
    static constexpr bool USES_ENTITY_COLLISIONS = true;
    static constexpr bool USES_ROTATION = true;
    static constexpr int ROT_POOL_FACTOR = 2;  // (pg_render.h ROT_POOL: 2 x 16 records) at 16 records 0.6 % of the frames fall back to the per-band path
    static constexpr bool DRAWS_GRID = false;  // the grid holds only SPACE
    static constexpr bool CUSTOM_BACKGROUND = true;
    static constexpr int SPAWN_CAP = 256;
    static constexpr int AUX_WORDS = EF_COUNT * SPAWN_CAP;
    static constexpr int ENT_CAP_T0 = 64, ENT_CAP_T1 = 128, ENT_CAP_T2 = 256;

    // object ids starpilot.cpp:12-21
    static constexpr int BULLET_PLAYER = 1, BULLET2 = 2, BULLET3 = 3, FLYER = 4, METEOR = 5, CLOUD = 6, TURRET = 7, FAST_FLYER = 8, FINISH_LINE = 9;
    static constexpr int SHOOTER_WIN_TIME = 500;
    static constexpr int NUM_BASIC_OBJECTS = 9;
    static constexpr int NUM_SHIP_THEMES = 7;
    static constexpr float V_SCALE = 2.0f / 5.0f;
    static constexpr float HP_SLOW_V = .5f;  // init_hps starpilot.cpp:206
    static constexpr int HP_MAX_GROUP_SIZE = 5, HP_MIN_ENEMY_DELTA_T = 10, HP_MAX_ENEMY_DELTA_T = 30;
    static constexpr float HP_SPAWN_RIGHT_THRESHOLD = 0.9f;

#define SP_N_SPAWNERS(G) (G).gsi0
#define SP_NEXT_SPAWN_TIME(G) (G).gsi1  // spawn_time of the list's tail (-1: list empty)

    PG_DEV static bool center_agent(const GameOptions &) { return false; }  // options.center_agent = false, starpilot.cpp:330

    static void construct(EnvHdr &G) {  // Game::Game, BAG ctor (BAG:22-46), StarPilotGame ctor (starpilot.cpp:48-53)
        G = EnvHdr{};
        G.timeout = 1000;
        G.last_reward = -1;
        G.done = 1;
        G.visibility = 16;
        G.min_visibility = 0;
        G.mixrate = 0.5f;
        G.maxspeed = 0.5f;
        G.max_jump = 0.5f;
        G.default_action = 4;
        G.last_move_action = 7;
        G.out_of_bounds_object = INVALID_OBJ;
        G.has_useful_vel_info = 1;
        G.random_agent_start = 1;
        G.main_width = 16;
        G.main_height = 16;
        G.rand_idx = MT_N;
        G.lvl_rand_idx = MT_N;
        G.gsi1 = -1;
    }

    // ---- init_hps starpilot.cpp:148-227 as functions of the distribution mode ---------------------------------------
    PG_DEV static float hp_vs(int dm, int type) {
        if (type == FAST_FLYER) return (float)1.5;
        if (type == BULLET_PLAYER || type == BULLET3) return 2;
        if (type == BULLET2) return dm == EasyMode ? (float)1.25 : 2.0f;
        if (type == FLYER) return dm == EasyMode ? (float).75 : 1.0f;
        return 1;
    }
    PG_DEV static float hp_healths(int dm, int type) {
        if (type == METEOR) return 500;
        if (type == TURRET) return dm == ExtremeMode ? 10.0f : 5.0f;
        if (type == FLYER) return dm == ExtremeMode ? 5.0f : 2.0f;
        if (type == FAST_FLYER) return dm == ExtremeMode ? 2.0f : 1.0f;
        return 0;
    }
    PG_DEV static float hp_bullet_r(int dm) { return dm == ExtremeMode ? 1.0f / 5 : (float)(1.0f / 2.5); }
    PG_DEV static float hp_object_r(int type) { return (type == TURRET || type == METEOR || type == CLOUD) ? 2.0f : 0.5f; }
    PG_DEV static float hp_object_prob_weight(int dm, int type) {
        if (type == BULLET_PLAYER || type == BULLET2 || type == BULLET3) return 0;
        if (type == FLYER) return 3;
        if (dm == EasyMode && (type == METEOR || type == CLOUD || type == TURRET || type == FAST_FLYER)) return 0;
        return 1;
    }
    PG_DEV static bool is_lethal(int type) {  // starpilot.cpp:341-345
        return type == FLYER || type == FAST_FLYER || type == BULLET2 || type == BULLET3 || type == TURRET || type == METEOR;
    }
    PG_DEV static bool is_destructible(int type) { return type == FLYER || type == FAST_FLYER || type == TURRET || type == METEOR; }  // :347-349

    // entity slots the next step can need: every entity that fires adds a bullet; a player bullet can add one
    // explosion when it hits and one when its target dies; <= 3 spawn groups overlap in time; the player's own
    // bullet; the finish line; the reserved slot
    template <class E>
    PG_DEV static int slots_needed_next_step(E &e) {
        const int n = e.G.n_ents;
        const int t1 = e.G.cur_time + 1;
        int extra = 0;
        for (int c = 0; c < ((n + 63) >> 6); c++) {
            const uint64_t fires = PG_BALLOT(l, ({
                                                 const int i = (c << 6) + l;
                                                 bool f = false;
                                                 if (i < n) {
                                                     const int ft = e.ei(EF_FIRE_TIME, i), st = e.ei(EF_SPAWN_TIME, i);
                                                     if (ft > 0) f = e.etype(i) == TURRET ? ((t1 - st) % ft == 0) : (t1 - st == ft);
                                                 }
                                                 f;
                                             }));
            const uint64_t pb = PG_BALLOT(l, ((c << 6) + l) < n && e.etype((c << 6) + l) == BULLET_PLAYER);
            extra += pg_popc64(fires) + 2 * pg_popc64(pb);
        }
        return n + extra + 3 + 1 + 1 + 2 + 1;
    }

    template <class E>
    PG_DEV static void choose_world_dim(E &) {}

    // ---- physics hooks: BasicAbstractGame defaults -------------------------------------------------------------------
    template <class E>
    PG_DEV static bool is_blocked(E &e, int, int target, bool) { return target == WALL_OBJ || target == e.G.out_of_bounds_object; }
    template <class E>
    PG_DEV static bool is_blocked_ents(E &e, int src, int target, bool h) { return is_blocked(e, e.etype(src), e.etype(target), h); }
    PG_DEV static bool will_reflect(int, int) { return false; }
    template <class E>
    PG_DEV static bool may_interact(E &e, int s, int t, bool h) { return is_blocked(e, s, t, h); }
    template <class E>
    PG_DEV static void handle_agent_collision(E &e, int obj) {  // starpilot.cpp:126-136
        const int t = e.etype(obj);
        if (t == FINISH_LINE) {
            e.G.done = 1;
            e.G.reward += 10.0f;
            e.G.level_complete = 1;
        } else if (is_lethal(t)) {
            e.G.done = 1;
        }
    }
    template <class E>
    PG_DEV static void handle_grid_collision(E &, int, int, int, int) {}
    template <class E>
    PG_DEV static void handle_collision(E &e, int src, int target) {  // starpilot.cpp:138-146
        const int tt = e.etype(target);
        if (e.etype(src) == BULLET_PLAYER && tt != CLOUD && is_destructible(tt)) {
            e.set_flag(src, MF_WILL_ERASE, true);
            e.ef(EF_HEALTH, target) -= 1;
            e.add_entity(e.ex(src), e.ey(src), e.evx(target), e.evy(target), (float)(.5 * (double)e.erx(src)), EXPLOSION);
        }
    }
    template <class E>
    PG_DEV static void set_action_xy(E &e, int move_action) {  // BAG:658-662
        e.G.action_vx = (float)(move_action / 3 - 1);
        e.G.action_vy = (float)(move_action % 3 - 1);
        e.G.action_vrot = 0;
    }
    template <class E>
    PG_DEV static void update_agent_velocity(E &e) { e.bag_update_agent_velocity(1.0f); }
    template <class E>
    PG_DEV static void choose_center(E &e, float &cx, float &cy) {
        cx = e.ex(e.G.agent);
        cy = e.ey(e.G.agent);
    }

    // Entity::face_direction reference src/entity.cpp:84-88 (atan2f by overload resolution)
    template <class E>
    PG_DEV static void face_direction(E &e, int i, float dx, float dy, float rotation_offset) {
        if (dx != 0 || dy != 0) e.ef(EF_ROTATION, i) = -1 * pg_atan2f(dy, dx) + rotation_offset;
    }

    // ---- level generation ------------------------------------------------------------------------------------------------
    template <class E>
    PG_DEV static float rand_pos(E &e, float r, float min, float max) {  // BAG:1100-1108
        if (!(min <= max)) e.fail(PGE_ASSERT);
        if (max - min <= 2 * r) return (max + min) / 2;
        const float range = max - min;
        return (range - 2 * r) * e.rand01() + r + min;
    }

    template <class E>
    PG_DEV static void add_spawners(E &e) {  // starpilot.cpp:229-325
        EnvHdr &G = e.G;
        const int dm = e.opt.distribution_mode;
        uint32_t *aux = e.aux();
        StarScratch &sc = e.s->scratch;
        float total_prob_weight = 0;
        for (int i = 2; i < NUM_BASIC_OBJECTS; i++) total_prob_weight += hp_object_prob_weight(dm, i);
        int t = 1 + e.randint(HP_MIN_ENEMY_DELTA_T, HP_MAX_ENEMY_DELTA_T);
        const bool can_spawn_left = dm != EasyMode;
        const int tmp_slot = 1;  // free slot of the LDS table (only the agent exists during a reset)
        int n_sp = 0;
        for (int i = 0; t <= SHOOTER_WIN_TIME; i++) {
            int group_size = 1;
            const float start_weight = e.rand01() * total_prob_weight;
            float curr_weight = start_weight;
            int type;
            for (type = 2; type < NUM_BASIC_OBJECTS; type++) {
                curr_weight -= hp_object_prob_weight(dm, type);
                if (curr_weight <= 0) break;
            }
            if (type >= NUM_BASIC_OBJECTS) type = NUM_BASIC_OBJECTS - 1;
            const float r = hp_object_r(type);
            int flyer_theme = 0;
            if (type == FLYER || type == FAST_FLYER) {
                group_size = e.randint(0, HP_MAX_GROUP_SIZE) + 1;
                flyer_theme = e.randn(NUM_SHIP_THEMES);
            }
            const float y_pos = rand_pos(e, r, 0, (float)G.main_height);
            for (int j = 0; j < group_size; j++) {
                const int spawn_time = t + j * 5;
                int fire_time = e.randint(10, 100);
                const float k = 2 * PG_PI / 4;
                float theta = (float)(((double)e.rand01() - .5) * (double)k);
                float v_scale = hp_vs(dm, type);
                if (e.randint(0, 2) == 1) theta = 0;
                const float health = hp_healths(dm, type);
                if (type == METEOR || type == CLOUD) {
                    theta = 0;
                    v_scale = HP_SLOW_V;
                    fire_time = -1;
                } else if (type == TURRET) {
                    theta = 0;
                    v_scale = HP_SLOW_V;
                    fire_time = e.randint(20, 30);
                }
                v_scale *= V_SCALE;
                float vx = (float)(-1 * pg_cos_d((double)theta) * (double)v_scale);
                const float vy = (float)(pg_sin_d((double)theta) * (double)v_scale);
                bool spawn_right = true;
                float x_pos;
                if (type == FLYER || type == FAST_FLYER) {
                    if (e.rand01() > HP_SPAWN_RIGHT_THRESHOLD && can_spawn_left) spawn_right = false;
                }
                if (spawn_right) {
                    x_pos = G.main_width + r;
                } else {
                    x_pos = -r;
                    vx *= -1;
                }
                e.ent_init(tmp_slot, x_pos, y_pos, vx, vy, r, r, type);
                e.ei(EF_FIRE_TIME, tmp_slot) = fire_time;
                e.ei(EF_SPAWN_TIME, tmp_slot) = spawn_time;
                e.ef(EF_HEALTH, tmp_slot) = health;
                if (type == CLOUD) {
                    e.set_render_z(tmp_slot, 1);
                    e.choose_random_theme(tmp_slot);
                } else if (type == METEOR) {
                    e.choose_random_theme(tmp_slot);
                } else if (type == FLYER || type == FAST_FLYER) {
                    e.set_image_theme(tmp_slot, flyer_theme);
                    e.ef(EF_ROTATION, tmp_slot) = ((vx > 0) ? -1 : 1) * PG_PI / 2;
                } else if (type == TURRET) {
                    e.choose_random_theme(tmp_slot);
                    e.match_aspect_ratio(tmp_slot);
                }
                if (n_sp >= SPAWN_CAP) {
                    e.fail(PGE_ENT_OVERFLOW);
                } else {
                    PG_SYNC();
                    PG_FOR_LANES(l) {
                        if (l < EF_COUNT) aux[l * SPAWN_CAP + n_sp] = e.s->ent[l * E::CAPACITY + tmp_slot];
                    }
                    sc.key[n_sp] = spawn_time;
                    sc.perm[n_sp] = (uint8_t)n_sp;
                    n_sp++;
                    PG_SYNC();
                }
            }
            t += e.randint(HP_MIN_ENEMY_DELTA_T, HP_MAX_ENEMY_DELTA_T);
        }
        SP_N_SPAWNERS(G) = n_sp;
    }

    // std::sort(spawners.begin(), spawners.end(), spawn_cmp) -- starpilot.cpp:29-31,334.  libstdc++ (GCC 11)
    // bits/stl_algo.h std::__sort: introsort (median of three moved to the front, unguarded partition, ranges of
    // <= 16 left alone) followed by the final insertion sort; run on the permutation.  The heapsort fallback at
    // depth 2*floor(log2 n) is not restated: the env reports an error if it would be taken.
    PG_DEV static bool cmp(const StarScratch &sc, int a, int b) { return PG_UNIFORM_I(sc.key[a] > sc.key[b]) != 0; }  // spawn_cmp on generation indices
    PG_DEV static void unguarded_linear_insert(StarScratch &sc, int last) {
        const int val = sc.perm[last];
        int next = last - 1;
        while (cmp(sc, val, sc.perm[next])) {
            sc.perm[last] = sc.perm[next];
            last = next;
            next--;
        }
        sc.perm[last] = (uint8_t)val;
    }
    PG_DEV static void insertion_sort(StarScratch &sc, int first, int last) {
        if (first == last) return;
        for (int i = first + 1; i != last; i++) {
            if (cmp(sc, sc.perm[i], sc.perm[first])) {
                const int val = sc.perm[i];
                for (int k = i; k > first; k--) sc.perm[k] = sc.perm[k - 1];
                sc.perm[first] = (uint8_t)val;
            } else {
                unguarded_linear_insert(sc, i);
            }
        }
    }
    template <class E>
    PG_DEV static void sort_spawners(E &e) {
        StarScratch &sc = e.s->scratch;
        const int n = SP_N_SPAWNERS(e.G);
        if (n == 0) return;
        int lg = 0;
        while ((1 << (lg + 1)) <= n) lg++;
        // __introsort_loop with its tail recursion on [cut, last) turned into an explicit stack of pending ranges
        int sp = 0;
        sc.stack[0][0] = 0;
        sc.stack[0][1] = (int16_t)n;
        sc.stack[0][2] = (int16_t)(lg * 2);
        sp = 1;
        while (sp > 0) {
            sp--;
            const int first = PG_UNIFORM_I(sc.stack[sp][0]);
            int last = PG_UNIFORM_I(sc.stack[sp][1]);
            int depth_limit = PG_UNIFORM_I(sc.stack[sp][2]);
            // ranges popped from the stack are processed in the order the recursion would reach them: the
            // recursive call on [cut, last) runs BEFORE the loop continues on [first, cut), but the two ranges are
            // disjoint, so the order does not change the result
            while (last - first > 16) {
                if (depth_limit == 0) {
                    e.fail(PGE_ASSERT);
                    return;
                }
                --depth_limit;
                const int mid = first + (last - first) / 2;
                const int a = first + 1, b = mid, c = last - 1;
                int pick;  // __move_median_to_first
                if (cmp(sc, sc.perm[a], sc.perm[b])) {
                    if (cmp(sc, sc.perm[b], sc.perm[c])) pick = b;
                    else if (cmp(sc, sc.perm[a], sc.perm[c])) pick = c;
                    else pick = a;
                } else if (cmp(sc, sc.perm[a], sc.perm[c])) pick = a;
                else if (cmp(sc, sc.perm[b], sc.perm[c])) pick = c;
                else pick = b;
                {
                    const uint8_t t = sc.perm[first];
                    sc.perm[first] = sc.perm[pick];
                    sc.perm[pick] = t;
                }
                int lo = first + 1, hi = last;
                for (;;) {  // __unguarded_partition
                    while (cmp(sc, sc.perm[lo], sc.perm[first])) ++lo;
                    --hi;
                    while (cmp(sc, sc.perm[first], sc.perm[hi])) --hi;
                    if (!(lo < hi)) break;
                    const uint8_t t = sc.perm[lo];
                    sc.perm[lo] = sc.perm[hi];
                    sc.perm[hi] = t;
                    ++lo;
                }
                if (sp >= 31) {
                    e.fail(PGE_ASSERT);
                    return;
                }
                sc.stack[sp][0] = (int16_t)lo;
                sc.stack[sp][1] = (int16_t)last;
                sc.stack[sp][2] = (int16_t)depth_limit;
                sp++;
                last = lo;
            }
        }
        if (n > 16) {  // __final_insertion_sort
            insertion_sort(sc, 0, 16);
            for (int i = 16; i != n; i++) unguarded_linear_insert(sc, i);
        } else {
            insertion_sort(sc, 0, n);
        }
        PG_SYNC();
        // apply the permutation to the HBM table, one field at a time (all reads of a field before its writes)
        uint32_t *aux = e.aux();
        for (int f = 0; f < EF_COUNT; f++) {
            PG_LANE_ARR(uint32_t, v, 4);
            PG_FOR_LANES(l) {
                for (int q = 0; q < 4; q++) {
                    const int j = l + 64 * q;
                    PG_LA(v, q, l) = j < n ? aux[f * SPAWN_CAP + sc.perm[j]] : 0u;
                }
            }
            PG_SYNC();
            PG_FOR_LANES(l) {
                for (int q = 0; q < 4; q++) {
                    const int j = l + 64 * q;
                    if (j < n) aux[f * SPAWN_CAP + j] = PG_LA(v, q, l);
                }
            }
            PG_SYNC();
        }
        SP_NEXT_SPAWN_TIME(e.G) = sc.key[sc.perm[n - 1]];
    }

    template <class E>
    PG_DEV static void game_reset(E &e) {  // starpilot.cpp:327-339
        e.bag_game_reset();
        EnvHdr &G = e.G;
        G.maxspeed = e.opt.distribution_mode == ExtremeMode ? (float)0.5 : (float)0.75;  // init_hps
        SP_N_SPAWNERS(G) = 0;
        SP_NEXT_SPAWN_TIME(G) = -1;
        add_spawners(e);
        sort_spawners(e);
        const int ag = G.agent;
        e.ef(EF_ROTATION, ag) = PG_PI / 2;
        e.choose_random_theme(ag);
        PG_SYNC();
    }

    template <class E>
    PG_DEV static void game_step(E &e) {  // starpilot.cpp:363-430
        e.bag_game_step();
        EnvHdr &G = e.G;
        const int dm = e.opt.distribution_mode;
        const bool is_firing = G.special_action != 0;
        PG_SYNC();
        {
            // entities that fire or die this step, visited from the highest index down (the loop bound is the size
            // at entry; bullets and explosions are appended behind it)
            const int n0 = G.n_ents;
            const int ag = G.agent;
            for (int c = (n0 - 1) >> 6; c >= 0; c--) {
                uint64_t m = PG_BALLOT(l, ({
                                           const int i = (c << 6) + l;
                                           bool w = false;
                                           if (i < n0) {
                                               const uint32_t mm = e.meta(i);
                                               const int type = meta_type(mm);
                                               if (type != PLAYER) {
                                                   const int ft = e.ei(EF_FIRE_TIME, i), st = e.ei(EF_SPAWN_TIME, i);
                                                   if (ft > 0) w = type == TURRET ? ((G.cur_time - st) % ft == 0) : (G.cur_time - st == ft);  // should_fire :351-361
                                                   w = w || (e.ef(EF_HEALTH, i) <= 0 && is_destructible(type) && !(mm & MF_WILL_ERASE));
                                               }
                                           }
                                           w;
                                       }));
                while (m) {
                    const int i = (c << 6) + pg_highest(m);
                    m &= ~(1ull << (i & 63));
                    const int type = e.etype(i);
                    const int ft = e.ei(EF_FIRE_TIME, i), st = e.ei(EF_SPAWN_TIME, i);
                    const bool fire = ft > 0 && (type == TURRET ? ((G.cur_time - st) % ft == 0) : (G.cur_time - st == ft));
                    if (fire) {
                        const int bullet_type = type == TURRET ? BULLET3 : BULLET2;
                        const float bullet_r = hp_bullet_r(dm);
                        float b_vx = e.ex(ag) - e.ex(i);
                        float b_vy = e.ey(ag) - e.ey(i);
                        const float bv_scale = (float)((double)(hp_vs(dm, bullet_type) * V_SCALE) / pg_sqrt((double)(b_vx * b_vx + b_vy * b_vy)));
                        b_vx = b_vx * bv_scale;
                        b_vy = b_vy * bv_scale;
                        const int nb = e.add_entity(e.ex(i), e.ey(i), b_vx, b_vy, bullet_r, bullet_type);
                        face_direction(e, nb, b_vx, b_vy, -1 * PG_PI / 2);
                    }
                    if (e.ef(EF_HEALTH, i) <= 0 && is_destructible(type) && !e.eflag(i, MF_WILL_ERASE)) {
                        e.add_entity(e.ex(i), e.ey(i), e.evx(i), e.evy(i), (float)(.5 * (double)e.erx(i)), EXPLOSION);  // spawn_child BAG:225-231
                        G.reward += 1.0f;
                        e.set_flag(i, MF_WILL_ERASE, true);
                    }
                    PG_SYNC();
                }
            }
        }
        // spawners whose time has come move from the tail of the HBM list into the entity table
        while (SP_N_SPAWNERS(G) > 0 && G.cur_time == SP_NEXT_SPAWN_TIME(G)) {
            const uint32_t *aux = e.aux();
            const int k = SP_N_SPAWNERS(G) - 1;
            const int i = G.n_ents;
            if (i >= E::CAPACITY - 1) {
                e.fail(PGE_ENT_OVERFLOW);
                break;
            }
            PG_FOR_LANES(l) {
                if (l < EF_COUNT) e.s->ent[l * E::CAPACITY + i] = aux[l * SPAWN_CAP + k];
            }
            G.n_ents = i + 1;
            SP_N_SPAWNERS(G) = k;
            SP_NEXT_SPAWN_TIME(G) = k > 0 ? (int)aux[EF_SPAWN_TIME * SPAWN_CAP + (k - 1)] : -1;
            PG_SYNC();
        }
        if (is_firing) {
            const int ag = G.agent;
            const float bullet_r = hp_bullet_r(dm);
            const float theta = G.special_action == 2 ? PG_PI : 0;
            const float v_scale = hp_vs(dm, BULLET_PLAYER) * V_SCALE;
            const float vx = (float)(pg_cos_d((double)theta) * (double)v_scale);
            const float vy = (float)(pg_sin_d((double)theta) * (double)v_scale);
            const float x_off = (float)((double)e.erx(ag) * pg_cos_d((double)theta));
            const int b = e.add_entity(e.ex(ag) + x_off, e.ey(ag), vx, vy, bullet_r, BULLET_PLAYER);
            e.set_flag(b, MF_COLLIDES, true);
            face_direction(e, b, vx, vy, 0);
            e.ef(EF_ROTATION, b) -= PG_PI / 2;
        }
        if (G.cur_time == SHOOTER_WIN_TIME) {
            const int f = e.add_entity_rxy((float)G.main_width, (float)(G.main_height / 2), -1 * HP_SLOW_V * V_SCALE, 0, 2, (float)(G.main_height / 2), FINISH_LINE);
            e.choose_random_theme(f);
            e.match_aspect_ratio_h(f);
            e.ex(f) = G.main_width + e.erx(f);
        }
        PG_SYNC();
    }

    // ---- drawing hooks ---------------------------------------------------------------------------------------------------
    template <class E>
    PG_DEV static int image_for_type(E &, int type) { return type < 0 ? -type : type; }
    template <class E>
    PG_DEV static int theme_for_grid_obj(E &, int) { return 0; }
    PG_DEV static RectD adjusted_image_rect(int, RectD rect) { return rect; }
    PG_DEV static bool should_preserve_type_themes(int) { return false; }
    template <class E>
    PG_DEV static bool should_draw_entity(E &, int) { return true; }
    template <class E>
    PG_DEV static float tile_aspect_ratio(E &, int) { return 0; }
    // game_draw override starpilot.cpp:108-124: 18 square tiles of the background image scroll left with time;
    // candidates = the tiles around the one under screen column 0 (tile_image BAG:854-865 with tile_ratio 1)
    template <class R>
    PG_DEV static int background_rects(R &r, RectD (&rects)[4]) {
        const float scale = (float)(R::FRAME_H / r.G.main_height);
        const float bg_k = 3;
        const float t = (float)r.G.cur_time;
        const float char_dim = 5;  // BAG:24
        const float x_off = -t * scale * HP_SLOW_V * 2 / char_dim;
        const RectD rect = {(double)x_off, (double)(-R::FRAME_H * (bg_k - 1) / 2), (double)(R::FRAME_H * bg_k * 18.0f), (double)(R::FRAME_H * bg_k)};
        int num_tiles = (int)(rect.w / (rect.h * (double)1.0f));
        if (num_tiles < 1) num_tiles = 1;
        const float tile_width = (float)(rect.w / num_tiles);
        const float tile_height = (float)rect.h;
        int i0 = (int)(-x_off / tile_width) - 1;
        _Pragma("unroll") for (int k = 0; k < 3; k++) {  // fixed slots (w = 0: tile off the strip): a running index would put the array in scratch memory
            const int i = i0 + k;
            const bool on = i >= 0 && i < num_tiles;
            rects[k].x = rect.x + (double)(tile_width * i);
            rects[k].y = rect.y;
            rects[k].w = on ? (double)tile_width : 0.0;
            rects[k].h = (double)tile_height;
        }
        return 3;
    }
};

}  // namespace pgamd
