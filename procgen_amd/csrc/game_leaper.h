// game_leaper.h -- Leaper rules as a policy for Env<> / Renderer<> (reference procgen/src/games/leaper.cpp).
// Frogger: lanes of cars (road cells) and logs (water cells) on a 15x15 grid, the agent hops one cell at a time
// (velocity decays over NSTEP frames), the finish line is an entity drawn as a row of tiles.  A reset fast-forwards
// the lane spawners for main_width / min_speed (= 300) iterations, so the entity table right after a reset holds
// everything that ever spawned (nothing is erased until the first step): the arenas are sized for that.
#pragma once
#include "pg_game_defaults.h"

namespace pgamd {

struct Leaper : BagDefaults<Leaper> {
    static constexpr int GAME_ID = GAME_LEAPER;
    static constexpr const char *NAME = "leaper";
    static constexpr int MAX_CELLS = 20 * 20;  // leaper.cpp:103-116
    static constexpr bool USES_ROTATION = true;  // cars driving left are turned by 180 degrees (a negative scale), the frog by +-90
    static constexpr int RENDER_MIN_WAVES = 4;  // with the 16-record rotation pool the arena is 9.7 KB: four render waves per SIMD at <= 128 VGPRs measured +9 % over the pool alone (27.0 -> 29.4 M) on the same box (profiles/r05_rot_pool_ab.txt)
    static constexpr bool USES_TILED_ENTITIES = true;
    static constexpr int WIDE_ROWS = 8;  // 161 instead of 170 VGPRs in the renderer: three waves per SIMD instead of two
    // reset: <= 15 cars x 5 lanes + 16 logs x 5 lanes in the worst case, typically < 90 (its 300-400 spawner rounds never
    // erase); steps: <= 10 spawns, and the first step's erase pass drops everything that left the world.  The reset kernel
    // has the big table (SPLIT_RESET), the step kernels take the tier the entity count asks for.
    static constexpr int ENT_CAP_T0 = 64, ENT_CAP_T1 = 128, ENT_CAP_T2 = 256;
    static constexpr bool SPLIT_RESET = true;
    static constexpr int RESET_CAP = ENT_CAP_T2;
    template <class E>
    PG_DEV static int slots_needed_next_step(E &e) { return e.G.n_ents + e.G.gsi1 + e.G.gsi3 + 3; }  // (gsi1 / gsi3 = LP_N_ROAD / LP_N_WATER)  // one spawn per lane and step at most

    static constexpr int LOG = 1, ROAD = 2, WATER = 3, CAR = 4, FINISH_LINE = 5;
    PG_HOSTDEV static bool use_block_asset(int t) { return t == WATER || t == ROAD; }  // leaper.cpp:87-89: generated as a rect texture (use_generated_assets)
    static constexpr float MONSTER_RADIUS = 0.25f, LOG_RADIUS = 0.45f;
    static constexpr int NSTEP = 5;
    static constexpr float MAX_SPEED = (float)(2 / (NSTEP - 1.0));
    static constexpr float VEL_DECAY = MAX_SPEED / NSTEP;

#define LP_BOTTOM_ROAD_Y(G) (G).gsi0
#define LP_N_ROAD(G) (G).gsi1
#define LP_BOTTOM_WATER_Y(G) (G).gsi2
#define LP_N_WATER(G) (G).gsi3
#define LP_GOAL_Y(G) (G).gsi4
    // lane speeds: road lanes 0-4 in gsf0-4, water lanes 0-2 in gsf5-7, water lanes 3-4 as the bits of gsi5-6
    PG_DEV static float road_speed(const EnvHdr &G, int k) { 
        const float a0 = pg_opaque_f(G.gsf0), a1 = pg_opaque_f(G.gsf1), a2 = pg_opaque_f(G.gsf2), a3 = pg_opaque_f(G.gsf3), a4 = pg_opaque_f(G.gsf4);
        return k == 0 ? a0 : (k == 1 ? a1 : (k == 2 ? a2 : (k == 3 ? a3 : a4)));
    }
    PG_DEV static void set_road_speed(EnvHdr &G, int k, float v) {  // every field is rewritten: an if / else chain of stores becomes one store at a computed offset (see pg_opaque_f)
        G.gsf0 = k == 0 ? v : G.gsf0;
        G.gsf1 = k == 1 ? v : G.gsf1;
        G.gsf2 = k == 2 ? v : G.gsf2;
        G.gsf3 = k == 3 ? v : G.gsf3;
        G.gsf4 = k >= 4 ? v : G.gsf4;
    }
    PG_DEV static float water_speed(const EnvHdr &G, int k) {
        const float a0 = pg_opaque_f(G.gsf5), a1 = pg_opaque_f(G.gsf6), a2 = pg_opaque_f(G.gsf7);
        const int b3 = pg_opaque_i(G.gsi5), b4 = pg_opaque_i(G.gsi6);
        return k == 0 ? a0 : (k == 1 ? a1 : (k == 2 ? a2 : __builtin_bit_cast(float, k == 3 ? b3 : b4)));
    }
    PG_DEV static void set_water_speed(EnvHdr &G, int k, float v) {
        G.gsf5 = k == 0 ? v : G.gsf5;
        G.gsf6 = k == 1 ? v : G.gsf6;
        G.gsf7 = k == 2 ? v : G.gsf7;
        G.gsi5 = k == 3 ? __builtin_bit_cast(int, v) : G.gsi5;
        G.gsi6 = k >= 4 ? __builtin_bit_cast(int, v) : G.gsi6;
    }

    PG_DEV static bool center_agent(const GameOptions &) { return false; }  // options.center_agent = false, leaper.cpp:125

    static void construct(EnvHdr &G) {  // leaper.cpp:35-38
        construct_defaults(G);
        G.maxspeed = MAX_SPEED;
        G.timeout = 500;
    }
    template <class E>
    PG_DEV static void choose_world_dim(E &e) {  // leaper.cpp:103-116
        const int dm = e.opt.distribution_mode;
        const int wd = dm == EasyMode ? 9 : (dm == HardMode ? 15 : 20);
        e.G.main_width = wd;
        e.G.main_height = wd;
    }
    template <class E>
    PG_DEV static void handle_agent_collision(E &e, int obj) {  // leaper.cpp:77-85
        const int t = e.etype(obj);
        const int ag = e.G.agent;
        if (t == CAR) {
            e.G.done = 1;
        } else if (t == FINISH_LINE && e.evx(ag) == 0 && e.evy(ag) == 0) {
            e.G.reward += 10.0f;
            e.G.done = 1;
            e.G.level_complete = 1;
        }
    }
    PG_DEV static bool should_preserve_type_themes(int type) { return type == PLAYER; }  // leaper.cpp:91-93

    PG_DEV static void decay_vel(float &vel) {  // leaper.cpp:208-214, sign() :23-25
        const float x = (float)(1.0 * (double)vel);
        const float vel_sign = x > 0 ? +1.0f : (x == 0 ? 0.0f : -1.0f);
        vel = (float)(pg_fabs((double)vel) - (double)VEL_DECAY);
        if (vel < 0) vel = 0;
        vel = vel * vel_sign;
    }
    template <class E>
    PG_DEV static void update_agent_velocity(E &e) {  // leaper.cpp:216-231
        EnvHdr &G = e.G;
        const int ag = G.agent;
        float vx = e.evx(ag), vy = e.evy(ag);
        if (vx == 0 && vy == 0) {
            if (G.action_vx != 0) {
                vx = G.maxspeed * G.action_vx;
                e.set_image_theme(ag, 1);
                e.ef(EF_ROTATION, ag) = (vx > 0 ? 1 : -1) * PG_PI / 2;
            } else if (G.action_vy != 0) {
                vy = G.maxspeed * G.action_vy;
                e.set_image_theme(ag, 1);
                e.ef(EF_ROTATION, ag) = vy > 0 ? 0 : PG_PI;
            }
        }
        decay_vel(vx);
        decay_vel(vy);
        e.evx(ag) = vx;
        e.evy(ag) = vy;
    }

    // leaper.cpp:177-206: each lane may emit one entity at the edge it moves away from, unless something is there
    template <class E>
    PG_DEV static void spawn_entities(E &e) {
        EnvHdr &G = e.G;
        const int n_road = LP_N_ROAD(G), n_water = LP_N_WATER(G);
        {
            // Most rounds no lane spawns anything (spawn probability = |speed| / 6 or / 2, a few percent): each lane then makes
            // exactly one draw, so the round's draws are looked at side by side, one wave lane per road / water lane, and
            // consumed together.  A round with a spawn attempt takes the reference's loop below (an attempt makes further
            // draws and may or may not add an entity).  A level's 300-400 spawner rounds are the reset's critical path.
            PG_LANE_VAR(uint32_t, u);
            const int L = n_road + n_water;
            if (L > 0 && e.rand_peek_lanes(L, u)) {
                const float r0 = pg_opaque_f(G.gsf0), r1 = pg_opaque_f(G.gsf1), r2 = pg_opaque_f(G.gsf2), r3 = pg_opaque_f(G.gsf3), r4 = pg_opaque_f(G.gsf4);
                const float w0 = pg_opaque_f(G.gsf5), w1 = pg_opaque_f(G.gsf6), w2 = pg_opaque_f(G.gsf7);
                const float w3 = __builtin_bit_cast(float, pg_opaque_i(G.gsi5)), w4 = __builtin_bit_cast(float, pg_opaque_i(G.gsi6));
                const uint64_t attempts = PG_BALLOT(l, ({
                                                        bool a = false;
                                                        if (l < L) {
                                                            const bool car = l < n_road;
                                                            const int k = car ? l : l - n_road;
                                                            const float speed = car ? (k == 0 ? r0 : (k == 1 ? r1 : (k == 2 ? r2 : (k == 3 ? r3 : r4))))
                                                                                    : (k == 0 ? w0 : (k == 1 ? w1 : (k == 2 ? w2 : (k == 3 ? w3 : w4))));
                                                            const float spawn_prob = (float)(pg_fabs((double)speed) / (car ? 6.0 : 2.0));
                                                            a = (float)((double)PG_LV(u, l) / 4294967296.0) < spawn_prob;
                                                        }
                                                        a;
                                                    }));
                if (attempts == 0) {
                    e.rand_skip(L);
                    return;
                }
            }
        }
        for (int lane = 0; lane < n_road + n_water; lane++) {
            const bool car = lane < n_road;
            const int k = car ? lane : lane - n_road;
            const float speed = car ? road_speed(G, k) : water_speed(G, k);
            const float spawn_prob = (float)(pg_fabs((double)speed) / (car ? 6.0 : 2.0));
            if (e.rand01() < spawn_prob) {
                const int i = G.n_ents;
                if (i >= E::CAPACITY - 1) {
                    e.fail(PGE_ENT_OVERFLOW);
                    return;
                }
                if (car) {
                    const float x = speed > 0 ? (-1 * MONSTER_RADIUS) : (G.main_width + MONSTER_RADIUS);
                    e.ent_init(i, x, (float)(LP_BOTTOM_ROAD_Y(G) + k + 0.5), speed, 0, 2 * MONSTER_RADIUS, MONSTER_RADIUS, CAR);
                    e.choose_random_theme(i);
                    if (speed < 0) e.ef(EF_ROTATION, i) = PG_PI;
                } else {
                    const float x = speed > 0 ? (-1 * LOG_RADIUS) : (G.main_width + LOG_RADIUS);
                    e.ent_init(i, x, (float)(LP_BOTTOM_WATER_Y(G) + k + 0.5), speed, 0, LOG_RADIUS, LOG_RADIUS, LOG);
                }
                PG_SYNC();
                if (!e.has_any_collision(i, 0)) G.n_ents = i + 1;
            }
        }
    }

    template <class E>
    PG_DEV static float rand_sign(E &e) { return e.rand01() < 0.5 ? 1.0f : -1.0f; }  // leaper.cpp:95-101
    template <class E>
    PG_DEV static float randrange(E &e, float low, float high) { return e.rand01() * (high - low) + low; }  // randgen.cpp:29-31
    template <class E>
    PG_DEV static int choose_extra_space(E &e) { return e.opt.distribution_mode == EasyMode ? 0 : e.randn(2); }  // leaper.cpp:118-120

    template <class E>
    PG_DEV static void game_reset(E &e) {  // leaper.cpp:122-175
        e.bag_game_reset();
        EnvHdr &G = e.G;
        const int dm = e.opt.distribution_mode;
        const int ag = G.agent;
        e.ey(ag) = e.ery(ag);
        float min_car_speed = 0.05f, max_car_speed = 0.2f, min_log_speed = 0.05f, max_log_speed = 0.1f;
        if (dm == EasyMode) {
            min_car_speed = 0.03f; max_car_speed = 0.12f; min_log_speed = 0.025f; max_log_speed = 0.075f;
        } else if (dm == ExtremeMode) {
            min_car_speed = 0.1f; max_car_speed = 0.3f; min_log_speed = 0.1f; max_log_speed = 0.2f;
        }
        LP_BOTTOM_ROAD_Y(G) = choose_extra_space(e) + 1;
        const int max_diff = dm == EasyMode ? 3 : 4;
        const int difficulty = e.randn(max_diff + 1);
        const int extra_lane_option = dm == EasyMode ? 0 : e.randn(4);
        const int num_road_lanes = difficulty + (extra_lane_option == 2 ? 1 : 0);
        for (int lane = 0; lane < num_road_lanes; lane++) {
            const float sgn = rand_sign(e);  // left operand of the product draws first (pinned by the oracle vs the compiled reference)
            const float mag = randrange(e, min_car_speed, max_car_speed);
            set_road_speed(G, lane, sgn * mag);
            e.fill_elem(0, LP_BOTTOM_ROAD_Y(G) + lane, G.main_width, 1, ROAD);
        }
        LP_N_ROAD(G) = num_road_lanes;
        LP_BOTTOM_WATER_Y(G) = LP_BOTTOM_ROAD_Y(G) + num_road_lanes + choose_extra_space(e) + 1;
        const int num_water_lanes = difficulty + (extra_lane_option == 3 ? 1 : 0);
        int curr_sign = (int)rand_sign(e);
        for (int lane = 0; lane < num_water_lanes; lane++) {
            set_water_speed(G, lane, curr_sign * randrange(e, min_log_speed, max_log_speed));
            curr_sign *= -1;
            e.fill_elem(0, LP_BOTTOM_WATER_Y(G) + lane, G.main_width, 1, WATER);
        }
        LP_N_WATER(G) = num_water_lanes;
        LP_GOAL_Y(G) = LP_BOTTOM_WATER_Y(G) + num_water_lanes + 1;
        const float lim = G.main_width / (min_car_speed < min_log_speed ? min_car_speed : min_log_speed);
        // leaper.cpp:171-174: the lanes are filled by running the spawners for main_width / min speed = 300-400 steps.  The
        // agent -- the only smart_step entity, at rest, and no entity type blocks or reflects it (BAG defaults) -- makes
        // the same basic_step_object every time: its inputs (its own box, the grid) do not change.  Once one round has
        // left it exactly where it was, the remaining rounds only need Entity::step for every entity (a reset used to
        // take 1.5 ms of a lone wave, almost all of it the agent's 400 object steps).
        bool agent_idle = false;
        int total_rounds = (int)lim;  // for (int i = 0; i < lim; i++)
        if ((float)total_rounds < lim) total_rounds++;
        for (int i = 0; i < total_rounds; i++) {
            if (agent_idle) {
                // Rounds in which no lane makes a spawn attempt (most of them: the attempt probability is a few percent per
                // lane) consist of one draw per lane and one Entity::step per entity.  The draws of the next 64 / L rounds are
                // looked at together (wave lane j = round j / L, road / water lane j % L); the leading attempt-free rounds are
                // then taken at once: their draws skipped, every entity stepped that many times inside one lane section.
                const int L = LP_N_ROAD(G) + LP_N_WATER(G);
                int R = L > 0 ? 64 / L : 64;
                if (R > total_rounds - i) R = total_rounds - i;
                int k = 0;
                PG_LANE_VAR(uint32_t, u);
                if (L == 0) {
                    k = R;
                } else if (R > 1 && e.rand_peek_lanes(R * L, u)) {
                    const int n_road = LP_N_ROAD(G);
                    const float r0 = pg_opaque_f(G.gsf0), r1 = pg_opaque_f(G.gsf1), r2 = pg_opaque_f(G.gsf2), r3 = pg_opaque_f(G.gsf3), r4 = pg_opaque_f(G.gsf4);
                    const float w0 = pg_opaque_f(G.gsf5), w1 = pg_opaque_f(G.gsf6), w2 = pg_opaque_f(G.gsf7);
                    const float w3 = __builtin_bit_cast(float, pg_opaque_i(G.gsi5)), w4 = __builtin_bit_cast(float, pg_opaque_i(G.gsi6));
                    const uint32_t inv = (uint32_t)(((1u << 16) + (uint32_t)L - 1u) / (uint32_t)L);  // j / L for j < 64
                    const uint64_t attempts = PG_BALLOT(l, ({
                                                            bool a = false;
                                                            if (l < R * L) {
                                                                const int lane = l - (int)(((uint32_t)l * inv) >> 16) * L;
                                                                const bool car = lane < n_road;
                                                                const int kk = car ? lane : lane - n_road;
                                                                const float speed = car ? (kk == 0 ? r0 : (kk == 1 ? r1 : (kk == 2 ? r2 : (kk == 3 ? r3 : r4))))
                                                                                        : (kk == 0 ? w0 : (kk == 1 ? w1 : (kk == 2 ? w2 : (kk == 3 ? w3 : w4))));
                                                                const float spawn_prob = (float)(pg_fabs((double)speed) / (car ? 6.0 : 2.0));
                                                                a = (float)((double)PG_LV(u, l) / 4294967296.0) < spawn_prob;
                                                            }
                                                            a;
                                                        }));
                    k = attempts == 0 ? R : (int)(((uint32_t)pg_ctz64(attempts) * inv) >> 16);
                }
                if (k > 0) {
                    e.rand_skip(k * L);
                    const int n = G.n_ents, a = G.agent;
                    for (int base = 0; base < n; base += 64) {
                        PG_FOR_LANES(l) {
                            const int idx = base + l;
                            if (idx < n) {
                                if (idx == a) {
                                    for (int t = 0; t < k; t++) e.ent_step(idx);
                                } else {
                                    float x = e.ex(idx);
                                    const float vx = e.evx(idx);
                                    for (int t = 0; t < k; t++) x += vx;
                                    e.ex(idx) = x;
                                    e.ei(EF_LIFE_TIME, idx) += k;
                                }
                            }
                        }
                    }
                    PG_SYNC();
                    i += k - 1;
                    continue;
                }
            }
            spawn_entities(e);
            PG_SYNC();
            if (agent_idle) {
                // Entity::step of a car / log as spawn_entities creates them (vy = 0, no spin, friction = grow_rate =
                // alpha_decay = 1, no expiry): x += vx and life_time += 1, every other member keeps its bits
                const int n = G.n_ents, a = G.agent;
                for (int base = 0; base < n; base += 64) {
                    PG_FOR_LANES(l) {
                        const int idx = base + l;
                        if (idx < n) {
                            if (idx == a) {
                                e.ent_step(idx);
                            } else {
                                e.ex(idx) += e.evx(idx);
                                e.ei(EF_LIFE_TIME, idx) += 1;
                            }
                        }
                    }
                }
                PG_SYNC();
            } else {
                const int a = G.agent;
                const float x0 = e.ex(a), y0 = e.ey(a), vx0 = e.evx(a), vy0 = e.evy(a);
                e.step_entities();
                const float x1 = e.ex(a), y1 = e.ey(a), vx1 = e.evx(a), vy1 = e.evy(a);
                agent_idle = vx0 == 0 && vy0 == 0 && __builtin_bit_cast(uint32_t, x0) == __builtin_bit_cast(uint32_t, x1) &&
                             __builtin_bit_cast(uint32_t, y0) == __builtin_bit_cast(uint32_t, y1) &&
                             __builtin_bit_cast(uint32_t, vx0) == __builtin_bit_cast(uint32_t, vx1) && __builtin_bit_cast(uint32_t, vy0) == __builtin_bit_cast(uint32_t, vy1);
            }
        }
        e.add_entity_rxy((float)(G.main_width / 2.0), (float)(LP_GOAL_Y(G) - .5), 0, 0, (float)(G.main_width / 2.0), (float).5, FINISH_LINE);
        PG_SYNC();
    }

    template <class E>
    PG_DEV static void game_step(E &e) {  // leaper.cpp:241-275
        EnvHdr &G = e.G;
        {
            const int th = meta_image_theme(e.meta(G.agent));
            if (th >= 1) e.set_image_theme(G.agent, (th + 1) % NSTEP);
        }
        e.bag_game_step();
        spawn_entities(e);
        PG_SYNC();
        const int ag = G.agent;
        bool standing_on_log = false;
        float log_vx = 0.0f;
        {
            const float margin = -1 * e.erx(ag);
            const int n = G.n_ents;
            const float ax = e.ex(ag), ay = e.ey(ag), arx = e.erx(ag), ary = e.ery(ag);
            for (int c = ((n + 63) >> 6) - 1; c >= 0 && !standing_on_log; c--) {  // the last match in list order gives log_vx
                const uint64_t m = PG_BALLOT(l, ({
                                                 const int i = (c << 6) + l;
                                                 bool hit = false;
                                                 if (i < n && e.etype(i) == LOG) {
                                                     const float tx = (arx + e.erx(i)) + margin;
                                                     const float ty = (ary + e.ery(i)) + margin;
                                                     hit = (pg_fabsf(ax - e.ex(i)) < tx) && (pg_fabsf(ay - e.ey(i)) < ty);
                                                 }
                                                 hit;
                                             }));
                if (m) {
                    standing_on_log = true;
                    log_vx = e.evx((c << 6) + pg_highest(m));
                }
            }
        }
        if (e.get_obj((int)e.ex(ag), (int)e.ey(ag)) == WATER) {
            if (!standing_on_log && e.evx(ag) == 0 && e.evy(ag) == 0) G.done = 1;
        }
        if (standing_on_log) e.ex(ag) += log_vx;
        PG_SYNC();
        if (e.is_out_of_bounds(ag)) G.done = 1;
    }

    PG_DEV static RectD adjusted_image_rect(int type, RectD rect) {  // leaper.cpp:233-239
        if (type == PLAYER) return adjust_rect(rect, 0, -.275, 1, 1.55);
        return rect;
    }
    template <class E>
    PG_DEV static float tile_aspect_ratio(E &e, int i) { return e.etype(i) == FINISH_LINE ? 1.0f : 0.0f; }  // leaper.cpp:69-75
};

}  // namespace pgamd
