// game_bossfight.h -- BossFight rules as a policy for Env<> / Renderer<> (reference procgen/src/games/bossfight.cpp).
// The most entity-dense game: a boss fires bullet patterns (sin/cos of a per-step angle), every enemy bullet
// leaves a fading, spinning laser trail each step (up to ~150 entities), player bullets and meteor barriers are
// colliders of the entity-entity pass, shields reflect bullets, rounds switch attack modes.
#pragma once
#include "pg_game_defaults.h"
#include "pg_math.h"

namespace pgamd {

struct BossFight : BagDefaults<BossFight> {
    static constexpr int GAME_ID = GAME_BOSSFIGHT;
    static constexpr const char *NAME = "bossfight";
    static constexpr int MAX_CELLS = 20 * 20;  // bossfight.cpp:66-67
    static constexpr bool USES_ENTITY_COLLISIONS = true;
    static constexpr bool USES_ROTATION = true;  // bullets and trails spin (vrot)
    // rotation records per frame, in sixteens (pg_render.h ROT_POOL): dozens of turned bullets are on screen at once.  3 x 16 since round 6:
    // 64 records (a record per lane) made the arena 10 640 bytes, nine LDS granules = 14 frames per CU; 48 make it 9104, eight granules =
    // 16 frames per CU -- the VGPR bound -- and the frames with more turned sprites than records take the per-band path: +2.7 % on the same
    // box (profiles/r06_call35_bossfight_rot_pool.txt; at 32 records one frame in six falls back and the game is slower, round 5)
    static constexpr int ROT_POOL_FACTOR = 3;
    static constexpr bool DRAWS_GRID = false;
    // (tier 0 at 120 slots -- 11 008 bytes, one LDS granule less -- measured +-0: profiles/r06_call27_ab.txt)
    static constexpr int ENT_CAP_T0 = 128, ENT_CAP_T1 = 256, ENT_CAP_T2 = 512;
    static constexpr int RENDER_CMD_SETS = 2;  // frames with more than 64 visible entities are common

    static constexpr int PLAYER_BULLET = 1, BOSS = 2, SHIELDS = 3, ENEMY_BULLET = 4, LASER_TRAIL = 5, REFLECTED_BULLET = 6, BARRIER = 7;
    static constexpr float BOSS_R = 3;
    static constexpr int NUM_ATTACK_MODES = 4, NUM_LASER_THEMES = 3, BOTTOM_MARGIN = 6, BOSS_VEL_TIMEOUT = 20, BOSS_DAMAGED_TIMEOUT = 40;
    static constexpr int VULNERABLE_DURATION = 500;
    static constexpr float BASE_FIRE_PROB = 0.1f;

#define BF_LAST_FIRE_TIME(G) (G).gsi0
#define BF_TIME_TO_SWAP(G) (G).gsi1
#define BF_CURR_VEL_TIMEOUT(G) (G).gsi2
#define BF_DAMAGED_UNTIL(G) (G).gsi3
#define BF_ROUND_NUM(G) (G).gsi4
#define BF_PACKED(G) (G).gsi5  // attack_modes 2b x 5 | num_rounds 3b @10 | round_health 4b @13 | invulnerable 3b @17 | player_laser 2b @20 | boss_laser 2b @22 | barriers_right 1b @24
#define BF_SHIELDS_UP(G) (G).gsi6
#define BF_ATTACK_MODE(G) (G).gsi7
#define BF_RAND_PCT(G) (G).gsf0
#define BF_RAND_FIRE_PCT(G) (G).gsf1
#define BF_RAND_PCT_X(G) (G).gsf2
#define BF_RAND_PCT_Y(G) (G).gsf3
    PG_DEV static int attack_modes_at(const EnvHdr &G, int k) { return (BF_PACKED(G) >> (2 * k)) & 3; }
    PG_DEV static int num_rounds(const EnvHdr &G) { return (BF_PACKED(G) >> 10) & 7; }
    PG_DEV static int round_health(const EnvHdr &G) { return (BF_PACKED(G) >> 13) & 15; }
    PG_DEV static int invulnerable_duration(const EnvHdr &G) { return (BF_PACKED(G) >> 17) & 7; }
    PG_DEV static int player_laser_theme(const EnvHdr &G) { return (BF_PACKED(G) >> 20) & 3; }
    PG_DEV static int boss_laser_theme(const EnvHdr &G) { return (BF_PACKED(G) >> 22) & 3; }
    template <class O>
    PG_DEV static float boss_bullet_vel(const O &opt) { return opt.distribution_mode == EasyMode ? (float).5 : (float).75; }

    PG_DEV static bool center_agent(const GameOptions &) { return false; }  // bossfight.cpp:211

    static void construct(EnvHdr &G) {  // bossfight.cpp:63-71
        construct_defaults(G);
        G.timeout = 4000;
        G.main_width = 20;
        G.main_height = 20;
        G.mixrate = (float).5;
        G.maxspeed = 0.85f;
    }

    // entity slots the next step can need: one trail per enemy bullet, <= 8 new bullets, the player's bullet, one
    // explosion per bullet (hits, barrier collisions) and per barrier, the damaged-mode explosion, the reserved slot
    template <class E>
    PG_DEV static int slots_needed_next_step(E &e) {
        const int n = e.G.n_ents;
        int extra = 0;
        for (int c = 0; c < ((n + 63) >> 6); c++) {
            const uint64_t eb = PG_BALLOT(l, ((c << 6) + l) < n && e.etype((c << 6) + l) == ENEMY_BULLET);
            const uint64_t ex = PG_BALLOT(l, ((c << 6) + l) < n && (e.etype((c << 6) + l) == PLAYER_BULLET || e.etype((c << 6) + l) == BARRIER));
            extra += 2 * pg_popc64(eb) + pg_popc64(ex);
        }
        return n + extra + 8 + 8 + 1 + 1 + 1 + 1;
    }

    template <class E>
    PG_DEV static int find_type(E &e, int type) {  // the boss / the shields: unique, never erased
        const int n = e.G.n_ents;
        for (int c = 0; c < ((n + 63) >> 6); c++) {
            const uint64_t m = PG_BALLOT(l, ((c << 6) + l) < n && e.etype((c << 6) + l) == type);
            if (m) return (c << 6) + pg_ctz64(m);
        }
        e.fail(PGE_ASSERT);
        return 0;
    }

    template <class E>
    PG_DEV static void handle_agent_collision(E &e, int obj) {  // bossfight.cpp:109-120
        const int t = e.etype(obj);
        if (t == BOSS || t == BARRIER || t == ENEMY_BULLET) e.G.done = 1;
    }
    template <class E>
    PG_DEV static void prepare_boss(E &e, int boss) {  // bossfight.cpp:194-201
        EnvHdr &G = e.G;
        BF_SHIELDS_UP(G) = 1;
        BF_CURR_VEL_TIMEOUT(G) = BOSS_VEL_TIMEOUT;
        BF_TIME_TO_SWAP(G) = invulnerable_duration(G);
        BF_ATTACK_MODE(G) = attack_modes_at(G, BF_ROUND_NUM(G) % num_rounds(G));
        e.evx(boss) = 0;
        e.evy(boss) = 0;
    }
    template <class E>
    PG_DEV static void handle_collision(E &e, int src, int target) {  // bossfight.cpp:129-192
        EnvHdr &G = e.G;
        const int st = e.etype(src), tt = e.etype(target);
        if (st == PLAYER_BULLET) {
            bool will_erase = false;
            if (tt == SHIELDS) {
                if (BF_SHIELDS_UP(G)) {
                    e.meta(src) = (e.meta(src) & ~M_TYPE_MASK) | (uint32_t)REFLECTED_BULLET;
                    const float theta = (float)((double)PG_PI * (1.25 + .5 * (double)BF_RAND_PCT(G)));
                    e.evy(src) = (float)(1 * pg_sin_d((double)theta) * .5);
                    e.evx(src) = (float)(1 * pg_cos_d((double)theta) * .5);
                    e.ei(EF_EXPIRE_TIME, src) = 4;
                    e.ei(EF_LIFE_TIME, src) = 0;
                    e.ef(EF_ALPHA_DECAY, src) = 0.8f;
                }
            } else if (tt == BOSS) {
                if (!BF_SHIELDS_UP(G)) {
                    e.ef(EF_HEALTH, target) -= 1;
                    will_erase = true;
                    if ((int)e.ef(EF_HEALTH, target) % round_health(G) == 0) {
                        G.reward += 1.0f;
                        if (e.ef(EF_HEALTH, target) == 0) {
                            G.done = 1;
                            G.reward += 10.0f;
                            G.level_complete = 1;
                        } else {
                            BF_ROUND_NUM(G) += 1;
                            prepare_boss(e, target);
                            BF_CURR_VEL_TIMEOUT(G) = BOSS_DAMAGED_TIMEOUT;
                            BF_DAMAGED_UNTIL(G) = G.cur_time + BOSS_DAMAGED_TIMEOUT;
                        }
                    }
                }
            }
            if (will_erase && !e.eflag(src, MF_WILL_ERASE)) {
                e.set_flag(src, MF_WILL_ERASE, true);
                const int x = e.add_entity(e.ex(src), e.ey(src), 0, 0, (float)(.5 * (double)e.erx(src)), EXPLOSION);  // spawn_child BAG:225-231
                e.evx(x) = e.evx(target);
                e.evy(x) = e.evy(target);
            }
        } else if (st == BARRIER) {
            if (tt == ENEMY_BULLET || tt == PLAYER_BULLET) {
                e.set_flag(target, MF_WILL_ERASE, true);
                e.add_entity(e.ex(target), e.ey(target), 0, 0, (float)(.5 * (double)e.erx(target)), EXPLOSION);
            } else if (tt == LASER_TRAIL) {
                e.set_flag(target, MF_WILL_ERASE, true);
            }
            if (e.ef(EF_HEALTH, src) <= 0) {
                if (!e.eflag(src, MF_WILL_ERASE)) {
                    const int x = e.add_entity(e.ex(src), e.ey(src), 0, 0, (float)(.5 * (double)e.erx(src)), EXPLOSION);
                    e.evx(x) = e.evx(src);
                    e.evy(x) = e.evy(src);
                }
                e.set_flag(src, MF_WILL_ERASE, true);
            }
        }
    }

    template <class E>
    PG_DEV static void spawn_barriers(E &e) {  // bossfight.cpp:318-336
        EnvHdr &G = e.G;
        const int num_barriers = e.randn(3) + 1;
        for (int k = 0; k < num_barriers; k++) {
            const float barrier_r = 0.6f;
            const float min_barrier_y = (float)((double)(2 * e.ery(G.agent) + barrier_r) + .5);
            const float ent_y = e.rand01() * (BOTTOM_MARGIN - min_barrier_y - barrier_r) + min_barrier_y;
            const float ent_x = e.rand01() * (G.main_width - 2 * barrier_r) + barrier_r;
            const int i = G.n_ents;
            if (i >= E::CAPACITY - 1) {
                e.fail(PGE_ENT_OVERFLOW);
                return;
            }
            e.ent_init(i, ent_x, ent_y, 0, 0, barrier_r, barrier_r, BARRIER);
            e.choose_random_theme(i);
            e.match_aspect_ratio(i);
            e.ef(EF_HEALTH, i) = 3;
            e.set_flag(i, MF_COLLIDES, true);
            PG_SYNC();
            if (!e.has_any_collision(i, 0)) G.n_ents = i + 1;
        }
    }

    template <class E>
    PG_DEV static void game_reset(E &e) {  // bossfight.cpp:203-262
        e.bag_game_reset();
        EnvHdr &G = e.G;
        BF_DAMAGED_UNTIL(G) = 0;
        BF_LAST_FIRE_TIME(G) = 0;
        const int max_extra_invulnerable = e.opt.distribution_mode == EasyMode ? 1 : 3;
        const int boss = e.add_entity((float)(G.main_width / 2), (float)(G.main_height / 2), 0, 0, BOSS_R, BOSS);
        e.choose_random_theme(boss);
        e.match_aspect_ratio(boss);
        e.add_entity_rxy(e.ex(boss), e.ey(boss), 0, 0, (float)(1.2 * (double)e.erx(boss)), (float)(1.2 * (double)e.ery(boss)), SHIELDS);
        const int rh = e.randn(9) + 1;
        const int nr = 1 + e.randn(5);
        const int inv = 2 + e.randn(max_extra_invulnerable + 1);
        e.ef(EF_HEALTH, boss) = (float)(rh * nr);
        e.choose_random_theme(G.agent);
        const int plt = e.randn(NUM_LASER_THEMES);
        const int blt = e.randn(NUM_LASER_THEMES);
        uint32_t packed = ((uint32_t)nr << 10) | ((uint32_t)rh << 13) | ((uint32_t)inv << 17) | ((uint32_t)plt << 20) | ((uint32_t)blt << 22);
        for (int i = 0; i < nr; i++) packed |= (uint32_t)e.randn(NUM_ATTACK_MODES) << (2 * i);
        BF_PACKED(G) = (int)packed;
        BF_ROUND_NUM(G) = 0;
        prepare_boss(e, boss);
        const int ag = G.agent;
        e.erx(ag) = (float).75;
        e.match_aspect_ratio(ag);
        PG_SYNC();
        e.reposition_agent();
        e.ey(ag) = e.ery(ag);
        if ((double)e.rand01() > .5) BF_PACKED(G) |= 1 << 24;  // barriers_moves_right = randbool()
        PG_SYNC();
        spawn_barriers(e);
        PG_SYNC();
    }

    template <class E>
    PG_DEV static void boss_fire(E &e, int boss, float bullet_r, float vel, float theta) {  // bossfight.cpp:264-269
        const int b = e.add_entity(e.ex(boss), e.ey(boss), (float)((double)vel * pg_cos_d((double)theta)), (float)((double)vel * pg_sin_d((double)theta)), bullet_r, ENEMY_BULLET);
        e.set_image_theme(b, boss_laser_theme(e.G));
        e.ei(EF_EXPIRE_TIME, b) = 50;
        e.ef(EF_VROT, b) = PG_PI / 8;
    }

    template <class E>
    PG_DEV static void game_step(E &e) {  // bossfight.cpp:338-414
        e.bag_game_step();
        EnvHdr &G = e.G;
        PG_SYNC();
        const int boss = find_type(e, BOSS), shields = find_type(e, SHIELDS), ag = G.agent;
        e.ex(shields) = e.ex(boss);
        e.ey(shields) = e.ey(boss);
        BF_RAND_PCT(G) = e.rand01();
        BF_RAND_FIRE_PCT(G) = e.rand01();
        BF_RAND_PCT_X(G) = e.rand01();
        BF_RAND_PCT_Y(G) = e.rand01();
        if (BF_CURR_VEL_TIMEOUT(G) <= 0) {
            const float dest_x = BF_RAND_PCT_X(G) * (G.main_width - 2 * BOSS_R) + BOSS_R;
            const float dest_y = BF_RAND_PCT_Y(G) * (G.main_height - 2 * BOSS_R - BOTTOM_MARGIN) + BOSS_R + BOTTOM_MARGIN;
            e.evx(boss) = (dest_x - e.ex(boss)) / BOSS_VEL_TIMEOUT;
            e.evy(boss) = (dest_y - e.ey(boss)) / BOSS_VEL_TIMEOUT;
            BF_CURR_VEL_TIMEOUT(G) = BOSS_VEL_TIMEOUT;
            if (BF_TIME_TO_SWAP(G) > 0) {
                BF_TIME_TO_SWAP(G) -= 1;
            } else {
                BF_TIME_TO_SWAP(G) = BF_SHIELDS_UP(G) ? VULNERABLE_DURATION : invulnerable_duration(G);
                BF_SHIELDS_UP(G) = !BF_SHIELDS_UP(G);
            }
        } else {
            BF_CURR_VEL_TIMEOUT(G) -= 1;
        }
        if (G.special_action == 1 && (G.cur_time - BF_LAST_FIRE_TIME(G)) >= 3) {
            const int b = e.add_entity(e.ex(ag), e.ey(ag), 0, 1, (float).25, PLAYER_BULLET);
            e.set_image_theme(b, player_laser_theme(G));
            e.set_flag(b, MF_COLLIDES, true);
            e.ei(EF_EXPIRE_TIME, b) = 25;
            BF_LAST_FIRE_TIME(G) = G.cur_time;
        }
        const int ct = G.cur_time;
        const float bv = boss_bullet_vel(e.opt);
        if (BF_DAMAGED_UNTIL(G) >= ct) {  // damaged_mode :309-315
            if (ct % 3 == 0) {
                const float pos_x = e.ex(boss) + (2 * BF_RAND_PCT_X(G) - 1) * e.erx(boss);
                const float pos_y = e.ey(boss) + (2 * BF_RAND_PCT_Y(G) - 1) * e.ery(boss);
                e.add_entity(pos_x, pos_y, 0, 0, (float).75, EXPLOSION);
            }
        } else if (BF_SHIELDS_UP(G)) {  // active_attack :277-327
            const int mode = BF_ATTACK_MODE(G);
            if (mode == 0) {
                if (ct % 8 == 0)
                    for (int i = 0; i < 5; i++) boss_fire(e, boss, (float).5, bv, (float)((double)PG_PI * 1.5 + (double)((i - 2) * PG_PI / 8)));
            } else if (mode == 1) {
                const int dt = 5;
                if (ct % dt == 0) {
                    int k = ct / dt;
                    k = 8 - (k % 16);
                    if (k < 0) k = -k;
                    for (int i = 0; i < 4; i++) boss_fire(e, boss, (float).5, bv, (float)((double)PG_PI * (1.25 + .5 * k / 8.0) + (double)(i * PG_PI / 2)));
                }
            } else if (mode == 2) {
                if (ct % 10 == 0) {
                    const int num_bullets = 8;
                    const float offset = BF_RAND_PCT(G) * 2 * PG_PI;
                    for (int i = 0; i < num_bullets; i++) {
                        const float theta = 2 * PG_PI / num_bullets * i + offset;
                        boss_fire(e, boss, (float).5, bv, theta);
                    }
                }
            } else if (mode == 3) {
                if (ct % 4 == 0) boss_fire(e, boss, (float).5, bv, PG_PI * (1 + BF_RAND_PCT(G)));
            }
        } else {  // passive_attack_mode :271-275
            if (BF_RAND_FIRE_PCT(G) < BASE_FIRE_PROB) boss_fire(e, boss, (float).5, bv, PG_PI * (1 + BF_RAND_PCT(G)));
        }
        PG_SYNC();
        {
            // every enemy bullet leaves a trail, highest index first; the trails are appended in that order.
            // Lane-parallel: bullet with rank r (from the top) writes slot n0 + r.
            const int n0 = G.n_ents;
            const int blt = boss_laser_theme(G);
            int appended = 0;
            for (int c = (n0 - 1) >> 6; c >= 0; c--) {
                const uint64_t m = PG_BALLOT(l, ((c << 6) + l) < n0 && e.etype((c << 6) + l) == ENEMY_BULLET);
                const int cnt = pg_popc64(m);
                if (n0 + appended + cnt > E::CAPACITY - 1) {
                    e.fail(PGE_ENT_OVERFLOW);
                    break;
                }
                PG_FOR_LANES(l) {
                    if ((m >> l) & 1ull) {
                        const int i = (c << 6) + l;
                        const int rank = pg_popc64(m >> l) - 1;  // bullets above this one in the chunk
                        const int t = n0 + appended + rank;
                        const float v_trail = (float).5;
                        e.ent_init(t, e.ex(i), e.ey(i), e.evx(i) * v_trail, e.evy(i) * v_trail, e.erx(i), e.ery(i), LASER_TRAIL);
                        e.ef(EF_ALPHA_DECAY, t) = 0.7f;
                        e.set_image_type(t, ENEMY_BULLET);
                        e.set_image_theme(t, blt);
                        e.ef(EF_VROT, t) = e.ef(EF_VROT, i);
                        e.ef(EF_ROTATION, t) = e.ef(EF_ROTATION, i);
                        e.ei(EF_EXPIRE_TIME, t) = 8;
                    }
                }
                appended += cnt;
            }
            G.n_ents = n0 + appended;
        }
        PG_SYNC();
    }

    template <class R>
    PG_DEV static bool should_draw_entity(R &r, int i) {  // bossfight.cpp:122-127
        if (r.etype(i) == SHIELDS) return BF_SHIELDS_UP(r.G) != 0;
        return true;
    }
};

}  // namespace pgamd
