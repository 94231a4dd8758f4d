#include "state_io.h"

#include <cstring>
#include <sstream>

#include "assets.h"
#include "host_state.h"

namespace pgamd {

namespace {

constexpr int32_t END_OF_BUFFER = (int32_t)0xCAFECAFE;  // reference src/vecgame.cpp:6

// EF_META packing (pg_env.h)
constexpr uint32_t M_TYPE_MASK = 0x3ffu;
constexpr int M_IMG_SHIFT = 10, M_THEME_SHIFT = 18, M_Z_SHIFT = 22;
constexpr uint32_t MF_WILL_ERASE = 1u << 24, MF_COLLIDES = 1u << 25, MF_REFLECTED = 1u << 26, MF_ABS_COORDS = 1u << 27, MF_SMART_STEP = 1u << 28,
                   MF_AVOIDS = 1u << 29, MF_AUTO_ERASE = 1u << 30;

struct Writer {  // reference src/buffer.h WriteBuffer
    char *data;
    size_t offset = 0, length;
    bool ok = true;
    void raw(const void *p, size_t n) {
        if (offset + n > length) {
            ok = false;
            return;
        }
        memcpy(data + offset, p, n);
        offset += n;
    }
    void i(int32_t v) { raw(&v, 4); }
    void f(float v) { raw(&v, 4); }
    void s(const std::string &v) {
        i((int32_t)v.size());
        raw(v.data(), v.size());
    }
};
struct Reader {  // reference src/buffer.h ReadBuffer
    const char *data;
    size_t offset = 0, length;
    bool ok = true;
    void raw(void *p, size_t n) {
        if (offset + n > length) {
            ok = false;
            memset(p, 0, n);
            return;
        }
        memcpy(p, data + offset, n);
        offset += n;
    }
    int32_t i() {
        int32_t v;
        raw(&v, 4);
        return v;
    }
    float f() {
        float v;
        raw(&v, 4);
        return v;
    }
    std::string s() {
        int32_t n = i();
        if (!ok || n < 0 || offset + (size_t)n > length) {
            ok = false;
            return std::string();
        }
        std::string v(data + offset, (size_t)n);
        offset += (size_t)n;
        return v;
    }
};

// RandGen::serialize reference src/randgen.cpp:100-107: is_seeded, then the iostream text of std::mt19937
// (libstdc++ operator<<: the 624 state words and the position, separated by single spaces)
void write_rng(Writer &w, bool seeded, const uint32_t *mt, int idx) {
    w.i(seeded ? 1 : 0);
    std::string txt;
    txt.reserve(MT_N * 11 + 8);
    char buf[16];
    for (int k = 0; k < MT_N; k++) {
        int n = snprintf(buf, sizeof buf, "%u ", mt[k]);
        txt.append(buf, (size_t)n);
    }
    int n = snprintf(buf, sizeof buf, "%d", idx);
    txt.append(buf, (size_t)n);
    w.s(txt);
}
bool read_rng(Reader &r, int *seeded, uint32_t *mt, int *idx) {
    *seeded = r.i();
    std::string txt = r.s();
    if (!r.ok) return false;
    std::istringstream is(txt);
    for (int k = 0; k < MT_N; k++) {
        unsigned long long v;
        if (!(is >> v)) return false;
        mt[k] = (uint32_t)v;
    }
    long long p;
    if (!(is >> p)) return false;
    *idx = (int)p;
    return true;
}

bool effective_center_agent(int game_id, const GameOptions &opt, const EnvHdr &h) {
    if (game_id == GAME_BIGFISH || game_id == GAME_STARPILOT || game_id == GAME_LEAPER || game_id == GAME_PLUNDER || game_id == GAME_BOSSFIGHT || game_id == GAME_CHASER) return h.initial_reset_complete ? false : opt.center_agent != 0;  // bigfish.cpp:64, starpilot.cpp:330
    if (game_id == GAME_MAZE || game_id == GAME_MINER || game_id == GAME_HEIST || game_id == GAME_DODGEBALL)  // maze.cpp:66, miner.cpp:140, heist.cpp:119
        return h.initial_reset_complete ? opt.distribution_mode == MemoryMode : opt.center_agent != 0;
    return opt.center_agent != 0;
}

}  // namespace


bool serialize_state(int game_id, const GameOptions &handle_opt, int game_n, const EnvSnapshot &s, char *data, int length, int *written, std::string *err) {
    Writer w{data, 0, (size_t)(length < 0 ? 0 : length)};
    const EnvHdr &h = s.hdr;
    const GameOptions opt = env_options(handle_opt, h.opt_bits, h.opt_debug_mode);  // the env's own options (reference: every Game has its own)
    const std::string name = game_name_from_id(game_id);
    // Game::serialize reference src/game.cpp:170-229
    w.i(0);  // SERIALIZE_VERSION
    w.s(name);
    w.i(opt.paint_vel_info);
    w.i(opt.use_generated_assets);
    w.i(opt.use_monochrome_assets);
    w.i(opt.restrict_themes);
    w.i(opt.use_backgrounds);
    w.i(effective_center_agent(game_id, opt, h) ? 1 : 0);
    w.i(opt.debug_mode);
    w.i(opt.distribution_mode);
    w.i(opt.use_sequential_levels);
    w.i(0);  // use_easy_jump
    w.i(0);  // plain_assets
    w.i(0);  // physics_mode
    w.i(h.grid_step);
    w.i(s.hdr.level_seed_low);
    w.i(s.hdr.level_seed_high);
    w.i(0);  // game_type
    w.i(game_n);
    write_rng(w, true, s.rng.data() + MT_STRIDE, h.lvl_rand_idx);  // level_seed_rand_gen
    write_rng(w, h.initial_reset_complete != 0, s.rng.data(), h.rand_idx);
    w.f(h.reward);
    w.i(h.done);
    w.i(h.level_complete);
    w.i(h.action);
    w.i(h.timeout);
    w.i(h.current_level_seed);
    w.i(h.prev_level_seed);
    w.i(h.episodes_remaining);
    w.i(h.episode_done);
    w.i(h.last_reward_timer);
    w.f(h.last_reward);
    w.i(h.default_action);
    w.i((int32_t)hash_str_uint32(name));  // fixed_asset_seed, reference src/vecgame.cpp:324-327
    w.i(h.cur_time);
    w.i(0);  // is_waiting_for_step
    // BasicAbstractGame::serialize BAG:1169-1223
    w.i(h.main_width * h.main_height);  // grid_size
    w.i(h.n_ents);
    auto write_entities = [&](const uint32_t *tab, int cap, int count) {
    auto W = [&](int f, int i) { return tab[(size_t)f * cap + i]; };
    for (int i = 0; i < count; i++) {  // Entity::serialize reference src/entity.cpp:90-137
        const uint32_t m = W(EF_META, i);
        auto wf = [&](int f) {
            uint32_t v = W(f, i);
            w.raw(&v, 4);
        };
        wf(EF_X); wf(EF_Y); wf(EF_VX); wf(EF_VY); wf(EF_RX); wf(EF_RY);
        w.i((int32_t)(m & M_TYPE_MASK));
        w.i((int32_t)((m >> M_IMG_SHIFT) & 0xffu));
        w.i((int32_t)((m >> M_THEME_SHIFT) & 0xfu));
        w.i((int32_t)((m >> M_Z_SHIFT) & 3u) - 1);
        w.i((m & MF_WILL_ERASE) != 0);
        w.i((m & MF_COLLIDES) != 0);
        wf(EF_COLLISION_MARGIN); wf(EF_ROTATION); wf(EF_VROT);
        w.i((m & MF_REFLECTED) != 0);
        wf(EF_FIRE_TIME); wf(EF_SPAWN_TIME); wf(EF_LIFE_TIME); wf(EF_EXPIRE_TIME);
        w.i((m & MF_ABS_COORDS) != 0);
        wf(EF_FRICTION);
        w.i((m & MF_SMART_STEP) != 0);
        w.i((m & MF_AVOIDS) != 0);
        w.i((m & MF_AUTO_ERASE) != 0);
        wf(EF_ALPHA); wf(EF_HEALTH); wf(EF_THETA); wf(EF_GROW_RATE); wf(EF_ALPHA_DECAY); wf(EF_CLIMBER_SPAWN_X);
    }
    };
    write_entities(s.ents.data(), s.ent_cap, h.n_ents);
    w.i(0);  // use_procgen_background
    w.i(h.background_index);
    w.f(h.bg_tile_ratio);
    w.f(h.bg_pct_x);
    w.f(5.0f);  // char_dim, BAG:24
    w.i(h.last_move_action);
    w.i(h.move_action);
    w.i(h.special_action);
    w.f(h.mixrate);
    w.f(h.maxspeed);
    w.f(h.max_jump);
    w.f(h.action_vx);
    w.f(h.action_vy);
    w.f(h.action_vrot);
    w.f(h.center_x);
    w.f(h.center_y);
    w.i(h.random_agent_start);
    w.i(h.has_useful_vel_info);
    w.i(h.step_rand_int);
    {   // asset_rand_gen is never seeded on the PNG-asset path: is_seeded = 0 + the default-constructed mt19937 (seed 5489)
        HostMT def;
        def.seed(5489);
        write_rng(w, false, def.mt, def.idx);
    }
    w.i(h.main_width);
    w.i(h.main_height);
    w.i(h.out_of_bounds_object);
    w.f(h.unit);
    w.f(h.view_dim);
    w.f(h.x_off);
    w.f(h.y_off);
    w.f(h.visibility);
    w.f(h.min_visibility);
    w.i(h.main_width);  // Grid::serialize reference src/grid.h:69-73
    w.i(h.main_height);
    const int cells = h.main_width * h.main_height;
    w.i(cells);
    const bool wide_cells = game_id == GAME_CHASER;  // u16 cells (object ids above 255)
    if ((size_t)cells * (wide_cells ? 2 : 1) > s.grid.size()) {
        if (err) *err = "grid larger than the snapshot";
        return false;
    }
    auto cell_at = [&](int c) -> int { return wide_cells ? (int)reinterpret_cast<const uint16_t *>(s.grid.data())[c] : (int)s.grid[c]; };
    for (int c = 0; c < cells; c++) w.i((int32_t)cell_at(c));
    // game tails
    if (game_id == GAME_COINRUN) {  // reference src/games/coinrun.cpp:500-509
        w.f(h.gsf0);
        w.i(h.gsi0);
        w.i(h.gsi1 ? 1 : 0);
        w.i(h.gsi2 ? 1 : 0);
        w.i(h.gsi3 ? 1 : 0);
        w.f(h.gsf1);
        w.f(h.gsf2);
    } else if (game_id == GAME_BIGFISH) {  // reference src/games/bigfish.cpp:109-113
        w.i(h.gsi0);
        w.f(h.gsf0);
    } else if (game_id == GAME_MAZE) {  // reference src/games/maze.cpp:126-130
        w.i(h.gsi0);
        w.i(h.gsi1);
    } else if (game_id == GAME_MINER) {  // reference src/games/miner.cpp:309-312
        w.i(h.gsi0);
    } else if (game_id == GAME_JUMPER) {  // reference src/games/jumper.cpp:451-460
        w.i(h.gsi0);
        w.i(h.gsi1);
        w.i(h.gsi2);
        w.i(h.gsi3 ? 1 : 0);
        w.i(h.gsi4 ? 1 : 0);
        w.i(h.gsi5);
        w.f(h.gsf0);
    } else if (game_id == GAME_CHASER) {  // reference src/games/chaser.cpp:392-403: free_cells / is_space_vec follow from the walls
        const int MAZE_WALL = 5;
        int nfree = 0;
        for (int c = 0; c < cells; c++) nfree += cell_at(c) != MAZE_WALL;
        w.i(nfree);
        for (int c = 0; c < cells; c++)
            if (cell_at(c) != MAZE_WALL) w.i(c);
        w.i(cells);
        for (int c = 0; c < cells; c++) w.i(cell_at(c) != MAZE_WALL ? 1 : 0);
        w.i(75);       // eat_timeout
        w.i(50);       // egg_timeout
        w.i(h.gsi0);   // eat_time
        w.i(h.gsi4);   // total_enemies
        w.i(h.gsi1);   // total_orbs
        w.i(h.gsi2);   // orbs_collected
        w.i(h.gsi3);   // maze_dim
    } else if (game_id == GAME_BOSSFIGHT) {  // reference src/games/bossfight.cpp:416-441
        const int p = h.gsi5, nr = (p >> 10) & 7;
        w.i(nr);
        for (int k = 0; k < nr; k++) w.i((p >> (2 * k)) & 3);
        w.i(h.gsi0);              // last_fire_time
        w.i(h.gsi1);              // time_to_swap
        w.i((p >> 17) & 7);       // invulnerable_duration
        w.i(500);                 // vulnerable_duration
        w.i(nr);                  // num_rounds
        w.i(h.gsi4);              // round_num
        w.i((p >> 13) & 15);      // round_health
        w.i(20);                  // boss_vel_timeout
        w.i(h.gsi2);              // curr_vel_timeout
        w.i(h.gsi7);              // attack_mode
        w.i((p >> 20) & 3);       // player_laser_theme
        w.i((p >> 22) & 3);       // boss_laser_theme
        w.i(h.gsi3);              // damaged_until_time
        w.i(h.gsi6 ? 1 : 0);      // shields_are_up
        w.i((p >> 24) & 1);       // barriers_moves_right
        w.f(0.1f);                // base_fire_prob
        w.f(opt.distribution_mode == EasyMode ? .5f : .75f);  // boss_bullet_vel
        w.f(0.1f);                // barrier_vel
        w.f(0.025f);              // barrier_spawn_prob
        w.f(h.gsf0);
        w.f(h.gsf1);
        w.f(h.gsf2);
        w.f(h.gsf3);
    } else if (game_id == GAME_DODGEBALL) {  // reference src/games/dodgeball.cpp:450-459
        w.f(h.gsf0);
        w.f(h.gsf1);
        w.f(h.gsf2);
        w.f(h.gsf3);
        w.i(h.gsi0);
        w.i(h.gsi1);
        w.i(h.gsi2);
    } else if (game_id == GAME_NINJA) {  // reference src/games/ninja.cpp:385-395
        w.i(h.gsi0 ? 1 : 0);
        w.i(h.gsi1 ? 1 : 0);
        w.i(h.gsi2);
        w.i(h.gsi3);
        w.f(h.gsf0);
        w.f(h.gsf1);
        w.f(h.gsf2);
        w.f(h.gsf3);
    } else if (game_id == GAME_HEIST) {  // reference src/games/heist.cpp:202-207
        w.i(h.gsi0);
        w.i(h.gsi1);
        w.i(h.gsi0);  // has_keys.size() == num_keys
        for (int k = 0; k < h.gsi0; k++) w.i((h.gsi2 >> k) & 1);
    } else if (game_id == GAME_PLUNDER) {  // reference src/games/plunder.cpp:241-257 (vectors = count + values, bools as ints)
        const float vels[5] = {h.gsf0, h.gsf1, h.gsf2, h.gsf3, h.gsf4};
        w.i(h.gsi0);
        w.i(5);
        for (int k = 0; k < 5; k++) w.i((h.gsi2 >> k) & 1);
        w.i(6);
        for (int k = 0; k < 6; k++) w.i((h.gsi3 >> k) & 1);
        w.i(6);
        for (int k = 0; k < 6; k++) w.i((h.gsi4 >> (3 * k)) & 7);
        w.i(5);
        for (int k = 0; k < 5; k++) w.f(vels[k]);
        w.i(5);   // num_lanes
        w.i(2);   // num_current_ship_types
        w.i(h.gsi1);
        w.i(20);  // target_quota
        w.f(h.gsf5);
        w.f(opt.distribution_mode == EasyMode ? 1.5f : 1.0f);  // r_scale
        w.f(0.06f);                                             // spawn_prob
        w.f(2.0f);                                              // legend_r
        w.f(h.gsf6);
    } else if (game_id == GAME_LEAPER) {  // reference src/games/leaper.cpp:277-284 (write_vector_float = count + values)
        const float road[5] = {h.gsf0, h.gsf1, h.gsf2, h.gsf3, h.gsf4};
        float water[5] = {h.gsf5, h.gsf6, h.gsf7, 0, 0};
        memcpy(&water[3], &h.gsi5, 4);
        memcpy(&water[4], &h.gsi6, 4);
        if (h.gsi1 < 0 || h.gsi1 > 5 || h.gsi3 < 0 || h.gsi3 > 5) {
            if (err) *err = "leaper: malformed lane tables";
            return false;
        }
        w.i(h.gsi0);
        w.i(h.gsi1);
        for (int k = 0; k < h.gsi1; k++) w.f(road[k]);
        w.i(h.gsi2);
        w.i(h.gsi3);
        for (int k = 0; k < h.gsi3; k++) w.f(water[k]);
        w.i(h.gsi4);
    } else if (game_id == GAME_FRUITBOT) {  // reference src/games/fruitbot.cpp:264-269
        w.f(h.gsf0);
        w.f(h.gsf1);
        w.i(h.gsi0);
    } else if (game_id == GAME_STARPILOT) {  // reference src/games/starpilot.cpp:432-435: write_entities(spawners)
        const int cell_bytes = (16 * 16 + 15) & ~15;
        const int spawn_cap = 256;
        const uint32_t *tab = reinterpret_cast<const uint32_t *>(s.grid.data() + cell_bytes);
        if ((size_t)cell_bytes + (size_t)EF_COUNT * spawn_cap * 4 > s.grid.size() || h.gsi0 < 0 || h.gsi0 > spawn_cap) {
            if (err) *err = "starpilot: malformed spawner table";
            return false;
        }
        w.i(h.gsi0);
        write_entities(tab, spawn_cap, h.gsi0);
    } else if (game_id == GAME_CLIMBER) {  // reference src/games/climber.cpp:318-327
        w.i(h.gsi1 ? 1 : 0);
        w.i(h.gsi2 ? 1 : 0);
        w.i(h.gsi3);
        w.i(h.gsi4);
        w.i(h.gsi0);
        w.f(h.gsf1);
        w.f(h.gsf2);
    }
    w.i(END_OF_BUFFER);
    if (!w.ok) {
        if (err) *err = "fassert failed 'offset + sizeof(int) <= length' (state buffer too small)";
        return false;
    }
    *written = (int)w.offset;
    return true;
}

bool deserialize_state(int game_id, const GameOptions &opt, EnvSnapshot *s, const char *data, int length, std::string *err, int *game_n_out) {
    Reader r{data, 0, (size_t)(length < 0 ? 0 : length)};
    EnvHdr &h = s->hdr;
    auto bad = [&](const char *what) {
        if (err) *err = what;
        return false;
    };
    if (r.i() != 0) return bad("fassert failed 'SERIALIZE_VERSION == b->read_int()'");
    if (r.s() != game_name_from_id(game_id)) return bad("fassert failed 'game_name == b->read_string()'");
    // reference src/game.cpp:233-246: the env adopts the options the state was saved under -- per env here (EnvHdr::opt_bits /
    // opt_debug_mode), the distribution_mode included when the handle's kernel instantiation serves it (every mode of a game but
    // caveflyer's memory mode, pg_defs.h kernel_id_for).  use_generated_assets selects the assets and the renderer of the whole handle: a
    // state that differs in it is refused, with the reason.
    int o[9];
    for (int k = 0; k < 9; k++) o[k] = r.i();
    if (o[7] != opt.distribution_mode) {
        const bool known = o[7] == EasyMode || o[7] == HardMode || (o[7] == ExtremeMode && game_has_extreme_mode(game_id)) || (o[7] == MemoryMode && game_has_memory_mode(game_id));
        if (!known) return bad("set_state: the state carries a distribution_mode this game does not have");
        if (kernel_id_for(game_id, o[7]) != kernel_id_for(game_id, opt.distribution_mode))
            return bad("set_state: the state was saved under another distribution_mode than this handle's (caveflyer's memory mode has kernels of its own; make a handle with that mode)");
    }
    if (o[1] != opt.use_generated_assets) return bad("set_state: the state was saved under another use_generated_assets setting than this handle's (the option selects the assets and kernels of the handle)");
    {
        GameOptions eo = opt;
        eo.paint_vel_info = o[0] != 0;
        eo.use_monochrome_assets = o[2] != 0;
        eo.restrict_themes = o[3] != 0;
        eo.use_backgrounds = o[4] != 0;
        eo.center_agent = o[5] != 0;
        eo.use_sequential_levels = o[8] != 0;
        eo.distribution_mode = o[7];
        h.opt_bits = env_option_bits(eo);
        h.opt_debug_mode = o[6];
    }
    r.i();  // use_easy_jump
    r.i();  // plain_assets
    r.i();  // physics_mode
    h.grid_step = r.i();
    h.level_seed_low = r.i();  // adopted per env, as the reference does (src/game.cpp:247-248): the env goes on drawing its levels from the range the state was saved under
    h.level_seed_high = r.i();
    // (an empty range is adopted like any other, as the reference does: its next reset that draws a level seed divides by the range,
    // RandGen::randint src/randgen.cpp:6-11 called from src/game.cpp:101 -- a zero range ends the process there (SIGFPE; here the device-side
    // check of game_reset_full), a negative one wraps in unsigned arithmetic, which the kernels' draw restates)
    r.i();  // game_type
    {
        const int saved_game_n = r.i();  // game_n: adopted (reference src/game.cpp:253); it only names the env in a warning there, but get_state writes it back
        if (game_n_out) *game_n_out = saved_game_n;
    }
    int seeded;
    if (!read_rng(r, &seeded, s->rng.data() + MT_STRIDE, &h.lvl_rand_idx)) return bad("set_state: malformed level_seed_rand_gen");
    if (!read_rng(r, &seeded, s->rng.data(), &h.rand_idx)) return bad("set_state: malformed rand_gen");
    h.initial_reset_complete = 1;
    h.reward = r.f();
    h.done = r.i();
    h.level_complete = r.i();
    h.action = r.i();
    h.timeout = r.i();
    h.current_level_seed = r.i();
    h.prev_level_seed = r.i();
    h.episodes_remaining = r.i();
    h.episode_done = r.i();
    h.last_reward_timer = r.i();
    h.last_reward = r.f();
    h.default_action = r.i();
    r.i();  // fixed_asset_seed
    h.cur_time = r.i();
    r.i();  // is_waiting_for_step
    r.i();  // grid_size
    const int n = r.i();
    const int cap = s->ent_cap;
    if (!r.ok || n < 0 || n > cap - 1) return bad("set_state: entity count exceeds the table capacity");
    h.n_ents = n;
    h.agent = -1;
    int last_player = -1;
    auto read_entities = [&](uint32_t *tab, int cap, int n) {
    auto W = [&](int f, int i) -> uint32_t & { return tab[(size_t)f * cap + i]; };
    for (int i = 0; i < n; i++) {
        auto rf = [&](int f) { r.raw(&W(f, i), 4); };
        rf(EF_X); rf(EF_Y); rf(EF_VX); rf(EF_VY); rf(EF_RX); rf(EF_RY);
        const int type = r.i(), image_type = r.i(), image_theme = r.i(), render_z = r.i();
        uint32_t m = ((uint32_t)type & M_TYPE_MASK) | (((uint32_t)image_type & 0xffu) << M_IMG_SHIFT) | (((uint32_t)image_theme & 0xfu) << M_THEME_SHIFT) |
                     (((uint32_t)(render_z + 1) & 3u) << M_Z_SHIFT);
        if (r.i()) m |= MF_WILL_ERASE;
        if (r.i()) m |= MF_COLLIDES;
        rf(EF_COLLISION_MARGIN); rf(EF_ROTATION); rf(EF_VROT);
        if (r.i()) m |= MF_REFLECTED;
        rf(EF_FIRE_TIME); rf(EF_SPAWN_TIME); rf(EF_LIFE_TIME); rf(EF_EXPIRE_TIME);
        if (r.i()) m |= MF_ABS_COORDS;
        rf(EF_FRICTION);
        if (r.i()) m |= MF_SMART_STEP;
        if (r.i()) m |= MF_AVOIDS;
        if (r.i()) m |= MF_AUTO_ERASE;
        rf(EF_ALPHA); rf(EF_HEALTH); rf(EF_THETA); rf(EF_GROW_RATE); rf(EF_ALPHA_DECAY); rf(EF_CLIMBER_SPAWN_X);
        W(EF_META, i) = m;
        if (type == PLAYER) last_player = i;
    }
    };
    read_entities(s->ents.data(), cap, n);
    h.agent = last_player;  // find_entity_index returns the LAST match, BAG:1133-1143
    if (h.agent < 0) return bad("fassert failed 'agent_idx >= 0'");
    r.i();  // use_procgen_background
    h.background_index = r.i();
    h.bg_tile_ratio = r.f();
    h.bg_pct_x = r.f();
    r.f();  // char_dim
    h.last_move_action = r.i();
    h.move_action = r.i();
    h.special_action = r.i();
    h.mixrate = r.f();
    h.maxspeed = r.f();
    h.max_jump = r.f();
    h.action_vx = r.f();
    h.action_vy = r.f();
    h.action_vrot = r.f();
    h.center_x = r.f();
    h.center_y = r.f();
    h.random_agent_start = r.i();
    h.has_useful_vel_info = r.i();
    h.step_rand_int = r.i();
    {
        uint32_t tmp[MT_N];
        int idx;
        if (!read_rng(r, &seeded, tmp, &idx)) return bad("set_state: malformed asset_rand_gen");
    }
    h.main_width = r.i();
    h.main_height = r.i();
    h.out_of_bounds_object = r.i();
    h.unit = r.f();
    h.view_dim = r.f();
    h.x_off = r.f();
    h.y_off = r.f();
    h.visibility = r.f();
    h.min_visibility = r.f();
    const int gw = r.i(), gh = r.i(), cnt = r.i();
    const bool wide_cells = game_id == GAME_CHASER;
    if (!r.ok || gw != h.main_width || gh != h.main_height || cnt != gw * gh || (size_t)cnt * (wide_cells ? 2 : 1) > s->grid.size()) return bad("set_state: malformed grid");
    for (int c = 0; c < cnt; c++) {
        if (wide_cells) reinterpret_cast<uint16_t *>(s->grid.data())[c] = (uint16_t)r.i();
        else s->grid[c] = (uint8_t)r.i();
    }
    if (game_id == GAME_COINRUN) {
        h.gsf0 = r.f();
        h.gsi0 = r.i();
        h.gsi1 = r.i() > 0;
        h.gsi2 = r.i() > 0;
        h.gsi3 = r.i() > 0;
        h.gsf1 = r.f();
        h.gsf2 = r.f();
    } else if (game_id == GAME_BIGFISH) {
        h.gsi0 = r.i();
        h.gsf0 = r.f();
    } else if (game_id == GAME_MAZE) {
        h.gsi0 = r.i();
        h.gsi1 = r.i();
    } else if (game_id == GAME_MINER) {
        h.gsi0 = r.i();
    } else if (game_id == GAME_JUMPER) {
        h.gsi0 = r.i();
        h.gsi1 = r.i();
        h.gsi2 = r.i();
        h.gsi3 = r.i() > 0;
        h.gsi4 = r.i() > 0;
        h.gsi5 = r.i();
        h.gsf0 = r.f();
    } else if (game_id == GAME_CHASER) {
        const int nf = r.i();
        if (!r.ok || nf < 0 || nf > cnt) return bad("set_state: chaser free_cells");
        for (int k = 0; k < nf; k++) r.i();
        const int nsv = r.i();
        if (!r.ok || nsv != cnt) return bad("set_state: chaser is_space_vec");
        for (int k = 0; k < nsv; k++) r.i();
        r.i();  // eat_timeout
        r.i();  // egg_timeout
        h.gsi0 = r.i();
        h.gsi4 = r.i();
        h.gsi1 = r.i();
        h.gsi2 = r.i();
        h.gsi3 = r.i();
        h.gsi5 = nf;
    } else if (game_id == GAME_BOSSFIGHT) {
        const int nm = r.i();
        if (!r.ok || nm < 0 || nm > 5) return bad("set_state: bossfight attack_modes");
        uint32_t p = 0;
        for (int k = 0; k < nm; k++) p |= ((uint32_t)r.i() & 3u) << (2 * k);
        h.gsi0 = r.i();
        h.gsi1 = r.i();
        p |= ((uint32_t)r.i() & 7u) << 17;   // invulnerable_duration
        r.i();                                // vulnerable_duration
        p |= ((uint32_t)r.i() & 7u) << 10;   // num_rounds
        h.gsi4 = r.i();
        p |= ((uint32_t)r.i() & 15u) << 13;  // round_health
        r.i();                                // boss_vel_timeout
        h.gsi2 = r.i();
        h.gsi7 = r.i();
        p |= ((uint32_t)r.i() & 3u) << 20;
        p |= ((uint32_t)r.i() & 3u) << 22;
        h.gsi3 = r.i();
        h.gsi6 = r.i() > 0;
        p |= (r.i() > 0 ? 1u : 0u) << 24;
        h.gsi5 = (int)p;
        r.f(); r.f(); r.f(); r.f();
        h.gsf0 = r.f();
        h.gsf1 = r.f();
        h.gsf2 = r.f();
        h.gsf3 = r.f();
    } else if (game_id == GAME_DODGEBALL) {
        h.gsf0 = r.f();
        h.gsf1 = r.f();
        h.gsf2 = r.f();
        h.gsf3 = r.f();
        h.gsi0 = r.i();
        h.gsi1 = r.i();
        h.gsi2 = r.i();
    } else if (game_id == GAME_NINJA) {
        h.gsi0 = r.i() > 0;
        h.gsi1 = r.i() > 0;
        h.gsi2 = r.i();
        h.gsi3 = r.i();
        h.gsf0 = r.f();
        h.gsf1 = r.f();
        h.gsf2 = r.f();
        h.gsf3 = r.f();
    } else if (game_id == GAME_HEIST) {
        h.gsi0 = r.i();
        h.gsi1 = r.i();
        const int nk = r.i();
        if (!r.ok || nk < 0 || nk > 3) return bad("set_state: heist has_keys");
        h.gsi2 = 0;
        for (int k = 0; k < nk; k++) h.gsi2 |= (r.i() ? 1 : 0) << k;
    } else if (game_id == GAME_PLUNDER) {
        h.gsi0 = r.i();
        if (r.i() != 5) return bad("set_state: plunder lane_directions");
        h.gsi2 = 0;
        for (int k = 0; k < 5; k++) h.gsi2 |= (r.i() ? 1 : 0) << k;
        if (r.i() != 6) return bad("set_state: plunder target_bools");
        h.gsi3 = 0;
        for (int k = 0; k < 6; k++) h.gsi3 |= (r.i() ? 1 : 0) << k;
        if (r.i() != 6) return bad("set_state: plunder image_permutation");
        h.gsi4 = 0;
        for (int k = 0; k < 6; k++) h.gsi4 |= (r.i() & 7) << (3 * k);
        if (r.i() != 5) return bad("set_state: plunder lane_vels");
        float vels[5];
        for (int k = 0; k < 5; k++) vels[k] = r.f();
        h.gsf0 = vels[0]; h.gsf1 = vels[1]; h.gsf2 = vels[2]; h.gsf3 = vels[3]; h.gsf4 = vels[4];
        r.i();  // num_lanes
        r.i();  // num_current_ship_types
        h.gsi1 = r.i();
        r.i();  // target_quota
        h.gsf5 = r.f();
        r.f();  // r_scale
        r.f();  // spawn_prob
        r.f();  // legend_r
        h.gsf6 = r.f();
    } else if (game_id == GAME_LEAPER) {
        float road[5] = {0, 0, 0, 0, 0}, water[5] = {0, 0, 0, 0, 0};
        h.gsi0 = r.i();
        h.gsi1 = r.i();
        if (!r.ok || h.gsi1 < 0 || h.gsi1 > 5) return bad("set_state: leaper road lane count");
        for (int k = 0; k < h.gsi1; k++) road[k] = r.f();
        h.gsi2 = r.i();
        h.gsi3 = r.i();
        if (!r.ok || h.gsi3 < 0 || h.gsi3 > 5) return bad("set_state: leaper water lane count");
        for (int k = 0; k < h.gsi3; k++) water[k] = r.f();
        h.gsi4 = r.i();
        h.gsf0 = road[0]; h.gsf1 = road[1]; h.gsf2 = road[2]; h.gsf3 = road[3]; h.gsf4 = road[4];
        h.gsf5 = water[0]; h.gsf6 = water[1]; h.gsf7 = water[2];
        memcpy(&h.gsi5, &water[3], 4);
        memcpy(&h.gsi6, &water[4], 4);
    } else if (game_id == GAME_FRUITBOT) {
        h.gsf0 = r.f();
        h.gsf1 = r.f();
        h.gsi0 = r.i();
    } else if (game_id == GAME_STARPILOT) {  // read_entities(spawners), starpilot.cpp:437-442
        const int cell_bytes = (16 * 16 + 15) & ~15;
        const int spawn_cap = 256;
        const int ns = r.i();
        if (!r.ok || ns < 0 || ns > spawn_cap || (size_t)cell_bytes + (size_t)EF_COUNT * spawn_cap * 4 > s->grid.size())
            return bad("set_state: spawner count exceeds the table capacity");
        uint32_t *tab = reinterpret_cast<uint32_t *>(s->grid.data() + cell_bytes);
        read_entities(tab, spawn_cap, ns);
        h.gsi0 = ns;
        h.gsi1 = ns > 0 ? (int)tab[(size_t)EF_SPAWN_TIME * spawn_cap + (ns - 1)] : -1;
    } else if (game_id == GAME_CLIMBER) {
        h.gsi1 = r.i() > 0;
        h.gsi2 = r.i() > 0;
        h.gsi3 = r.i();
        h.gsi4 = r.i();
        h.gsi0 = r.i();
        h.gsf1 = r.f();
        h.gsf2 = r.f();
    }
    if (r.i() != END_OF_BUFFER || !r.ok) return bad("fassert failed 'b.read_int() == END_OF_BUFFER'");
    h.error = 0;
    h.grid_dirty = 0;
    return true;
}

}  // namespace pgamd
