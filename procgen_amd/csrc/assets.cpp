#include "assets.h"

#include "host_state.h"
#include "pg_assetgen.h"

#include <sys/stat.h>

#include <algorithm>
#include <cctype>
#include <cstring>

namespace pgamd {

static const char *GAME_NAMES[NUM_GAMES] = {"bigfish", "bossfight", "caveflyer", "chaser", "climber", "coinrun", "dodgeball", "fruitbot",
                                             "heist", "jumper", "leaper", "maze", "miner", "ninja", "plunder", "starpilot"};

uint32_t hash_str_uint32(const std::string &str) {  // reference src/vecgame.cpp:156-167
    uint32_t hash = 0x811c9dc5u;
    for (unsigned char c : str) {
        hash ^= c;
        hash *= 0x1000193u;
    }
    return hash;
}

void generate_game_assets(const std::string &game_name, bool (*use_block_asset)(int type), HostAssets *out) {
    GameAssetsDev &t = out->table;
    memset(&t, 0, sizeof(t));
    out->pixels.assign((size_t)MAX_ASSETS * 64 * 64, 0u);
    out->image_names.clear();
    struct Mt {
        HostMT m;
        uint32_t u32() { return m.next(); }
    };
    for (int type = 0; type < MAX_ASSETS; type++) {
        Mt rng;
        rng.m.seed((int)(hash_str_uint32(game_name) + (uint32_t)type));  // asset_rand_gen.seed(fixed_asset_seed + type), BAG:101
        int cnt[64], xa[64];
        assetgen::MemPainter mp{out->pixels.data() + (size_t)type * 4096, 64, 64, cnt, xa, 0u, 0u};
        assetgen::Gen<Mt, assetgen::MemPainter> gen{rng, mp};
        gen.generate_resource(64, 64, 0, 5, use_block_asset(type));
        t.img[type].off = (uint32_t)type * 4096u;
        t.img[type].w = t.img[type].h = 64;
        bool opaque = true;
        for (int k = 0; k < 4096; k++) opaque = opaque && (out->pixels[(size_t)type * 4096 + k] >> 24) == 0xffu;
        t.img[type].opaque = opaque ? 1u : 0u;
        for (int th = 0; th < MAX_IMAGE_THEMES; th++) t.type_theme_img[type][th] = (int16_t)type;
        t.type_num_themes[type] = 1;
        out->image_names.push_back("generated:" + std::to_string(type));
    }
    t.img[MAX_ASSETS].off = 0;  // the background canvas: pixels in DevCtx::gen_bg
    t.img[MAX_ASSETS].w = t.img[MAX_ASSETS].h = 500;
    t.img[MAX_ASSETS].opaque = 1;
    out->image_names.push_back("generated:background|bg");
    t.n_bg = 1;
    t.bg_img[0] = (int16_t)MAX_ASSETS;
    t.ref_w = t.ref_h = 64;
    finish_asset_tables(t);
}

int game_id_from_name(const std::string &name) {
    for (int i = 0; i < NUM_GAMES; i++)
        if (name == GAME_NAMES[i]) return i;
    return -1;
}
const char *game_name_from_id(int id) { return (id >= 0 && id < NUM_GAMES) ? GAME_NAMES[id] : "?"; }

static std::string lower(std::string s) {
    std::transform(s.begin(), s.end(), s.begin(), [](unsigned char c) { return (char)std::tolower(c); });
    return s;
}

// background groups: reference src/resources.cpp:817-953
static const char *SPACE_BGS[] = {"deep_space_01", "spacegen_01", "milky_way_01", "ez_space_lite_01", "meyespace_v1_01", "eye_nebula_01", "deep_sky_01",
                                  "space_nebula_01", "Background-1", "Background-2", "Background-3", "Background-4", "parallax-space-backgound"};
static const char *PLATFORM_BGS[] = {"alien_bg", "another_world_bg", "back_cave", "caverns", "cyberpunk_bg", "parallax_forest", "scifi_bg", "scifi2_bg",
                                     "living_tissue_bg", "airadventurelevel1", "airadventurelevel2", "airadventurelevel3", "airadventurelevel4",
                                     "cave_background", "blue_desert", "blue_grass", "blue_land", "blue_shroom", "colored_desert", "colored_grass",
                                     "colored_land", "colored_shroom", "landscape1", "landscape2", "landscape3", "landscape4", "battleback1",
                                     "battleback2", "battleback3", "battleback4", "battleback5", "battleback6", "battleback7", "battleback8",
                                     "battleback9", "battleback10", "sunrise"};
static const char *PLATFORM2_BGS[] = {"beach1", "beach2", "beach3", "beach4", "fantasy1", "fantasy2", "fantasy3", "fantasy4", "candy1", "candy2", "candy3", "candy4"};

static void platform_backgrounds(std::vector<std::string> *out) {
    for (const char *n : PLATFORM_BGS) out->push_back(std::string("platform_backgrounds/") + n + ".png");
    for (const char *n : PLATFORM2_BGS) out->push_back(std::string("platform_backgrounds_2/") + n + ".png");
    for (const char *n : SPACE_BGS) out->push_back(std::string("space_backgrounds/") + n + ".png");  // resources.cpp:950-953
}

static void topdown_backgrounds(std::vector<std::string> *out) {  // reference src/resources.cpp:900-911
    out->push_back("topdown_backgrounds/floortiles.png");
    for (int k = 1; k <= 8; k++) out->push_back("topdown_backgrounds/backgrounddetailed" + std::to_string(k) + ".png");
}

static void reserved_assets(std::vector<SpriteName> *s) {  // reference BAG:416-430
    for (int k = 0; k < 5; k++) s->push_back({EXPLOSION + k, 0, "misc_assets/explosion" + std::to_string(k + 1) + ".png"});
    s->push_back({TRAIL, 0, "misc_assets/iconCircle_white.png"});
}

bool game_asset_names(int game_id, std::vector<SpriteName> *sprites, std::vector<std::string> *backgrounds) {
    sprites->clear();
    backgrounds->clear();
    auto add_themes = [&](int type, const std::vector<std::string> &paths) {
        for (size_t t = 0; t < paths.size(); t++) sprites->push_back({type, (int)t, paths[t]});
    };
    if (game_id == GAME_COINRUN) {  // reference src/games/coinrun.cpp:33-35,60-62,72-121
        const std::vector<std::string> enemies = {"slimeBlock", "slimePurple", "slimeBlue", "slimeGreen", "mouse", "snail", "ladybug", "wormGreen", "wormPink"};
        const std::vector<std::string> colors = {"Beige", "Blue", "Green", "Pink", "Yellow"};
        const std::vector<std::string> grounds = {"Dirt", "Grass", "Planet", "Sand", "Snow", "Stone"};
        const int ptypes[4] = {0, 9, 12, 13};
        const char *pnames[4] = {"stand", "jump", "walk1", "walk2"};
        for (int k = 0; k < 4; k++) {
            std::vector<std::string> v;
            for (auto &c : colors) v.push_back("kenney/Players/128x256/" + c + "/alien" + c + "_" + pnames[k] + ".png");
            add_themes(ptypes[k], v);
        }
        std::vector<std::string> e1, e2, top, mid;
        for (auto &e : enemies) e1.push_back("kenney/Enemies/" + e + ".png");
        for (auto &e : enemies) e2.push_back("kenney/Enemies/" + e + "_move.png");
        for (auto &g : grounds) top.push_back("kenney/Ground/" + g + "/" + lower(g) + "Mid.png");
        for (auto &g : grounds) mid.push_back("kenney/Ground/" + g + "/" + lower(g) + "Center.png");
        add_themes(6, e1);
        add_themes(7, e2);
        add_themes(1, {"kenney/Items/coinGold.png"});
        add_themes(16, top);
        add_themes(15, mid);
        add_themes(18, {"kenney/Tiles/lavaTop_low.png"});
        add_themes(17, {"kenney/Tiles/lava.png"});
        add_themes(2, {"kenney/Enemies/sawHalf.png"});
        add_themes(3, {"kenney/Enemies/sawHalf_move.png"});
        add_themes(20, {"kenney/Tiles/boxCrate.png", "kenney/Tiles/boxCrate_double.png", "kenney/Tiles/boxCrate_single.png", "kenney/Tiles/boxCrate_warning.png"});
        platform_backgrounds(backgrounds);
    } else if (game_id == GAME_BIGFISH) {  // reference src/games/bigfish.cpp:33-46, src/resources.cpp:921-932
        add_themes(0, {"misc_assets/fishTile_072.png"});
        add_themes(2, {"misc_assets/fishTile_074.png", "misc_assets/fishTile_078.png", "misc_assets/fishTile_080.png"});
        for (const char *n : {"water1", "water2", "water3", "water4", "underwater1", "underwater2", "underwater3"})
            backgrounds->push_back(std::string("water_backgrounds/") + n + ".png");
    } else if (game_id == GAME_CLIMBER) {  // reference src/games/climber.cpp:42-88
        const char *pcol[4] = {"Blue", "Green", "Grey", "Red"};
        const int ptypes[4] = {0, 9, 12, 13};
        const char *pnames[4] = {"stand", "walk4", "walk1", "walk2"};
        for (int k = 0; k < 4; k++) {
            std::vector<std::string> v;
            for (auto c : pcol) v.push_back(std::string("platformer/player") + c + "_" + pnames[k] + ".png");
            add_themes(ptypes[k], v);
        }
        add_themes(16, {"platformer/tileBlue_05.png", "platformer/tileGreen_05.png", "platformer/tileYellow_06.png", "platformer/tileBrown_06.png"});
        add_themes(15, {"platformer/tileBlue_08.png", "platformer/tileGreen_08.png", "platformer/tileYellow_09.png", "platformer/tileBrown_09.png"});
        add_themes(6, {"platformer/enemySwimming_1.png"});
        add_themes(7, {"platformer/enemySwimming_2.png"});
        add_themes(1, {"platformer/yellowCrystal.png"});
        platform_backgrounds(backgrounds);
    } else if (game_id == GAME_MINER) {  // reference src/games/miner.cpp:37-55
        add_themes(0, {"misc_assets/robot_greenDrive1.png"});
        add_themes(1, {"misc_assets/elementStone007.png"});
        add_themes(2, {"misc_assets/gemBlue.png"});
        add_themes(6, {"misc_assets/window.png"});
        add_themes(9, {"misc_assets/dirt.png"});
        add_themes(10, {"misc_assets/tile_bricksGrey.png"});
        platform_backgrounds(backgrounds);
    } else if (game_id == GAME_JUMPER) {  // reference src/games/jumper.cpp:46-80
        add_themes(0, {"misc_assets/bunny2_ready.png"});
        add_themes(2, {"misc_assets/spikeMan_stand.png"});
        add_themes(1, {"misc_assets/carrot.png"});
        add_themes(9, {"misc_assets/bunny2_jump.png"});
        add_themes(12, {"misc_assets/bunny2_walk1.png"});
        add_themes(13, {"misc_assets/bunny2_walk2.png"});
        add_themes(10, {"misc_assets/bunny2_walk1.png"});
        add_themes(11, {"misc_assets/bunny2_walk2.png"});
        add_themes(7, {"platformer/tileBlue_05.png", "platformer/tileGreen_05.png", "platformer/tileYellow_06.png", "platformer/tileBrown_06.png"});
        add_themes(6, {"platformer/tileBlue_08.png", "platformer/tileGreen_08.png", "platformer/tileYellow_09.png", "platformer/tileBrown_09.png"});
        platform_backgrounds(backgrounds);
    } else if (game_id == GAME_CAVEFLYER) {  // reference src/games/caveflyer.cpp:31-53
        add_themes(1, {"misc_assets/ufoGreen2.png"});
        add_themes(2, {"misc_assets/meteorBrown_big1.png"});
        add_themes(3, {"misc_assets/ufoRed2.png"});
        add_themes(4, {"misc_assets/laserBlue02.png"});
        add_themes(5, {"misc_assets/enemyShipBlue4.png"});
        add_themes(0, {"misc_assets/playerShip1_red.png"});
        add_themes(8, {"misc_assets/groundA.png"});
        add_themes(9, {"misc_assets/towerDefense_tile295.png"});
        for (const char *n : SPACE_BGS) backgrounds->push_back(std::string("space_backgrounds/") + n + ".png");
    } else if (game_id == GAME_CHASER) {  // reference src/games/chaser.cpp:51-73, src/resources.cpp:913-918
        add_themes(0, {"misc_assets/enemyFloating_1b.png"});
        add_themes(6, {"misc_assets/enemyFlying_1.png"});
        add_themes(7, {"misc_assets/enemyFlying_2.png"});
        add_themes(8, {"misc_assets/enemyFlying_3.png"});
        add_themes(2, {"misc_assets/yellowCrystal.png"});
        add_themes(3, {"misc_assets/enemyWalking_1b.png"});
        add_themes(4, {"misc_assets/enemySpikey_1b.png"});
        add_themes(5, {"misc_assets/tileStone_slope.png"});
        backgrounds->push_back("topdown_backgrounds/floortiles.png");
    } else if (game_id == GAME_BOSSFIGHT) {  // reference src/games/bossfight.cpp:73-107
        add_themes(0, {"misc_assets/playerShip1_blue.png", "misc_assets/playerShip1_green.png", "misc_assets/playerShip2_orange.png", "misc_assets/playerShip3_red.png"});
        add_themes(2, {"misc_assets/enemyShipBlack1.png", "misc_assets/enemyShipBlue2.png", "misc_assets/enemyShipGreen3.png", "misc_assets/enemyShipRed4.png"});
        add_themes(4, {"misc_assets/laserGreen14.png", "misc_assets/laserRed11.png", "misc_assets/laserBlue09.png"});
        add_themes(1, {"misc_assets/laserGreen14.png", "misc_assets/laserRed11.png", "misc_assets/laserBlue09.png"});
        add_themes(3, {"misc_assets/shield2.png"});
        add_themes(7, {"misc_assets/spaceMeteors_001.png", "misc_assets/spaceMeteors_002.png", "misc_assets/spaceMeteors_003.png", "misc_assets/spaceMeteors_004.png",
                       "misc_assets/meteorGrey_big1.png", "misc_assets/meteorGrey_big2.png", "misc_assets/meteorGrey_big3.png", "misc_assets/meteorGrey_big4.png"});
        for (const char *n : SPACE_BGS) backgrounds->push_back(std::string("space_backgrounds/") + n + ".png");
    } else if (game_id == GAME_DODGEBALL) {  // reference src/games/dodgeball.cpp:49-88
        auto series = [](const std::string &stem, int n) {
            std::vector<std::string> v;
            for (int i = 1; i <= n; i++) v.push_back("misc_assets/" + stem + std::to_string(i) + ".png");
            return v;
        };
        add_themes(0, {"misc_assets/character12.png"});
        add_themes(3, {"misc_assets/ball_soccer1.png"});
        add_themes(4, series("character", 11));
        add_themes(5, {"misc_assets/blockRed.png"});
        add_themes(6, {"misc_assets/ball_soccer2.png"});
        add_themes(7, {"misc_assets/blockGreen.png"});
        add_themes(1, {"misc_assets/tileStone_slope2.png"});
        add_themes(10, {"misc_assets/tileStone_slope2.png"});
        add_themes(8, series("spaceEffect", 9));
        topdown_backgrounds(backgrounds);
    } else if (game_id == GAME_NINJA) {  // reference src/games/ninja.cpp:45-75
        add_themes(20, {"misc_assets/tile_bricksGrey.png", "misc_assets/tile_bricksGrown.png", "misc_assets/tile_bricksRed.png"});
        add_themes(1, {"platformer/shroom1.png", "platformer/shroom2.png", "platformer/shroom3.png", "platformer/shroom4.png", "platformer/shroom5.png", "platformer/shroom6.png"});
        add_themes(0, {"platformer/zombie_idle.png"});
        add_themes(9, {"platformer/zombie_jump.png"});
        add_themes(12, {"platformer/zombie_walk1.png"});
        add_themes(13, {"platformer/zombie_walk2.png"});
        add_themes(6, {"misc_assets/bomb.png"});
        add_themes(7, {"misc_assets/saw.png"});
        add_themes(14, {"misc_assets/bomb.png"});
        platform_backgrounds(backgrounds);
    } else if (game_id == GAME_HEIST) {  // reference src/games/heist.cpp:41-57
        add_themes(51, {"kenney/Ground/Dirt/dirtCenter.png"});
        add_themes(9, {"misc_assets/gemYellow.png"});
        add_themes(0, {"misc_assets/spaceAstronauts_008.png"});
        add_themes(2, {"misc_assets/keyBlue.png", "misc_assets/keyGreen.png", "misc_assets/keyRed.png"});
        add_themes(1, {"misc_assets/lock_blue.png", "misc_assets/lock_green.png", "misc_assets/lock_red.png"});
        topdown_backgrounds(backgrounds);
    } else if (game_id == GAME_PLUNDER) {  // reference src/games/plunder.cpp:45-63, src/resources.cpp:933-940
        add_themes(7, {"misc_assets/ship_1.png", "misc_assets/ship_2.png", "misc_assets/ship_3.png", "misc_assets/ship_4.png", "misc_assets/ship_5.png", "misc_assets/ship_6.png"});
        add_themes(1, {"misc_assets/cannonBall.png"});
        add_themes(6, {"misc_assets/panel_wood.png"});
        add_themes(3, {"misc_assets/target_red2.png"});
        for (int k = 1; k <= 4; k++) backgrounds->push_back("water_backgrounds/water" + std::to_string(k) + ".png");
    } else if (game_id == GAME_LEAPER) {  // reference src/games/leaper.cpp:40-66
        add_themes(2, {"misc_assets/roadTile6b.png"});
        add_themes(3, {"misc_assets/terrainTile6.png"});
        add_themes(4, {"misc_assets/car_yellow_5.png", "misc_assets/car_black_1.png", "misc_assets/car_blue_2.png", "misc_assets/car_green_3.png", "misc_assets/car_red_4.png"});
        add_themes(1, {"misc_assets/elementWood044.png"});
        add_themes(0, {"misc_assets/frog1.png", "misc_assets/frog2.png", "misc_assets/frog4.png", "misc_assets/frog6.png", "misc_assets/frog7.png"});
        add_themes(5, {"misc_assets/finish2.png"});
        topdown_backgrounds(backgrounds);
    } else if (game_id == GAME_FRUITBOT) {  // reference src/games/fruitbot.cpp:42-78
        auto series = [](const std::string &stem, int n) {
            std::vector<std::string> v;
            for (int i = 1; i <= n; i++) v.push_back("misc_assets/" + stem + std::to_string(i) + ".png");
            return v;
        };
        add_themes(0, {"misc_assets/robot_3Dblue.png"});
        add_themes(1, {"misc_assets/tileStone_slope.png"});
        add_themes(2, {"misc_assets/tileStone_slope.png"});
        add_themes(3, {"misc_assets/keyRed2.png"});
        add_themes(4, series("food", 6));
        add_themes(7, series("fruit", 6));
        add_themes(10, {"misc_assets/fenceYellow.png"});
        add_themes(11, {"misc_assets/lockRed2.png"});
        add_themes(12, series("present", 3));
        topdown_backgrounds(backgrounds);
    } else if (game_id == GAME_STARPILOT) {  // reference src/games/starpilot.cpp:55-106, src/resources.cpp:828-845
        auto numbered = [](const std::string &stem, int from, int to, int width) {
            std::vector<std::string> v;
            for (int i = from; i <= to; i++) {
                std::string n = std::to_string(i);
                while ((int)n.size() < width) n = "0" + n;
                v.push_back("misc_assets/" + stem + n + ".png");
            }
            return v;
        };
        add_themes(0, {"misc_assets/playerShip2_blue.png"});
        add_themes(1, {"misc_assets/towerDefense_tile295.png"});
        add_themes(2, {"misc_assets/towerDefense_tile296.png"});
        add_themes(3, {"misc_assets/towerDefense_tile297.png"});
        add_themes(4, numbered("spaceShips_", 1, 7, 3));
        add_themes(8, numbered("spaceShips_", 1, 7, 3));
        {
            std::vector<std::string> m = numbered("spaceMeteors_", 1, 4, 3), g = numbered("meteorGrey_big", 1, 4, 1);
            m.insert(m.end(), g.begin(), g.end());
            add_themes(5, m);
        }
        add_themes(6, numbered("spaceEffect", 1, 9, 1));
        add_themes(7, {"misc_assets/spaceStation_018.png", "misc_assets/spaceStation_019.png"});
        add_themes(9, numbered("spaceRockets_", 1, 4, 3));
        for (const char *n : SPACE_BGS) backgrounds->push_back(std::string("space_backgrounds/") + n + ".png");
    } else if (game_id == GAME_MAZE) {  // reference src/games/maze.cpp:26-38, src/resources.cpp:900-911
        add_themes(51, {"kenney/Ground/Sand/sandCenter.png"});
        add_themes(2, {"misc_assets/cheese.png"});
        add_themes(0, {"kenney/Enemies/mouse_move.png"});
        topdown_backgrounds(backgrounds);
    } else {
        return false;
    }
    reserved_assets(sprites);
    return true;
}

static bool file_exists(const std::string &p) {
    struct stat st;
    return !p.empty() && stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}

static bool gather_images(int game_id, const std::string &resource_root, const AtlasPack *pack, std::vector<SpriteName> *sprites,
                          std::vector<std::string> *bgs, AtlasPack *out, std::string *err) {
    if (!game_asset_names(game_id, sprites, bgs)) {
        if (err) *err = std::string("no asset table for game ") + game_name_from_id(game_id);
        return false;
    }
    auto fetch = [&](const std::string &path, ImageFormat fmt) -> bool {
        const std::string key = fmt == IMG_RGB32 ? path + "|bg" : path;
        if (out->images.count(key)) return true;
        if (pack) {
            auto it = pack->images.find(key);
            if (it == pack->images.end()) {
                if (err) *err = "atlas pack lacks " + key;
                return false;
            }
            out->images[key] = it->second;
            return true;
        }
        Image im;
        if (!decode_png(resource_root + path, fmt, &im, err)) return false;
        out->images[key] = std::move(im);
        return true;
    };
    for (auto &s : *sprites)
        if (s.theme < MAX_IMAGE_THEMES && !fetch(s.path, IMG_ARGB32_PM)) return false;
    for (auto &b : *bgs)
        if (!fetch(b, IMG_RGB32)) return false;
    return true;
}

bool bake_game_atlas(int game_id, const std::string &resource_root, const std::string &atlas_path, std::string *err) {
    std::vector<SpriteName> sprites;
    std::vector<std::string> bgs;
    AtlasPack out;
    if (!gather_images(game_id, resource_root, nullptr, &sprites, &bgs, &out, err)) return false;
    return out.save(atlas_path, err);
}

bool load_game_assets(int game_id, const std::string &resource_root, const std::string &atlas_path, HostAssets *out, std::string *err) {
    std::vector<SpriteName> sprites;
    std::vector<std::string> bgs;
    AtlasPack pack, imgs;
    const bool use_pack = file_exists(atlas_path);
    if (use_pack && !pack.load(atlas_path, err)) return false;
    if (!gather_images(game_id, resource_root, use_pack ? &pack : nullptr, &sprites, &bgs, &imgs, err)) return false;

    GameAssetsDev &t = out->table;
    memset(&t, 0, sizeof(t));
    for (auto &row : t.type_theme_img)
        for (auto &v : row) v = -1;
    out->pixels.clear();
    out->image_names.clear();
    std::map<std::string, int> index;
    auto place = [&](const std::string &key) -> int {
        auto it = index.find(key);
        if (it != index.end()) return it->second;
        const Image &im = imgs.images.at(key);
        const int idx = (int)out->image_names.size();
        if (idx >= MAX_GAME_IMAGES) return -1;
        t.img[idx].off = (uint32_t)out->pixels.size();
        t.img[idx].w = (uint16_t)im.w;
        t.img[idx].h = (uint16_t)im.h;
        bool opaque = true;
        for (uint32_t px : im.px)
            if ((px >> 24) != 0xffu) {
                opaque = false;
                break;
            }
        t.img[idx].opaque = opaque ? 1u : 0u;
        out->pixels.insert(out->pixels.end(), im.px.begin(), im.px.end());
        out->image_names.push_back(key);
        index[key] = idx;
        return idx;
    };
    for (auto &s : sprites) {
        if (s.type < 0 || s.type >= MAX_ASSETS) {
            if (err) *err = "asset table overflow";
            return false;
        }
        if (s.theme >= MAX_IMAGE_THEMES) {  // counted in asset_num_themes (BAG:114-116) but never drawable (fassert BAG:888)
            if (t.type_num_themes[s.type] < s.theme + 1) t.type_num_themes[s.type] = (uint8_t)(s.theme + 1);
            continue;
        }
        const int idx = place(s.path);
        if (idx < 0) {
            if (err) *err = "asset table overflow";
            return false;
        }
        t.type_theme_img[s.type][s.theme] = (int16_t)idx;
        if (t.type_num_themes[s.type] < s.theme + 1) t.type_num_themes[s.type] = (uint8_t)(s.theme + 1);
    }
    int ref_type = -1;  // the game's wall tile, when it has one: its size is the renderer's reference cell-image size
    if (game_id == GAME_COINRUN || game_id == GAME_CLIMBER) ref_type = 15;
    if (game_id == GAME_MAZE || game_id == GAME_HEIST) ref_type = 51;
    if (game_id == GAME_MINER) ref_type = 9;
    if (game_id == GAME_FRUITBOT) ref_type = 2;
    if (game_id == GAME_LEAPER) ref_type = 2;
    if (game_id == GAME_NINJA) ref_type = 20;
    if (game_id == GAME_DODGEBALL) ref_type = 10;
    if (game_id == GAME_CHASER) ref_type = 5;
    if (game_id == GAME_CAVEFLYER) ref_type = 8;
    if (game_id == GAME_JUMPER) ref_type = 6;
    if (ref_type >= 0 && t.type_theme_img[ref_type][0] >= 0) {
        t.ref_w = t.img[t.type_theme_img[ref_type][0]].w;
        t.ref_h = t.img[t.type_theme_img[ref_type][0]].h;
    } else {   // most common sprite size
        std::map<std::pair<int, int>, int> hist;
        for (auto &sp : sprites) {
            if (sp.theme >= MAX_IMAGE_THEMES) continue;
            const Image &im = imgs.images.at(sp.path);
            hist[{im.w, im.h}]++;
        }
        int best = -1;
        for (auto &kv : hist)
            if (kv.second > best) {
                best = kv.second;
                t.ref_w = kv.first.first;
                t.ref_h = kv.first.second;
            }
    }
    t.n_bg = (int32_t)bgs.size();
    if (t.n_bg > MAX_BACKGROUNDS) {
        if (err) *err = "too many backgrounds";
        return false;
    }
    for (size_t i = 0; i < bgs.size(); i++) {
        const int idx = place(bgs[i] + "|bg");
        if (idx < 0) {
            if (err) *err = "asset table overflow";
            return false;
        }
        t.bg_img[i] = (int16_t)idx;
    }
    finish_asset_tables(t);
    return true;
}

}  // namespace pgamd
