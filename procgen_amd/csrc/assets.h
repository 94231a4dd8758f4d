// assets.h -- host side of the sprite atlas: which image files a game uses (the reference's asset_for_type /
// load_background_images tables), decoding them (PNG tree or baked .atlas pack), and flattening them into the
// HBM blob + descriptor table the kernels sample from.  Replaces images_load() + initialize_asset_if_necessary()
// (reference src/resources.cpp:30-954, src/basic-abstract-game.cpp:79-123).
#pragma once
#include <string>
#include <vector>

#include "image_io.h"
#include "pg_defs.h"

namespace pgamd {

struct SpriteName {
    int type, theme;
    std::string path;  // relative to resource_root
};

// sprite (type, theme) -> file and the background group of one game
bool game_asset_names(int game_id, std::vector<SpriteName> *sprites, std::vector<std::string> *backgrounds);

struct HostAssets {
    GameAssetsDev table;
    std::vector<uint32_t> pixels;
    std::vector<std::string> image_names;  // parallel to table.img (backgrounds carry a "|bg" suffix)
};

// Loads from `atlas_path` when that file exists, else decodes PNGs under `resource_root`.
bool load_game_assets(int game_id, const std::string &resource_root, const std::string &atlas_path, HostAssets *out, std::string *err);
// use_generated_assets (reference BAG:79-123 with options.use_generated_assets): no image file is read.  Image t (t < MAX_ASSETS) is
// the 64 x 64 sprite AssetGen paints for object type t from a generator seeded with fixed_asset_seed + t -- the same for every
// theme, env and episode -- as a rect texture when use_block_asset(t), else a shape on transparent ground; one theme per type,
// aspect ratio 1.  Image MAX_ASSETS describes the per-env 500 x 500 background canvas (its pixels live in DevCtx::gen_bg).
void generate_game_assets(const std::string &game_name, bool (*use_block_asset)(int type), HostAssets *out);
// Decodes the game's PNGs and writes the pack.
bool bake_game_atlas(int game_id, const std::string &resource_root, const std::string &atlas_path, std::string *err);

uint32_t hash_str_uint32(const std::string &str);  // FNV-1a of a game name = fixed_asset_seed (reference src/vecgame.cpp:156-167,324-327)
int game_id_from_name(const std::string &name);
const char *game_name_from_id(int id);

}  // namespace pgamd
