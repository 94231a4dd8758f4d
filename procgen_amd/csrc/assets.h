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
// Decodes the game's PNGs and writes the pack.
bool bake_game_atlas(int game_id, const std::string &resource_root, const std::string &atlas_path, std::string *err);

int game_id_from_name(const std::string &name);
const char *game_name_from_id(int id);

}  // namespace pgamd
