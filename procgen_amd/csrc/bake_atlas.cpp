// bake_atlas -- decodes the PNGs one game needs and writes them as a .atlas pack (see image_io.h), so that a
// machine without the PNG tree (the GPU box) can construct the environment.  Run by __graft_entry__.build().
#include <cstdio>

#include "assets.h"

int main(int argc, char **argv) {
    if (argc != 4) {
        fprintf(stderr, "usage: bake_atlas <game> <resource_root/> <out.atlas>\n");
        return 2;
    }
    const int gid = pgamd::game_id_from_name(argv[1]);
    std::string root = argv[2], err;
    if (!root.empty() && root.back() != '/') root += '/';
    if (gid < 0 || !pgamd::bake_game_atlas(gid, root, argv[3], &err)) {
        fprintf(stderr, "bake_atlas: %s\n", gid < 0 ? "unknown game" : err.c_str());
        return 1;
    }
    return 0;
}
