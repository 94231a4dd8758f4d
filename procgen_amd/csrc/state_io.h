// state_io.h -- the reference's get_state / set_state wire format (reference src/vecgame.cpp:437-457,
// src/game.cpp:170-278, src/basic-abstract-game.cpp:1152-1288, src/entity.cpp:90-177, src/randgen.cpp:100-114,
// src/buffer.h), produced from / applied to the HBM-resident state of one env.  Host code.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "pg_defs.h"

namespace pgamd {

struct EnvSnapshot {  // host copy of one env's device state
    EnvHdr hdr;
    std::vector<uint32_t> ents;  // [EF_COUNT][ent_cap]
    int ent_cap = 0;
    std::vector<uint32_t> rng;   // [2][MT_STRIDE]
    std::vector<uint8_t> grid;   // grid_bytes (u8 cells)
};

// Serializes into the reference's byte stream. Returns false (with *err) when the buffer is too small.
bool serialize_state(int game_id, const GameOptions &opt, int game_n, const EnvSnapshot &s, char *data, int length, int *written, std::string *err);
// Parses a reference byte stream into the snapshot (sizes of s.ents / s.rng / s.grid must be pre-set).
// game_n_out (may be null): the env index the state was saved at -- the reference adopts it (src/game.cpp:253) and writes it back in get_state
bool deserialize_state(int game_id, const GameOptions &opt, EnvSnapshot *s, const char *data, int length, std::string *err, int *game_n_out = nullptr);

}  // namespace pgamd
