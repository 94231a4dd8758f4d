#!/bin/bash
# Rebuilds the device objects of the named games only (after an edit that cannot change the other games' code, e.g. inside a policy header or
# under an `if constexpr` on a policy constant), marks the rest up to date and relinks procgen_amd/csrc/build/libenv.so.
# usage: tools/quick_game_build.sh CoinRun [BigFish ...]      (BUILD=<dir> EXTRA=<flags> as for the Makefile)
set -e
cd "$(dirname "$0")/../procgen_amd/csrc"
B=${BUILD:-build}
objs=""
for g in "$@"; do rm -f $B/kernels_$g.o; objs="$objs $B/kernels_$g.o"; done
make -s -j8 ARCH=gfx950 BUILD=$B EXTRA="$EXTRA" $objs $B/kernels.o $B/libenv_hip.o $B/state_io.o $B/assets.o $B/image_io.o 2>&1 | grep -E "error|Error" || true
make -s -t ARCH=gfx950 BUILD=$B $(ls $B/kernels_*.o) >/dev/null
rm -f $B/libenv.so
make -s ARCH=gfx950 BUILD=$B EXTRA="$EXTRA" $B/libenv.so 2>&1 | grep -E "error|Error" || true
ls -la $B/libenv.so
