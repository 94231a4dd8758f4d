// games.h -- the game policies compiled into the kernels (one kernel instantiation per game).
#pragma once
#include "game_bigfish.h"
#include "game_bossfight.h"
#include "game_caveflyer.h"
#include "game_chaser.h"
#include "game_climber.h"
#include "game_coinrun.h"
#include "game_dodgeball.h"
#include "game_fruitbot.h"
#include "game_heist.h"
#include "game_jumper.h"
#include "game_leaper.h"
#include "game_maze.h"
#include "game_miner.h"
#include "game_ninja.h"
#include "game_plunder.h"
#include "game_starpilot.h"

// (tests/emu can be built for a subset while iterating: -D'PG_FOR_EACH_GAME(X)=X(CoinRun)')
#ifndef PG_FOR_EACH_GAME
#define PG_FOR_EACH_GAME(X) X(CoinRun) X(BigFish) X(Maze) X(Climber) X(Miner) X(StarPilot) X(FruitBot) X(Leaper) X(Plunder) X(Heist) X(Ninja) X(Dodgeball) X(BossFight) X(Chaser) X(CaveFlyer) X(Jumper) X(CaveFlyerMemory)
#endif
