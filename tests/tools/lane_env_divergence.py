"""
Planning aid for DESIGN.md section 8 item 1 (not a test): how alike is the physics work of neighbouring envs?

A lane = env step kernel runs 64 envs per wavefront; a wave takes as long as its slowest lane.  This replays the
oracle and, per step and per group of 64 consecutive envs, compares the mean with the maximum of a per-env work
proxy for BasicAbstractGame::step_entities (reference src/basic-abstract-game.cpp:593-656,1086-1098):
    A: sum over smart_step entities of 2 * max(4, int(4 * |v|)) sub_steps (grid probes only, entity scans filtered out
       by may_interact as in coinrun / maze), plus a tenth of a sub_step per entity for Entity::step
    B: the same sub_steps each scanning all n entities (games whose entities do collide), plus 2 n
mean / max is the SIMT efficiency bound of that kernel; the fraction of env-steps with a reset (which would leave
the wave for the wave = env kernel) is reported beside it.
"""
import sys, os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, "oracle"))
import oracle_env  # noqa: E402


def main(game="coinrun", n=256, steps=300):
    env = oracle_env.OracleEnv(n, game, rand_seed=23)
    rng = np.random.RandomState(0)
    eff, eff_b, resets = [], [], 0
    for t in range(steps):
        _, _, first = env.observe()
        if t:
            resets += int(first.sum())
        work = np.zeros(n)
        work_b = np.zeros(n)
        for e in range(n):
            ents = env.entities(e)
            ne = len(ents)
            v = ents[:, 2:4].view(np.float32)
            smart = ents[:, 22] != 0
            sub = np.maximum(4, (4 * np.sqrt((v.astype(np.float64) ** 2).sum(axis=1))).astype(int))
            work[e] = (2 * sub[smart]).sum() + 0.1 * ne
            work_b[e] = (2 * sub[smart]).sum() * max(ne, 1) + 2 * ne
        for g in range(0, n, 64):
            eff.append(work[g:g + 64].mean() / work[g:g + 64].max())
            eff_b.append(work_b[g:g + 64].mean() / work_b[g:g + 64].max())
        env.act(rng.randint(0, 15, size=(n,), dtype=np.int32))
    print(f"{game}: mean/max work per 64-env group: A {np.mean(eff):.2f} (p10 {np.percentile(eff, 10):.2f})  B {np.mean(eff_b):.2f} (p10 {np.percentile(eff_b, 10):.2f}), "
          f"resets per env-step = {resets / (n * (steps - 1)):.4f}")


if __name__ == "__main__":
    for g in (sys.argv[1:] or ["coinrun"]):
        main(g)
