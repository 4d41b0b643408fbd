#!/usr/bin/env python3
"""How often would DartWalker3d-v1's self-collision check (walker3d.py:26) fire during valid episodes?
Rolls the CPU oracle (ground contacts only) with random actions and tests every NON-adjacent pair of link boxes for
overlap (separating-axis test) after every env step.  Diagnostic for DESIGN.md's deviation table."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
from dart_env_amd.model_card import card_for
import oracle_lib as ol


def boxes_overlap(Ta, ha, Tb, hb):
    Ra, Rb = Ta[:3, :3], Tb[:3, :3]
    t = Ra.T @ (Tb[:3, 3] - Ta[:3, 3])
    R = Ra.T @ Rb
    A = np.abs(R) + 1e-12
    for i in range(3):
        if abs(t[i]) > ha[i] + A[i] @ hb: return False
    for j in range(3):
        if abs(t @ R[:, j]) > ha @ A[:, j] + hb[j]: return False
    for i in range(3):
        for j in range(3):
            ra = ha[(i + 1) % 3] * A[(i + 2) % 3, j] + ha[(i + 2) % 3] * A[(i + 1) % 3, j]
            rb = hb[(j + 1) % 3] * A[i, (j + 2) % 3] + hb[(j + 2) % 3] * A[i, (j + 1) % 3]
            if abs(t[(i + 2) % 3] * R[(i + 1) % 3, j] - t[(i + 1) % 3] * R[(i + 2) % 3, j]) > ra + rb: return False
    return True


def main(episodes=40, seed=0):
    card = card_for("DartWalker3d-v1", all_bodies_collide=True)
    shapes = [(card.shape_body[s], np.array(card.shape_pose[s]).reshape(4, 4), np.array(card.shape_size[s][:]) / 2)
              for s in range(card.nshapes)]
    parent = [card.parent[b] for b in range(card.nbodies)]
    w = ol.OracleWorld(card)
    rs = np.random.RandomState(seed)
    steps = hits = eps_hit = 0
    lens = []
    for ep in range(episodes):
        w.reset()
        n = card.ndofs
        w.set_state(rs.uniform(-.005, .005, n), rs.uniform(-.005, .005, n))
        hit_ep = False
        for t in range(1000):
            ob, r, done = w.env_step(rs.uniform(-1, 1, card.act_dim))
            if done: break
            steps += 1
            poses = {b: w.body_pose(b) for b in range(card.nbodies)}
            hit = False
            for i, (bi, Pi, hi) in enumerate(shapes):
                for (bj, Pj, hj) in shapes[i + 1:]:
                    if parent[bi] == bj or parent[bj] == bi or bi == bj: continue
                    if boxes_overlap(poses[bi] @ Pi, hi, poses[bj] @ Pj, hj): hit = True; pair = (bi, bj)
            if hit:
                hits += 1; hit_ep = True
        lens.append(t); eps_hit += hit_ep
    print("episodes %d, mean length %.1f, valid env-steps %d, steps with a non-adjacent box overlap %d (%.2f%%), episodes affected %d"
          % (episodes, np.mean(lens), steps, hits, 100.0 * hits / max(steps, 1), eps_hit))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 40)
