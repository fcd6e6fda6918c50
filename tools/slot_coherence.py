#!/usr/bin/env python3
"""How alike the particles of one wave are: for every 64 consecutive slots of the particle store (what one
wave of a tiled kernel processes in one trip), the number of distinct cells floor(position) among them, at the
start of the benchmark scene and after --steps steps.  Two lanes with the same cell add to the same LDS words
in all 27 (mass pass) / 81 (momentum pass) atomics of the trip.

A slab handle that owns the whole domain is used because it returns its particles in slot order.

    python tools/slot_coherence.py [--steps 220] [--noise 0]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import scene_uniforms
from flow_amd import Handle, _lib


def report(h, n, grid, label):
    rec, ids = h.download_particles()
    cell = (np.floor(rec[:, 0]).astype(np.int64) * grid[1] + np.floor(rec[:, 1]).astype(np.int64)) * grid[2] \
        + np.floor(rec[:, 2]).astype(np.int64)
    tile = ((np.floor(rec[:, 0]).astype(np.int64) // 8) * (grid[1] // 8) + np.floor(rec[:, 1]).astype(np.int64) // 8) * (grid[2] // 8) \
        + np.floor(rec[:, 2]).astype(np.int64) // 8
    waves = cell[: n // 64 * 64].reshape(-1, 64)
    s = np.sort(waves, axis=1)
    distinct = 1 + (np.diff(s, axis=1) != 0).sum(axis=1)
    # the largest group of lanes with one cell, per wave
    runs = np.ones_like(s)
    for k in range(1, 64):
        runs[:, k] = np.where(s[:, k] == s[:, k - 1], runs[:, k - 1] + 1, 1)
    largest = runs.max(axis=1)
    per_tile = np.bincount(tile)
    occ = per_tile[per_tile > 0]
    per_cell = np.bincount(cell)
    print("%s: %d occupied tiles (mean %.0f particles), %.2f particles per occupied cell" %
          (label, occ.size, occ.mean(), per_cell[per_cell > 0].mean()))
    print("  distinct cells per wave of 64 slots: mean %.1f, 10th percentile %d, median %d" %
          (distinct.mean(), np.percentile(distinct, 10), np.median(distinct)))
    print("  largest same-cell group per wave: mean %.2f, 90th percentile %d, max %d" %
          (largest.mean(), np.percentile(largest, 90), largest.max()))
    print("  P(same cell) of two slots d apart:", "  ".join("d=%d %.3f" % (d, float((cell[:-d] == cell[d:]).mean())) for d in (1, 2, 3, 4, 8, 16, 32, 64, 128, 256, 1024)))
    # waves that take every S-th slot of a block of 64 * S consecutive slots (lane l of wave w: slot S * l + w)
    for S in (2, 4, 8, 16):
        m = n // (64 * S) * (64 * S)
        ws = cell[:m].reshape(-1, 64, S).transpose(0, 2, 1).reshape(-1, 64)
        ss = np.sort(ws, axis=1)
        ds = 1 + (np.diff(ss, axis=1) != 0).sum(axis=1)
        print("  waves of every %2d-th slot: %.1f distinct cells per wave" % (S, ds.mean()))
    # what a permutation of whole granules of G consecutive slots inside each tile would leave (the stores of a
    # re-bin stay coalesced in runs of G particles)
    order0 = np.argsort(tile, kind="stable")
    t_sorted = tile[order0]
    first = np.r_[0, np.flatnonzero(np.diff(t_sorted)) + 1]
    counts = np.diff(np.r_[first, t_sorted.size])
    idx_in_tile = np.arange(t_sorted.size) - np.repeat(first, counts)
    c_of = np.repeat(counts, counts)
    for G in (4, 8, 16, 32):
        granules = (c_of + G - 1) // G
        g = idx_in_tile // G
        mult = np.where(granules % 257 != 0, 257, 263)
        pg = (g * mult) % granules
        o = np.lexsort((idx_in_tile % G, pg, t_sorted))
        wg = cell[order0][o][: n // 64 * 64].reshape(-1, 64)
        sg = np.sort(wg, axis=1)
        dg = 1 + (np.diff(sg, axis=1) != 0).sum(axis=1)
        within = cell[: n // G * G].reshape(-1, G)
        sw = np.sort(within, axis=1)
        dw = 1 + (np.diff(sw, axis=1) != 0).sum(axis=1)
        print("  granules of %2d slots: %.2f distinct cells inside one; permuted inside the tiles: %.1f distinct cells per wave" %
              (G, dw.mean(), dg.mean()))
    # the same for the particles of each wave put in random order inside their tile (what a shuffle would give)
    order = np.lexsort((np.random.default_rng(1).random(cell.size), tile))
    w2 = cell[order][: n // 64 * 64].reshape(-1, 64)
    s2 = np.sort(w2, axis=1)
    d2 = 1 + (np.diff(s2, axis=1) != 0).sum(axis=1)
    print("  (random order inside each tile would give mean %.1f distinct cells per wave)" % d2.mean())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=220)
    ap.add_argument("--noise", type=float, default=0.0)
    ap.add_argument("--particles-log2", type=int, default=24)
    ap.add_argument("--grid", default="256,256,256")
    a = ap.parse_args()
    grid = tuple(int(x) for x in a.grid.split(","))
    n = 1 << a.particles_log2
    h = Handle(grid, n, algorithm=_lib.ALGO_TILED, rebin_interval=64, slab=(0, grid[0]))
    h.generate_particles(_lib.SCENE_ELLIPSOID, n, seed=0xF10E5EED + 4)
    h.set_domain_particles(n)
    h.set_uniforms(**scene_uniforms(n, grid[0] * grid[1] * grid[2], "ellipsoid", a.noise))
    h.step(1, 0.1)
    report(h, n, grid, "after 1 step")
    done = 1
    for at in (a.steps // 2, a.steps):
        h.step(at - done, 0.1)
        done = at
        report(h, n, grid, "after %d steps" % at)
    print("re-bins performed:", h.timings()["rebin"]["performed"])
    h.close()


if __name__ == "__main__":
    main()
