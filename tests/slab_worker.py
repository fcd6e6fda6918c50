"""One rank of a slab-decomposed run, launched by tests through torch.distributed.run.

--backend oracle : the protocol of flow_amd.slab.SlabSimulation over gloo with the CPU oracle
                   standing in for the HIP library (test infrastructure; runs without a GPU)
--backend gpu    : the real library; every rank uses cuda:0 and the transport is gloo with host
                   staging (a one-GPU rehearsal of the multi-GPU path)
Each rank writes its resident particles and owned grid planes to --out/rank<r>.npz.
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from flow_amd import _lib  # noqa: E402
from flow_amd.slab import SlabSimulation, balanced_plane_starts, equal_plane_starts  # noqa: E402
from helpers import DEFAULT_UNIFORMS, moving_particles  # noqa: E402


class OracleSlabBackend:
    """Backend interface of SlabSimulation implemented with the CPU oracle on a full-size grid
    of which only this slab's planes (and halos) are ever non-zero."""

    def __init__(self, grid, x0, nx):
        from oracle import binding
        self.binding = binding
        self.grid, self.x0, self.nx = grid, x0, nx
        self.orc = binding.Oracle(grid)
        self.rec = np.zeros((0, 28), np.float32)
        self.ids = np.zeros(0, np.int32)
        self.uniform_kw = {}
        self.domain_particles = 0
        self.modes = None
        self.plane_words = grid[1] * grid[2] * 4
        self.torch_device = torch.device("cpu")

    def _u(self, whole_domain=False):
        """The oracle loops over uniforms.numParticles records; only the mode branches of g2p,
        which get their record count separately, see the particle count of the whole domain."""
        n = self.domain_particles if (whole_domain and self.domain_particles) else self.rec.shape[0]
        return self.binding.make_uniforms(self.grid, n, **self.uniform_kw)

    def set_domain_particles(self, n):
        self.domain_particles = int(n)

    def set_uniforms(self, **kw):
        self.uniform_kw = kw

    def set_mode_uniforms(self, modes):
        self.modes = self.binding.ModeUniforms.from_buffer_copy(bytes(modes))

    def run_kernel(self, k):
        if k == _lib.KERNEL_CLEAR_GRID:
            self.orc.clear_grid()
        elif k == _lib.KERNEL_P2G1:
            self.orc.p2g1(self.rec, self._u())
        elif k == _lib.KERNEL_P2G2:
            self.orc.p2g2(self.rec, self._u())
        elif k == _lib.KERNEL_UPDATE_GRID:
            self.orc.update_grid(self._u())
        elif k == _lib.KERNEL_G2P:
            if self.modes is None:
                self.orc.g2p(self.rec, self._u())
            else:
                self.orc.g2p_modes(self.rec, self._u(whole_domain=True), self.modes,
                                   ids=np.ascontiguousarray(self.ids), count=self.rec.shape[0])

    def upload(self, records, ids):
        self.rec, self.ids = np.ascontiguousarray(records, np.float32).copy(), np.asarray(ids, np.int32).copy()

    def download(self):
        return self.rec.copy(), self.ids.copy()

    def resident(self):
        return self.rec.shape[0]

    def _plane_x(self, plane):
        return {_lib.PLANE_LOW_HALO: self.x0 - 1, _lib.PLANE_LOW_OWNED: self.x0,
                _lib.PLANE_HIGH_OWNED: self.x0 + self.nx - 1, _lib.PLANE_HIGH_HALO: self.x0 + self.nx}[plane]

    def _plane_view(self, plane):
        per = self.grid[1] * self.grid[2]
        x = self._plane_x(plane)
        return self.orc.cell[x * per:(x + 1) * per]

    def new_plane_buffer(self, planes):
        return torch.zeros((planes, self.plane_words), dtype=torch.int32)

    def plane_read(self, plane, out):
        out.copy_(torch.from_numpy(self._plane_view(plane).reshape(-1).copy()))

    def plane_add(self, plane, src, mask):
        view = self._plane_view(plane).view(np.uint32)
        add = src.numpy().reshape(-1, 4).view(np.uint32)
        for c in range(4):
            if mask & (1 << c):
                view[:, c] += add[:, c]

    def _pair_view(self, side):
        per = self.grid[1] * self.grid[2]
        x = self.x0 - 1 if side == _lib.SIDE_LOW else self.x0 + self.nx - 1
        return self.orc.cell[x * per:(x + 2) * per]

    @staticmethod
    def _components(mask):
        return [c for c in range(4) if mask & (1 << c)]

    def new_halo_buffer(self, mask):
        return torch.zeros(2 * self.grid[1] * self.grid[2] * len(self._components(mask)), dtype=torch.int32)

    def halo_pack(self, side, mask, out):
        out.copy_(torch.from_numpy(np.ascontiguousarray(self._pair_view(side)[:, self._components(mask)]).reshape(-1)))

    def halo_add(self, side, mask, src):
        comps = self._components(mask)
        view = self._pair_view(side).view(np.uint32)
        add = src.numpy().reshape(-1, len(comps)).view(np.uint32)
        for k, c in enumerate(comps):
            view[:, c] += add[:, k]

    def _pack(self, rows):
        r = self.rec[rows]
        out = np.zeros((len(rows), 24), np.float32)
        out[:, 0:3], out[:, 3:6] = r[:, 0:3], r[:, 4:7]
        for c in range(3):
            out[:, 6 + 3 * c:9 + 3 * c] = r[:, 8 + 4 * c:11 + 4 * c]
        out[:, 15], out[:, 16] = r[:, 3], r[:, 7]
        out[:, 17:20], out[:, 20:23] = r[:, 20:23], r[:, 24:27]
        out[:, 23] = self.ids[rows].view(np.float32)
        return out

    def find_leavers(self, plane_starts):
        self._plane_starts = list(plane_starts)
        ix = np.trunc(self.rec[:, 0]).astype(np.int64)
        leaving = (ix < self.x0) | (ix >= self.x0 + self.nx)
        world = len(plane_starts) - 1
        dest = np.clip(np.searchsorted(np.asarray(plane_starts), ix, side="right") - 1, 0, world - 1)
        return torch.from_numpy(np.bincount(dest[leaving], minlength=world).astype(np.int64))

    def pack_leavers(self, counts):
        got, packed = self.extract_leavers(self._plane_starts)
        assert np.array_equal(got, np.asarray(counts))
        return packed

    def extract_leavers(self, plane_starts):
        ix = np.trunc(self.rec[:, 0]).astype(np.int64)
        leaving = (ix < self.x0) | (ix >= self.x0 + self.nx)
        world = len(plane_starts) - 1
        dest = np.clip(np.searchsorted(np.asarray(plane_starts), ix, side="right") - 1, 0, world - 1)
        rows = np.nonzero(leaving)[0]
        rows = rows[np.argsort(dest[rows], kind="stable")]
        counts = np.bincount(dest[rows], minlength=world).astype(np.int64)
        packed = torch.from_numpy(self._pack(rows))
        self.rec, self.ids = np.ascontiguousarray(self.rec[~leaving]), self.ids[~leaving]
        return counts, packed

    def new_record_buffer(self, count):
        return torch.zeros((count, 24), dtype=torch.float32)

    def insert_arrivals(self, records):
        p = records.numpy()
        r = np.zeros((p.shape[0], 28), np.float32)
        r[:, 0:3], r[:, 4:7] = p[:, 0:3], p[:, 3:6]
        for c in range(3):
            r[:, 8 + 4 * c:11 + 4 * c] = p[:, 6 + 3 * c:9 + 3 * c]
        r[:, 3], r[:, 7] = p[:, 15], p[:, 16]
        r[:, 20:23], r[:, 24:27] = p[:, 17:20], p[:, 20:23]
        self.rec = np.concatenate([self.rec, r])
        self.ids = np.concatenate([self.ids, np.ascontiguousarray(p[:, 23]).view(np.int32)])

    def grid_sums(self):
        per = self.grid[1] * self.grid[2]
        return self.orc.cell[self.x0 * per:(self.x0 + self.nx) * per].astype(np.int64).sum(axis=0)

    def owned_grid(self):
        per = self.grid[1] * self.grid[2]
        return (self.orc.cell[self.x0 * per:(self.x0 + self.nx) * per].copy(),
                self.orc.cellf[self.x0 * per:(self.x0 + self.nx) * per].copy())

    def synchronize(self):
        pass

    def close(self):
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", choices=("oracle", "gpu"), required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--grid", default="24,12,12")
    ap.add_argument("--particles", type=int, default=2500)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--vel", type=float, default=3.0)
    ap.add_argument("--balanced", action="store_true")
    ap.add_argument("--algorithm", choices=("tiled", "scatter"), default="tiled")
    ap.add_argument("--mode-case", default=None, help="a case name of tests/mode_cases.py (grid 128,64,64)")
    ap.add_argument("--dt", type=float, default=DEFAULT_UNIFORMS["dt"],
                    help="a large dt makes particles jump over whole slabs in one step")
    ap.add_argument("--recut-every", type=int, default=0,
                    help="gpu backend: SlabSimulation.rebalance() after every k-th step (slabs change width)")
    ap.add_argument("--leaver-capacity-of-rank1", type=int, default=-1,
                    help="oracle backend: rank 1 claims a leaver buffer of this many records; the run must then "
                         "end with the capacity error on EVERY rank (marker files), not with a hang")
    args = ap.parse_args()
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    grid = tuple(int(g) for g in args.grid.split(","))
    n = args.particles
    mode_uni, mode_fields = {}, None
    if args.mode_case:
        import mode_cases
        assert grid == mode_cases.GRID
        mode_uni, mode_fields = mode_cases.case(args.mode_case)
        rec = mode_cases.particles(n, seed=41, vel=args.vel)
    else:
        rec = moving_particles(n, grid, seed=41, vel=args.vel)   # every rank builds the same scene
    lo, hi = rank * n // world, (rank + 1) * n // world          # and contributes one id range
    if args.balanced:
        starts = balanced_plane_starts(rec[lo:hi, 0], grid[0], world, dist)
    else:
        starts = equal_plane_starts(grid[0], world)
    x0, nx = starts[rank], starts[rank + 1] - starts[rank]
    if args.backend == "oracle":
        backend = OracleSlabBackend(grid, x0, nx)
        if args.leaver_capacity_of_rank1 >= 0 and rank == 1:
            backend.leaver_capacity = lambda: args.leaver_capacity_of_rank1
        sim = SlabSimulation(grid, starts, rank, world, dist=dist, backend=backend)
    else:
        algo = _lib.ALGO_TILED if args.algorithm == "tiled" else _lib.ALGO_SCATTER
        sim = SlabSimulation(grid, starts, rank, world, max_particles=n, device=0, algorithm=algo,
                             rebin_interval=2, dist=dist, slab_max_nx=grid[0] if args.recut_every else 0)
    sim.scatter_initial_particles(rec[lo:hi], np.arange(lo, hi, dtype=np.int32))
    if mode_fields is not None:
        sim.set_mode_uniforms(mode_cases.fill(_lib.ModeUniforms, mode_fields))
    try:
        for k in range(args.steps):
            sim.set_uniforms(**dict(DEFAULT_UNIFORMS, dt=args.dt, time=0.25 * k, **mode_uni))
            sim.step()
            if args.recut_every and k % args.recut_every == args.recut_every - 1 and k < args.steps - 1:
                sim.rebalance(0.2)
        sim.synchronize()
    except _lib.FlowMpmError as exc:
        if args.leaver_capacity_of_rank1 < 0 or exc.code != _lib.ERR_CAPACITY:
            raise
        with open(os.path.join(args.out, "rank%d.capacity" % rank), "w") as f:
            f.write(str(exc))
        return      # no collective may follow: every rank left the step at the same point
    totals = sim.conservation()
    out_rec, out_ids = sim.download()
    starts, x0, nx = sim.plane_starts, sim.x0, sim.nx        # a re-cut may have moved them
    if args.backend == "oracle":
        cell, cellf = sim.backend.owned_grid()
    else:
        per = grid[1] * grid[2]
        cell = sim.backend.handle.download_grid_i32()[per:(nx + 1) * per]
        cellf = sim.backend.handle.download_grid_f32()[per:(nx + 1) * per]
    np.savez(os.path.join(args.out, "rank%d.npz" % rank), rec=out_rec, ids=out_ids, cell=cell, cellf=cellf,
             starts=np.asarray(starts), totals=totals, max_hop=sim.max_hop, recuts=getattr(sim, "recuts", 0))
    sim.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
