"""x-slab decomposition of the MLS-MPM step over the GPUs of one node.

The reference runs on one GPU; this module has no counterpart there.  The cell index is
x-major (src/mls-mpm/mlsMpmSimulator.js:286-290), so a slab of x-planes is one contiguous
block of both grids, and a particle with floor(x) = ix touches planes ix-1 .. ix+1
(SIM:307, 315-319): every slab keeps one halo plane on each side.

One process per GPU.  Per step and per neighbour (point-to-point send/recv over RCCL, i.e.
one xGMI link per neighbour; no all-reduce on the data path):

  clearGrid, p2g1
  exchange A  both boundary planes (halo + first owned) go to the neighbour, which adds the
              MASS component into its matching planes            -> mass complete for p2g2
              (4 bytes per cell travel)
  p2g2
  exchange B  same planes again, the neighbour adds the MOMENTUM components (12 bytes per cell)
  updateGrid  on owned and halo planes alike (redundant on halos, saves a third exchange)
  g2p
  migrate     particles whose floor(x) left the slab are counted per destination on the device
              right after g2p; the counts are exchanged (one small all_gather of the count
              matrix, read by the host) at the start of the NEXT step, after that step's
              clearGrid has been enqueued, so that the GPU clears the grid while the host waits
              for the counts; then the leavers are packed and sent straight to their owner

The grid is fixed-point i32 and the merges are integer adds, so the result is bit-identical
to the single-GPU grid for any number of slabs; tests assert exactly that.

`SlabSimulation` only speaks to a *backend* object (kernels + plane access + migration
pieces) and to `torch.distributed`, so the protocol itself is exercised on CPU with the
gloo backend in the tests, and runs on GPUs with `GpuSlabBackend`.
"""
import os
import sys

import numpy as np
import torch

from . import _lib

RECORD_WORDS = _lib.MIGRATE_WORDS


def equal_plane_starts(grid_x, world):
    return [round(r * grid_x / world) for r in range(world + 1)]


def balanced_plane_starts(x_positions, grid_x, world, dist=None, device=None):
    """Slab boundaries that give every rank about the same number of particles: the global
    histogram of floor(x) is summed over ranks and cut at equal quantiles.  Every slab gets
    at least 2 planes."""
    hist = np.bincount(np.clip(np.floor(x_positions).astype(np.int64), 0, grid_x - 1), minlength=grid_x)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.from_numpy(hist.astype(np.int64))
        if device is None and dist.get_backend() == "nccl":
            device = torch.device("cuda", torch.cuda.current_device())
        if device is not None:
            t = t.to(device)
        dist.all_reduce(t)
        hist = t.cpu().numpy()
    return plane_starts_from_histogram(hist, world)


def plane_starts_from_histogram(hist, world):
    """The balanced cut from the particles-per-plane histogram of the whole domain (what
    flowmpm_plane_histogram returns, summed over the slabs)."""
    hist = np.asarray(hist, dtype=np.int64)
    grid_x = hist.shape[0]
    cum = np.cumsum(hist)
    total = int(cum[-1])
    starts = [0]
    for r in range(1, world):
        cut = int(np.searchsorted(cum, total * r / world, side="left")) + 1
        cut = max(cut, starts[-1] + 2)
        cut = min(cut, grid_x - 2 * (world - r))
        starts.append(cut)
    starts.append(grid_x)
    return starts


def limited_plane_starts(hist, current, max_nx, budget):
    """The balanced cut, approached from the current one: boundary r moves toward its balanced place only as
    far as at most `budget` particles change hands through it (one plane always may), and no slab becomes
    wider than max_nx[r] or narrower than 2 planes.  The rule of the library's slab groups
    (flow_amd/csrc/kernels_migrate.h), so that every host cuts alike."""
    hist = np.asarray(hist, dtype=np.int64)
    world = len(current) - 1
    want = plane_starts_from_histogram(hist, world)
    nxt = [int(c) for c in current]
    for r in range(1, world):
        cut = nxt[r]
        step = 1 if want[r] > cut else -1
        moved = 0
        while cut != want[r]:
            plane = cut if step > 0 else cut - 1
            if moved > 0 and moved + int(hist[plane]) > budget:
                break
            moved += int(hist[plane])
            cut += step
        nxt[r] = cut
    for r in range(1, world):
        nxt[r] = max(nxt[r], nxt[r - 1] + 2)
    for r in range(world - 1, 0, -1):
        nxt[r] = min(nxt[r], nxt[r + 1] - 2)
    if any(nxt[r + 1] - nxt[r] > max_nx[r] for r in range(world)):
        return [int(c) for c in current]
    return nxt


class _Transport:
    """Point-to-point exchange over torch.distributed.  RCCL moves device tensors directly;
    gloo (CPU rehearsals) cannot send device tensors, so they are staged through the host."""

    def __init__(self, dist):
        self.dist = dist
        self.stage = dist.get_backend() == "gloo"

    def post(self, sends, recvs):
        """sends: list of (tensor, peer); recvs: list of (tensor, peer).  Starts the transfers
        (RCCL: on its own stream, after what the current stream holds so far) and returns what
        complete() needs.  Work enqueued on the current stream before complete() overlaps them."""
        if not sends and not recvs:
            return None
        dist = self.dist
        ops, staged, keep = [], [], []
        for t, peer in sends:
            buf = t.cpu() if (self.stage and t.is_cuda) else t
            keep.append(buf)
            ops.append(dist.P2POp(dist.isend, buf, peer))
        for t, peer in recvs:
            if self.stage and t.is_cuda:
                buf = torch.empty(t.shape, dtype=t.dtype)
                staged.append((t, buf))
            else:
                buf = t
            ops.append(dist.P2POp(dist.irecv, buf, peer))
        return dist.batch_isend_irecv(ops), staged, keep

    def complete(self, posted):
        """Makes the received tensors usable on the current stream."""
        if posted is None:
            return
        works, staged, _ = posted
        for req in works:
            req.wait()
        for t, buf in staged:
            t.copy_(buf)

    def exchange(self, sends, recvs):
        self.complete(self.post(sends, recvs))


class GpuSlabBackend:
    """The device-side pieces of one slab: a flowmpm handle enqueueing on torch's current
    stream (so RCCL operations issued through torch order against the kernels) plus torch
    tensors as send/receive staging."""

    def __init__(self, grid, slab_x0, slab_nx, max_particles, device, algorithm, rebin_interval, slab_max_nx=0):
        from .simulator import Handle
        self.torch_device = torch.device("cuda", device)
        torch.cuda.set_device(self.torch_device)
        stream = torch.cuda.current_stream(self.torch_device).cuda_stream
        self.handle = Handle(grid, max_particles, device=device, algorithm=algorithm,
                             rebin_interval=rebin_interval, slab=(slab_x0, slab_nx), stream=stream,
                             external_stream=True, slab_max_nx=slab_max_nx)
        self.slab_max_nx = max(slab_max_nx, slab_nx)
        self.plane_words = grid[1] * grid[2] * 4
        self.max_particles = max_particles
        self._leavers = torch.empty((max(max_particles // 4, 1024), RECORD_WORDS), dtype=torch.float32,
                                    device=self.torch_device)
        self._counts = None

    def set_uniforms(self, **kw):
        self.handle.set_uniforms(**kw)

    def set_mode_uniforms(self, modes):
        self.handle.set_mode_uniforms(modes)

    def set_domain_particles(self, n):
        self.handle.set_domain_particles(n)

    def run_kernel(self, kernel_id):
        self.handle.run_kernel(kernel_id)

    def upload(self, records, ids):
        self.handle.upload_particles(records, ids)

    def download(self):
        return self.handle.download_particles()

    def resident(self):
        return self.handle.num_particles()[1]

    def new_plane_buffer(self, planes):
        return torch.empty((planes, self.plane_words), dtype=torch.int32, device=self.torch_device)

    def plane_read(self, plane, out):
        self.handle.plane_copy_out(plane, out.data_ptr())

    def plane_add(self, plane, src, mask):
        self.handle.plane_add_i32(plane, src.data_ptr(), mask)

    def new_halo_buffer(self, mask):
        return torch.empty(self.handle.halo_words(mask), dtype=torch.int32, device=self.torch_device)

    def halo_pack(self, side, mask, out):
        self.handle.halo_pack(side, mask, out.data_ptr())

    def halo_add(self, side, mask, src):
        self.handle.halo_add(side, mask, src.data_ptr())

    def extract_leavers(self, plane_starts):
        counts = self.handle.extract_leavers(plane_starts, self._leavers.data_ptr(), self._leavers.shape[0])
        return counts, self._leavers[: int(counts.sum())]

    def find_leavers(self, plane_starts):
        """Enqueues the classification; returns the device tensor the counts will appear in."""
        if self._counts is None or self._counts.shape[0] != len(plane_starts) - 1:
            self._counts = torch.zeros(len(plane_starts) - 1, dtype=torch.int64, device=self.torch_device)
        self.handle.find_leavers(plane_starts, self._counts.data_ptr())
        return self._counts

    def counts_tensor(self, world):
        """The device tensor the classification at the end of a step writes its counts into."""
        if self._counts is None or self._counts.shape[0] != world:
            self._counts = torch.zeros(world, dtype=torch.int64, device=self.torch_device)
        return self._counts

    def refind_leavers(self, plane_starts):
        """The classification again, on the slots the particles occupy now (a re-bin may have
        moved them since find_leavers); the counts are the same and are not exported again."""
        self.handle.find_leavers(plane_starts, 0)

    def pack_leavers(self, counts):
        self.handle.pack_leavers(counts, self._leavers.data_ptr(), self._leavers.shape[0])
        return self._leavers[: int(np.asarray(counts).sum())]

    def new_record_buffer(self, count):
        return torch.empty((count, RECORD_WORDS), dtype=torch.float32, device=self.torch_device)

    def leaver_capacity(self):
        return int(self._leavers.shape[0])

    def grid_sums(self):
        return self.handle.grid_sums()

    def insert_arrivals(self, records):
        self.handle.insert_arrivals(records.data_ptr(), records.shape[0])

    def synchronize(self):
        self.handle.synchronize()

    def close(self):
        self.handle.close()


class SlabSimulation:
    def __init__(self, grid, plane_starts, rank, world, max_particles=0, device=0,
                 algorithm=_lib.ALGO_TILED, rebin_interval=1, dist=None, torch_device=None, backend=None,
                 slab_max_nx=0):
        self.grid = tuple(int(g) for g in grid)
        self.plane_starts = [int(s) for s in plane_starts]
        assert len(self.plane_starts) == world + 1 and self.plane_starts[0] == 0
        assert self.plane_starts[-1] == self.grid[0]
        self.rank, self.world, self.dist = rank, world, dist
        self.x0 = self.plane_starts[rank]
        self.nx = self.plane_starts[rank + 1] - self.x0
        self.backend = backend if backend is not None else GpuSlabBackend(
            self.grid, self.x0, self.nx, max_particles, device, algorithm, rebin_interval, slab_max_nx)
        self.recuts = 0
        self.transport = _Transport(dist) if (dist is not None and world > 1) else None
        # the tiled algorithm's two-pass p2g (mass, then summed momentum); other backends keep
        # the reference's p2g1 / p2g2
        fused = backend is None and algorithm == _lib.ALGO_TILED
        self.p2g_kernels = ((_lib.KERNEL_P2G_MASS, _lib.KERNEL_P2G_MOMENTUM) if fused
                            else (_lib.KERNEL_P2G1, _lib.KERNEL_P2G2))
        # the device library's updateGrid keeps the float grid, so only the fixed-point grid is cleared
        self.clear_kernel = _lib.KERNEL_CLEAR_FIXED_GRID if backend is None else _lib.KERNEL_CLEAR_GRID
        # The library's tiled mass pass can skip the leavers itself and add the arrivals later, so it
        # is enqueued before the host waits for the migration counts.
        # (FLOWMPM_SLAB_OVERLAP=0 keeps every stage whole, for A/B runs on a multi-GPU node.)
        self.split_mass_pass = fused and os.environ.get("FLOWMPM_SLAB_OVERLAP", "1") != "0"
        self.low = rank - 1 if rank > 0 else None
        self.high = rank + 1 if rank < world - 1 else None
        b = self.backend
        masks = (_lib.COMP_MASS, _lib.COMP_MOMENTUM)
        self._pending_counts = None
        self._pending_matrix = None
        self._count_bufs = None
        self._inbox = {}            # peer -> receive buffer for its migrating particles
        # RCCL on device tensors: the count exchange starts right after the classification
        # (FLOWMPM_SLAB_EARLY_COUNTS=0 keeps it where the counts are needed, for A/B runs)
        self.early_counts = (self.transport is not None and not self.transport.stage and backend is None
                             and os.environ.get("FLOWMPM_SLAB_EARLY_COUNTS", "1") != "0")
        self.migrated_out = 0       # particles this rank has sent away so far
        self.max_hop = 0            # farthest destination so far, in slabs (1 = a neighbour)
        # What every rank knows about every rank, so that an overflow anywhere is seen everywhere:
        # [leaver-buffer records, max_particles] per rank (gathered once) and the resident count per
        # rank (gathered once, then kept up to date from the count matrices, which every rank sees).
        self._limits = None
        self._residents = None
        self._send = {(side, m): b.new_halo_buffer(m) for side in ("low", "high") for m in masks}
        self._recv = {(side, m): b.new_halo_buffer(m) for side in ("low", "high") for m in masks}

    # ---- set-up ----
    def set_uniforms(self, **kw):
        self.backend.set_uniforms(**kw)

    def set_mode_uniforms(self, modes):
        """The mode uniforms act per particle (keyed on the particle id, which migrates with
        the particle), so every slab takes the same record."""
        self.backend.set_mode_uniforms(modes)

    def scatter_initial_particles(self, records, ids):
        """Every rank brings any subset of the global particles; each record is routed to the
        rank that owns floor(x), then uploaded there with its global id."""
        records = np.ascontiguousarray(records, dtype=np.float32).reshape(-1, _lib.PARTICLE_STRIDE)
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        self._residents = None      # a new particle set: the bookkeeping starts over
        # uniforms.numParticles of the whole domain (mlsMpmSimulator.js:2513-2522): g2p's
        # hide-percentage branch divides the particle id by it
        total = records.shape[0]
        if self.world > 1:
            t = torch.tensor([total], dtype=torch.int64, device=self._comm_device())
            self.dist.all_reduce(t)
            total = int(t.item())
        if total > 0 and hasattr(self.backend, "set_domain_particles"):
            self.backend.set_domain_particles(total)
        starts = np.asarray(self.plane_starts)
        dest = np.clip(np.searchsorted(starts, np.floor(records[:, 0]).astype(np.int64), side="right") - 1,
                       0, self.world - 1)
        if self.world == 1:
            self.backend.upload(records, ids)
            return
        order = np.argsort(dest, kind="stable")
        counts = np.bincount(dest, minlength=self.world).astype(np.int64)
        packed = np.empty((records.shape[0], _lib.PARTICLE_STRIDE + 1), dtype=np.float32)
        packed[:, :_lib.PARTICLE_STRIDE] = records[order]
        packed[:, _lib.PARTICLE_STRIDE] = ids[order].view(np.float32)
        mine, offs = [], np.concatenate([[0], np.cumsum(counts)])
        dev = self._comm_device()
        matrix = self._all_gather_counts(counts, dev)
        send_t = torch.from_numpy(packed).to(dev)
        sends, recvs, inbox = [], [], []
        for peer in range(self.world):
            if peer == self.rank:
                mine.append(packed[offs[peer]:offs[peer + 1]])
                continue
            if counts[peer]:
                sends.append((send_t[offs[peer]:offs[peer + 1]], peer))
            k = int(matrix[peer, self.rank])
            if k:
                buf = torch.empty((k, _lib.PARTICLE_STRIDE + 1), dtype=torch.float32, device=dev)
                inbox.append(buf)
                recvs.append((buf, peer))
        self.transport.exchange(sends, recvs)
        mine += [b.cpu().numpy() for b in inbox]
        allrec = np.concatenate(mine, axis=0) if mine else np.zeros((0, _lib.PARTICLE_STRIDE + 1), np.float32)
        self.backend.upload(np.ascontiguousarray(allrec[:, :_lib.PARTICLE_STRIDE]),
                            np.ascontiguousarray(allrec[:, _lib.PARTICLE_STRIDE]).view(np.int32))

    def generate_initial_particles(self, scene, total, seed=0xF10E5EED, fill=0.25):
        """The whole seeded scene generated on this rank's device, of which the slab keeps the particles
        of its own planes (flowmpm_generate_particles): no host records, no exchange.  Library backend
        only.  Returns the number of particles this rank holds."""
        self._residents = None
        self.backend.set_domain_particles(int(total))
        return self.backend.handle.generate_particles(scene, int(total), seed=seed, fill=fill)

    def _comm_device(self):
        if self.dist is not None and self.dist.get_backend() == "nccl":
            return self.backend.torch_device
        return torch.device("cpu")

    def _all_gather_counts(self, counts, dev):
        """counts: numpy array or tensor of `world` int64.  Returns the world x world matrix
        (row = sender) on the host."""
        t = counts if torch.is_tensor(counts) else torch.from_numpy(np.asarray(counts, dtype=np.int64))
        t = t.to(dev)
        out = torch.empty(self.world * self.world, dtype=torch.int64, device=dev)
        self.dist.all_gather_into_tensor(out, t)
        return out.cpu().numpy().reshape(self.world, self.world)

    # ---- the step ----
    def _post_planes(self, mask):
        """Both boundary planes of each side go to that side's neighbour, which adds them into
        its matching pair: my LOW pair (planes x0-1, x0) is the low neighbour's HIGH pair.
        Packs and starts the transfers; _add_planes finishes."""
        if self.transport is None:
            return None
        b = self.backend
        sends, recvs = [], []
        if self.low is not None:
            b.halo_pack(_lib.SIDE_LOW, mask, self._send["low", mask])
            sends.append((self._send["low", mask], self.low))
            recvs.append((self._recv["low", mask], self.low))
        if self.high is not None:
            b.halo_pack(_lib.SIDE_HIGH, mask, self._send["high", mask])
            sends.append((self._send["high", mask], self.high))
            recvs.append((self._recv["high", mask], self.high))
        return self.transport.post(sends, recvs)

    def _add_planes(self, mask, posted):
        if self.transport is None:
            return
        b = self.backend
        self.transport.complete(posted)
        if self.low is not None:
            b.halo_add(_lib.SIDE_LOW, mask, self._recv["low", mask])
        if self.high is not None:
            b.halo_add(_lib.SIDE_HIGH, mask, self._recv["high", mask])

    def _exchange_planes(self, mask):
        self._add_planes(mask, self._post_planes(mask))

    def _begin_migration(self):
        if self.transport is None:
            return
        self._pending_counts = self.backend.find_leavers(self.plane_starts)
        self._start_early_counts()

    def _start_early_counts(self):
        if not self.early_counts:
            return
        try:
            self._pending_matrix = self._start_count_gather(self._pending_counts)
        except (RuntimeError, TypeError, NotImplementedError) as exc:
            # every rank runs the same library, so every rank lands here together; the
            # exchange then happens where the counts are needed
            print("flow_amd.slab: early count exchange unavailable (%s); using the blocking one"
                  % (str(exc).splitlines()[0] if str(exc) else type(exc).__name__), file=sys.stderr)
            self.early_counts = False
            self._pending_matrix = None

    def _start_count_gather(self, counts_dev):
        """The count exchange and its copy to the host, started as soon as the classification is
        enqueued and kept off the stream the kernels run on: by the time the next step needs the
        matrix it has long arrived, and the host never waits with an empty queue behind it."""
        dev = self.backend.torch_device
        if self._count_bufs is None:
            self._count_bufs = (torch.empty(self.world * self.world, dtype=torch.int64, device=dev),
                                torch.empty(self.world * self.world, dtype=torch.int64).pin_memory(),
                                torch.cuda.Stream(device=dev))
        out, host, side = self._count_bufs
        work = self.dist.all_gather_into_tensor(out, counts_dev, async_op=True)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            if work is not None:
                work.wait()                     # the side stream waits for the collective, not the host
            host.copy_(out, non_blocking=True)
            done = torch.cuda.Event()
            done.record(side)
        return done

    def _finish_migration(self):
        """Completes the migration begun after the last g2p (a no-op when none is pending)."""
        counts_dev, self._pending_counts = self._pending_counts, None
        if counts_dev is None:
            return
        b = self.backend
        done, self._pending_matrix = self._pending_matrix, None
        if done is not None:
            done.synchronize()
            matrix = self._count_bufs[1].numpy().reshape(self.world, self.world).copy()
        else:
            matrix = self._all_gather_counts(counts_dev, self._comm_device())
        counts = matrix[self.rank]
        self._check_capacities(matrix)
        self.migrated_out += int(counts.sum())
        for peer in np.nonzero(counts)[0]:
            self.max_hop = max(self.max_hop, abs(int(peer) - self.rank))
        packed = b.pack_leavers(counts)
        if not matrix.any():
            return
        offs = np.concatenate([[0], np.cumsum(counts)])
        sends, recvs, inbox = [], [], []
        for peer in range(self.world):
            if peer == self.rank:
                continue
            if counts[peer]:
                sends.append((packed[int(offs[peer]):int(offs[peer + 1])], peer))
            k = int(matrix[peer, self.rank])
            if k:
                buf = self._inbox_for(peer, k)
                inbox.append(buf)
                recvs.append((buf, peer))
        self.transport.exchange(sends, recvs)
        for buf in inbox:
            b.insert_arrivals(buf)

    def _inbox_for(self, peer, count):
        """The first `count` records of the receive buffer kept for `peer`: allocated once with headroom (a
        slab's flow through a cut changes by a few per cent from step to step), so a step allocates nothing."""
        held = self._inbox.get(peer)
        if held is None or held.shape[0] < count:
            held = self.backend.new_record_buffer(max(4096, 2 * count))
            self._inbox[peer] = held
        return held[:count]

    def _check_capacities(self, matrix):
        """Every rank holds the whole count matrix, so every rank can see whether ANY rank is about to
        overflow its leaver buffer (its row sum) or its particle store (what it keeps plus its column
        sum) -- and all of them raise together, before any point-to-point transfer is posted.  A rank
        that failed alone would leave its neighbours waiting for its records for ever."""
        if self._limits is None or self._residents is None:
            b = self.backend
            mine = np.array([getattr(b, "leaver_capacity", lambda: 2 ** 62)(),
                             getattr(b, "max_particles", 2 ** 62) or 2 ** 62, b.resident()], dtype=np.int64)
            dev = self._comm_device()
            out = torch.empty(self.world * 3, dtype=torch.int64, device=dev)
            self.dist.all_gather_into_tensor(out, torch.from_numpy(mine).to(dev))
            table = out.cpu().numpy().reshape(self.world, 3)
            self._limits, self._residents = table[:, 0:2].copy(), table[:, 2].copy()
        leaving, arriving = matrix.sum(axis=1), matrix.sum(axis=0)
        after = self._residents - leaving + arriving
        problems = []
        for r in range(self.world):
            if leaving[r] > self._limits[r, 0]:
                problems.append("rank %d sends %d particles away in one step but its leaver buffer holds %d"
                                % (r, leaving[r], self._limits[r, 0]))
            if after[r] > self._limits[r, 1]:
                problems.append("rank %d would hold %d particles but was created for %d"
                                % (r, after[r], self._limits[r, 1]))
        if problems:
            raise _lib.FlowMpmError(_lib.ERR_CAPACITY, "slab migration: " + "; ".join(problems))
        self._residents = after

    def step(self):
        b = self.backend
        p2g = self.p2g_kernels
        if not self.split_mass_pass:
            b.run_kernel(self.clear_kernel)           # touches no particle: runs while the counts travel
            self._finish_migration()
            b.run_kernel(p2g[0])
            self._exchange_planes(_lib.COMP_MASS)
            b.run_kernel(p2g[1])
            self._exchange_planes(_lib.COMP_MOMENTUM)
            b.run_kernel(_lib.KERNEL_UPDATE_GRID)
            b.run_kernel(_lib.KERNEL_G2P)
            self._begin_migration()
            return
        # The library's tiled stages in six calls (flowmpm_slab_step_*), so that neither the host's wait for
        # the migration counts nor a halo exchange leaves the GPU without work: the mass pass of the
        # particles that stay runs while the counts travel (the leavers are skipped on the device, the
        # arrivals added once they are in); the mass halo travels while the work items that cannot see the
        # boundary planes run their momentum pass; the momentum halo travels during updateGrid and the
        # interior part of g2p, and updateGrid is repeated over the boundary tiles once it has been added.
        h = b.handle
        ptr = lambda side, m, bufs: bufs[side, m].data_ptr() if (self.low if side == "low" else self.high) is not None else None
        MASS, MOM = _lib.COMP_MASS, _lib.COMP_MOMENTUM
        h.slab_step_begin()
        self._finish_migration()
        h.slab_step_pack_mass(ptr("low", MASS, self._send), ptr("high", MASS, self._send))
        mass_halo = self._post_halo(MASS)
        h.slab_step_interior()
        self._complete(mass_halo)
        h.slab_step_boundary(ptr("low", MASS, self._recv), ptr("high", MASS, self._recv),
                             ptr("low", MOM, self._send), ptr("high", MOM, self._send))
        momentum_halo = self._post_halo(MOM)
        h.slab_step_grid()
        self._complete(momentum_halo)
        if self.transport is None:
            h.slab_step_end(None, None)
            return
        counts = b.counts_tensor(self.world)
        h.slab_step_end(ptr("low", MOM, self._recv), ptr("high", MOM, self._recv), self.plane_starts, counts.data_ptr())
        self._pending_counts = counts
        self._start_early_counts()

    def _post_halo(self, mask):
        """Starts the exchange of the halo pairs the library has packed into the send buffers."""
        if self.transport is None:
            return None
        sends, recvs = [], []
        if self.low is not None:
            sends.append((self._send["low", mask], self.low))
            recvs.append((self._recv["low", mask], self.low))
        if self.high is not None:
            sends.append((self._send["high", mask], self.high))
            recvs.append((self._recv["high", mask], self.high))
        return self.transport.post(sends, recvs)

    def _complete(self, posted):
        if self.transport is not None:
            self.transport.complete(posted)

    def rebalance(self, max_move_fraction=0.1):
        """Re-cuts the slabs at the particle quantiles of the moment (a collective: every rank calls it,
        every few steps): the per-plane histogram is summed over the ranks, every rank derives the same new
        cut from it (limited_plane_starts), re-cuts its handle (flowmpm_set_slab; the handle must have been
        created with slab_max_nx) and starts the migration of the planes that changed hands, which the
        next step completes.  Returns True when the cut moved."""
        if self.transport is None:
            return False
        self._finish_migration()
        h = self.backend.handle
        hist = torch.from_numpy(h.plane_histogram()).to(self._comm_device())
        self.dist.all_reduce(hist)
        hist = hist.cpu().numpy()
        total = int(hist.sum())
        limits = torch.tensor([self.backend.slab_max_nx], dtype=torch.int64, device=self._comm_device())
        gathered = torch.empty(self.world, dtype=torch.int64, device=self._comm_device())
        self.dist.all_gather_into_tensor(gathered, limits)
        new = limited_plane_starts(hist, self.plane_starts, [int(v) for v in gathered.cpu()],
                                   int(max_move_fraction * total / self.world))
        if new == self.plane_starts:
            return False
        self.plane_starts = new
        self.x0, self.nx = new[self.rank], new[self.rank + 1] - new[self.rank]
        h.set_slab(self.x0, self.nx)
        self.recuts += 1
        self._begin_migration()
        return True

    def resident(self):
        self._finish_migration()
        return self.backend.resident()

    def synchronize(self):
        """Also completes the migration of the last step (a collective: every rank calls it)."""
        self._finish_migration()
        self.backend.synchronize()

    def conservation(self):
        """Global [grid x, y, z momentum words, grid mass words, particle count] as int64: one
        small all-reduce, for a reporting interval, not for every step.  The grid sums are
        those of the last step's p2g; the mass sum is exact and independent of the number of
        slabs."""
        self._finish_migration()
        local = np.concatenate([self.backend.grid_sums(), [self.backend.resident()]]).astype(np.int64)
        if self.transport is None:
            return local
        t = torch.from_numpy(local).to(self._comm_device())
        self.dist.all_reduce(t)
        return t.cpu().numpy()

    def download(self):
        """(records, global ids) of the particles resident on this rank."""
        self._finish_migration()
        return self.backend.download()

    def close(self):
        self.backend.close()


class LocalSlabGroup:
    """All slabs of a decomposition driven by ONE process in lock-step on one device: the stage order of
    SlabSimulation.step() (the mass pass split around the migration, momentum pass and g2p in an
    interior and a boundary launch around the halo merges) with device buffers handed from slab to
    slab in place of a transport.  Every handle enqueues on the same stream, so the order of the calls
    is the order on the device.  What the full-size decomposition tests and the single-GPU slab
    measurements run on; results equal the single domain's bit for bit, as with any transport."""

    def __init__(self, grid, plane_starts, max_particles, device=0, rebin_interval=64):
        import torch
        self.torch = torch
        self.grid = tuple(int(g) for g in grid)
        self.plane_starts = [int(x) for x in plane_starts]
        self.world = len(self.plane_starts) - 1
        caps = max_particles if hasattr(max_particles, "__len__") else [max_particles] * self.world
        self.backends = [GpuSlabBackend(self.grid, self.plane_starts[r], self.plane_starts[r + 1] - self.plane_starts[r],
                                        int(caps[r]), device, _lib.ALGO_TILED, rebin_interval)
                         for r in range(self.world)]
        self.handles = [b.handle for b in self.backends]
        masks = (_lib.COMP_MASS, _lib.COMP_MOMENTUM)
        self._halo = [{(side, m): b.new_halo_buffer(m) for side in (_lib.SIDE_LOW, _lib.SIDE_HIGH) for m in masks}
                      for b in self.backends]
        self._pending = False
        self.migrated = 0
        self.max_hop = 0

    def set_uniforms(self, **kw):
        for b in self.backends:
            b.set_uniforms(**kw)

    def _run(self, kernel_id):
        for b in self.backends:
            b.run_kernel(kernel_id)

    def _pack(self, mask):
        for r, b in enumerate(self.backends):
            if r > 0:
                b.halo_pack(_lib.SIDE_LOW, mask, self._halo[r][_lib.SIDE_LOW, mask])
            if r < self.world - 1:
                b.halo_pack(_lib.SIDE_HIGH, mask, self._halo[r][_lib.SIDE_HIGH, mask])

    def _add(self, mask):
        for r, b in enumerate(self.backends):
            if r > 0:          # my LOW pair is the low neighbour's HIGH pair
                b.halo_add(_lib.SIDE_LOW, mask, self._halo[r - 1][_lib.SIDE_HIGH, mask])
            if r < self.world - 1:
                b.halo_add(_lib.SIDE_HIGH, mask, self._halo[r + 1][_lib.SIDE_LOW, mask])

    def _migrate(self):
        if not self._pending:
            return
        self._pending = False
        matrix = self.torch.stack([b._counts for b in self.backends]).cpu().numpy()      # the one host wait
        packed = [b.pack_leavers(matrix[r]) for r, b in enumerate(self.backends)]
        for r in range(self.world):
            offs = np.concatenate([[0], np.cumsum(matrix[r])])
            for peer in np.nonzero(matrix[r])[0]:
                self.migrated += int(matrix[r, peer])
                self.max_hop = max(self.max_hop, abs(int(peer) - r))
                self.backends[peer].insert_arrivals(packed[r][int(offs[peer]):int(offs[peer + 1])])

    def _buf(self, r, side, mask):
        """Slab r's own packed halo pair of `side` (None at the ends of the domain)."""
        if (side == _lib.SIDE_LOW and r == 0) or (side == _lib.SIDE_HIGH and r == self.world - 1):
            return None
        return self._halo[r][side, mask].data_ptr()

    def _from(self, r, side, mask):
        """What slab r receives on `side`: the neighbour's packed pair of the facing side."""
        if side == _lib.SIDE_LOW:
            return self._buf(r - 1, _lib.SIDE_HIGH, mask) if r > 0 else None
        return self._buf(r + 1, _lib.SIDE_LOW, mask) if r < self.world - 1 else None

    def step(self):
        LOW, HIGH, MASS, MOM = _lib.SIDE_LOW, _lib.SIDE_HIGH, _lib.COMP_MASS, _lib.COMP_MOMENTUM
        hs = self.handles
        for h in hs:
            h.slab_step_begin()
        self._migrate()
        for r, h in enumerate(hs):
            h.slab_step_pack_mass(self._buf(r, LOW, MASS), self._buf(r, HIGH, MASS))
        for h in hs:
            h.slab_step_interior()
        for r, h in enumerate(hs):        # every mass pair has been packed: the merges may read them
            h.slab_step_boundary(self._from(r, LOW, MASS), self._from(r, HIGH, MASS), self._buf(r, LOW, MOM),
                                 self._buf(r, HIGH, MOM))
        for h in hs:
            h.slab_step_grid()
        for r, (h, b) in enumerate(zip(hs, self.backends)):
            h.slab_step_end(self._from(r, LOW, MOM), self._from(r, HIGH, MOM), self.plane_starts,
                            b.counts_tensor(self.world).data_ptr())
        self._pending = True

    def finish(self):
        """Completes the migration of the last step; the particles then sit on their owners."""
        self._migrate()
        for b in self.backends:
            b.synchronize()

    def residents(self):
        self._migrate()
        return [b.resident() for b in self.backends]

    def conservation(self):
        """[grid x, y, z momentum words, grid mass words, particles] summed over the slabs."""
        self._migrate()
        return sum(np.concatenate([b.grid_sums(), [b.resident()]]).astype(np.int64) for b in self.backends)

    def close(self):
        for b in self.backends:
            b.close()
