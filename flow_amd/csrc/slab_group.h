// A group of slab handles driven by ONE host thread (declared in include/flowmpm.h, "slab groups").
//
// The reference's caller is a single-threaded JavaScript program (src/app.js:72-80, 156), so a host in
// that language drives a decomposed domain through one object: the group owns one slab handle per
// x-slab, steps them in lock-step with the flowmpm_slab_step_* sequence, and moves the halo pairs and
// the migrating particles itself.  Two transports:
//   - all slabs on one device: they share one stream, a slab's packed halo pair is read by its
//     neighbour where it lies, and leavers are inserted straight from the sender's pack buffer;
//   - one slab per device: RCCL point-to-point (ncclSend / ncclRecv inside ncclGroupStart / End, one
//     communicator per device from ncclCommInitAll) on a communication stream per device, ordered
//     against the compute stream with events, so that an exchange is in flight while the interior
//     work items run.  librccl.so is loaded when such a group is created, not before.
// Integer halo merges make the result independent of the number of slabs (see kernels_scatter.h).
//
// This file is included by flowmpm.hip after the definition of flowmpm_sim.
#pragma once

#include <dlfcn.h>

#include <memory>

namespace slabgroup {

// ---- the few RCCL entry points, resolved at run time -------------------------------------------
typedef struct ncclComm *ncclComm_t;
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string error;
    bool load()
    {
        if (lib) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) { error = std::string("cannot load librccl.so: ") + dlerror(); return false; }
        auto sym = [&](const char *n) -> void * {
            void *p = dlsym(lib, n);
            if (!p && error.empty()) error = std::string("librccl.so lacks ") + n;
            return p;
        };
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
        Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        if (!error.empty()) { dlclose(lib); lib = nullptr; return false; }
        return true;
    }
};
constexpr int kNcclInt32 = 2;      // ncclInt32 (rccl.h: ncclInt8 0, ncclUint8 1, ncclInt32 2, ...)

inline Rccl &rccl()
{
    static Rccl r;
    return r;
}

struct Member {
    flowmpm_sim *sim = nullptr;
    int device = 0;
    // packed halo pairs: [side][0 mass, 1 momentum]; recv only with RCCL (on one device the neighbour's
    // send buffer is read where it lies)
    int *send[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    int *recv[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    float *leavers = nullptr;        // records this slab sends away in a step, grouped by destination
    long long leaver_cap = 0;
    float *inbox = nullptr;          // RCCL: records received in a step
    long long inbox_cap = 0;
    long long *d_counts = nullptr;   // leaver counts per destination, written by the classification
    long long *h_counts = nullptr;   // pinned copy
    hipEvent_t counts_ready = nullptr;
    // RCCL: communication stream and the events that order it against the compute stream
    hipStream_t comm_stream = nullptr;
    hipEvent_t packed = nullptr, received = nullptr;
    ncclComm_t comm = nullptr;
};

}  // namespace slabgroup

struct flowmpm_slab_group {
    int n = 0;
    bool rccl = false;
    std::vector<int> starts;
    std::vector<slabgroup::Member> m;
    hipStream_t shared_stream = nullptr;      // one device: every slab enqueues here
    bool pending = false;                     // a classification is in flight (its counts not yet read)
    long long migrated = 0;
    int max_hop = 0;
    std::string err;
};

namespace slabgroup {

inline int gfail(flowmpm_slab_group *g, int code, const std::string &msg)
{
    if (g) g->err = msg;
    g_last_error = msg;
    return code;
}

#define GROUP_HIP(g, expr)                                                                     \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return slabgroup::gfail(g, FLOWMPM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

#define GROUP_SIM(g, r, expr)                                                                  \
    do {                                                                                       \
        int rc_ = (expr);                                                                      \
        if (rc_ != FLOWMPM_OK)                                                                 \
            return slabgroup::gfail(g, rc_, "slab " + std::to_string(r) + ": " + flowmpm_last_error((g)->m[r].sim)); \
    } while (0)

#define GROUP_NCCL(g, expr)                                                                    \
    do {                                                                                       \
        int n_ = (expr);                                                                       \
        if (n_ != 0)                                                                           \
            return slabgroup::gfail(g, FLOWMPM_ERR_HIP, std::string(#expr) + ": " + slabgroup::rccl().GetErrorString(n_)); \
    } while (0)

inline int *send_of(flowmpm_slab_group *g, int r, int side, int which)
{
    if ((side == 0 && r == 0) || (side == 1 && r == g->n - 1)) return nullptr;
    return g->m[r].send[side][which];
}

// What slab r merges on `side`: with RCCL its own receive buffer, on one device the facing pair of the neighbour.
inline int *recv_of(flowmpm_slab_group *g, int r, int side, int which)
{
    if ((side == 0 && r == 0) || (side == 1 && r == g->n - 1)) return nullptr;
    if (g->rccl) return g->m[r].recv[side][which];
    return side == 0 ? g->m[r - 1].send[1][which] : g->m[r + 1].send[0][which];
}

// RCCL: the halo pairs packed on the compute streams travel on the communication streams.
inline int exchange_halo(flowmpm_slab_group *g, int which, long long words)
{
    if (!g->rccl || g->n < 2) return FLOWMPM_OK;
    Rccl &R = rccl();
    for (int r = 0; r < g->n; r++) {
        Member &M = g->m[r];
        GROUP_HIP(g, hipSetDevice(M.device));
        GROUP_HIP(g, hipEventRecord(M.packed, M.sim->stream));
        GROUP_HIP(g, hipStreamWaitEvent(M.comm_stream, M.packed, 0));
    }
    GROUP_NCCL(g, R.GroupStart());
    for (int r = 0; r < g->n; r++) {
        Member &M = g->m[r];
        for (int side = 0; side < 2; side++) {
            const int peer = side == 0 ? r - 1 : r + 1;
            if (peer < 0 || peer >= g->n) continue;
            GROUP_NCCL(g, R.Send(M.send[side][which], (size_t)words, kNcclInt32, peer, M.comm, M.comm_stream));
            GROUP_NCCL(g, R.Recv(M.recv[side][which], (size_t)words, kNcclInt32, peer, M.comm, M.comm_stream));
        }
    }
    GROUP_NCCL(g, R.GroupEnd());
    for (int r = 0; r < g->n; r++) {
        Member &M = g->m[r];
        GROUP_HIP(g, hipSetDevice(M.device));
        GROUP_HIP(g, hipEventRecord(M.received, M.comm_stream));
    }
    return FLOWMPM_OK;
}

// The compute streams wait for the exchange started last.
inline int await_halo(flowmpm_slab_group *g)
{
    if (!g->rccl || g->n < 2) return FLOWMPM_OK;
    for (int r = 0; r < g->n; r++) {
        GROUP_HIP(g, hipSetDevice(g->m[r].device));
        GROUP_HIP(g, hipStreamWaitEvent(g->m[r].sim->stream, g->m[r].received, 0));
    }
    return FLOWMPM_OK;
}

// Completes the migration the last step began: reads the count matrix (the one host wait of a step),
// packs every slab's leavers and hands them to their owners.
inline int migrate(flowmpm_slab_group *g)
{
    if (!g->pending) return FLOWMPM_OK;
    g->pending = false;
    const int n = g->n;
    std::vector<int64_t> matrix((size_t)n * n);
    for (int r = 0; r < n; r++) {
        GROUP_HIP(g, hipEventSynchronize(g->m[r].counts_ready));
        for (int p = 0; p < n; p++) matrix[(size_t)r * n + p] = g->m[r].h_counts[p];
    }
    // capacities, before anything moves
    for (int r = 0; r < n; r++) {
        int64_t out = 0, in = 0;
        for (int p = 0; p < n; p++) { out += matrix[(size_t)r * n + p]; in += matrix[(size_t)p * n + r]; }
        if (out > g->m[r].leaver_cap)
            return gfail(g, FLOWMPM_ERR_CAPACITY, "slab " + std::to_string(r) + " sends " + std::to_string(out) +
                         " particles away in one step, its buffer holds " + std::to_string(g->m[r].leaver_cap));
        if (g->m[r].sim->resident - out + in > g->m[r].sim->cfg.max_particles)
            return gfail(g, FLOWMPM_ERR_CAPACITY, "slab " + std::to_string(r) + " would hold more particles than it was created for");
        if (g->rccl && in > g->m[r].inbox_cap)
            return gfail(g, FLOWMPM_ERR_CAPACITY, "slab " + std::to_string(r) + " receives more particles in one step than its inbox holds");
    }
    for (int r = 0; r < n; r++)
        GROUP_SIM(g, r, flowmpm_pack_leavers(g->m[r].sim, &matrix[(size_t)r * n], n, g->m[r].leavers, g->m[r].leaver_cap));
    bool any = false;
    for (int r = 0; r < n; r++)
        for (int p = 0; p < n; p++)
            if (matrix[(size_t)r * n + p]) {
                any = true;
                g->migrated += matrix[(size_t)r * n + p];
                g->max_hop = std::max(g->max_hop, std::abs(p - r));
            }
    if (!any) return FLOWMPM_OK;
    constexpr long long kWords = FLOWMPM_MIGRATE_WORDS;
    if (!g->rccl) {
        // one device, one stream: the owner reads the records where the sender packed them
        for (int r = 0; r < n; r++) {
            long long off = 0;
            for (int p = 0; p < n; p++) {
                const long long k = matrix[(size_t)r * n + p];
                if (k) GROUP_SIM(g, p, flowmpm_insert_arrivals(g->m[p].sim, g->m[r].leavers + off * kWords, k));
                off += k;
            }
        }
        return FLOWMPM_OK;
    }
    Rccl &R = rccl();
    for (int r = 0; r < n; r++) {
        Member &M = g->m[r];
        GROUP_HIP(g, hipSetDevice(M.device));
        GROUP_HIP(g, hipEventRecord(M.packed, M.sim->stream));
        GROUP_HIP(g, hipStreamWaitEvent(M.comm_stream, M.packed, 0));
    }
    GROUP_NCCL(g, R.GroupStart());
    for (int r = 0; r < n; r++) {
        Member &M = g->m[r];
        long long off = 0, in_off = 0;
        for (int p = 0; p < n; p++) {
            const long long k = matrix[(size_t)r * n + p];
            if (k) GROUP_NCCL(g, R.Send(M.leavers + off * kWords, (size_t)(k * kWords), kNcclInt32, p, M.comm, M.comm_stream));
            off += k;
            const long long kin = matrix[(size_t)p * n + r];
            if (kin) GROUP_NCCL(g, R.Recv(M.inbox + in_off * kWords, (size_t)(kin * kWords), kNcclInt32, p, M.comm, M.comm_stream));
            in_off += kin;
        }
    }
    GROUP_NCCL(g, R.GroupEnd());
    for (int r = 0; r < n; r++) {
        Member &M = g->m[r];
        long long in = 0;
        for (int p = 0; p < n; p++) in += matrix[(size_t)p * n + r];
        GROUP_HIP(g, hipSetDevice(M.device));
        GROUP_HIP(g, hipEventRecord(M.received, M.comm_stream));
        GROUP_HIP(g, hipStreamWaitEvent(M.sim->stream, M.received, 0));
        if (in) GROUP_SIM(g, r, flowmpm_insert_arrivals(M.sim, M.inbox, in));
    }
    return FLOWMPM_OK;
}

inline int step_once(flowmpm_slab_group *g)
{
    const int n = g->n;
    int64_t mass_words = 0, momentum_words = 0;
    GROUP_SIM(g, 0, flowmpm_halo_words(g->m[0].sim, FLOWMPM_COMP_MASS, &mass_words));
    GROUP_SIM(g, 0, flowmpm_halo_words(g->m[0].sim, FLOWMPM_COMP_MOMENTUM, &momentum_words));
    for (int r = 0; r < n; r++) GROUP_SIM(g, r, flowmpm_slab_step_begin(g->m[r].sim));
    int rc = migrate(g);
    if (rc) return rc;
    for (int r = 0; r < n; r++)
        GROUP_SIM(g, r, flowmpm_slab_step_pack_mass(g->m[r].sim, send_of(g, r, 0, 0), send_of(g, r, 1, 0)));
    if ((rc = exchange_halo(g, 0, mass_words))) return rc;
    for (int r = 0; r < n; r++) GROUP_SIM(g, r, flowmpm_slab_step_interior(g->m[r].sim));
    if ((rc = await_halo(g))) return rc;
    for (int r = 0; r < n; r++)
        GROUP_SIM(g, r, flowmpm_slab_step_boundary(g->m[r].sim, recv_of(g, r, 0, 0), recv_of(g, r, 1, 0), send_of(g, r, 0, 1),
                                                   send_of(g, r, 1, 1)));
    if ((rc = exchange_halo(g, 1, momentum_words))) return rc;
    for (int r = 0; r < n; r++) GROUP_SIM(g, r, flowmpm_slab_step_grid(g->m[r].sim));
    if ((rc = await_halo(g))) return rc;
    for (int r = 0; r < n; r++) {
        slabgroup::Member &M = g->m[r];
        GROUP_SIM(g, r, flowmpm_slab_step_end(M.sim, recv_of(g, r, 0, 1), recv_of(g, r, 1, 1), n > 1 ? g->starts.data() : nullptr,
                                              n, M.d_counts));
        if (n > 1) {
            GROUP_HIP(g, hipSetDevice(M.device));
            GROUP_HIP(g, hipMemcpyAsync(M.h_counts, M.d_counts, (size_t)n * sizeof(long long), hipMemcpyDeviceToHost, M.sim->stream));
            GROUP_HIP(g, hipEventRecord(M.counts_ready, M.sim->stream));
        }
    }
    g->pending = n > 1;
    return FLOWMPM_OK;
}

inline void release(flowmpm_slab_group *g)
{
    for (auto &M : g->m) {
        (void)hipSetDevice(M.device);
        if (M.sim) {
            if (M.sim->stream) (void)hipStreamSynchronize(M.sim->stream);
            (void)flowmpm_destroy(M.sim);
        }
        if (M.comm_stream) { (void)hipStreamSynchronize(M.comm_stream); }
        if (M.comm && rccl().CommDestroy) (void)rccl().CommDestroy(M.comm);
        for (int side = 0; side < 2; side++)
            for (int w = 0; w < 2; w++) {
                if (M.send[side][w]) (void)hipFree(M.send[side][w]);
                if (M.recv[side][w]) (void)hipFree(M.recv[side][w]);
            }
        if (M.leavers) (void)hipFree(M.leavers);
        if (M.inbox) (void)hipFree(M.inbox);
        if (M.d_counts) (void)hipFree(M.d_counts);
        if (M.h_counts) (void)hipHostFree(M.h_counts);
        if (M.counts_ready) (void)hipEventDestroy(M.counts_ready);
        if (M.packed) (void)hipEventDestroy(M.packed);
        if (M.received) (void)hipEventDestroy(M.received);
        if (M.comm_stream) (void)hipStreamDestroy(M.comm_stream);
    }
    if (g->shared_stream) {
        (void)hipSetDevice(g->m.empty() ? 0 : g->m[0].device);
        (void)hipStreamDestroy(g->shared_stream);
    }
    delete g;
}

}  // namespace slabgroup

extern "C" {

int flowmpm_slab_group_create(const flowmpm_config *base, const int32_t *plane_starts, int32_t n_slabs,
                              const int32_t *devices, const int64_t *max_particles, flowmpm_slab_group **out)
{
    using namespace slabgroup;
    if (!base || !plane_starts || !max_particles || !out) return gfail(nullptr, FLOWMPM_ERR_INVALID, "null argument");
    *out = nullptr;
    if (n_slabs < 1 || n_slabs > mpm::kMaxRanks) return gfail(nullptr, FLOWMPM_ERR_INVALID, "between 1 and 64 slabs");
    if (base->algorithm != FLOWMPM_ALGO_TILED) return gfail(nullptr, FLOWMPM_ERR_INVALID, "a slab group runs FLOWMPM_ALGO_TILED");
    if (plane_starts[0] != 0 || plane_starts[n_slabs] != base->grid_x)
        return gfail(nullptr, FLOWMPM_ERR_INVALID, "plane_starts must run from 0 to grid_x");
    for (int r = 0; r < n_slabs; r++)
        if (plane_starts[r + 1] - plane_starts[r] < 2) return gfail(nullptr, FLOWMPM_ERR_INVALID, "a slab needs at least 2 planes");
    bool one_device = true, distinct = true;
    for (int r = 0; r < n_slabs; r++) {
        const int d = devices ? devices[r] : base->device;
        one_device = one_device && d == (devices ? devices[0] : base->device);
        for (int q = 0; q < r; q++) distinct = distinct && (devices ? devices[q] : base->device) != d;
    }
    if (!one_device && !distinct)
        return gfail(nullptr, FLOWMPM_ERR_INVALID, "the slabs of a group sit all on one device or each on a device of its own");
    flowmpm_slab_group *g = new flowmpm_slab_group();
    g->n = n_slabs;
    g->rccl = !one_device;
    g->starts.assign(plane_starts, plane_starts + n_slabs + 1);
    g->m.resize(n_slabs);
    auto body = [&]() -> int {
        if (g->rccl && !rccl().load()) return gfail(g, FLOWMPM_ERR_HIP, rccl().error);
        if (!g->rccl) {
            GROUP_HIP(g, hipSetDevice(devices ? devices[0] : base->device));
            GROUP_HIP(g, hipStreamCreateWithFlags(&g->shared_stream, hipStreamNonBlocking));
        }
        for (int r = 0; r < n_slabs; r++) {
            Member &M = g->m[r];
            M.device = devices ? devices[r] : base->device;
            flowmpm_config cfg = *base;
            cfg.device = M.device;
            cfg.slab_x0 = plane_starts[r];
            cfg.slab_nx = plane_starts[r + 1] - plane_starts[r];
            cfg.max_particles = max_particles[r];
            cfg.stream = g->shared_stream;
            cfg.external_stream = g->rccl ? 0 : 1;
            int rc = flowmpm_create(&cfg, &M.sim);
            if (rc) return gfail(g, rc, "slab " + std::to_string(r) + ": " + flowmpm_last_error(nullptr));
            GROUP_HIP(g, hipSetDevice(M.device));
            for (int which = 0; which < 2; which++) {
                int64_t words = 0;
                GROUP_SIM(g, r, flowmpm_halo_words(M.sim, which == 0 ? FLOWMPM_COMP_MASS : FLOWMPM_COMP_MOMENTUM, &words));
                for (int side = 0; side < 2; side++) {
                    if ((side == 0 && r == 0) || (side == 1 && r == n_slabs - 1)) continue;
                    GROUP_HIP(g, hipMalloc(reinterpret_cast<void **>(&M.send[side][which]), (size_t)words * sizeof(int)));
                    if (g->rccl)
                        GROUP_HIP(g, hipMalloc(reinterpret_cast<void **>(&M.recv[side][which]), (size_t)words * sizeof(int)));
                }
            }
            if (n_slabs > 1) {
                M.leaver_cap = std::max<long long>(4096, max_particles[r] / 4);
                GROUP_HIP(g, hipMalloc(reinterpret_cast<void **>(&M.leavers), (size_t)M.leaver_cap * FLOWMPM_MIGRATE_WORDS * sizeof(float)));
                if (g->rccl) {
                    M.inbox_cap = M.leaver_cap;
                    GROUP_HIP(g, hipMalloc(reinterpret_cast<void **>(&M.inbox), (size_t)M.inbox_cap * FLOWMPM_MIGRATE_WORDS * sizeof(float)));
                }
                GROUP_HIP(g, hipMalloc(reinterpret_cast<void **>(&M.d_counts), (size_t)n_slabs * sizeof(long long)));
                GROUP_HIP(g, hipHostMalloc(reinterpret_cast<void **>(&M.h_counts), (size_t)n_slabs * sizeof(long long), hipHostMallocDefault));
                GROUP_HIP(g, hipEventCreateWithFlags(&M.counts_ready, hipEventDisableTiming));
            }
            if (g->rccl) {
                GROUP_HIP(g, hipStreamCreateWithFlags(&M.comm_stream, hipStreamNonBlocking));
                GROUP_HIP(g, hipEventCreateWithFlags(&M.packed, hipEventDisableTiming));
                GROUP_HIP(g, hipEventCreateWithFlags(&M.received, hipEventDisableTiming));
            }
        }
        if (g->rccl) {
            std::vector<ncclComm_t> comms(n_slabs);
            std::vector<int> devs(n_slabs);
            for (int r = 0; r < n_slabs; r++) devs[r] = g->m[r].device;
            GROUP_NCCL(g, rccl().CommInitAll(comms.data(), n_slabs, devs.data()));
            for (int r = 0; r < n_slabs; r++) g->m[r].comm = comms[r];
        }
        return FLOWMPM_OK;
    };
    int rc = body();
    if (rc) {
        const std::string msg = g->err;
        release(g);
        g_last_error = msg;
        return rc;
    }
    *out = g;
    return FLOWMPM_OK;
}

int flowmpm_slab_group_destroy(flowmpm_slab_group *g)
{
    if (g) slabgroup::release(g);
    return FLOWMPM_OK;
}

const char *flowmpm_slab_group_last_error(const flowmpm_slab_group *g) { return g ? g->err.c_str() : g_last_error.c_str(); }

int flowmpm_slab_group_size(flowmpm_slab_group *g, int32_t *n_slabs, int32_t *uses_rccl)
{
    if (!g) return slabgroup::gfail(nullptr, FLOWMPM_ERR_INVALID, "null group");
    if (n_slabs) *n_slabs = g->n;
    if (uses_rccl) *uses_rccl = g->rccl ? 1 : 0;
    return FLOWMPM_OK;
}

int flowmpm_slab_group_slab(flowmpm_slab_group *g, int32_t r, flowmpm_sim **out)
{
    if (!g || !out || r < 0 || r >= g->n) return slabgroup::gfail(g, FLOWMPM_ERR_INVALID, "no such slab");
    *out = g->m[r].sim;
    return FLOWMPM_OK;
}

// Routes host records to the slabs that own floor(x); record i becomes particle id i.
int flowmpm_slab_group_upload(flowmpm_slab_group *g, const float *aos, int64_t count)
{
    using namespace slabgroup;
    if (!g || count < 0 || (count > 0 && !aos)) return gfail(g, FLOWMPM_ERR_INVALID, "bad particle array");
    if (count > 2147483647LL) return gfail(g, FLOWMPM_ERR_INVALID, "particle ids must fit 31 bits");
    int rc = migrate(g);        // nothing of an earlier particle set stays in flight
    if (rc) return rc;
    std::vector<std::vector<float>> rec(g->n);
    std::vector<std::vector<int32_t>> ids(g->n);
    for (int64_t i = 0; i < count; i++) {
        const float *p = aos + i * FLOWMPM_PARTICLE_STRIDE;
        const int ix = (int)std::floor(p[0]);
        int owner = 0;
        for (int r = 0; r < g->n; r++)
            if (ix >= g->starts[r]) owner = r;
        rec[owner].insert(rec[owner].end(), p, p + FLOWMPM_PARTICLE_STRIDE);
        ids[owner].push_back((int32_t)i);
    }
    for (int r = 0; r < g->n; r++) {
        GROUP_SIM(g, r, flowmpm_upload_particles(g->m[r].sim, rec[r].data(), ids[r].data(), (int64_t)ids[r].size()));
        GROUP_SIM(g, r, flowmpm_set_domain_particles(g->m[r].sim, std::max<int64_t>(count, 1)));
    }
    return FLOWMPM_OK;
}

int flowmpm_slab_group_generate(flowmpm_slab_group *g, int32_t scene, uint32_t seed, int64_t count, double fill)
{
    using namespace slabgroup;
    if (!g) return gfail(nullptr, FLOWMPM_ERR_INVALID, "null group");
    int rc = migrate(g);
    if (rc) return rc;
    int64_t total = 0;
    for (int r = 0; r < g->n; r++) {
        int64_t kept = 0;
        GROUP_SIM(g, r, flowmpm_generate_particles(g->m[r].sim, scene, seed, 0, count, fill, &kept));
        GROUP_SIM(g, r, flowmpm_set_domain_particles(g->m[r].sim, std::max<int64_t>(count, 1)));
        total += kept;
    }
    if (total != count) return gfail(g, FLOWMPM_ERR_INVALID, "the slabs kept " + std::to_string(total) + " of " + std::to_string(count) + " particles");
    return FLOWMPM_OK;
}

int flowmpm_slab_group_set_uniforms(flowmpm_slab_group *g, const flowmpm_uniforms *u)
{
    if (!g || !u) return slabgroup::gfail(g, FLOWMPM_ERR_INVALID, "null argument");
    for (int r = 0; r < g->n; r++) GROUP_SIM(g, r, flowmpm_set_uniforms(g->m[r].sim, u));
    return FLOWMPM_OK;
}

int flowmpm_slab_group_set_mode_uniforms(flowmpm_slab_group *g, const flowmpm_mode_uniforms *m)
{
    if (!g || !m) return slabgroup::gfail(g, FLOWMPM_ERR_INVALID, "null argument");
    for (int r = 0; r < g->n; r++) GROUP_SIM(g, r, flowmpm_set_mode_uniforms(g->m[r].sim, m));
    return FLOWMPM_OK;
}

// renderer.computeAsync([...5 kernels]) (SIM:2538-2539) over the decomposed domain, n_steps times.
int flowmpm_slab_group_step(flowmpm_slab_group *g, int32_t n_steps, double dtime)
{
    using namespace slabgroup;
    if (!g || n_steps < 0) return gfail(g, FLOWMPM_ERR_INVALID, "bad step count");
    std::vector<float> time0(g->n);
    for (int r = 0; r < g->n; r++) time0[r] = g->m[r].sim->uniforms.time;
    int rc = FLOWMPM_OK;
    for (int k = 0; k < n_steps && rc == FLOWMPM_OK; k++) {
        for (int r = 0; r < g->n; r++) g->m[r].sim->params.time = (float)((double)time0[r] + (double)k * dtime);
        rc = step_once(g);
    }
    for (int r = 0; r < g->n; r++) g->m[r].sim->params.time = time0[r];
    return rc;
}

// Completes the migration of the last step (the particles then sit on their owners) and waits for the device(s).
int flowmpm_slab_group_synchronize(flowmpm_slab_group *g)
{
    using namespace slabgroup;
    if (!g) return gfail(nullptr, FLOWMPM_ERR_INVALID, "null group");
    int rc = migrate(g);
    if (rc) return rc;
    for (int r = 0; r < g->n; r++) GROUP_SIM(g, r, flowmpm_synchronize(g->m[r].sim));
    return FLOWMPM_OK;
}

int flowmpm_slab_group_get_num_particles(flowmpm_slab_group *g, int64_t *total, int64_t *per_slab)
{
    using namespace slabgroup;
    if (!g) return gfail(nullptr, FLOWMPM_ERR_INVALID, "null group");
    int rc = migrate(g);
    if (rc) return rc;
    int64_t sum = 0;
    for (int r = 0; r < g->n; r++) {
        sum += g->m[r].sim->resident;
        if (per_slab) per_slab[r] = g->m[r].sim->resident;
    }
    if (total) *total = sum;
    return FLOWMPM_OK;
}

// All records in particle-id order (ids 0 .. count - 1 as uploaded or generated).
int flowmpm_slab_group_download_particles(flowmpm_slab_group *g, float *aos, int64_t capacity, int64_t *count)
{
    using namespace slabgroup;
    if (!g) return gfail(nullptr, FLOWMPM_ERR_INVALID, "null group");
    int rc = migrate(g);
    if (rc) return rc;
    int64_t total = 0;
    for (int r = 0; r < g->n; r++) total += g->m[r].sim->resident;
    if (count) *count = total;
    if (!aos || capacity < total) return gfail(g, FLOWMPM_ERR_INVALID, "output array too small");
    std::vector<float> rec;
    std::vector<int32_t> ids;
    for (int r = 0; r < g->n; r++) {
        const int64_t k = g->m[r].sim->resident;
        if (k == 0) continue;
        rec.resize((size_t)k * FLOWMPM_PARTICLE_STRIDE);
        ids.resize((size_t)k);
        int64_t got = 0;
        GROUP_SIM(g, r, flowmpm_download_particles(g->m[r].sim, rec.data(), ids.data(), k, &got));
        for (int64_t i = 0; i < got; i++) {
            if (ids[i] < 0 || ids[i] >= total) return gfail(g, FLOWMPM_ERR_INVALID, "a particle id lies outside 0 .. count - 1");
            memcpy(aos + (int64_t)ids[i] * FLOWMPM_PARTICLE_STRIDE, rec.data() + i * FLOWMPM_PARTICLE_STRIDE,
                   FLOWMPM_PARTICLE_STRIDE * sizeof(float));
        }
    }
    return FLOWMPM_OK;
}

// One per-particle field in particle-id order (what the reference's renderers bind).
int flowmpm_slab_group_download_field(flowmpm_slab_group *g, int32_t field, float *out, int64_t capacity)
{
    using namespace slabgroup;
    if (!g) return gfail(nullptr, FLOWMPM_ERR_INVALID, "null group");
    int rc = migrate(g);
    if (rc) return rc;
    int64_t total = 0;
    for (int r = 0; r < g->n; r++) total += g->m[r].sim->resident;
    if (!out || capacity < total) return gfail(g, FLOWMPM_ERR_INVALID, "output array too small");
    std::vector<float> vals;
    std::vector<int32_t> ids;
    for (int r = 0; r < g->n; r++) {
        flowmpm_sim *s = g->m[r].sim;
        const long long k = s->resident;
        if (k == 0) continue;
        int slot, width;
        if (field_columns(s, field, &slot, &width)) return gfail(g, FLOWMPM_ERR_INVALID, "unknown field");
        vals.resize((size_t)k * width);
        ids.resize((size_t)k);
        GROUP_HIP(g, hipSetDevice(s->cfg.device));
        // slot order plus the ids; the host puts them in id order
        const size_t row_bytes = (size_t)(width + 1) * sizeof(float);
        void *stage = nullptr;
        size_t got = 0;
        if (stage_device(s, (size_t)k * row_bytes, &stage, &got)) return gfail(g, FLOWMPM_ERR_HIP, s->err);
        const long long rows_max = std::max<long long>(1, (long long)(got / row_bytes));
        for (long long lo = 0; lo < k; lo += rows_max) {
            const long long rows = std::min(rows_max, k - lo);
            float *d_vals = static_cast<float *>(stage);
            int *d_ids = reinterpret_cast<int *>(d_vals + rows * width);
            hipLaunchKernelGGL(mpm::gather_field_slots_kernel, dim3(blocks_for(rows, 256)), dim3(256), 0, s->stream, s->store, lo,
                               rows, slot, width, d_vals, d_ids);
            GROUP_HIP(g, hipGetLastError());
            GROUP_HIP(g, hipMemcpyAsync(ids.data() + lo, d_ids, (size_t)rows * sizeof(int), hipMemcpyDeviceToHost, s->stream));
            if (copy_out_and_wait(s, vals.data() + lo * width, d_vals, (size_t)rows * width * sizeof(float), "group field"))
                return gfail(g, FLOWMPM_ERR_HIP, s->err);
        }
        for (long long i = 0; i < k; i++) {
            if (ids[i] < 0 || ids[i] >= total) return gfail(g, FLOWMPM_ERR_INVALID, "a particle id lies outside 0 .. count - 1");
            for (int w = 0; w < width; w++) out[(int64_t)ids[i] * width + w] = vals[(size_t)i * width + w];
        }
    }
    return FLOWMPM_OK;
}

// [grid x, y, z momentum words, grid mass words, particles] summed over the slabs (64-bit).
int flowmpm_slab_group_conservation(flowmpm_slab_group *g, int64_t sums[5])
{
    using namespace slabgroup;
    if (!g || !sums) return gfail(g, FLOWMPM_ERR_INVALID, "null argument");
    int rc = migrate(g);
    if (rc) return rc;
    for (int k = 0; k < 5; k++) sums[k] = 0;
    for (int r = 0; r < g->n; r++) {
        int64_t part[4];
        GROUP_SIM(g, r, flowmpm_grid_sums(g->m[r].sim, part));
        for (int k = 0; k < 4; k++) sums[k] += part[k];
        sums[4] += g->m[r].sim->resident;
    }
    return FLOWMPM_OK;
}

int flowmpm_slab_group_migration_stats(flowmpm_slab_group *g, int64_t *migrated, int32_t *max_hop)
{
    if (!g) return slabgroup::gfail(nullptr, FLOWMPM_ERR_INVALID, "null group");
    if (migrated) *migrated = g->migrated;
    if (max_hop) *max_hop = g->max_hop;
    return FLOWMPM_OK;
}

int flowmpm_slab_group_plane_starts(flowmpm_slab_group *g, int32_t *plane_starts)
{
    if (!g || !plane_starts) return slabgroup::gfail(g, FLOWMPM_ERR_INVALID, "null argument");
    for (int r = 0; r <= g->n; r++) plane_starts[r] = g->starts[r];
    return FLOWMPM_OK;
}

int flowmpm_slab_group_rebalance(flowmpm_slab_group *g, double max_move_fraction, int32_t *changed)
{
    using namespace slabgroup;
    if (!g) return gfail(nullptr, FLOWMPM_ERR_INVALID, "null group");
    if (changed) *changed = 0;
    if (!(max_move_fraction > 0.0 && max_move_fraction <= 1.0)) return gfail(g, FLOWMPM_ERR_INVALID, "0 < max_move_fraction <= 1");
    if (g->n < 2) return FLOWMPM_OK;
    int rc = migrate(g);                  // every particle sits on its owner
    if (rc) return rc;
    const int gx = g->m[0].sim->cfg.grid_x;
    std::vector<int64_t> hist((size_t)gx, 0), part((size_t)gx);
    int64_t total = 0;
    for (int r = 0; r < g->n; r++) {
        GROUP_SIM(g, r, flowmpm_plane_histogram(g->m[r].sim, part.data(), gx));
        for (int x = 0; x < gx; x++) { hist[(size_t)x] += part[(size_t)x]; total += part[(size_t)x]; }
    }
    if (total == 0) return FLOWMPM_OK;
    std::vector<int> max_nx((size_t)g->n);
    for (int r = 0; r < g->n; r++) max_nx[(size_t)r] = g->m[r].sim->plane_cap - 2;
    std::vector<int> next = mpm::limited_plane_starts(hist, g->starts, max_nx, (int64_t)(max_move_fraction * (double)total / g->n));
    if (next == g->starts) return FLOWMPM_OK;
    for (int r = 0; r < g->n; r++) GROUP_SIM(g, r, flowmpm_set_slab(g->m[r].sim, next[(size_t)r], next[(size_t)r + 1] - next[(size_t)r]));
    g->starts = next;
    // the classification under the new cut: the next step's migration moves the planes that changed hands
    for (int r = 0; r < g->n; r++) {
        Member &M = g->m[r];
        GROUP_SIM(g, r, flowmpm_find_leavers(M.sim, g->starts.data(), g->n, M.d_counts));
        GROUP_HIP(g, hipSetDevice(M.device));
        GROUP_HIP(g, hipMemcpyAsync(M.h_counts, M.d_counts, (size_t)g->n * sizeof(long long), hipMemcpyDeviceToHost, M.sim->stream));
        GROUP_HIP(g, hipEventRecord(M.counts_ready, M.sim->stream));
    }
    g->pending = true;
    if (changed) *changed = 1;
    return FLOWMPM_OK;
}

// RCCL in a world of one: a communicator on `device`, a buffer sent to itself inside one group call and
// compared.  What a host checks before it creates a group over several devices.
int flowmpm_rccl_selftest(int32_t device, int64_t words)
{
    using namespace slabgroup;
    if (words < 1 || words > (1 << 28)) return gfail(nullptr, FLOWMPM_ERR_INVALID, "bad word count");
    Rccl &R = rccl();
    if (!R.load()) return gfail(nullptr, FLOWMPM_ERR_HIP, R.error);
    flowmpm_slab_group *none = nullptr;
    GROUP_HIP(none, hipSetDevice(device));
    ncclComm_t comm = nullptr;
    const int dev = device;
    GROUP_NCCL(none, R.CommInitAll(&comm, 1, &dev));
    int *src = nullptr, *dst = nullptr;
    hipStream_t stream = nullptr;
    std::vector<int> host((size_t)words), back((size_t)words);
    for (int64_t i = 0; i < words; i++) host[(size_t)i] = (int)(i * 2654435761u);
    int rc = FLOWMPM_OK;
    auto body = [&]() -> int {
        GROUP_HIP(none, hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        GROUP_HIP(none, hipMalloc(reinterpret_cast<void **>(&src), (size_t)words * sizeof(int)));
        GROUP_HIP(none, hipMalloc(reinterpret_cast<void **>(&dst), (size_t)words * sizeof(int)));
        GROUP_HIP(none, hipMemcpyAsync(src, host.data(), (size_t)words * sizeof(int), hipMemcpyHostToDevice, stream));
        GROUP_HIP(none, hipMemsetAsync(dst, 0, (size_t)words * sizeof(int), stream));
        GROUP_NCCL(none, R.GroupStart());
        GROUP_NCCL(none, R.Send(src, (size_t)words, kNcclInt32, 0, comm, stream));
        GROUP_NCCL(none, R.Recv(dst, (size_t)words, kNcclInt32, 0, comm, stream));
        GROUP_NCCL(none, R.GroupEnd());
        GROUP_HIP(none, hipMemcpyAsync(back.data(), dst, (size_t)words * sizeof(int), hipMemcpyDeviceToHost, stream));
        GROUP_HIP(none, hipStreamSynchronize(stream));
        if (back != host) return gfail(nullptr, FLOWMPM_ERR_HIP, "rccl selftest: the received words differ from the sent ones");
        return FLOWMPM_OK;
    };
    rc = body();
    if (src) (void)hipFree(src);
    if (dst) (void)hipFree(dst);
    if (stream) (void)hipStreamDestroy(stream);
    (void)R.CommDestroy(comm);
    return rc;
}

}  // extern "C"
