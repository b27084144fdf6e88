// gpusort_mgpu.hpp — multi-GPU sharded sort behind the C-ABI (include/gpusort.h, "multi-GPU" section); part of
// the gpusort_capi.hip translation unit (it drives the handle's internals: prologue, pass launchers, sort_impl).
//
// One process per GPU.  BASELINE.json configs[3]: "MSD bucket split + RCCL Alltoallv across 8 MI355X then per-GPU
// OneSweep"; the reference has no multi-GPU code.  Pipeline of gs_onesweep_sort_sharded, everything
// enqueued on the caller's stream:
//   1. GlobalHistogram + Scan of the shard on its top byte (the sort's own kernels, position-segment chains),
//      folded to 256 counts on the device
//   2. all-gather of the 256 counts (RCCL ncclAllGather over xGMI)
//   3. msd_plan_kernel: splitters, per-peer send/receive counts, overflow check — on the device
//   4. ONE small device-to-host copy (the plan: a few dozen words) and ONE event wait: RCCL's send/recv take their
//      counts as host integers, so this wait is inherent in the exchange, and it is the only one
//   5. the stable DigitBinningPass on the top byte groups the shard by destination (reusing the scan state of 1) — and, inside a
//      destination, by top byte: this IS the first pass of the receiver's two-level sort (hybrid_kernels.hpp, pass A)
//   6. bucket exchange: grouped ncclSend / ncclRecv pairs — point-to-point on all xGMI links at once.  Pairs: the KEYS go
//      first, on the caller's stream; the VALUES follow on a second communicator (ncclCommSplit) and a second stream, so
//      that step 7 starts on the keys while the values are still on the links.  Round 6: ONE MESSAGE PER (peer, top byte) — all
//      of a call's messages inside one ncclGroup — so that the receiver chooses where each (source, top byte) segment lands: a
//      bucket that will be offered the two-level plan is landed BIN-MAJOR (top byte, then source rank, then the source's order:
//      exactly the stable order of a top-byte partition of the concatenated sources) in the local sort's alternate buffer,
//   7. and the local sort of the received bucket starts at pass B (sort_impl's `pregrouped`): histogram sweep + Scan, the
//      DigitBinningPass on byte 2, the bucket-local sort — 20 B/key instead of 28: round 5 did the top-byte partition twice, once
//      per side of the exchange.  Buckets below the plan's size, the 12-bit split and ncclAllToAllv (one buffer pair per call)
//      keep the source-major layout and the full local sort; a bucket whose keys void the plan on the device (skew) is copied to
//      the caller's buffer by hy_void_copy_kernel and takes the four LSD passes
//   8. a one-word all-gather of every rank's status closes the call (see FAILURES)
// Why grouped send/recv and not ncclAllToAllv (rccl.h:815, the call BASELINE.json names): RCCL implements AllToAllv as
// exactly this group of sends and receives, but through one entry point that takes ONE buffer pair — keys and values
// would be two calls on one communicator, serialised, and the call cannot skip empty peers or put a peer's two arrays
// back to back.  gs_mgpu_options::alltoallv = 1 (or gs_mgpu_set_alltoallv) switches the exchange to ncclAllToAllv (one call per array) for comparison on
// hardware; results are identical.
// FAILURES.  A rank that fails alone must not leave its peers inside a collective.  Before the gather (histogram,
// allocation, launch errors): the rank gathers a POISONED row (MSD_POISON in bin 0), the plan kernel of every rank sees
// it, and all ranks return GS_ERR_COMM together right after the call's one host wait.  After the plan: the rank still
// runs its exchange as planned (its peers' receives complete), reports its status in the closing all-gather, and
// returns its own error; the peers learn it from gs_mgpu_check() — GS_ERR_COMM — at their next synchronisation.  A
// context that has failed is torn down with ncclCommAbort instead of ncclCommDestroy.
// Skewed shards (a top-byte bucket would not fit a rank): the split is redone at the 12-bit prefix (4096 bins, shard
// ordered by its top two bytes); if that does not fit either, every rank returns GS_ERR_SIZE together.
// RCCL is loaded lazily (dlopen "librccl.so.1") so that single-GPU users of libgpusort.so do not need it; tests
// inject a host-staged transport (gs_mgpu_create_with_transport) to run several ranks on one GPU.
#include <dlfcn.h>

#include <vector>

#include "msd_kernels.hpp"

namespace {

// ---- RCCL, bound at run time ------------------------------------------------------------------------------
struct Rccl {
    typedef struct { char internal[128]; } UniqueId;  // ncclUniqueId (rccl.h:40-43)
    int (*GetUniqueId)(UniqueId*);
    int (*CommInitRank)(void** comm, int nranks, UniqueId id, int rank);
    int (*CommDestroy)(void* comm);
    int (*AllGather)(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t s);
    int (*Send)(const void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t s);
    int (*Recv)(void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t s);
    int (*GroupStart)();
    int (*GroupEnd)();
    const char* (*GetErrorString)(int);
    int (*CommSplit)(void* comm, int color, int key, void** newcomm, void* config);  // optional (NCCL >= 2.18)
    int (*CommAbort)(void* comm);                                                     // optional
    int (*AllToAllv)(const void* send, const size_t* sendcounts, const size_t* sdispls, void* recv, const size_t* recvcounts,
                     const size_t* rdispls, int dtype, void* comm, hipStream_t s);    // optional (RCCL extension, rccl.h:815)
    bool ok;
};
constexpr int kNcclUint8 = 1, kNcclUint32 = 3;  // ncclDataType_t (rccl.h:460-464)
static_assert(GS_MGPU_UNIQUE_ID_BYTES == 128, "ncclUniqueId is 128 bytes");

Rccl* rccl() {
    static Rccl r = [] {
        Rccl x{};
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return x;
#define GS_SYM(field, name) x.field = reinterpret_cast<decltype(x.field)>(dlsym(h, name))
        GS_SYM(GetUniqueId, "ncclGetUniqueId");
        GS_SYM(CommInitRank, "ncclCommInitRank");
        GS_SYM(CommDestroy, "ncclCommDestroy");
        GS_SYM(AllGather, "ncclAllGather");
        GS_SYM(Send, "ncclSend");
        GS_SYM(Recv, "ncclRecv");
        GS_SYM(GroupStart, "ncclGroupStart");
        GS_SYM(GroupEnd, "ncclGroupEnd");
        GS_SYM(GetErrorString, "ncclGetErrorString");
        GS_SYM(CommSplit, "ncclCommSplit");
        GS_SYM(CommAbort, "ncclCommAbort");
        GS_SYM(AllToAllv, "ncclAllToAllv");
#undef GS_SYM
        x.ok = x.GetUniqueId && x.CommInitRank && x.CommDestroy && x.AllGather && x.Send && x.Recv && x.GroupStart && x.GroupEnd;
        return x;
    }();
    return r.ok ? &r : nullptr;
}

thread_local int g_last_rccl_error = 0;

struct RcclTransport {
    void* comm;
    uint32_t rank, world;
    int alltoallv;  // exchange through ncclAllToAllv instead of grouped send / recv (gs_mgpu_options::alltoallv, gs_mgpu_set_alltoallv)
};

int rccl_all_gather_u32(void* user, const void* d_send, void* d_recv, size_t count, void* stream) {
    RcclTransport* t = static_cast<RcclTransport*>(user);
    const int e = rccl()->AllGather(d_send, d_recv, count, kNcclUint32, t->comm, static_cast<hipStream_t>(stream));
    if (e) g_last_rccl_error = e;
    return e;
}

// keys and values in ONE group: every (array, peer) pair is a send and a receive that progress together
int rccl_exchange(void* user, uint32_t n_arrays, const void* const* d_send, void* const* d_recv, const uint32_t* elem_bytes,
                  const uint32_t* send_counts, const uint32_t* send_displs, const uint32_t* recv_counts,
                  const uint32_t* recv_displs, void* stream) {
    RcclTransport* t = static_cast<RcclTransport*>(user);
    hipStream_t s = static_cast<hipStream_t>(stream);
    Rccl* r = rccl();
    for (uint32_t a = 0; a < n_arrays; ++a) {  // own bucket: a copy inside the device
        const size_t eb = elem_bytes[a];
        if (send_counts[t->rank] &&
            hipMemcpyAsync(static_cast<char*>(d_recv[a]) + (size_t)recv_displs[t->rank] * eb,
                           static_cast<const char*>(d_send[a]) + (size_t)send_displs[t->rank] * eb,
                           (size_t)send_counts[t->rank] * eb, hipMemcpyDeviceToDevice, s) != hipSuccess)
            return -1;
    }
    if (t->world == 1) return 0;
    if (t->alltoallv && r->AllToAllv) {  // one call per array; counts and displacements in bytes (the own bucket went by copy)
        std::vector<size_t> sc(t->world), sd(t->world), rc(t->world), rd(t->world);
        int e = 0;
        for (uint32_t a = 0; a < n_arrays && !e; ++a) {
            const size_t eb = elem_bytes[a];
            for (uint32_t p = 0; p < t->world; ++p) {
                sc[p] = p == t->rank ? 0 : (size_t)send_counts[p] * eb;
                rc[p] = p == t->rank ? 0 : (size_t)recv_counts[p] * eb;
                sd[p] = (size_t)send_displs[p] * eb;
                rd[p] = (size_t)recv_displs[p] * eb;
            }
            e = r->AllToAllv(d_send[a], sc.data(), sd.data(), d_recv[a], rc.data(), rd.data(), kNcclUint8, t->comm, s);
        }
        if (e) g_last_rccl_error = e;
        return e;
    }
    int e = r->GroupStart();
    for (uint32_t a = 0; a < n_arrays && !e; ++a) {
        const size_t eb = elem_bytes[a];
        for (uint32_t p = 0; p < t->world && !e; ++p) {
            if (p == t->rank) continue;
            if (send_counts[p])
                e = r->Send(static_cast<const char*>(d_send[a]) + (size_t)send_displs[p] * eb, (size_t)send_counts[p] * eb, kNcclUint8,
                            (int)p, t->comm, s);
            if (!e && recv_counts[p])
                e = r->Recv(static_cast<char*>(d_recv[a]) + (size_t)recv_displs[p] * eb, (size_t)recv_counts[p] * eb, kNcclUint8, (int)p,
                            t->comm, s);
        }
    }
    const int e2 = r->GroupEnd();
    if (e || e2) g_last_rccl_error = e ? e : e2;
    return e ? e : e2;
}

}  // namespace

struct gs_mgpu {
    uint32_t rank, world, shard_keys, capacity, value_bytes;
    gs_mode mode;
    gs_onesweep* sorter;       // local engine: scan state for `capacity` keys
    gs_mgpu_transport transport;
    RcclTransport rccl_state;  // when the transport is RCCL
    RcclTransport rccl_state2; // ... and the second communicator (values), if ncclCommSplit exists
    gs_mgpu_transport transport2;  // the transport the values travel on (== transport without a second communicator)
    bool owns_comm;
    hipStream_t s2;            // second stream: the value exchange and the closing status gather
    hipEvent_t ev_part, ev_vals, ev_tail;
    uint32_t *d_status;        // [0] this rank's status of the running call, [1 .. world] every rank's (gathered)
    uint32_t* h_status;        // pinned mirror
    int overlap;               // values on the second stream (gs_mgpu_options::overlap, default 1)
    int alltoallv;             // the RCCL transport exchanges through ncclAllToAllv (gs_mgpu_options::alltoallv)
    int by_bin;                // the grouped exchange goes one message per (peer, top byte) (gs_mgpu_options::by_bin, default 1)
    int failed;                // a call on this context has failed: destroy aborts the communicators
    bool call_complete = true; // the last gs_onesweep_sort_sharded ran through its closing status gather and returned GS_OK on this
                               // rank: only then are d_status[1..world] THIS call's words (gs_mgpu_check reads nothing otherwise)
    int debug_fail;            // test hook: 1 = fail before the gather, 2 = fail after the plan (next call only)
    uint32_t *part_keys;       // shard grouped by destination; alt buffer of the local sort afterwards
    void* part_vals;
    uint32_t *d_hist, *d_table, *d_plan;  // nbins, world x nbins, plan words
    uint32_t* h_plan;          // pinned mirror of the plan
    uint32_t* h_table;         // pinned mirror of the gathered [source][top byte] table (world x 256): the exchange goes bin by bin
    uint32_t last_pregrouped;  // the last call landed its bucket bin-major and its local sort skipped the top-byte pass
    hipEvent_t ev_plan, ev[5];
    int force_exchange;        // run split + exchange even with one rank (tests)
    float last_ms[4];
    uint64_t last_sent, last_recv;
    uint32_t last_fine;
    bool prof_pending;
};

namespace {

gs_status mgpu_alloc(gs_mgpu* c) {
    const size_t scratch = c->shard_keys > c->capacity ? c->shard_keys : c->capacity;
    GS_HIP(hipMalloc(&c->part_keys, scratch * sizeof(uint32_t)));
    if (c->value_bytes) GS_HIP(hipMalloc(&c->part_vals, scratch * (size_t)c->value_bytes));
    GS_HIP(hipMalloc(&c->d_hist, 4096 * sizeof(uint32_t)));
    GS_HIP(hipMalloc(&c->d_table, (size_t)c->world * 4096 * sizeof(uint32_t)));
    GS_HIP(hipMalloc(&c->d_plan, gs::plan_words(c->world) * sizeof(uint32_t)));
    GS_HIP(hipHostMalloc(&c->h_plan, gs::plan_words(c->world) * sizeof(uint32_t), hipHostMallocDefault));
    GS_HIP(hipHostMalloc(&c->h_table, (size_t)c->world * gs::RADIX * sizeof(uint32_t), hipHostMallocDefault));
    GS_HIP(hipEventCreateWithFlags(&c->ev_plan, hipEventDisableTiming));
    for (auto& e : c->ev) GS_HIP(hipEventCreate(&e));
    GS_HIP(hipStreamCreateWithFlags(&c->s2, hipStreamNonBlocking));
    GS_HIP(hipEventCreateWithFlags(&c->ev_part, hipEventDisableTiming));
    GS_HIP(hipEventCreateWithFlags(&c->ev_vals, hipEventDisableTiming));
    GS_HIP(hipEventCreateWithFlags(&c->ev_tail, hipEventDisableTiming));
    GS_HIP(hipMalloc(&c->d_status, (c->world + 1) * sizeof(uint32_t)));
    GS_HIP(hipMemset(c->d_status, 0, (c->world + 1) * sizeof(uint32_t)));
    GS_HIP(hipHostMalloc(&c->h_status, (c->world + 1) * sizeof(uint32_t), hipHostMallocDefault));
    return GS_OK;
}

gs_status mgpu_new(gs_mgpu** out, uint32_t rank, uint32_t world, uint32_t shard_keys, uint32_t capacity, gs_mode mode,
                   uint32_t value_bytes, const gs_mgpu_options* options) {
    if (!out) return GS_ERR_ARG;
    *out = nullptr;
    gs_mgpu_options o;
    gs_mgpu_options_default(&o);
    if (options) {
        if (options->struct_size != sizeof(gs_mgpu_options)) return GS_ERR_ARG;
        o = *options;
    }
    if (world == 0 || world > gs::MSD_MAX_WORLD || rank >= world) return GS_ERR_ARG;
    if (shard_keys == 0 || shard_keys > GS_MAX_KEYS || capacity < shard_keys || capacity > GS_MAX_KEYS) return GS_ERR_SIZE;
    gs_mgpu* c = new (std::nothrow) gs_mgpu();
    if (!c) return GS_ERR_ARG;
    c->rank = rank; c->world = world; c->shard_keys = shard_keys; c->capacity = capacity;
    c->mode = mode; c->value_bytes = mode == GS_MODE_PAIRS ? value_bytes : 0;
    c->force_exchange = o.force_exchange ? 1 : 0;
    c->overlap = o.overlap ? 1 : 0;
    c->alltoallv = o.alltoallv ? 1 : 0;
    c->by_bin = o.by_bin ? 1 : 0;
    gs_status st = gs_onesweep_create_ex(&c->sorter, capacity, mode, value_bytes, o.sorter.struct_size ? &o.sorter : nullptr);
    if (st == GS_OK) st = mgpu_alloc(c);
    if (st != GS_OK) { gs_mgpu_destroy(c); return st; }
    *out = c;
    return GS_OK;
}

// steps 1-4 for one granularity: histogram of the shard -> gathered table -> plan on the host.  fine = 12-bit prefix.
// A rank whose own part fails (n == 0 shards have none) still gathers — a poisoned row — so that nobody waits for it; the
// plan of every rank then says PLAN_PEER_FAILED.  *local = the rank's own error, if any.
gs_status mgpu_plan(gs_mgpu* c, const void* d_keys, uint32_t n, gs_key_type kt, hipStream_t s, bool fine, PassPlan* pp, gs_status* local) {
    gs_onesweep* h = c->sorter;
    const uint32_t nbins = fine ? 4096u : gs::RADIX;
    gs_status st = GS_OK;
    if (c->debug_fail == 1) st = GS_ERR_HIP;  // test hook: as if the histogram launch had failed
    if (st == GS_OK && n) st = fine ? prologue(h, d_keys, n, kt, s, 2, 2, pp) : prologue(h, d_keys, n, kt, s, 3, 1, pp);
    if (st == GS_OK && n) {
        hipLaunchKernelGGL(gs::msd_fold_kernel, dim3(nbins / 256), dim3(256), 0, s, h->slab + SLAB_HIST, nbins, c->d_hist);
        if (fine) {  // no pass follows this prologue: hand HIST back zeroed
            if (zero_hist(h, s) != hipSuccess) st = GS_ERR_HIP;
            h->hist_dirty = false;
        }
    } else if (st == GS_OK) {
        if (hipMemsetAsync(c->d_hist, 0, 4096 * sizeof(uint32_t), s) != hipSuccess) st = GS_ERR_HIP;
    }
    if (st != GS_OK) {  // poison the row: every rank's plan will say so
        *local = st;
        static const uint32_t poison = gs::MSD_POISON;
        (void)hipMemsetAsync(c->d_hist, 0, 4096 * sizeof(uint32_t), s);
        (void)hipMemcpyAsync(c->d_hist, &poison, sizeof(uint32_t), hipMemcpyHostToDevice, s);
    }
    if (c->transport.all_gather_u32(c->transport.user, c->d_hist, c->d_table, nbins, s) != 0) return GS_ERR_COMM;
    hipLaunchKernelGGL(gs::msd_plan_kernel, dim3(1), dim3(256), 0, s, c->d_table, nbins, c->world, c->rank, c->capacity, c->d_plan);
    GS_HIP(hipMemcpyAsync(c->h_plan, c->d_plan, gs::plan_words(c->world) * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    if (!fine) GS_HIP(hipMemcpyAsync(c->h_table, c->d_table, (size_t)c->world * gs::RADIX * sizeof(uint32_t), hipMemcpyDeviceToHost, s));  // 8 KiB at world 8
    GS_HIP(hipEventRecord(c->ev_plan, s));
    GS_HIP(hipEventSynchronize(c->ev_plan));  // the ONE host wait of the pipeline: send/recv counts are host integers
    return GS_OK;
}

}  // namespace

extern "C" {

int gs_last_rccl_error(void) { return g_last_rccl_error; }

gs_status gs_mgpu_get_unique_id(uint8_t id[GS_MGPU_UNIQUE_ID_BYTES]) {
    if (!id) return GS_ERR_ARG;
    Rccl* r = rccl();
    if (!r) return GS_ERR_COMM;
    Rccl::UniqueId u;
    const int e = r->GetUniqueId(&u);
    if (e) { g_last_rccl_error = e; return GS_ERR_COMM; }
    memcpy(id, u.internal, GS_MGPU_UNIQUE_ID_BYTES);
    return GS_OK;
}

void gs_mgpu_options_default(gs_mgpu_options* o) {
    if (!o) return;
    memset(o, 0, sizeof(*o));
    o->struct_size = (uint32_t)sizeof(*o);
    o->overlap = 1;
    o->by_bin = 1;
}

gs_status gs_mgpu_create(gs_mgpu** out, const uint8_t id[GS_MGPU_UNIQUE_ID_BYTES], uint32_t rank, uint32_t world,
                         uint32_t shard_keys, uint32_t capacity, gs_mode mode, uint32_t value_bytes) {
    return gs_mgpu_create_ex(out, id, rank, world, shard_keys, capacity, mode, value_bytes, nullptr);
}

gs_status gs_mgpu_set_alltoallv(gs_mgpu* c, int on) {
    if (!c) return GS_ERR_ARG;
    c->alltoallv = on ? 1 : 0;
    c->rccl_state.alltoallv = c->alltoallv;
    c->rccl_state2.alltoallv = c->alltoallv;
    return GS_OK;
}

gs_status gs_mgpu_create_ex(gs_mgpu** out, const uint8_t id[GS_MGPU_UNIQUE_ID_BYTES], uint32_t rank, uint32_t world,
                            uint32_t shard_keys, uint32_t capacity, gs_mode mode, uint32_t value_bytes, const gs_mgpu_options* options) {
    if (!id) return GS_ERR_ARG;
    Rccl* r = rccl();
    if (!r) return GS_ERR_COMM;
    gs_status st = mgpu_new(out, rank, world, shard_keys, capacity, mode, value_bytes, options);
    if (st != GS_OK) return st;
    gs_mgpu* c = *out;
    Rccl::UniqueId u;
    memcpy(u.internal, id, GS_MGPU_UNIQUE_ID_BYTES);
    void* comm = nullptr;
    const int e = r->CommInitRank(&comm, (int)world, u, (int)rank);  // collective: every rank of the job is in here
    if (e) {
        g_last_rccl_error = e;
        gs_mgpu_destroy(c);
        *out = nullptr;
        return GS_ERR_COMM;
    }
    const int a2av = c->alltoallv;
    c->rccl_state = RcclTransport{comm, rank, world, a2av};
    c->owns_comm = true;
    c->transport = gs_mgpu_transport{&c->rccl_state, rccl_all_gather_u32, rccl_exchange};
    c->transport2 = c->transport;
    c->rccl_state2 = RcclTransport{nullptr, rank, world, a2av};
    if (world > 1 && c->value_bytes && c->overlap && r->CommSplit) {  // collective: the values' own communicator
        void* comm2 = nullptr;
        const int e2 = r->CommSplit(comm, 0, (int)rank, &comm2, nullptr);
        if (e2 == 0 && comm2) {
            c->rccl_state2.comm = comm2;
            c->transport2 = gs_mgpu_transport{&c->rccl_state2, rccl_all_gather_u32, rccl_exchange};
        } else {
            g_last_rccl_error = e2;  // not fatal: keys and values share the first communicator, one after the other
        }
    }
    return GS_OK;
}

gs_status gs_mgpu_create_with_transport(gs_mgpu** out, const gs_mgpu_transport* t, uint32_t rank, uint32_t world,
                                        uint32_t shard_keys, uint32_t capacity, gs_mode mode, uint32_t value_bytes) {
    return gs_mgpu_create_with_transport_ex(out, t, rank, world, shard_keys, capacity, mode, value_bytes, nullptr);
}

gs_status gs_mgpu_create_with_transport_ex(gs_mgpu** out, const gs_mgpu_transport* t, uint32_t rank, uint32_t world,
                                           uint32_t shard_keys, uint32_t capacity, gs_mode mode, uint32_t value_bytes, const gs_mgpu_options* options) {
    if (!t || !t->all_gather_u32 || !t->exchange) return GS_ERR_ARG;
    gs_status st = mgpu_new(out, rank, world, shard_keys, capacity, mode, value_bytes, options);
    if (st != GS_OK) return st;
    (*out)->transport = *t;
    (*out)->transport2 = *t;
    (*out)->owns_comm = false;
    return GS_OK;
}

gs_status gs_mgpu_destroy(gs_mgpu* c) {
    if (!c) return GS_ERR_ARG;
    if (c->owns_comm && rccl()) {
        // a context that has failed may have collectives in flight that will never complete: abort, do not drain
        auto drop = [&](void* comm) {
            if (!comm) return;
            if (c->failed && rccl()->CommAbort) (void)rccl()->CommAbort(comm);
            else (void)rccl()->CommDestroy(comm);
        };
        drop(c->rccl_state2.comm);
        drop(c->rccl_state.comm);
    }
    if (c->s2) (void)hipStreamDestroy(c->s2);
    if (c->ev_part) (void)hipEventDestroy(c->ev_part);
    if (c->ev_vals) (void)hipEventDestroy(c->ev_vals);
    if (c->ev_tail) (void)hipEventDestroy(c->ev_tail);
    if (c->d_status) (void)hipFree(c->d_status);
    if (c->h_status) (void)hipHostFree(c->h_status);
    if (c->sorter) (void)gs_onesweep_destroy(c->sorter);
    if (c->part_keys) (void)hipFree(c->part_keys);
    if (c->part_vals) (void)hipFree(c->part_vals);
    if (c->d_hist) (void)hipFree(c->d_hist);
    if (c->d_table) (void)hipFree(c->d_table);
    if (c->d_plan) (void)hipFree(c->d_plan);
    if (c->h_plan) (void)hipHostFree(c->h_plan);
    if (c->h_table) (void)hipHostFree(c->h_table);
    if (c->ev_plan) (void)hipEventDestroy(c->ev_plan);
    for (auto& e : c->ev)
        if (e) (void)hipEventDestroy(e);
    delete c;
    return GS_OK;
}

gs_onesweep* gs_mgpu_sorter(gs_mgpu* c) { return c ? c->sorter : nullptr; }

gs_status gs_mgpu_set_force_exchange(gs_mgpu* c, int on) {
    if (!c) return GS_ERR_ARG;
    c->force_exchange = on ? 1 : 0;
    return GS_OK;
}

gs_status gs_onesweep_sort_sharded(gs_mgpu* c, const void* d_keys, const void* d_vals, uint32_t n, gs_key_type kt,
                                   void* d_out_keys, void* d_out_vals, uint32_t* out_n, void* stream) {
    // (an EMPTY shard may come with null input pointers — an empty tensor has none — and still takes part in every collective;
    //  returning early here would leave the peers waiting in the all-gather)
    if (!c || !d_out_keys || !out_n || misaligned(d_out_keys)) return GS_ERR_ARG;
    if (n != 0 && (!d_keys || misaligned(d_keys))) return GS_ERR_ARG;
    if ((int)kt < 0 || (int)kt > 2) return GS_ERR_ARG;
    if (n > c->shard_keys) return GS_ERR_SIZE;
    const uint32_t vb = c->value_bytes;
    if (vb) {
        if (!d_out_vals || misaligned(d_out_vals)) return GS_ERR_ARG;
        if (n != 0 && (!d_vals || misaligned(d_vals))) return GS_ERR_ARG;
    } else if (d_vals || d_out_vals) {
        return GS_ERR_MODE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    gs_onesweep* h = c->sorter;
    *out_n = 0;
    c->prof_pending = false;
    c->call_complete = false;  // until this call's closing gather is behind us: the status words on the device are an EARLIER call's
    c->last_sent = c->last_recv = 0;
    c->last_fine = 0;
    c->last_pregrouped = 0;
    GS_HIP(hipEventRecord(c->ev[0], s));
    uint32_t n_recv = n;
    hipEvent_t values_ready = nullptr;  // pairs with the values on the second stream: the local sort's passes wait for it
    gs_status local = GS_OK;            // this rank's own error after the plan: reported to the peers, returned at the end
    if (c->world == 1 && !c->force_exchange) {
        if (n) GS_HIP(hipMemcpyAsync(d_out_keys, d_keys, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
        if (n && vb) GS_HIP(hipMemcpyAsync(d_out_vals, d_vals, (size_t)n * vb, hipMemcpyDeviceToDevice, s));
        GS_HIP(hipEventRecord(c->ev[1], s));
        GS_HIP(hipEventRecord(c->ev[2], s));
    } else {
        // An empty shard still takes part in every collective; the kernels need n >= 1, so an empty shard
        // contributes a zero histogram directly.
        PassPlan pp{};
        bool fine = false;
        gs_status early = GS_OK;  // this rank's own error BEFORE the gather: its row is poisoned, everybody stops together
        gs_status st = mgpu_plan(c, d_keys, n, kt, s, false, &pp, &early);
        if (st != GS_OK) { c->failed = 1; return st; }
        if (!c->h_plan[gs::PLAN_PEER_FAILED] && c->h_plan[gs::PLAN_OVERFLOW]) {
            // every rank sees the same gathered table and takes the same decision: split at the 12-bit prefix
            fine = true;
            if (n && early == GS_OK) {  // the top-byte scan state is abandoned: its histogram region must be handed back zeroed
                if (zero_hist(h, s) != hipSuccess) early = GS_ERR_HIP;
                h->hist_dirty = false;
            }
            st = mgpu_plan(c, d_keys, n, kt, s, true, &pp, &early);
            if (st != GS_OK) { c->failed = 1; return st; }
            if (!c->h_plan[gs::PLAN_PEER_FAILED] && c->h_plan[gs::PLAN_OVERFLOW]) return GS_ERR_SIZE;  // on every rank alike: raise the capacity
        }
        if (c->h_plan[gs::PLAN_PEER_FAILED]) {  // on every rank alike: nobody enters the exchange
            c->debug_fail = 0;
            c->failed = 1;
            if (n && h->hist_dirty) { (void)zero_hist(h, s); h->hist_dirty = false; }
            return early != GS_OK ? early : GS_ERR_COMM;
        }
        c->last_fine = fine ? 1u : 0u;
        const uint32_t W = c->world;
        const uint32_t* send = c->h_plan + gs::PLAN_HEADER;
        const uint32_t* recv = send + W;
        n_recv = c->h_plan[gs::PLAN_NRECV];
        if (n_recv > c->capacity) return GS_ERR_SIZE;
        // ---- from here on the peers are committed to the exchange: an error of this rank is carried through it ----
        if (c->debug_fail == 2) local = GS_ERR_HIP;  // test hook: as if the partition pass had failed to launch
        c->debug_fail = 0;
        // Layout of the received bucket (this rank's own decision): bin-major in the local sort's ALTERNATE buffer if that sort will be
        // offered the two-level plan — its top-byte pass is then behind us — source-major in the caller's buffer otherwise.
        // Messages per (peer, top byte): whenever the exchange can carry them (not the 12-bit split, not ncclAllToAllv).
        const bool by_bin = c->by_bin && !fine && !(c->owns_comm && c->alltoallv);
        // (the bin-major path groups the shard INTO the output buffers: a caller whose input overlaps its output keeps the other layout)
        auto overlaps = [](const void* a, size_t na, const void* b, size_t nb) {
            const uintptr_t x = reinterpret_cast<uintptr_t>(a), y = reinterpret_cast<uintptr_t>(b);
            return a && b && x < y + nb && y < x + na;
        };
        const bool aliased = overlaps(d_keys, (size_t)n * 4, d_out_keys, (size_t)c->capacity * 4) ||
                             (vb && overlaps(d_vals, (size_t)n * vb, d_out_vals, (size_t)c->capacity * vb));
        const bool pre = by_bin && !aliased && n_recv != 0 && sort_route(h, n_recv, kt, vb).hy;
        c->last_pregrouped = pre ? 1u : 0u;
        uint32_t* const split_keys = pre ? static_cast<uint32_t*>(d_out_keys) : c->part_keys;  // where the shard is grouped = the send buffer
        void* const split_vals = pre ? d_out_vals : c->part_vals;
        uint32_t* const land_keys = pre ? c->part_keys : static_cast<uint32_t*>(d_out_keys);    // where the bucket lands
        void* const land_vals = pre ? c->part_vals : d_out_vals;
        // group the shard by destination (stable)
        if (n && local == GS_OK) {
            const BinLauncher fn = g_shapes[h->shape].fn[h->rank_mode][vb_index(vb)][kt];
            if (!fn) {
                local = GS_ERR_ARG;
            } else if (!fine) {
                fn(s, pp.grid, const_cast<uint32_t*>(static_cast<const uint32_t*>(d_keys)), split_keys, const_cast<void*>(d_vals),
                   split_vals, h->slab + SLAB_DESC, h->slab + SLAB_COUNTERS, h->slab + SLAB_INFO, h->slab + gs::SLAB_HSUB,
                   h->slab + SLAB_STATUS, n, 24, gs::BM_ZERO_HIST);
                if (hipGetLastError() != hipSuccess) local = GS_ERR_HIP;
                h->hist_dirty = false;
            } else {  // order by the top two bytes: every 12-bit prefix range is contiguous (the output buffers are the scratch)
                local = gs_onesweep_digit_pass(h, d_keys, d_out_keys, d_vals, d_out_vals, n, 2, kt, 0, s);
                if (local == GS_OK) local = gs_onesweep_digit_pass(h, d_out_keys, c->part_keys, d_out_vals, c->part_vals, n, 3, kt, 0, s);
            }
        }
        // (from the plan on NOTHING returns before the closing status gather: a rank that left here would strand its peers in
        //  ncclRecv or in that gather — every error is kept in `local`, the collectives are entered regardless, and the error is
        //  returned behind them)
        auto note = [&](gs_status e) { if (local == GS_OK) local = e; };
        if (hipEventRecord(c->ev[1], s) != hipSuccess) note(GS_ERR_HIP);
        // bucket exchange
        std::vector<uint32_t> sd(W), rd(W);
        uint32_t a = 0, b = 0;
        for (uint32_t p = 0; p < W; ++p) {
            sd[p] = a; rd[p] = b;
            a += send[p]; b += recv[p];
            if (p != c->rank) { c->last_sent += (uint64_t)send[p] * (4 + vb); c->last_recv += (uint64_t)recv[p] * (4 + vb); }
        }
        // Rounds: round j carries, for every peer p, top byte first[p] + j of p's range (ONE round with whole buckets if the exchange
        // goes peer by peer, or with one rank: a single source is bin-major as it is).  Both sides derive every count from the same
        // gathered table (gs_msd_exchange_round: a host function, checked on its own in tests/test_capi.py); messages between two
        // ranks are matched in the order they are issued, bin by bin.
        const uint32_t* first = recv + W;              // first_bin[world + 1]
        const bool rounds_by_bin = by_bin && W > 1;
        uint32_t R = 1;
        std::vector<uint32_t> sc(W), sdp(W), rc(W), rdp(W);
        if (rounds_by_bin && gs_msd_exchange_round(c->h_table, W, c->rank, first, pre ? 1 : 0, 0, sc.data(), sdp.data(), rc.data(), rdp.data(), &R) != GS_OK) note(GS_ERR_ARG);
        // all rounds of one array set in ONE ncclGroup (nested groups merge): one launch, every link busy from the start
        auto exchange_all = [&](const gs_mgpu_transport& t, uint32_t n_arrays, const void* const* src_, void* const* dst_, const uint32_t* eb_, hipStream_t st_) -> int {
            if (!rounds_by_bin) return t.exchange(t.user, n_arrays, src_, dst_, eb_, send, sd.data(), recv, rd.data(), st_);
            const bool grouped = c->owns_comm && rccl() != nullptr;
            int e = grouped ? rccl()->GroupStart() : 0;
            gs::MsdSegments own{};  // this rank's own bytes: taken out of the rounds and copied by one launch per array (msd_copy_segments_kernel)
            for (uint32_t j = 0; j < R && !e; ++j) {
                if (gs_msd_exchange_round(c->h_table, W, c->rank, first, pre ? 1 : 0, j, sc.data(), sdp.data(), rc.data(), rdp.data(), nullptr) != GS_OK) { e = -1; break; }
                if (sc[c->rank] && own.n < gs::RADIX) {
                    own.seg[3 * own.n] = sdp[c->rank]; own.seg[3 * own.n + 1] = rdp[c->rank]; own.seg[3 * own.n + 2] = sc[c->rank];
                    ++own.n;
                    sc[c->rank] = 0; rc[c->rank] = 0;
                }
                e = t.exchange(t.user, n_arrays, src_, dst_, eb_, sc.data(), sdp.data(), rc.data(), rdp.data(), st_);
            }
            for (uint32_t a_ = 0; a_ < n_arrays && own.n && !e; ++a_) {
                const uint32_t bps = 64;  // blocks per segment: 64 x 256 threads x 4 B per trip = 64 KiB of a segment of a few MiB per trip
                hipLaunchKernelGGL(gs::msd_copy_segments_kernel, dim3(own.n * bps), dim3(256), 0, st_, static_cast<const uint32_t*>(src_[a_]),
                                   static_cast<uint32_t*>(dst_[a_]), own, eb_[a_] / 4u, bps);
                if (hipGetLastError() != hipSuccess) e = -1;
            }
            const int e2 = grouped ? rccl()->GroupEnd() : 0;
            if (grouped && (e || e2)) g_last_rccl_error = e ? e : e2;
            return e ? e : e2;
        };
        const void* src[2] = {split_keys, split_vals};
        void* dst[2] = {land_keys, land_vals};
        const uint32_t eb[2] = {4u, vb};
        hipStream_t tail = s;  // the stream the closing status gather goes on
        if (vb && c->overlap) {
            // keys on the caller's stream; values on the second stream (and the second communicator) behind the partition pass
            if (hipEventRecord(c->ev_part, s) != hipSuccess || hipStreamWaitEvent(c->s2, c->ev_part, 0) != hipSuccess) note(GS_ERR_HIP);
            if (exchange_all(c->transport, 1u, src, dst, eb, s) != 0) note(GS_ERR_COMM);
            if (exchange_all(c->transport2, 1u, src + 1, dst + 1, eb + 1, c->s2) != 0) note(GS_ERR_COMM);
            if (hipEventRecord(c->ev_vals, c->s2) != hipSuccess) note(GS_ERR_HIP);
            values_ready = c->ev_vals;
            tail = c->s2;
        } else if (exchange_all(c->transport, vb ? 2u : 1u, src, dst, eb, s) != 0) {
            note(GS_ERR_COMM);
        }
        if (hipEventRecord(c->ev[2], s) != hipSuccess) note(GS_ERR_HIP);
        // closing status gather: every rank learns whether some peer carried an error through the exchange
        c->h_status[0] = (uint32_t)local;
        if (hipMemcpyAsync(c->d_status, c->h_status, sizeof(uint32_t), hipMemcpyHostToDevice, tail) != hipSuccess ||
            (vb && c->overlap ? c->transport2 : c->transport).all_gather_u32((vb && c->overlap ? c->transport2 : c->transport).user,
                                                                             c->d_status, c->d_status + 1, 1, tail) != 0) {
            c->failed = 1;  // (the gather itself failed: nothing more this rank can do for its peers)
            return local != GS_OK ? local : GS_ERR_COMM;
        }
        if (tail != s) {  // the caller's stream must not run ahead of the side stream's last use of the context
            if (hipEventRecord(c->ev_tail, tail) != hipSuccess) note(GS_ERR_HIP);
        }
        if (local != GS_OK) {  // the peers are served; this rank's own result is not there
            if (tail != s) (void)hipStreamWaitEvent(s, c->ev_tail, 0);
            c->failed = 1;
            return local;
        }
    }
    // local sort of the received bucket; the partition buffers are free again and serve as alt.  A bucket that was landed bin-major
    // in the alternate buffer (c->last_pregrouped) starts at the two-level plan's second pass.
    if (n_recv) {
        gs_status st = (vb || c->last_pregrouped)
                           ? sort_impl(h, d_out_keys, d_out_vals, c->part_keys, c->part_vals, n_recv, kt, GS_ORDER_ASCENDING, s, vb, values_ready, c->last_pregrouped != 0)
                           : gs_onesweep_sort_keys(h, d_out_keys, c->part_keys, n_recv, kt, GS_ORDER_ASCENDING, s);
        if (st != GS_OK) { c->failed = 1; return st; }
    } else if (values_ready) {
        GS_HIP(hipStreamWaitEvent(s, values_ready, 0));
    }
    if (values_ready) GS_HIP(hipStreamWaitEvent(s, c->ev_tail, 0));  // everything of the call is behind the caller's stream
    GS_HIP(hipEventRecord(c->ev[3], s));
    *out_n = n_recv;
    c->prof_pending = true;
    c->call_complete = true;
    return GS_OK;
}

gs_status gs_mgpu_check(gs_mgpu* c, void* stream) {
    if (!c) return GS_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!c->call_complete) {
        // The last call returned an error on this rank before or at its closing gather (plan error, a failed peer, the gather
        // itself) or behind it (the local sort): d_status[1..world] still hold an EARLIER call's words — reading them would
        // report GS_OK and clear the failure latch while collectives may still be in flight.  The latch stays.
        GS_HIP(hipStreamSynchronize(s));
        return GS_ERR_COMM;
    }
    GS_HIP(hipMemcpyAsync(c->h_status, c->d_status, (c->world + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    GS_HIP(hipStreamSynchronize(s));
    if (c->world > 1 || c->force_exchange)
        for (uint32_t r = 0; r < c->world; ++r)
            if (c->h_status[1 + r] != GS_OK) { c->failed = 1; return GS_ERR_COMM; }  // a peer carried an error through the exchange
    const gs_status st = gs_onesweep_check(c->sorter, stream);
    if (st == GS_OK) c->failed = 0;  // the last call was clean on every rank: the context has recovered (teardown destroys, not aborts)
    return st;
}

gs_status gs_mgpu_debug_fail(gs_mgpu* c, int where) {
    if (!c || where < 0 || where > 2) return GS_ERR_ARG;
    c->debug_fail = where;
    return GS_OK;
}

gs_status gs_mgpu_get_profile(gs_mgpu* c, float ms[4], uint64_t* bytes_sent, uint64_t* bytes_received, uint32_t* fine_split) {
    if (!c || !ms || !c->prof_pending) return GS_ERR_ARG;
    GS_HIP(hipEventSynchronize(c->ev[3]));
    for (int i = 0; i < 3; ++i) GS_HIP(hipEventElapsedTime(&ms[i], c->ev[i], c->ev[i + 1]));
    GS_HIP(hipEventElapsedTime(&ms[3], c->ev[0], c->ev[3]));
    if (bytes_sent) *bytes_sent = c->last_sent;
    if (bytes_received) *bytes_received = c->last_recv;
    if (fine_split) *fine_split = c->last_fine;
    return GS_OK;
}

gs_status gs_mgpu_last_layout(gs_mgpu* c, uint32_t* bin_major) {
    if (!c || !bin_major) return GS_ERR_ARG;
    *bin_major = c->last_pregrouped;  // (host state of the last call: no synchronisation)
    return GS_OK;
}

gs_status gs_mgpu_last_plan(gs_mgpu* c, uint32_t* plan, uint32_t words) {
    if (!c || !plan || words < gs::plan_words(c->world)) return GS_ERR_ARG;
    memcpy(plan, c->h_plan, gs::plan_words(c->world) * sizeof(uint32_t));
    plan[3] = c->last_fine;  // (no synchronisation: the plan is on the host since the call's one event wait)
    return GS_OK;
}

gs_status gs_msd_plan(const uint32_t* table, uint32_t nbins, uint32_t world, uint32_t rank, uint32_t capacity, uint32_t* plan) {
    if (!table || !plan || nbins == 0 || world == 0 || world > nbins || rank >= world) return GS_ERR_ARG;
    std::vector<uint64_t> g(nbins, 0);
    for (uint32_t r = 0; r < world; ++r)
        for (uint32_t b = 0; b < nbins; ++b) g[b] += table[(size_t)r * nbins + b];
    uint32_t* send = plan + gs::PLAN_HEADER;
    uint32_t* recv = send + world;
    uint32_t* first = recv + world;
    gs_status st = gs_msd_splitters_n(g.data(), nbins, world, first);
    if (st != GS_OK) return st;
    uint64_t nrecv = 0, mx = 0;
    for (uint32_t d = 0; d < world; ++d) {
        uint64_t bucket = 0;
        uint32_t mine = 0;
        for (uint32_t b = first[d]; b < first[d + 1]; ++b) {
            bucket += g[b];
            mine += table[(size_t)rank * nbins + b];
        }
        send[d] = mine;
        mx = bucket > mx ? bucket : mx;
    }
    for (uint32_t q = 0; q < world; ++q) {
        uint32_t c = 0;
        for (uint32_t b = first[rank]; b < first[rank + 1]; ++b) c += table[(size_t)q * nbins + b];
        recv[q] = c;
        nrecv += c;
    }
    plan[gs::PLAN_NRECV] = (uint32_t)(nrecv > 0xffffffffull ? 0xffffffffull : nrecv);
    plan[gs::PLAN_OVERFLOW] = mx > capacity ? 1u : 0u;
    plan[gs::PLAN_MAXBUCKET] = (uint32_t)(mx > 0xffffffffull ? 0xffffffffull : mx);
    plan[3] = 0;
    return GS_OK;
}

gs_status gs_msd_exchange_round(const uint32_t* table, uint32_t world, uint32_t rank, const uint32_t* first_bin, int bin_major, uint32_t round,
                                uint32_t* send_counts, uint32_t* send_displs, uint32_t* recv_counts, uint32_t* recv_displs, uint32_t* rounds) {
    if (!table || !first_bin || world == 0 || world > gs::MSD_MAX_WORLD || rank >= world || !send_counts || !send_displs || !recv_counts || !recv_displs)
        return GS_ERR_ARG;
    constexpr uint32_t NB = gs::RADIX;
    if (first_bin[0] != 0 || first_bin[world] != NB) return GS_ERR_ARG;
    uint32_t R = 0;
    for (uint32_t p = 0; p < world; ++p) {
        if (first_bin[p + 1] < first_bin[p]) return GS_ERR_ARG;
        R = first_bin[p + 1] - first_bin[p] > R ? first_bin[p + 1] - first_bin[p] : R;
    }
    if (rounds) *rounds = R;
    const uint32_t f0 = first_bin[rank], nb = first_bin[rank + 1] - f0;
    const uint32_t* mine = table + (size_t)rank * NB;
    // where this rank's bytes start in its own grouped shard (the split pass wrote them in byte order), up to the bytes this round sends
    for (uint32_t p = 0; p < world; ++p) {
        const uint32_t bp = first_bin[p] + round;      // the byte of p's range this round carries
        const bool sends = bp < first_bin[p + 1];
        uint32_t off = 0;
        if (sends) for (uint32_t x = 0; x < bp; ++x) off += mine[x];
        send_counts[p] = sends ? mine[bp] : 0u;
        send_displs[p] = off;
    }
    // where (source q, my byte f0 + round) lands: bin-major = byte by byte, sources in rank order inside a byte (the stable top-byte
    // partition of the concatenated sources); source-major = source by source, a source's bytes in order (round 5's layout)
    for (uint32_t q = 0; q < world; ++q) {
        uint32_t off = 0, cnt = 0;
        if (round < nb) {
            const uint32_t b = f0 + round;
            cnt = table[(size_t)q * NB + b];
            if (bin_major) {
                for (uint32_t j = 0; j < round; ++j) for (uint32_t s = 0; s < world; ++s) off += table[(size_t)s * NB + f0 + j];
                for (uint32_t s = 0; s < q; ++s) off += table[(size_t)s * NB + b];
            } else {
                for (uint32_t s = 0; s < q; ++s) for (uint32_t j = 0; j < nb; ++j) off += table[(size_t)s * NB + f0 + j];
                for (uint32_t j = 0; j < round; ++j) off += table[(size_t)q * NB + f0 + j];
            }
        }
        recv_counts[q] = cnt;
        recv_displs[q] = off;
    }
    return GS_OK;
}

gs_status gs_debug_msd_plan_device(const uint32_t* h_table, uint32_t nbins, uint32_t world, uint32_t rank, uint32_t capacity,
                                   uint32_t* h_plan, void* stream) {
    if (!h_table || !h_plan || (nbins != 256 && nbins != 4096) || world == 0 || world > gs::MSD_MAX_WORLD || rank >= world) return GS_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    uint32_t *d_t = nullptr, *d_p = nullptr;
    GS_HIP(hipMalloc(&d_t, (size_t)world * nbins * 4));
    gs_status ret = GS_OK;
    if (hipMalloc(&d_p, gs::plan_words(world) * 4) != hipSuccess) ret = GS_ERR_HIP;
    if (ret == GS_OK && hipMemcpyAsync(d_t, h_table, (size_t)world * nbins * 4, hipMemcpyHostToDevice, s) != hipSuccess) ret = GS_ERR_HIP;
    if (ret == GS_OK) {
        hipLaunchKernelGGL(gs::msd_plan_kernel, dim3(1), dim3(256), 0, s, d_t, nbins, world, rank, capacity, d_p);
        if (hipMemcpyAsync(h_plan, d_p, gs::plan_words(world) * 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess)
            ret = GS_ERR_HIP;
    }
    (void)hipFree(d_t);
    if (d_p) (void)hipFree(d_p);
    return ret;
}

}  // extern "C"
