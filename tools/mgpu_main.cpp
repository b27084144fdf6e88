// mgpu_main.cpp — torch-free harness of the multi-GPU sharded sort (BASELINE.json configs[3]):
// one rank per GPU over the C-ABI of include/gpusort.h, nothing but HIP and libgpusort.so (which loads RCCL itself).
//
//   build/mgpu_main [--gpus N] [--log2 L] [--iters K] [--warmup W] [--pairs 0|4|8] [--mode fork|threads] [--preset P]
//                   [--slack 1.25] [--one-device]
//
//   --mode fork     one PROCESS per GPU (default; the deployment model): rank r = child r, hipSetDevice(r); rank 0 draws the
//                   communicator id (gs_mgpu_get_unique_id) and hands its 128 bytes to the others through pipes
//   --mode threads  one process, one THREAD per GPU (the ncclCommInitAll model: ranks share an address space; the id is
//                   passed in memory; every thread calls gs_mgpu_create = ncclCommInitRank concurrently)
//   --one-device    every rank on device 0 with world = 1 each is NOT a multi-rank run; with --gpus 1 the exchange path is
//                   forced (gs_mgpu_set_force_exchange) so that the whole pipeline runs on a one-GPU box
//   --ranks N --share-gpu   a REHEARSAL of world = N on ONE GPU: N rank threads time-share device 0 and exchange through an
//                   in-process transport (gs_mgpu_create_with_transport: every collective is a barrier + device-to-device copies; RCCL
//                   refuses several ranks on one device).  Everything else is the product pipeline at the real world size — N-way
//                   plan, count and displacement tables, N - 1 peers per rank, the closing status gather.  The throughput it prints is
//                   that of N ranks sharing one GPU's HBM, not a scaling number.
// Every rank generates its shard with the library's InitRandom (seed 10 + i + 1000 * rank: bench.py's convention), sorts K
// times (weak scaling: 2^L keys per GPU), checks its bucket (sorted; sizes add up; bucket borders ascend across ranks) and
// reports its phase times; rank 0 prints ONE JSON line: GKeys/s of the whole job (slowest rank), per-phase ms (max over
// ranks), bytes exchanged, GB/s per rank and per xGMI link against 153 GB/s, and the local sort's dominant-kernel roofline.
#include <hip/hip_runtime.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "gpusort.h"

namespace {

struct Options {
    int gpus = 1, log2 = 28, iters = 5, warmup = 1, pairs = 0, preset = 0;
    double slack = 1.25;
    bool threads = false, one_device = false, share_gpu = false;
};

// ---- in-process transport of --share-gpu: all ranks are threads of this process on device 0 ----
struct Barrier {
    std::mutex m;
    std::condition_variable cv;
    int n = 0, waiting = 0;
    unsigned long long gen = 0;
    void wait() {
        std::unique_lock<std::mutex> l(m);
        const unsigned long long g = gen;
        if (++waiting == n) { waiting = 0; ++gen; cv.notify_all(); }
        else cv.wait(l, [&] { return gen != g; });
    }
};
struct Bus {
    struct Pub { const void* send[2]; const uint32_t* send_displs; };
    Barrier bar;
    std::vector<Pub> pub;
};
struct BusRank { Bus* bus; uint32_t rank, world; };
int bus_all_gather(void* user, const void* d_send, void* d_recv, size_t count, void* stream) {
    BusRank* r = static_cast<BusRank*>(user);
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc = hipStreamSynchronize(s) == hipSuccess ? 0 : 1;  // what this rank sends is complete
    r->bus->pub[r->rank].send[0] = d_send;
    r->bus->bar.wait();
    for (uint32_t p = 0; p < r->world; ++p)
        if (hipMemcpyAsync(static_cast<char*>(d_recv) + (size_t)p * count * 4, r->bus->pub[p].send[0], count * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) rc = 1;
    if (hipStreamSynchronize(s) != hipSuccess) rc = 1;
    r->bus->bar.wait();  // nobody reuses its send buffer before everybody has read it
    return rc;
}
int bus_exchange(void* user, uint32_t n_arrays, const void* const* d_send, void* const* d_recv, const uint32_t* elem_bytes,
                 const uint32_t* send_counts, const uint32_t* send_displs, const uint32_t* recv_counts, const uint32_t* recv_displs, void* stream) {
    (void)send_counts;
    BusRank* r = static_cast<BusRank*>(user);
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc = hipStreamSynchronize(s) == hipSuccess ? 0 : 1;
    Bus::Pub& mine = r->bus->pub[r->rank];
    for (uint32_t a = 0; a < n_arrays && a < 2; ++a) mine.send[a] = d_send[a];
    mine.send_displs = send_displs;
    r->bus->bar.wait();
    for (uint32_t a = 0; a < n_arrays && a < 2; ++a)
        for (uint32_t p = 0; p < r->world; ++p) {  // peer p's elements for this rank start at ITS send_displs[this rank]
            const Bus::Pub& pp = r->bus->pub[p];
            if (recv_counts[p] && hipMemcpyAsync(static_cast<char*>(d_recv[a]) + (size_t)recv_displs[p] * elem_bytes[a],
                                                 static_cast<const char*>(pp.send[a]) + (size_t)pp.send_displs[r->rank] * elem_bytes[a],
                                                 (size_t)recv_counts[p] * elem_bytes[a], hipMemcpyDeviceToDevice, s) != hipSuccess) rc = 1;
        }
    if (hipStreamSynchronize(s) != hipSuccess) rc = 1;
    r->bus->bar.wait();
    return rc;
}
Bus g_bus;

struct RankResult {  // what a rank reports to rank 0 (plain data: crosses a pipe in fork mode)
    double ms_total = 0;            // mean wall time per sort (host clock around K sorts + sync)
    float phase[4] = {0, 0, 0, 0};  // split / exchange / local sort / total of the LAST sort (HIP events)
    float kern[GS_PROFILE_SLOTS] = {0};  // per-kernel times of one profiled local sort
    uint64_t sent = 0, recv = 0;
    uint32_t out_n = 0, first_key = 0, last_key = 0, fine = 0, sorted = 0, status = 0, two_level = 0, bin_major = 0;
};

#define CHECK_HIP(x)                                                                     \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) {                                                          \
            fprintf(stderr, "HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__);     \
            return 100;                                                                  \
        }                                                                                \
    } while (0)
#define CHECK_GS(x)                                                                                         \
    do {                                                                                                    \
        gs_status s_ = (x);                                                                                 \
        if (s_ != GS_OK) {                                                                                  \
            fprintf(stderr, "%s -> %s (hip %d, rccl %d) at %s:%d\n", #x, gs_status_string(s_), gs_last_hip_error(), \
                    gs_last_rccl_error(), __FILE__, __LINE__);                                              \
            return 200 + (int)s_;                                                                           \
        }                                                                                                   \
    } while (0)

int run_rank(const Options& o, int rank, const uint8_t* id, RankResult* out) {
    CHECK_HIP(hipSetDevice((o.one_device || o.share_gpu) ? 0 : rank));
    const uint32_t n = 1u << o.log2;
    const uint32_t cap = (uint32_t)std::min<double>((double)n * o.slack + 256.0, (double)GS_MAX_KEYS);
    const uint32_t vb = (uint32_t)o.pairs;
    gs_mgpu* ctx = nullptr;
    BusRank bus_rank{&g_bus, (uint32_t)rank, (uint32_t)o.gpus};
    if (o.share_gpu) {
        const gs_mgpu_transport t{&bus_rank, bus_all_gather, bus_exchange};
        CHECK_GS(gs_mgpu_create_with_transport(&ctx, &t, (uint32_t)rank, (uint32_t)o.gpus, n, cap, vb ? GS_MODE_PAIRS : GS_MODE_KEYS_ONLY, vb));
    } else
    CHECK_GS(gs_mgpu_create(&ctx, id, (uint32_t)rank, (uint32_t)o.gpus, n, cap, vb ? GS_MODE_PAIRS : GS_MODE_KEYS_ONLY, vb));
    if (o.gpus == 1) CHECK_GS(gs_mgpu_set_force_exchange(ctx, 1));  // a one-GPU box still runs split + exchange + sort
    hipStream_t s;
    CHECK_HIP(hipStreamCreate(&s));
    void *keys = nullptr, *vals = nullptr, *out_k = nullptr, *out_v = nullptr;
    CHECK_HIP(hipMalloc(&keys, (size_t)n * 4));
    CHECK_HIP(hipMalloc(&out_k, (size_t)cap * 4));
    if (vb) {
        CHECK_HIP(hipMalloc(&vals, (size_t)n * vb));
        CHECK_HIP(hipMalloc(&out_v, (size_t)cap * vb));
    }
    uint32_t out_n = 0;
    double wall = 0;
    for (int i = -o.warmup; i < o.iters; ++i) {
        CHECK_GS(gs_init_random(keys, vals, vb, (uint32_t)o.preset, (uint32_t)(10 + (i < 0 ? 5000 - i : i) + 1000 * rank), n, s));
        CHECK_HIP(hipStreamSynchronize(s));
        const auto t0 = std::chrono::steady_clock::now();
        CHECK_GS(gs_onesweep_sort_sharded(ctx, keys, vals, n, GS_KEY_UINT32, out_k, out_v, &out_n, s));
        CHECK_HIP(hipStreamSynchronize(s));
        if (i >= 0) wall += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    out->ms_total = wall / o.iters;
    out->status = (unsigned)gs_mgpu_check(ctx, s);
    CHECK_GS(gs_mgpu_get_profile(ctx, out->phase, &out->sent, &out->recv, &out->fine));
    CHECK_GS(gs_mgpu_last_layout(ctx, &out->bin_major));
    out->out_n = out_n;
    uint32_t err = 1;
    if (out_n) {
        CHECK_GS(gs_validate(out_k, vb == 4 ? out_v : nullptr, vb == 4 ? 4 : 0, out_n, GS_KEY_UINT32, GS_ORDER_ASCENDING, &err, s));
        CHECK_HIP(hipMemcpy(&out->first_key, out_k, 4, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(&out->last_key, static_cast<char*>(out_k) + (size_t)(out_n - 1) * 4, 4, hipMemcpyDeviceToHost));
    } else {
        err = 0;
    }
    out->sorted = err == 0;
    // one profiled LOCAL sort (the per-GPU OneSweep of n keys): per-kernel HIP events for the roofline of the dominant kernel
    gs_onesweep* local = gs_mgpu_sorter(ctx);
    void* alt = nullptr;
    void* valt = nullptr;
    CHECK_HIP(hipMalloc(&alt, (size_t)n * 4));
    if (vb) CHECK_HIP(hipMalloc(&valt, (size_t)n * vb));
    CHECK_GS(gs_onesweep_set_profiling(local, 1));
    CHECK_GS(gs_init_random(keys, vals, vb, (uint32_t)o.preset, 777u + (uint32_t)rank, n, s));
    if (vb) CHECK_GS(gs_onesweep_sort_pairs(local, keys, vals, alt, valt, n, GS_KEY_UINT32, GS_ORDER_ASCENDING, s));
    else CHECK_GS(gs_onesweep_sort_keys(local, keys, alt, n, GS_KEY_UINT32, GS_ORDER_ASCENDING, s));
    CHECK_GS(gs_onesweep_get_profile(local, out->kern));
    {   // the device's plan for that sort: on the two-level plan only slots 3 and 4 are DigitBinningPasses (5 = the bucket-local sort)
        uint32_t two_level = 0;
        CHECK_GS(gs_onesweep_last_plan(local, &two_level, nullptr, s));
        out->two_level = two_level;
    }
    (void)hipFree(alt);
    (void)hipFree(valt);
    (void)hipFree(keys);
    (void)hipFree(vals);
    (void)hipFree(out_k);
    (void)hipFree(out_v);
    (void)hipStreamDestroy(s);
    CHECK_GS(gs_mgpu_destroy(ctx));
    return 0;
}

void report(const Options& o, const std::vector<RankResult>& r, const std::vector<int>& rc) {
    const int W = o.gpus;
    const double n = (double)(1ull << o.log2);
    double ms = 0, ph[4] = {0, 0, 0, 0};
    uint64_t sent_max = 0, sent_sum = 0, total_out = 0;
    bool ok = true;
    for (int i = 0; i < W; ++i) {
        ok = ok && rc[i] == 0 && r[i].sorted && r[i].status == GS_OK;
        ms = std::max(ms, r[i].ms_total);
        for (int k = 0; k < 4; ++k) ph[k] = std::max<double>(ph[k], r[i].phase[k]);
        sent_max = std::max(sent_max, r[i].sent);
        sent_sum += r[i].sent;
        total_out += r[i].out_n;
    }
    ok = ok && total_out == (uint64_t)n * W;
    for (int i = 0, prev = -1; i < W; ++i) {  // bucket borders ascend across ranks
        if (!r[i].out_n) continue;
        if (prev >= 0 && r[prev].last_key > r[i].first_key) ok = false;
        prev = i;
    }
    const double vb = o.pairs, pass_ms = r[0].two_level ? (r[0].kern[3] + r[0].kern[4]) / 2.0 : (r[0].kern[3] + r[0].kern[4] + r[0].kern[5] + r[0].kern[6]) / 4.0;
    const double pass_gbs = (8.0 + 2.0 * vb) * n / (pass_ms * 1e-3) / 1e9;
    const double ex_gbs = ph[1] > 0 ? sent_max / (ph[1] * 1e-3) / 1e9 : 0.0;
    const int links = std::max(W - 1, 1);
    printf("{\"tool\": \"mgpu_main\", \"metric\": \"GKeys/s uint32 OneSweep, MSD split + RCCL bucket exchange + per-GPU OneSweep\", "
           "\"value\": %.4f, \"unit\": \"GKeys/s\", \"n_gpus\": %d, \"mode\": \"%s\", \"keys_per_gpu\": %.0f, \"value_bytes\": %d, "
           "\"iters\": %d, \"ms_per_sort\": %.4f, \"scaling\": \"weak\", \"verified\": %s, "
           "\"phase_ms_max_over_ranks\": {\"split\": %.4f, \"exchange\": %.4f, \"local_sort\": %.4f, \"total\": %.4f}, "
           "\"bytes_sent_off_rank\": {\"max\": %llu, \"sum\": %llu}, \"exchange_GBps_per_rank\": %.2f, \"exchange_GBps_per_link\": %.2f, "
           "\"xgmi_link_peak_GBps\": 153.0, \"links_used_per_rank\": %d, \"frac_of_link_peak\": %.4f, \"split\": \"%s\", \"bucket_layout\": \"%s\", "
           "\"local_sort_rank0\": {\"plan\": \"%s\", \"per_kernel_ms\": {\"global_histogram\": %.4f, \"scan\": %.4f, \"pass0\": %.4f, \"pass1\": %.4f, "
           "\"pass2\": %.4f, \"pass3\": %.4f, \"total\": %.4f}, \"roofline\": {\"bound\": \"hbm\", \"kernel\": \"one 8-bit DigitBinningPass\", "
           "\"achieved\": %.1f, \"peak\": 8000.0, \"unit\": \"GB/s\", \"frac\": %.4f}}, \"rank_exit_codes\": [",
           n * W / (ms * 1e-3) / 1e9, W, o.share_gpu ? "threads sharing ONE GPU over an in-process transport (rehearsal, not a scaling number)" : o.threads ? "threads" : "fork", n, o.pairs, o.iters, ms, ok ? "true" : "false", ph[0], ph[1], ph[2],
           ph[3], (unsigned long long)sent_max, (unsigned long long)sent_sum, ex_gbs, ex_gbs / links, W - 1, ex_gbs / links / 153.0, r[0].fine ? "12-bit prefix" : "top byte",
           r[0].bin_major ? "bin-major in the alternate buffer: the local sort starts at the two-level plan's second pass (phase local_sort)" : "source-major: full local sort",
           r[0].two_level ? "two-level (pass0 = top byte, pass1 = byte 2, pass2 = bucket-local sort)" : "four LSD passes",
           r[0].kern[1], r[0].kern[2], r[0].kern[3], r[0].kern[4], r[0].kern[5], r[0].kern[6], r[0].kern[7], pass_gbs, pass_gbs / 8000.0);
    for (int i = 0; i < W; ++i) printf("%s%d", i ? ", " : "", rc[i]);
    printf("]}\n");
}

}  // namespace

int main(int argc, char** argv) {
    Options o;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&](int def) { return i + 1 < argc ? atoi(argv[++i]) : def; };
        if (a == "--gpus") o.gpus = val(1);
        else if (a == "--log2") o.log2 = val(28);
        else if (a == "--iters") o.iters = std::max(1, val(5));
        else if (a == "--warmup") o.warmup = std::max(0, val(1));
        else if (a == "--pairs") o.pairs = val(0);
        else if (a == "--preset") o.preset = val(0);
        else if (a == "--slack") o.slack = i + 1 < argc ? atof(argv[++i]) : 1.25;
        else if (a == "--mode") o.threads = i + 1 < argc && std::string(argv[++i]) == "threads";
        else if (a == "--one-device") o.one_device = true;
        else if (a == "--ranks") o.gpus = val(1);
        else if (a == "--share-gpu") { o.share_gpu = true; o.threads = true; }
        else { fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    if (o.gpus < 1 || o.gpus > 64 || o.log2 < 10 || o.log2 > 29 || (o.pairs != 0 && o.pairs != 4 && o.pairs != 8)) {
        fprintf(stderr, "bad arguments\n");
        return 2;
    }
    std::vector<RankResult> res(o.gpus);
    std::vector<int> rc(o.gpus, 0);
    if (o.share_gpu) {
        g_bus.bar.n = o.gpus;
        g_bus.pub.resize(o.gpus);
    }
    if (o.threads) {
        // one process: rank 0's id in memory, every thread = one rank (gs_mgpu_create calls ncclCommInitRank concurrently)
        uint8_t id[GS_MGPU_UNIQUE_ID_BYTES] = {0};
        if (hipSetDevice(0) != hipSuccess || (!o.share_gpu && gs_mgpu_get_unique_id(id) != GS_OK)) {
            fprintf(stderr, "gs_mgpu_get_unique_id failed (rccl %d)\n", gs_last_rccl_error());
            return 3;
        }
        std::vector<std::thread> th;
        for (int r = 0; r < o.gpus; ++r) th.emplace_back([&, r] { rc[r] = run_rank(o, r, id, &res[r]); });
        for (auto& t : th) t.join();
        report(o, res, rc);
        return std::all_of(rc.begin(), rc.end(), [](int c) { return c == 0; }) ? 0 : 1;
    }
    // one process per GPU.  Children are forked BEFORE any HIP call of this process (a forked HIP context is unusable);
    // child 0 draws the id and sends it up, the parent relays it down to the other children; results come back through pipes.
    std::vector<int> up(o.gpus), down(o.gpus);
    std::vector<pid_t> pid(o.gpus);
    for (int r = 0; r < o.gpus; ++r) {
        int pu[2], pd[2];
        if (pipe(pu) != 0 || pipe(pd) != 0) { perror("pipe"); return 3; }
        pid[r] = fork();
        if (pid[r] < 0) { perror("fork"); return 3; }
        if (pid[r] == 0) {
            close(pu[0]);
            close(pd[1]);
            uint8_t id[GS_MGPU_UNIQUE_ID_BYTES];
            memset(id, 0, sizeof id);
            if (r == 0) {
                if (hipSetDevice(0) != hipSuccess || gs_mgpu_get_unique_id(id) != GS_OK) {
                    fprintf(stderr, "rank 0: gs_mgpu_get_unique_id failed (rccl %d)\n", gs_last_rccl_error());
                    _exit(3);
                }
                if (write(pu[1], id, sizeof id) != (ssize_t)sizeof id) _exit(4);
            } else if (read(pd[0], id, sizeof id) != (ssize_t)sizeof id) {
                _exit(4);
            }
            RankResult rr;
            const int code = run_rank(o, r, id, &rr);
            if (write(pu[1], &rr, sizeof rr) != (ssize_t)sizeof rr) _exit(5);
            _exit(code);
        }
        close(pu[1]);
        close(pd[0]);
        up[r] = pu[0];
        down[r] = pd[1];
    }
    uint8_t id[GS_MGPU_UNIQUE_ID_BYTES];
    bool relay_ok = read(up[0], id, sizeof id) == (ssize_t)sizeof id;
    for (int r = 1; r < o.gpus && relay_ok; ++r) relay_ok = write(down[r], id, sizeof id) == (ssize_t)sizeof id;
    for (int r = 0; r < o.gpus; ++r) {
        if (read(up[r], &res[r], sizeof(RankResult)) != (ssize_t)sizeof(RankResult)) rc[r] = 9;
        int st = 0;
        waitpid(pid[r], &st, 0);
        if (rc[r] == 0) rc[r] = WIFEXITED(st) ? WEXITSTATUS(st) : 10;
    }
    report(o, res, rc);
    return std::all_of(rc.begin(), rc.end(), [](int c) { return c == 0; }) ? 0 : 1;
}
