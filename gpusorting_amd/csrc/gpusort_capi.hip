// gpusort_capi.hip — host side of libgpusort.so: the C-ABI of include/gpusort.h
// over the gfx950 kernels of onesweep_kernels.hpp.
//
// Replaces the dispatch half of the reference's OneSweepDispatcher
// (GPUSortingCUDA/Sort/OneSweepDispatcher.cuh:301-391): state clear, the
// 1 + 1 + 4 launch sequence, validation read-back.  Differences by design:
//   - no clear launch at all (the reference: 6 cudaMemset, :301-309): the scan state is ONE contiguous slab that
//     the GlobalHistogram kernel zeroes while it reads the keys; no host sync inside the sort (:318);
//   - the pass plan (identity passes dropped, input buffer of each pass, skew handling, position chains) is made by
//     the Scan kernel on the device;
//   - descriptor rows = tiles + 1, so the last tile's publish to row tile+1
//     stays in bounds (the reference overruns by 256 words when size==maxSize);
//   - everything is enqueued on the caller's stream.
#include "../../include/gpusort.h"
#include "onesweep_kernels.hpp"
#include "mid_kernels.hpp"
#include "hybrid_kernels.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>

namespace {

thread_local int g_last_hip_error = 0;

#define GS_HIP(call)                                   \
    do {                                               \
        hipError_t e_ = (call);                        \
        if (e_ != hipSuccess) {                        \
            g_last_hip_error = (int)e_;                \
            return GS_ERR_HIP;                         \
        }                                              \
    } while (0)

// ---- tile-shape table -------------------------------------------------------
// One launcher per (shape, value bytes, key type).  Shape 0 is the default; the
// others exist for on-device tuning sweeps (u32 keys only).
using BinLauncher = void (*)(hipStream_t, uint32_t grid, uint32_t*, uint32_t*, void*, void*,
                             uint32_t* desc, uint32_t* counters, const uint32_t* info, uint32_t* hsub, uint32_t* status,
                             uint32_t n, uint32_t shift, uint32_t mode);

template <int THREADS, int KPT, int VB, int KT, int RANK, int VR = 1>
void launch_bin(hipStream_t s, uint32_t grid, uint32_t* ka, uint32_t* kb, void* va, void* vb, uint32_t* desc,
                uint32_t* counters, const uint32_t* info, uint32_t* hsub, uint32_t* status, uint32_t n, uint32_t shift,
                uint32_t mode) {
    hipLaunchKernelGGL((gs::digit_binning_kernel<THREADS, KPT, VB, KT, RANK, VR>), dim3(grid), dim3(THREADS), 0, s, ka, kb,
                       va, vb, desc, counters, info, hsub, status, n, shift, mode);
}
// the two-round form of the 8-byte-value pass (two workgroups per CU), launched beside the one-round form in full
// sorts; the pass's PF_SKEW flag decides on the device which of the two works.  [rank mode][key type]
#if defined(GS_MINIMAL) && defined(GS_MIN_PAIRS)
const BinLauncher g_vr2[2][3] = {{nullptr, nullptr, nullptr}, {launch_bin<512, 32, 8, 0, 1, 2>, nullptr, nullptr}};
#elif defined(GS_MINIMAL)  // experiment builds (tools/): u32 keys-only kernels of the three product shapes, nothing else — a 10 s compile
const BinLauncher g_vr2[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
#else
const BinLauncher g_vr2[2][3] = {{launch_bin<512, 32, 8, 0, 0, 2>, launch_bin<512, 32, 8, 1, 0, 2>, launch_bin<512, 32, 8, 2, 0, 2>},
                                 {launch_bin<512, 32, 8, 0, 1, 2>, launch_bin<512, 32, 8, 1, 1, 2>, launch_bin<512, 32, 8, 2, 1, 2>}};
#endif

// keys-only sorts of 32-bit keys on the default tile that the Scan kernel may plan on position chains (PF_POS, skewed keys): one
// launch per pass of the dual kernel — persistent workgroups that run the plain or the position-chain form, as planned.
// [last pass][key type]
// tile of the counting position-chain passes, as the Scan kernel takes it (bit 31: the plan's last pass runs on it as well).
// Keys-only: the full tile, counters packed 2 x 16 bit; pairs: 512 x 24 with 32-bit counters — and for 8-byte values in the last
// pass too (its two staging rounds run 9 % faster on the smaller tile, profiles/r04_pos_packed_counters.txt)
constexpr uint32_t POS_TILE = 512 * GS_POS_KPT;
inline uint32_t pos_tile_for(uint32_t vb) {
    return vb == 0 ? POS_TILE : (512u * GS_POSV_KPT) | ((vb == 8 && GS_POSV8_LAST_SMALL) ? 0x80000000u : 0u);
}
template <int KT, bool LAST>
void launch_dual(hipStream_t s, uint32_t grid, uint32_t* ka, uint32_t* kb, void* va, void* vb, uint32_t* desc, uint32_t* counters,
                 const uint32_t* info, uint32_t* hsub, uint32_t* status, uint32_t n, uint32_t shift, uint32_t mode) {
    hipLaunchKernelGGL((gs::digit_binning_dual_kernel<KT, LAST>), dim3(grid), dim3(512), 0, s, ka, kb, va, vb, desc, counters, info,
                       hsub, status, n, shift, mode);
}
template <int VB, int KT, bool LAST>
void launch_posv(hipStream_t s, uint32_t grid, uint32_t* ka, uint32_t* kb, void* va, void* vb, uint32_t* desc, uint32_t* counters,
                 const uint32_t* info, uint32_t* hsub, uint32_t* status, uint32_t n, uint32_t shift, uint32_t mode) {
    hipLaunchKernelGGL((gs::digit_binning_posv_kernel<VB, KT, LAST>), dim3(grid), dim3(512), 0, s, ka, kb, va, vb, desc, counters, info,
                       hsub, status, n, shift, mode);
}
// pairs on the two-level plan: the plain form of the pass as persistent workgroups [8-byte values][key type]; rank mode 1 only
template <int T, int K, int VB, int KT>
void launch_persist(hipStream_t s, uint32_t grid, uint32_t* ka, uint32_t* kb, void* va, void* vb, uint32_t* desc, uint32_t* counters,
                    const uint32_t* info, uint32_t* hsub, uint32_t* status, uint32_t n, uint32_t shift, uint32_t mode) {
    hipLaunchKernelGGL((gs::digit_binning_persist_kernel<T, K, VB, KT, 1>), dim3(grid), dim3(T), 0, s, ka, kb, va, vb, desc, counters, info, hsub,
                       status, n, shift, mode);
}
#ifdef GS_MINIMAL
const BinLauncher g_dual[2][3] = {{launch_dual<0, false>, nullptr, nullptr}, {launch_dual<0, true>, nullptr, nullptr}};
#ifdef GS_MIN_PAIRS
const BinLauncher g_posv[2][2][3] = {{{launch_posv<4, 0, false>, nullptr, nullptr}, {launch_posv<4, 0, true>, nullptr, nullptr}},
                                     {{launch_posv<8, 0, false>, nullptr, nullptr}, {launch_posv<8, 0, true>, nullptr, nullptr}}};
const BinLauncher g_persist[2][3] = {{launch_persist<1024, 16, 4, 0>, nullptr, nullptr}, {launch_persist<512, 32, 8, 0>, nullptr, nullptr}};
#else
const BinLauncher g_posv[2][2][3] = {{{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}}, {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}}};
const BinLauncher g_persist[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
#endif
#else
const BinLauncher g_persist[2][3] = {{launch_persist<1024, 16, 4, 0>, launch_persist<1024, 16, 4, 1>, launch_persist<1024, 16, 4, 2>},
                                     {launch_persist<512, 32, 8, 0>, launch_persist<512, 32, 8, 1>, launch_persist<512, 32, 8, 2>}};
// pairs: the position-chain form of the pass, launched beside the plain form(s) [4- / 8-byte values][last pass][key type]
const BinLauncher g_posv[2][2][3] = {{{launch_posv<4, 0, false>, launch_posv<4, 1, false>, launch_posv<4, 2, false>},
                                      {launch_posv<4, 0, true>, launch_posv<4, 1, true>, launch_posv<4, 2, true>}},
                                     {{launch_posv<8, 0, false>, launch_posv<8, 1, false>, launch_posv<8, 2, false>},
                                      {launch_posv<8, 0, true>, launch_posv<8, 1, true>, launch_posv<8, 2, true>}}};
const BinLauncher g_dual[2][3] = {{launch_dual<0, false>, launch_dual<1, false>, launch_dual<2, false>},
                                  {launch_dual<0, true>, launch_dual<1, true>, launch_dual<2, true>}};
#endif

struct Shape {
    int threads, kpt;
    BinLauncher fn[2][3][6];  // [rank mode][vb index 0/4/8][key type: 3 x 32-bit, 3 x 64-bit]; nullptr = not compiled
};

#define GS_ROWS(T, K, R)                                                                             \
    {                                                                                                \
        {launch_bin<T, K, 0, 0, R>, launch_bin<T, K, 0, 1, R>, launch_bin<T, K, 0, 2, R>},           \
            {launch_bin<T, K, 4, 0, R>, launch_bin<T, K, 4, 1, R>, launch_bin<T, K, 4, 2, R>},       \
            {launch_bin<T, K, 8, 0, R>, launch_bin<T, K, 8, 1, R>, launch_bin<T, K, 8, 2, R>},       \
    }
#define GS_ROWS_U32(T, K, R)                                                                         \
    {                                                                                                \
        {launch_bin<T, K, 0, 0, R>, nullptr, nullptr}, {launch_bin<T, K, 4, 0, R>, nullptr, nullptr}, \
            {launch_bin<T, K, 8, 0, R>, nullptr, nullptr},                                           \
    }
// every key type, 64-bit keys included (8-byte stage slots: tiles up to 8192 keys)
#define GS_ROWS64(T, K, R)                                                                                                   \
    {                                                                                                                        \
        {launch_bin<T, K, 0, 0, R>, launch_bin<T, K, 0, 1, R>, launch_bin<T, K, 0, 2, R>, launch_bin<T, K, 0, 3, R>,         \
         launch_bin<T, K, 0, 4, R>, launch_bin<T, K, 0, 5, R>},                                                              \
            {launch_bin<T, K, 4, 0, R>, launch_bin<T, K, 4, 1, R>, launch_bin<T, K, 4, 2, R>, launch_bin<T, K, 4, 3, R>,     \
             launch_bin<T, K, 4, 4, R>, launch_bin<T, K, 4, 5, R>},                                                          \
            {launch_bin<T, K, 8, 0, R>, launch_bin<T, K, 8, 1, R>, launch_bin<T, K, 8, 2, R>, launch_bin<T, K, 8, 3, R>,     \
             launch_bin<T, K, 8, 4, R>, launch_bin<T, K, 8, 5, R>},                                                          \
    }
#define GS_FULL64(T, K) {T, K, {GS_ROWS64(T, K, 0), GS_ROWS64(T, K, 1)}}
#define GS_ROWS_KEYS(T, K, R) {{launch_bin<T, K, 0, 0, R>, nullptr, nullptr}, {nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}}
#define GS_KEYSONLY(T, K) {T, K, {GS_ROWS_KEYS(T, K, 0), GS_ROWS_KEYS(T, K, 1)}}
#define GS_FULL(T, K) {T, K, {GS_ROWS(T, K, 0), GS_ROWS(T, K, 1)}}
#define GS_U32ONLY(T, K) {T, K, {GS_ROWS_U32(T, K, 0), GS_ROWS_U32(T, K, 1)}}

#ifdef GS_MINIMAL
#ifdef GS_MIN_PAIRS  // (tuning flavour with the uint32-key pairs kernels as well: the two-level plan for pairs, tools/)
#define GS_ROWS_KEYS64(T, K, R) {{launch_bin<T, K, 0, 0, R>, nullptr, nullptr, nullptr, nullptr, nullptr}, {launch_bin<T, K, 4, 0, R>, nullptr, nullptr, nullptr, nullptr, nullptr}, {launch_bin<T, K, 8, 0, R>, nullptr, nullptr, nullptr, nullptr, nullptr}}
#else
#define GS_ROWS_KEYS64(T, K, R) {{launch_bin<T, K, 0, 0, R>, nullptr, nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}}
#endif
#define GS_KEYSONLY64(T, K) {T, K, {GS_ROWS_KEYS64(T, K, 0), GS_ROWS_KEYS64(T, K, 1)}}
const Shape g_shapes[] = {GS_KEYSONLY64(512, 32), GS_KEYSONLY64(1024, 16), GS_KEYSONLY64(512, 16),
#ifdef GS_TUNING
                          GS_KEYSONLY64(256, 32), GS_KEYSONLY64(256, 16), GS_KEYSONLY64(512, 20),
#endif
};
#else
const Shape g_shapes[] = {
    GS_FULL(512, 32),   // default for keys-only and 8-byte values: 16384-key tiles, 2 workgroups per CU
    GS_FULL(1024, 16),  // default for 4-byte values (measured best, profiles/r01_sweep_v16_*)
    GS_FULL64(512, 16), // mid sizes (n <= mid_keys): 8192-key tiles, shorter per-tile latency, more workgroups;
                        // and the shape of 64-bit keys at every size (8-byte stage slots: 64 KiB per tile)
#ifdef GS_TUNING  // tuning build only (libgpusort_tuning.so)
    GS_U32ONLY(256, 32), GS_U32ONLY(256, 16),
    GS_U32ONLY(512, 20),  // 10 240-key tiles: 52 KiB of LDS, three workgroups per CU
#endif
};
#endif
constexpr int g_num_shapes = sizeof(g_shapes) / sizeof(g_shapes[0]);

inline int vb_index(uint32_t vb) { return vb == 0 ? 0 : vb == 4 ? 1 : 2; }
inline uint32_t div_up(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

using gs::SLAB_COUNTERS;
using gs::SLAB_DESC;
using gs::SLAB_HIST;
using gs::SLAB_INFO;
using gs::SLAB_STATUS;

constexpr uint32_t MIN_TILE = 4096;  // smallest tile of any compiled shape (sizing of the slab)
constexpr uint32_t KEY64_TILE = 8192;  // tile of every sort of 64-bit keys (MID_SHAPE: 8-byte stage slots, 64 KiB)
constexpr int MID_SHAPE = 2;         // g_shapes index used for n <= mid_keys(vb) unless the caller picked a shape
// profiles/r02_shape_by_size.txt (general path, back-to-back sorts): the 8192-key tile wins up to 2^25 keys for keys-only
// sorts (180 vs 194 us at 2^24, 293 vs 302 at 2^25, loses at 2^26) and for 8-byte values (whose big tile leaves one
// workgroup per CU), up to 2^23 with 4-byte values (1024 x 16 wins from 2^24)
inline uint32_t mid_keys(uint32_t vb) { return vb == 4 ? (1u << 23) : (1u << 25); }


// ---- two-level plan (hybrid_kernels.hpp) ----
using HyHistLauncher = void (*)(hipStream_t, uint32_t grid, const uint32_t* keys, uint32_t* slab, size_t used_words, uint32_t n, uint32_t seg_len0,
                                uint32_t per_wg, uint32_t wg_per_seg, uint32_t* slices, uint32_t cap);
template <int KT>
void launch_hy_hist(hipStream_t s, uint32_t grid, const uint32_t* keys, uint32_t* slab, size_t used_words, uint32_t n, uint32_t seg_len0,
                    uint32_t per_wg, uint32_t wg_per_seg, uint32_t* slices, uint32_t cap) {
    hipLaunchKernelGGL((gs::hy_histogram_kernel<KT>), dim3(grid), dim3(gs::HY_HIST_THREADS), 0, s, keys, slab, used_words, n, seg_len0, per_wg,
                       wg_per_seg, slices, cap);
}
using HyLocalLauncher = void (*)(hipStream_t, uint32_t grid, uint32_t* keys, const uint32_t* tab, uint32_t* slab, uint32_t n, uint32_t descending);
template <int KT, int T, int K>
void launch_hy_local(hipStream_t s, uint32_t grid, uint32_t* keys, const uint32_t* tab, uint32_t* slab, uint32_t n, uint32_t descending) {
    hipLaunchKernelGGL((gs::hy_local_sort_kernel<KT, T, K>), dim3(grid), dim3(T), 0, s, keys, tab, slab, n, descending);
}
// the local sort's workgroup by the mean bucket n / 65 536: it holds 1.5 x the mean at the top of its class (uniform keys stay within
// a few per cent of the mean; what does not fit sends the sort to the LSD passes).  [class][key type]
struct HyLocalClass { uint32_t max_n, cap; };
constexpr HyLocalClass g_hy_class[4] = {{1u << 27, 256 * 12}, {1u << 28, 512 * 12}, {1u << 29, 1024 * 12}, {GS_MAX_KEYS, 1024 * 24}};
#ifdef GS_MINIMAL
const HyHistLauncher g_hy_hist[3] = {launch_hy_hist<0>, nullptr, nullptr};
const HyLocalLauncher g_hy_local[4][3] = {{launch_hy_local<0, 256, 12>, nullptr, nullptr}, {launch_hy_local<0, 512, 12>, nullptr, nullptr},
                                          {launch_hy_local<0, 1024, 12>, nullptr, nullptr}, {launch_hy_local<0, 1024, 24>, nullptr, nullptr}};
#else
const HyHistLauncher g_hy_hist[3] = {launch_hy_hist<0>, launch_hy_hist<1>, launch_hy_hist<2>};
const HyLocalLauncher g_hy_local[4][3] = {{launch_hy_local<0, 256, 12>, launch_hy_local<1, 256, 12>, launch_hy_local<2, 256, 12>},
                                          {launch_hy_local<0, 512, 12>, launch_hy_local<1, 512, 12>, launch_hy_local<2, 512, 12>},
                                          {launch_hy_local<0, 1024, 12>, launch_hy_local<1, 1024, 12>, launch_hy_local<2, 1024, 12>},
                                          {launch_hy_local<0, 1024, 24>, launch_hy_local<1, 1024, 24>, launch_hy_local<2, 1024, 24>}};
#endif
using HyLocalPairsLauncher = void (*)(hipStream_t, uint32_t* keys, void* vals, const uint32_t* tab, const uint32_t* slab, uint32_t n, uint32_t descending);
template <int KT, int VB, int T, int K>
void launch_hy_local_pairs(hipStream_t s, uint32_t* keys, void* vals, const uint32_t* tab, const uint32_t* slab, uint32_t n, uint32_t descending) {
    hipLaunchKernelGGL((gs::hy_local_sort_pairs_kernel<KT, VB, T, K>), dim3(gs::HY_BINS), dim3(T), 0, s, keys, vals, tab, slab, n, descending);
}
// [8-byte values][class][key type]; the 24 576-pair class with 8-byte values does not fit a workgroup's LDS (nullptr: LSD passes)
#define GS_HYP_ROW(VB, T, K) {launch_hy_local_pairs<0, VB, T, K>, launch_hy_local_pairs<1, VB, T, K>, launch_hy_local_pairs<2, VB, T, K>}
#define GS_HYP_ROW0(VB, T, K) {launch_hy_local_pairs<0, VB, T, K>, nullptr, nullptr}
#if defined(GS_MINIMAL) && defined(GS_MIN_PAIRS)
const HyLocalPairsLauncher g_hy_local_pairs[2][4][3] = {{GS_HYP_ROW0(4, 256, 12), GS_HYP_ROW0(4, 512, 12), GS_HYP_ROW0(4, 1024, 12), GS_HYP_ROW0(4, 1024, 24)},
                                                        {GS_HYP_ROW0(8, 256, 12), GS_HYP_ROW0(8, 512, 12), GS_HYP_ROW0(8, 1024, 12), {nullptr, nullptr, nullptr}}};
#elif defined(GS_MINIMAL)
const HyLocalPairsLauncher g_hy_local_pairs[2][4][3] = {};
#else
const HyLocalPairsLauncher g_hy_local_pairs[2][4][3] = {{GS_HYP_ROW(4, 256, 12), GS_HYP_ROW(4, 512, 12), GS_HYP_ROW(4, 1024, 12), GS_HYP_ROW(4, 1024, 24)},
                                                        {GS_HYP_ROW(8, 256, 12), GS_HYP_ROW(8, 512, 12), GS_HYP_ROW(8, 1024, 12), {nullptr, nullptr, nullptr}}};
#endif
inline int hy_class(uint32_t n) { return n <= g_hy_class[0].max_n ? 0 : n <= g_hy_class[1].max_n ? 1 : n <= g_hy_class[2].max_n ? 2 : 3; }
constexpr uint32_t HY_MIN_PAIRS_DEFAULT = (1u << 25) + 1u;  // pairs: from where the position-chain plan (its fall-back) starts — at 2^25 pairs the two-level plan already wins (61.6 against 58.8, 44.6 against 40.4 GKeys/s with 4- / 8-byte values), at 2^24 it loses
constexpr uint32_t HY_MIN_KEYS_DEFAULT = 3u << 24;  // 50 M keys: measured, the LSD passes win at 2^25 (121 against 102 GKeys/s), the two-level plan at 2^26 (139 against 122): below, its 65 536 buckets are a few hundred keys each and a workgroup per bucket is mostly launch (profiles/r05_two_level_threshold.txt)

}  // namespace

struct gs_onesweep {
    uint32_t max_keys;
    gs_mode mode;
    uint32_t value_bytes;
    int shape;
    int shape_auto;  // 1 = the library picks (mid sizes use MID_SHAPE); 0 after gs_onesweep_set_shape / gs_onesweep_options::shape_*
    int small_path; // 1 = single-tile kernel for n <= SMALL_TILE (default), 0 = always the tiled path
    int mid_path;    // 1 = two-launch MSD + bucket sort for single-tile limit < n <= 2^20 (default), 0 = the six-launch path
    int skip_passes; // 1 = identity passes (one digit value for all keys) are dropped in pairs (default)
    int pos_chains;  // keys-only sorts of skewed 32-bit keys run every pass on position chains: 1 allowed (default), 0 never
    int key64_sweeps;       // 64-bit keys: 1 = one GlobalHistogram + Scan for all eight passes (default), 2 = one per word (gs_onesweep_options::key64_sweeps; A/B, tests)
    uint32_t pos_min_keys;  // ... from this many keys up (default 2^25 + 1: where the big tile shape takes over; gs_onesweep_options::position_chains_min_log2)
    int rank_mode;  // 0 ballot multi-split, 1 returning LDS atomic (needs the lane-order probe to pass)
    uint32_t* slab;
    size_t slab_words;
    uint32_t* partials;  // the histogram workgroups' tables: hist_blocks(max_keys) x HIST_TABLE_WORDS, summed by hist_reduce_kernel
    size_t partials_words;
    int profiling;
    hipEvent_t ev[GS_PROFILE_SLOTS + 1];
    bool ev_valid;
    bool profile_pending;
    void* trace_buf;   // experiment builds only (GS_EXP & 2): per-tile phase timestamps
    const void* msd_keys;  // shard whose top-byte histogram + scan currently sit in the slab (msd_prepare)
    uint32_t msd_n, msd_grid;
    gs_key_type msd_kt;
    uint32_t* pinned;  // 1024 + 8 words of pinned host memory for read-backs
    // geometry of the last tiled call, for gs_debug_check_state (tile 0 = the last call left no scan state)
    uint32_t last_n, last_tile, last_tile0, last_p0, last_np, last_dyn, last_desc_stride, last_pos_tile = 0;
    bool hist_dirty;   // a call failed between the histogram launch and the kernel that hands HIST back zeroed
    uint32_t hist_blocks_opt;  // gs_onesweep_options::hist_blocks (0 = the library picks)
    int first_pass_big;        // gs_onesweep_options::first_pass_big
    uint32_t debug_flags;      // gs_onesweep_options::debug_flags
    int plan;              // gs_onesweep_options::plan / gs_onesweep_set_plan: 0 the library picks, 1 LSD passes only, 2 two-level plan wherever it can run
    uint32_t hy_min_keys;  // plan 0: the two-level plan from this many keys up
    uint32_t* hy_tab;      // the two-level plan's tables (gs::HYT_WORDS), nullptr: the handle cannot run it (pairs, 64-bit keys only ...)
    uint32_t hy_grid;      // workgroups of its histogram kernel (a multiple of NCH)
    int last_hy;           // the last sort was enqueued with the two-level plan's launches (whether it RAN on it is the device's decision: gs_onesweep_last_plan)
    bool exp_keep_desc;  // experiment builds (GS_EXP & 1024): the histogram kernel leaves the descriptor rows alone
};

namespace {

size_t slab_words_for(uint32_t max_keys) {
    // descriptor rows: four passes on the smallest tile, or the eight passes of a 64-bit sort on its 8192-key tile
    // (two-level plan: its second pass has CHMAX chains — on 16 384-key tiles, from 2^20 keys up at the earliest: covered by the rows of
    //  the smallest tile as soon as max_keys / 4096 - max_keys / 16 384 >= 2 * CHMAX, i.e. from 2^12 x 171 keys)
    const size_t rows4 = 4 * ((size_t)div_up(max_keys, MIN_TILE) + 2 * gs::MAXCH + 2);
    const size_t rows8 = gs::MAX_PASSES * ((size_t)div_up(max_keys, KEY64_TILE) + 2 * gs::MAXCH + 2);
    return SLAB_DESC + (rows4 > rows8 ? rows4 : rows8) * (size_t)gs::RADIX;
}

using HistLauncher = void (*)(hipStream_t, uint32_t, const uint32_t*, uint32_t*, size_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t,
                              uint32_t*);
template <int KT>
void launch_hist(hipStream_t s, uint32_t blocks, const uint32_t* keys, uint32_t* slab, size_t used_words, uint32_t n,
                 uint32_t seg_len0, uint32_t p0, uint32_t np, uint32_t word, uint32_t allow_pos, uint32_t* partials) {
    hipLaunchKernelGGL((gs::global_histogram_kernel<KT>), dim3(blocks), dim3(gs::GHIST_THREADS), 0, s, keys, slab,
                       used_words, n, seg_len0, p0, np, word, allow_pos, partials);
    // the workgroups' tables -> the HIST region (one thread per bin)
    hipLaunchKernelGGL(gs::hist_reduce_kernel, dim3(np * gs::NCH * gs::RADIX / 64u), dim3(256), 0, s, partials, blocks,
                       np * gs::NCH * gs::RADIX, slab + SLAB_HIST);
}
inline hipError_t zero_hist(gs_onesweep* h, hipStream_t s) {  // the HIST region: four joint tables + what the keys look like as a whole
    return hipMemsetAsync(h->slab + SLAB_HIST, 0, gs::HIST_WORDS * sizeof(uint32_t), s);
}
const HistLauncher g_hist[6] = {launch_hist<0>, launch_hist<1>, launch_hist<2>, launch_hist<3>, launch_hist<4>, launch_hist<5>};
inline bool is_key64(gs_key_type kt) { return (int)kt >= 3; }

uint32_t hist_blocks(uint32_t n, uint32_t forced = 0) {
    // one chunk per workgroup at mid sizes (measured: 4/8/16 chunks per workgroup — fewer closing global atomics,
    // less parallelism — are slower: 11 -> 15-23 us at 2^16..2^20)
    // Above that ONE workgroup per CU (half of them up to 2^22 keys): every workgroup closes with one global atomic per
    // non-empty bin of its 4 x 4096-bin LDS histograms (~14 000 of them) — most of the kernel at mid sizes and still 6 %
    // of it at 2^28.  512 -> 256 workgroups: 30 -> 21 us at 2^21, 59 -> 51 us at 2^25, 179 -> 156 us at 2^27,
    // 300 -> 282 us at 2^28; counts that do not divide the CUs evenly (320, 384, 448) lose 10-35 %
    // (profiles/r02_hist_blocks_mid_sizes.txt).
    static const uint32_t cus = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        return (uint32_t)v;
    }();
    const uint32_t want = div_up(n, gs::HIST_CHUNK);
    const uint32_t cap = n <= (1u << 22) ? (cus + 1) / 2 : cus;
    if (forced > 0) return forced < want ? forced : want;  // gs_onesweep_options::hist_blocks (tuning aid)
    return want < 1 ? 1 : (want > cap ? cap : want);
}

// persistent workgroups of the position-chain pass: two per CU (76 KiB of LDS each)
uint32_t pos_grid() {
    static const uint32_t cus = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        return (uint32_t)v;
    }();
    return 2u * cus;
}

// workgroups of the two-level plan's histogram kernel: one per CU, a multiple of NCH (position segments get equal numbers of them)
uint32_t hy_grid_for_device() {
    const uint32_t cus = pos_grid() / 2u;
    return cus >= gs::NCH ? cus / gs::NCH * gs::NCH : gs::NCH;
}

// most workgroups the histogram kernel is ever launched with for a handle of max_keys keys (sizes its slices)
uint32_t hist_blocks_cap(uint32_t max_keys, uint32_t forced = 0) {
    uint32_t m = hist_blocks(max_keys);
    if (max_keys > (1u << 22)) { const uint32_t b = hist_blocks(1u << 22); m = b > m ? b : m; }
    if (forced > m) m = forced;
    return m;
}

bool misaligned(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) != 0; }

// Clears the scan state and runs GlobalHistogram + Scan for passes p0 .. p0+np-1
// (pass p0 over position segments, later passes over digit groups of the previous digit).
struct PassPlan {
    uint32_t grid, desc_stride;
    uint32_t grid0;  // grid of the plan's first pass (shape0_index)
};
// shape0_index >= 0: the plan's first pass runs on that (larger) tile shape, the others on shape_index
// hy: the sort may run on the two-level plan (hybrid_kernels.hpp): its histogram sweep replaces GlobalHistogram, its scan runs in front
// of the ordinary one, and both plans' launches follow (sort_impl)
gs_status prologue(gs_onesweep* h, const void* d_keys, uint32_t n, gs_key_type kt, hipStream_t s, uint32_t p0,
                   uint32_t np, PassPlan* plan, uint32_t scan_plan = 0, int shape_index = -1, uint32_t word = 0, int shape0_index = -1,
                   uint32_t pos_tile = POS_TILE, bool hy = false, bool pregrouped = false) {
    h->msd_keys = nullptr;  // whatever an earlier gs_onesweep_msd_prepare left in the slab is overwritten now
    h->last_hy = hy ? 1 : 0;
    const Shape& sh = g_shapes[shape_index < 0 ? h->shape : shape_index];
    const uint32_t tile = (uint32_t)sh.threads * sh.kpt;
    const uint32_t tile0 = shape0_index < 0 ? tile : (uint32_t)g_shapes[shape0_index].threads * g_shapes[shape0_index].kpt;
    const uint32_t tiles = div_up(n, tile < tile0 ? tile : tile0);
    // every chain: its tiles (+1 partial) + row 0; bit 2 of the plan: the sort may end up on the (smaller) position-chain tiles
    // (two-level plan: its second pass runs on CHMAX chains)
    const uint32_t rows = ((scan_plan & 4u) && (pos_tile & 0x7fffffffu) < tile ? div_up(n, pos_tile & 0x7fffffffu) : tiles) + (hy ? 2 * gs::CHMAX + 8 : 2 * gs::MAXCH + 2);
    const uint32_t desc_stride = rows * gs::RADIX;
    // (hy: the descriptor regions of LSD passes 2 and 3 are zeroed by the launch of LSD pass 1 (BM_ZERO_DESC23) if — and only if — those passes run)
    const size_t used_words = h->exp_keep_desc ? (size_t)SLAB_DESC : SLAB_DESC + (size_t)(hy ? 2u : np) * desc_stride;
    if (SLAB_DESC + (size_t)np * desc_stride > h->slab_words) return GS_ERR_SIZE;  // (cannot happen with the tiles the library picks)
    // position segments of the first pass: equal, multiples of the histogram chunk — and of the first pass's tile where that is a
    // multiple of the chunk (every shape the library picks): its chains then consist of whole tiles, 16 partial tiles fewer (at
    // mid sizes one launch round fewer: 2^24 keys are 1024 tiles of 16 384, two rounds on 512 slots)
    const uint32_t seg_unit = (tile0 % gs::HIST_CHUNK == 0u) ? tile0 : gs::HIST_CHUNK;
    const uint32_t seg_len0 = div_up(div_up(n, gs::NCH), seg_unit) * seg_unit;
    // no separate clear: the histogram kernel zeroes the scan state while it reads the keys (profile slot 0 stays 0)
    // hist_reduce_kernel OVERWRITES the np tables it sums; the tables it does not touch and the HX words (HX_SKEW and the keys'
    // OR / AND, which the histogram kernel sets with atomics) rely on the HIST region being zero between calls — the first pass
    // launched after the Scan, or the read-back entry points, hand it back zeroed.  A call that failed in between left it
    // dirty: zero it here, once.
    if (h->hist_dirty) GS_HIP(zero_hist(h, s));
    // (64-bit keys: passes 4..7 — or the second round's kernels — are charged to slot 6; the events of round 0 stay where they are)
    const bool rec = h->profiling && word == 0;
    if (rec) GS_HIP(hipEventRecord(h->ev[0], s));
    if (rec) GS_HIP(hipEventRecord(h->ev[1], s));
    h->hist_dirty = true;  // until the caller has launched whatever zeroes HIST again
    if (hy) {
        // one workgroup per CU at most, NCH position segments of equal numbers of workgroups, every workgroup >= one tile of keys
        const uint32_t seg_tiles = seg_len0 / tile0;
        const uint32_t wg_per_seg = seg_tiles < h->hy_grid / gs::NCH ? (seg_tiles ? seg_tiles : 1u) : h->hy_grid / gs::NCH;
        const uint32_t G = wg_per_seg * gs::NCH;
        const uint32_t per_wg = div_up(div_up(seg_len0, wg_per_seg), gs::HIST_CHUNK) * gs::HIST_CHUNK;
        g_hy_hist[kt](s, G, static_cast<const uint32_t*>(d_keys), h->slab, used_words, n, seg_len0, per_wg, wg_per_seg, h->partials, g_hy_class[hy_class(n)].cap);
        hipLaunchKernelGGL(gs::hy_reduce_kernel, dim3(gs::RADIX + gs::NCH), dim3(256), 0, s, h->partials, G, wg_per_seg, h->hy_tab, h->slab + SLAB_HIST);
        if (rec) GS_HIP(hipEventRecord(h->ev[2], s));
        hipLaunchKernelGGL(gs::hy_scan_kernel, dim3(1), dim3(1024), 0, s, h->slab, h->hy_tab, n, seg_len0, desc_stride, g_hy_class[hy_class(n)].cap, tile, pregrouped ? 1u : 0u);
    } else {
    g_hist[kt](s, hist_blocks(n, h->hist_blocks_opt), static_cast<const uint32_t*>(d_keys), h->slab, used_words, n, seg_len0, p0, np, word,
               (scan_plan & 4u) ? (h->pos_chains == 2 ? 3u : 1u) : 0u, h->partials);
    }
#if (GS_EXP & 2)
    GS_HIP(hipMemcpyAsync(h->slab + SLAB_STATUS + 8, &h->trace_buf, sizeof(void*), hipMemcpyHostToDevice, s));
#endif
    if (rec && !hy) GS_HIP(hipEventRecord(h->ev[2], s));
    if (np > 4)  // 64-bit keys: all eight passes from one sweep
        hipLaunchKernelGGL(gs::scan_kernel<8>, dim3(np), dim3(256), 0, s, h->slab + SLAB_HIST, h->slab + SLAB_DESC,
                           h->slab + SLAB_INFO, desc_stride, n, seg_len0, tile, scan_plan, pos_tile, tile0);
    else
        hipLaunchKernelGGL(gs::scan_kernel<4>, dim3(np), dim3(256), 0, s, h->slab + SLAB_HIST, h->slab + SLAB_DESC,
                           h->slab + SLAB_INFO, desc_stride, n, seg_len0, tile, scan_plan, pos_tile, tile0);
    if (rec) GS_HIP(hipEventRecord(h->ev[3], s));
    plan->grid = div_up(n, tile) + (hy ? gs::CHMAX : gs::MAXCH) + 1;  // chains end in partial tiles: at most one more tile per chain than n/tile
    plan->grid0 = div_up(n, tile0) + gs::MAXCH + 1;
    plan->desc_stride = desc_stride;
    h->last_n = n; h->last_tile = tile; h->last_tile0 = tile0; h->last_p0 = p0; h->last_np = np; h->last_dyn = (scan_plan & 2u) ? 1u : 0u; h->last_pos_tile = pos_tile;
    h->last_desc_stride = desc_stride;
    return GS_OK;
}

gs_status check_common(gs_onesweep* h, const void* a, const void* b, uint32_t n, gs_key_type kt, gs_order order) {
    if (!h || !a || !b || misaligned(a) || misaligned(b)) return GS_ERR_ARG;
    if ((int)kt < 0 || (int)kt > 5 || (int)order < 0 || (int)order > 1) return GS_ERR_ARG;
    if (n == 0 || n > h->max_keys || n > GS_MAX_KEYS) return GS_ERR_SIZE;
    return GS_OK;
}

// single-tile fast path: one launch, no scan state.  Three tile sizes: 8192 slots (every mode), 16384
// (keys-only and 4-byte values), 32768 (keys-only) — what fits 160 KiB of LDS.
using SmallLauncher = void (*)(hipStream_t, uint32_t*, void*, uint32_t, uint32_t, uint32_t*);
template <int T, int K, int VB, int KT, int RANK>
void launch_small(hipStream_t s, uint32_t* keys, void* vals, uint32_t n, uint32_t descending, uint32_t* status) {
    hipLaunchKernelGGL((gs::small_sort_kernel<T, K, VB, KT, RANK>), dim3(1), dim3(T), 0, s, keys, vals, n, descending, status);
}
#define GS_SMALL_ROW(T, K, VB, R) {launch_small<T, K, VB, 0, R>, launch_small<T, K, VB, 1, R>, launch_small<T, K, VB, 2, R>, nullptr, nullptr, nullptr}
#define GS_SMALL_ROW64(T, K, VB, R)                                                                                   \
    {launch_small<T, K, VB, 0, R>, launch_small<T, K, VB, 1, R>, launch_small<T, K, VB, 2, R>, launch_small<T, K, VB, 3, R>, \
     launch_small<T, K, VB, 4, R>, launch_small<T, K, VB, 5, R>}
#define GS_SMALL_NONE {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}
// [size class][rank mode][vb index][key type]; 64-bit keys: the classes up to 8192 slots.  The two smallest classes (256 x 4 and
// 256 x 8 slots) exist because a sort of 2^10 keys in the 8192-slot shape pays for 8192 slots in every pass: 10.5 us against
// 8.1 (profiles/r04_small_shapes.txt; the reference's size sweep starts there, GPUSortingD3D12/Tests.h:392-393,415-416)
#ifndef GS_MINIMAL
const SmallLauncher g_small[5][2][3][6] = {
    {{GS_SMALL_ROW64(256, 4, 0, 0), GS_SMALL_ROW64(256, 4, 4, 0), GS_SMALL_ROW64(256, 4, 8, 0)},
     {GS_SMALL_ROW64(256, 4, 0, 1), GS_SMALL_ROW64(256, 4, 4, 1), GS_SMALL_ROW64(256, 4, 8, 1)}},
    {{GS_SMALL_ROW64(256, 8, 0, 0), GS_SMALL_ROW64(256, 8, 4, 0), GS_SMALL_ROW64(256, 8, 8, 0)},
     {GS_SMALL_ROW64(256, 8, 0, 1), GS_SMALL_ROW64(256, 8, 4, 1), GS_SMALL_ROW64(256, 8, 8, 1)}},
    {{GS_SMALL_ROW64(512, 16, 0, 0), GS_SMALL_ROW64(512, 16, 4, 0), GS_SMALL_ROW64(512, 16, 8, 0)},
     {GS_SMALL_ROW64(512, 16, 0, 1), GS_SMALL_ROW64(512, 16, 4, 1), GS_SMALL_ROW64(512, 16, 8, 1)}},
    {{GS_SMALL_ROW(1024, 16, 0, 0), GS_SMALL_ROW(1024, 16, 4, 0), GS_SMALL_NONE},
     {GS_SMALL_ROW(1024, 16, 0, 1), GS_SMALL_ROW(1024, 16, 4, 1), GS_SMALL_NONE}},
    {{GS_SMALL_ROW(1024, 32, 0, 0), GS_SMALL_NONE, GS_SMALL_NONE},
     {GS_SMALL_ROW(1024, 32, 0, 1), GS_SMALL_NONE, GS_SMALL_NONE}},
};
#endif
inline SmallLauncher small_launcher(uint32_t n, int rank_mode, uint32_t vb, gs_key_type kt) {
#ifdef GS_MINIMAL
    return nullptr;
#else
    const int cls = n <= 1024 ? 0 : n <= 2048 ? 1 : n <= 8192 ? 2 : n <= 16384 ? 3 : n <= 32768 ? 4 : 5;
    return cls < 5 ? g_small[cls][rank_mode][vb_index(vb)][kt] : nullptr;  // nullptr: no single-tile kernel for this case
#endif
}

// mid sizes: two launches (mid_kernels.hpp).  [class][rank mode][vb index][key type]; classes by the bucket K2 can hold:
// 8192 keys (n <= 2^20, every value width; K1: <= 128 tiles of 8192), 16 384 (n <= 2^21, keys-only and 4-byte values; K1: <= 128
// tiles of 16 384), 32 768 (n <= 2^22, keys-only; K1: <= 256 tiles of 16 384 — the 32 768-key tile spilled there and kept
// half the CUs idle, 44 us of a 67 us sort, profiles/r03_mid_size_timeline.txt)
using MidLauncher = void (*)(hipStream_t, uint32_t n_tiles, uint32_t* keys, uint32_t* alt, void* vals, void* valt, uint32_t* scratch,
                             uint32_t* status, uint32_t n, uint32_t descending);
template <int VB, int KT, int RANK, int T, int K, int T2 = T, int K2 = K>
void launch_mid(hipStream_t s, uint32_t tiles, uint32_t* keys, uint32_t* alt, void* vals, void* valt, uint32_t* scratch, uint32_t* status,
                uint32_t n, uint32_t descending) {
    hipLaunchKernelGGL((gs::mid_msd_kernel<VB, KT, RANK, T, K, T2 * K2>), dim3(tiles), dim3(T), 0, s, keys, alt, vals, valt, scratch, status,
                       n, descending);
    hipLaunchKernelGGL((gs::bucket_sort_kernel<VB, KT, RANK, T2, K2>), dim3(gs::RADIX), dim3(T2), 0, s, keys, alt, vals, valt, scratch,
                       status, n, descending);
}
#ifndef GS_MINIMAL
#define GS_MID_ROW(VB, R, ...) {launch_mid<VB, 0, R, __VA_ARGS__>, launch_mid<VB, 1, R, __VA_ARGS__>, launch_mid<VB, 2, R, __VA_ARGS__>}
#define GS_MID_NONE {nullptr, nullptr, nullptr}
const MidLauncher g_mid[5][2][3][3] = {
    {{GS_MID_ROW(0, 0, 512, 16), GS_MID_ROW(4, 0, 512, 16), GS_MID_ROW(8, 0, 512, 16)},
     {GS_MID_ROW(0, 1, 512, 16), GS_MID_ROW(4, 1, 512, 16), GS_MID_ROW(8, 1, 512, 16)}},
    {{GS_MID_ROW(0, 0, 512, 32), GS_MID_ROW(4, 0, 512, 32), GS_MID_NONE}, {GS_MID_ROW(0, 1, 512, 32), GS_MID_ROW(4, 1, 512, 32), GS_MID_NONE}},
    {{GS_MID_ROW(0, 0, 512, 32, 1024, 32), GS_MID_NONE, GS_MID_NONE}, {GS_MID_ROW(0, 1, 512, 32, 1024, 32), GS_MID_NONE, GS_MID_NONE}},
    {{GS_MID_ROW(0, 0, 1024, 32, 1024, 34), GS_MID_NONE, GS_MID_NONE}, {GS_MID_ROW(0, 1, 1024, 32, 1024, 34), GS_MID_NONE, GS_MID_NONE}},
    {{GS_MID_NONE, GS_MID_ROW(4, 0, 512, 32, 512, 34), GS_MID_NONE}, {GS_MID_NONE, GS_MID_ROW(4, 1, 512, 32, 512, 34), GS_MID_NONE}},
};
#endif
// round 5: class 3 — keys-only up to 2^23 (K1: 256 tiles of 32 768, one per CU; K2 holds 34 816 keys: 6 % above the mean bucket) — and class 4 —
// 4-byte values up to 2^22 pairs (K1: 256 tiles of 16 384; K2 holds 17 408 pairs): 74.6 -> 102 GKeys/s at 2^23 keys, profiles/r05_mid_classes.txt
constexpr uint32_t g_mid_tile[5] = {512 * 16, 512 * 32, 512 * 32, 1024 * 32, 512 * 32};  // K1's tile
constexpr uint32_t g_mid_tiles[5] = {128, 128, 256, 256, 256};                 // ... and how many of them at most (<= MID_MAX_TILES; never more than fit the chip at once:
                                                                           // 512 tiles of 16 384 keys for 2^23 keys left half of them to be adopted one by one — 1.7 ms)
static_assert(g_mid_tiles[3] <= gs::MID_MAX_TILES && g_mid_tiles[4] <= gs::MID_MAX_TILES, "mid-size classes");
// class of a mid-size sort, -1: the general pipeline
inline int mid_class(uint32_t n, uint32_t vb) {
    if (n <= g_mid_tiles[0] * g_mid_tile[0]) return 0;
    if (n <= g_mid_tiles[1] * g_mid_tile[1] && vb != 8) return 1;
    if (n <= g_mid_tiles[2] * g_mid_tile[2] && vb == 0) return 2;
    if (n <= g_mid_tiles[3] * g_mid_tile[3] && vb == 0) return 3;
    if (n <= g_mid_tiles[4] * g_mid_tile[4] && vb == 4) return 4;
    return -1;
}

// Which way a sort of n elements goes — decided on the host from sizes, modes and options alone (what the KEYS look like is the
// device's business: identity passes, skew, the two-level plan's validity).  sort_impl enqueues accordingly; gs_onesweep_sort_sharded
// asks whether a bucket it is about to receive will be offered the two-level plan (`hy`) before it chooses the exchange's layout.
struct SortRoute {
    SmallLauncher small;  // != nullptr: one workgroup, one launch
    int mid_cls;          // >= 0: the two-launch mid-size route (mid_kernels.hpp), class index
    int shape, shape0;    // tile shapes of the general pipeline: passes 1.., first pass
    uint32_t dyn;         // 2: the Scan kernel plans the passes on the device (identity passes dropped, source buffers); 0: fixed ping-pong
    bool pos;             // the sort may be planned on position chains (PF_POS): every pass is launched in both chain forms
    bool hy;              // the sort is offered the two-level plan (hybrid_kernels.hpp): both plans' launches are enqueued
};
SortRoute sort_route(const gs_onesweep* h, uint32_t n, gs_key_type kt, uint32_t vb) {
    SortRoute r{};
    // routing by size: one workgroup up to 8192 keys; two launches (MSD pass + bucket sorts) up to 2^20 (2^22 pairs with 4-byte
    // values, 2^23 keys-only: mid_class); the general pipeline above.  (The 16 384- and 32 768-slot single-tile kernels serve when the mid-size route is switched off:
    // with it, 2^15 keys take 18 us instead of 34, profiles/r02_size_and_entropy_sweep.txt.)
    r.mid_cls = (h->mid_path && h->shape_auto && n > gs::SMALL_TILE && !is_key64(kt)) ? mid_class(n, vb) : -1;
#ifdef GS_MINIMAL
    r.mid_cls = -1;
#endif
    r.small = (h->small_path && r.mid_cls < 0) ? small_launcher(n, h->rank_mode, vb, kt) : nullptr;
    // 64-bit keys: 8-byte stage slots fit 8192-key tiles only (the mid-size shape), at every size
    // (a sort that may be planned on position chains — see `pos` below — runs on the default tile: the dual kernel's shapes)
    const bool pos_size = h->skip_passes && h->rank_mode == 1 && !is_key64(kt) && h->pos_chains != 0 && n >= h->pos_min_keys;
    r.shape = (h->shape_auto && n <= mid_keys(vb) && !pos_size) ? MID_SHAPE : h->shape;
    if (is_key64(kt) && !g_shapes[r.shape].fn[h->rank_mode][vb_index(vb)][kt]) r.shape = MID_SHAPE;
    const Shape& sh = g_shapes[r.shape];
    // Mid sizes, keys-only (2^22 < n <= 2^25: the 8192-key tile): the FIRST pass runs on the 16 384-key tile.  Its position segments are
    // whole tiles (prologue), so 2^24 keys are exactly 1024 tiles — two launch rounds on the 512 slots of that shape instead of three
    // rounds of 8192-key tiles on 768 — and its input is cold, which the larger tile streams better; the later passes' chains are
    // digit groups with a partial tile at each end, which overflow the round.  gs_onesweep_options::first_pass_big = 0 switches it off (A/B).
    r.shape0 = (h->first_pass_big && h->shape_auto && r.shape == MID_SHAPE && vb == 0 && !is_key64(kt) && n > (1u << 22) &&
                g_shapes[0].fn[h->rank_mode][0][kt] != nullptr) ? 0 : r.shape;
    // The scan kernel decides on the device which passes run and which buffer each one reads (identity passes
    // are dropped in pairs, see scan_kernel); every pass is handed (keys, alt) and the sort's order.
    r.dyn = h->skip_passes ? 2u : 0u;
    // The sort may run on position chains in every pass (PF_POS; decided on the device: the histogram kernel finds the
    // digit groups uneven, the Scan kernel plans accordingly) — every pass is then launched in both chain forms and the
    // plan says which one works.  Sorts of 32-bit keys on the big tile shape, LDS-atomic ranking; gs_onesweep_options::position_chains = 0
    // switches it off.
    r.pos = r.dyn && h->rank_mode == 1 && !is_key64(kt) && h->pos_chains != 0 && n >= h->pos_min_keys &&
            (vb == 0 ? g_dual[0][kt] : g_posv[vb == 8][0][kt]) != nullptr &&
            (vb == 4 ? sh.threads * sh.kpt == 16384 : (sh.threads == 512 && sh.kpt == 32));  // (the plan's last pass runs on 16 384-key tiles)
    // Two-level plan (hybrid_kernels.hpp): sorts of 32-bit keys — keys-only and pairs with 4- / 8-byte values — that may also run on
    // position chains (its fall-back when the keys turn out skewed) — the histogram sweep counts the 16-bit prefixes, and the device decides which plan runs.
    // (position_chains = 2 asks for the position-chain plan whatever the keys look like: only plan 2 overrides that)
    r.hy = r.pos && !r.small && r.mid_cls < 0 && h->hy_tab != nullptr && g_hy_hist[kt] != nullptr && h->plan != 1 &&
           (h->plan == 2 || (n >= (vb ? HY_MIN_PAIRS_DEFAULT : h->hy_min_keys) && h->pos_chains != 2)) &&
           (vb == 0 || (g_persist[vb == 8][kt] != nullptr && g_hy_local_pairs[vb == 8][hy_class(n)][kt] != nullptr)) &&
           (size_t)gs::SLAB_DESC + 4 * (size_t)(div_up(n, pos_tile_for(vb) & 0x7fffffffu) + 2 * gs::CHMAX + 8) * gs::RADIX <= h->slab_words;  // (prologue's row formula)
    return r;
}

// values_ready (multi-GPU): an event behind which d_vals is complete — the keys already are, so GlobalHistogram + Scan (which read
// keys only) run before the stream waits for it; the one- and two-launch routes wait first.
gs_status sort_impl(gs_onesweep* h, void* d_keys, void* d_vals, void* d_alt_keys, void* d_alt_vals, uint32_t n,
                    gs_key_type kt, gs_order order, hipStream_t s, uint32_t vb, hipEvent_t values_ready = nullptr, bool pregrouped = false) {
    // pregrouped (multi-GPU, gs_onesweep_sort_sharded): the input lies in the ALTERNATE buffers, already grouped by its top byte in
    // ascending order — the bucket exchange landed it bin by bin — and the caller made sure (hy_offered) that the sort is offered the
    // two-level plan: its pass A (the top-byte partition) is skipped, pass B reads the alternate buffers as it always does.  If the device
    // finds the plan void, hy_void_copy_kernel moves the input to the caller's buffers and the four LSD passes run as ever.
    const SortRoute route = sort_route(h, n, kt, vb);
    if (pregrouped && !route.hy) return GS_ERR_ARG;  // (the caller asks sort_route first)
    const int mid_cls = route.mid_cls;
    const bool use_mid = mid_cls >= 0;
    if (SmallLauncher small = route.small) {
        if (values_ready) GS_HIP(hipStreamWaitEvent(s, values_ready, 0));
        if (h->profiling) GS_HIP(hipEventRecord(h->ev[0], s));
        small(s, static_cast<uint32_t*>(d_keys), d_vals, n, order == GS_ORDER_DESCENDING ? 1u : 0u, h->slab + SLAB_STATUS);
        h->last_tile = 0;
        h->last_hy = 0;
        if (h->profiling)  // everything is charged to slot 0 (and the total)
            for (int e = 1; e <= 7; ++e) GS_HIP(hipEventRecord(h->ev[e], s));
        GS_HIP(hipGetLastError());
        h->profile_pending = h->profiling != 0;
        // the single-tile kernel has no spin and cannot time out: it sets the status word to OK
        return GS_OK;
    }
#ifndef GS_MINIMAL
    if (use_mid) {
        // one MSD pass + one LDS sort per top-byte bucket (a skewed top byte: the LSD passes inside the first kernel)
        if (values_ready) GS_HIP(hipStreamWaitEvent(s, values_ready, 0));
        if (h->profiling) GS_HIP(hipEventRecord(h->ev[0], s));
        g_mid[mid_cls][h->rank_mode][vb_index(vb)][kt](s, div_up(n, g_mid_tile[mid_cls]), static_cast<uint32_t*>(d_keys), static_cast<uint32_t*>(d_alt_keys),
                                              d_vals, d_alt_vals, h->slab + gs::SLAB_MID, h->slab + SLAB_STATUS, n,
                                              order == GS_ORDER_DESCENDING ? 1u : 0u);
        h->last_tile = 0;
        h->last_hy = 0;
        if (h->profiling)  // everything is charged to slot 0 (and the total)
            for (int e = 1; e <= 7; ++e) GS_HIP(hipEventRecord(h->ev[e], s));
        GS_HIP(hipGetLastError());
        h->profile_pending = h->profiling != 0;
        return GS_OK;
    }
#endif
    const int shape = route.shape, shape0 = route.shape0;
    const Shape& sh = g_shapes[shape];
    BinLauncher fn = sh.fn[h->rank_mode][vb_index(vb)][kt];
    if (!fn) return GS_ERR_ARG;
    BinLauncher fn0 = g_shapes[shape0].fn[h->rank_mode][vb_index(vb)][kt];
    const uint32_t dyn = route.dyn;
    const bool pos = route.pos, hy = route.hy;
    uint32_t* k[2] = {static_cast<uint32_t*>(d_keys), static_cast<uint32_t*>(d_alt_keys)};
    void* v[2] = {d_vals, d_alt_vals};
    // 64-bit keys: ONE GlobalHistogram + Scan plans all eight passes (eight joint tables from one sweep over the keys; the
    // chains of pass 4 are the groups of byte 3's values, as inside a word) — identity passes are dropped in pairs across the
    // whole key (keys below 2^32: four passes).  key64_sweeps = 2 (A/B) or a caller-picked tile too small for the slab's
    // eight descriptor regions: two rounds of histogram + scan + 4 passes — the low word's bytes, then (stable) the high
    // word's; each round leaves its result in the caller's buffers, only the last one carries the descending reversal.
    const bool one_sweep = is_key64(kt) && h->key64_sweeps == 1 &&
                           SLAB_DESC + (size_t)gs::MAX_PASSES * (div_up(n, (uint32_t)sh.threads * sh.kpt) + 2 * gs::MAXCH + 2) * gs::RADIX <= h->slab_words;
    const uint32_t rounds = (is_key64(kt) && !one_sweep) ? 2u : 1u;
    const uint32_t NP = one_sweep ? gs::MAX_PASSES : 4u;
#if (GS_EXP & (1024 | 2048))
    const uint32_t exp_mode = h->debug_flags & (256u | 512u | 1024u | 2048u);
    h->exp_keep_desc = (exp_mode & 256u) != 0u;
#else
    const uint32_t exp_mode = 0u;
#endif
    for (uint32_t word = 0; word < rounds; ++word) {
        const uint32_t desc_bit = (order == GS_ORDER_DESCENDING && word + 1 == rounds) ? 1u : 0u;
        PassPlan plan;
        gs_status st = prologue(h, pregrouped ? d_alt_keys : d_keys, n, kt, s, 0, NP, &plan, desc_bit | dyn | (pos ? 4u : 0u), shape, word, shape0, pos_tile_for(vb), hy, pregrouped);
        if (st != GS_OK) return st;
        if (values_ready && word == 0) GS_HIP(hipStreamWaitEvent(s, values_ready, 0));  // histogram + scan ran on the keys meanwhile
        if (pregrouped) {  // the plan may turn out void: the LSD passes read the caller's buffers (exits at once otherwise)
            const size_t kw = (size_t)n;  // key words
            hipLaunchKernelGGL(gs::hy_void_copy_kernel, dim3(pos_grid()), dim3(256), 0, s, h->slab, static_cast<const uint4*>(d_alt_keys), static_cast<uint4*>(d_keys),
                               kw / 4, static_cast<const uint32_t*>(d_alt_keys) + (kw & ~(size_t)3), static_cast<uint32_t*>(d_keys) + (kw & ~(size_t)3), (uint32_t)(kw & 3));
            if (vb) {
                const size_t vw = (size_t)n * (vb / 4);
                hipLaunchKernelGGL(gs::hy_void_copy_kernel, dim3(pos_grid()), dim3(256), 0, s, h->slab, static_cast<const uint4*>(d_alt_vals), static_cast<uint4*>(d_vals),
                                   vw / 4, static_cast<const uint32_t*>(d_alt_vals) + (vw & ~(size_t)3), static_cast<uint32_t*>(d_vals) + (vw & ~(size_t)3), (uint32_t)(vw & 3));
            }
        }
        // one launch of pass p: form `f` on `grid` workgroups, reading k[a] / v[a] and writing the other pair, with pass p's scan state
        auto launch = [&](BinLauncher f, uint32_t grid, uint32_t p, uint32_t a, uint32_t shift, uint32_t mode) {
            f(s, grid, k[a], k[a ^ 1u], v[a], v[a ^ 1u], h->slab + SLAB_DESC + (size_t)p * plan.desc_stride,
              h->slab + SLAB_COUNTERS + p * gs::COUNTERS_PER_PASS * gs::COUNTER_STRIDE, h->slab + SLAB_INFO + p * gs::INFO_STRIDE,
              h->slab + gs::SLAB_HSUB, h->slab + SLAB_STATUS, n, shift, mode);
        };
        const bool skip_local = (h->debug_flags & 0x40000000u) != 0u;  // (tuning builds, tools/hy_bringup.py: pass B's output stays as it is)
        for (uint32_t p = 0; p < NP; ++p) {
            const uint32_t a = dyn ? 0u : (p & 1u);
            const uint32_t mode = (dyn ? (desc_bit | gs::BM_PLANNED) : ((desc_bit && p == NP - 1) ? gs::BM_REVERSE : 0u)) | (p == 0 ? gs::BM_ZERO_HIST : 0u);
            if (pos && vb == 0) {
                // (1) keys-only sorts that may run on position chains: ONE launch per pass serves every plan (persistent workgroups, two per
                // CU).  Offered the two-level plan, the first two launches are pass A / pass B or LSD passes 0 / 1 — digit and chain count come
                // from the info block — the bucket-local sort follows them, and LSD passes 2 and 3 exit on PF_SKIP if it ran.
                launch(g_dual[p == 3][kt], pos_grid(), p, a, p * 8,
                       mode | ((hy && p < 2) ? gs::BM_INFO_SHIFT | gs::BM_INFO_CHAINS : 0u) | ((hy && p == 1) ? gs::BM_ZERO_DESC23 : 0u));
                if (hy && p == 1) {
                    if (h->profiling) GS_HIP(hipEventRecord(h->ev[5], s));  // slot 4 = pass B; slot 5: the local sort (+ LSD pass 2's launch); slot 6: LSD pass 3's
                    if (!skip_local) g_hy_local[hy_class(n)][kt](s, gs::HY_BINS, k[0], h->hy_tab, h->slab, n, desc_bit);
                }
            } else if (hy) {
                // (2) pairs that are offered the two-level plan: launches 0 and 1 = its two DigitBinningPasses (the plain form as persistent
                // workgroups: digit and chain count from the info block) or, on position chains, LSD passes 0 and 1 (the position-chain
                // form, which also serves LSD passes 2 and 3); the bucket-local sort sits between them.  The non-persistent plain forms
                // are not launched at all: whichever plan the device picks, one of these two forms is the one that works.
                if (p < 2) launch(g_persist[vb == 8][kt], pos_grid() / 2u, p, a, p * 8, mode | gs::BM_FORMS | gs::BM_INFO_SHIFT | gs::BM_INFO_CHAINS);
                launch(g_posv[vb == 8][p == 3][kt], pos_grid(), p, a, p * 8, (mode & ~gs::BM_ZERO_HIST) | gs::BM_FORMS | (p == 1 ? gs::BM_ZERO_DESC23 : 0u));
                if (p == 1) {
                    if (h->profiling) GS_HIP(hipEventRecord(h->ev[5], s));
                    if (!skip_local) g_hy_local_pairs[vb == 8][hy_class(n)][kt](s, k[0], v[0], h->hy_tab, h->slab, n, desc_bit);
                }
            } else {
                // (3) everything else: the plain form, one tile per workgroup.  8-byte values on the big tile come in two forms and the
                // pass's skew flag picks one (BinCfg::VROUNDS); pairs that may run on position chains add that form as a launch of its own.
                const bool two_forms = dyn && vb == 8 && !is_key64(kt) && sh.threads == 512 && sh.kpt == 32;
                launch(p == 0 ? fn0 : fn, p == 0 ? plan.grid0 : plan.grid, p, a, word * 32 + p * 8,
                       mode | (two_forms ? gs::BM_IF_EVEN : 0u) | ((pos && vb != 0) ? gs::BM_FORMS : 0u) | exp_mode);
                if (two_forms) launch(g_vr2[h->rank_mode][kt], plan.grid, p, a, word * 32 + p * 8, mode | gs::BM_IF_SKEW | ((pos && vb == 8) ? gs::BM_FORMS : 0u));
                if (pos && vb != 0) launch(g_posv[vb == 8][p == 3][kt], pos_grid(), p, a, p * 8, (mode & ~gs::BM_ZERO_HIST) | gs::BM_FORMS);
            }
            if (h->profiling && word == 0 && p < 4 && !(hy && p == 1)) GS_HIP(hipEventRecord(h->ev[4 + p], s));
        }
    }
    if (h->profiling && is_key64(kt)) GS_HIP(hipEventRecord(h->ev[7], s));  // slot 6 then holds pass 3 and everything behind it
    GS_HIP(hipGetLastError());
    h->hist_dirty = false;  // pass 0 (BM_ZERO_HIST) was launched: it zeroes HIST
    h->profile_pending = h->profiling != 0;
    return GS_OK;
}

}  // namespace

extern "C" gs_status gs_selftest_lds_atomic_order(uint32_t iters, uint32_t seed, uint64_t* h_failures, void* stream);
#ifdef GS_TUNING
extern "C" gs_status gs_debug_copy_floor(const void* d_in, void* d_out, uint32_t n, uint32_t threads, uint32_t kpt, void* stream);
#endif

namespace {
bool lds_atomic_order_ok() {
    static int cached[64];  // per device: 0 unknown, 1 ok, 2 failed
    static std::mutex guard;  // handles may be created from several host threads
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    std::lock_guard<std::mutex> lock(guard);
    if (cached[dev] == 0) {
        uint64_t fails = 1;
        const gs_status st = gs_selftest_lds_atomic_order(64, 0x9e3779b9u, &fails, nullptr);
        cached[dev] = (st == GS_OK && fails == 0) ? 1 : 2;
    }
    return cached[dev] == 1;
}
}  // namespace

extern "C" {

const char* gs_version(void) { return "gpusort-mi355x 0.1 (gfx950 OneSweep)"; }

const char* gs_status_string(gs_status s) {
    switch (s) {
        case GS_OK: return "ok";
        case GS_ERR_ARG: return "bad argument";
        case GS_ERR_SIZE: return "bad size";
        case GS_ERR_HIP: return "HIP runtime error";
        case GS_ERR_TIMEOUT: return "look-back timeout on device";
        case GS_ERR_MODE: return "mode / value width mismatch";
        case GS_ERR_NO_DEVICE: return "no GPU device";
        case GS_ERR_COMM: return "multi-GPU communication (RCCL) error";
    }
    return "unknown";
}

int gs_last_hip_error(void) { return g_last_hip_error; }

size_t gs_onesweep_temp_bytes(uint32_t max_keys) {
    // an upper bound over modes and options (default hist_blocks): slab + the histogram workgroups' slices + the two-level plan's tables
    const size_t slices = (size_t)hist_blocks_cap(max_keys) * gs::HIST_TABLE_WORDS, hy_slices = (size_t)hy_grid_for_device() * gs::HY_SLICE_WORDS;
    const bool hy = max_keys > (1u << 20);
    return (slab_words_for(max_keys) + (hy && hy_slices > slices ? hy_slices : slices) + (hy ? gs::HYT_WORDS : 0)) * sizeof(uint32_t);
}

uint32_t gs_onesweep_partition_size(gs_mode mode, uint32_t value_bytes) {
    const Shape& sh = g_shapes[(mode == GS_MODE_PAIRS && value_bytes == 4) ? 1 : 0];
    return (uint32_t)sh.threads * sh.kpt;
}

void gs_onesweep_options_default(gs_onesweep_options* o) {
    if (!o) return;
    memset(o, 0, sizeof(*o));
    o->struct_size = (uint32_t)sizeof(*o);
    o->rank_mode = -1;
    o->small_path = 1;
    o->mid_path = 1;
    o->skip_passes = 1;
    o->position_chains = 1;
    o->position_chains_min_log2 = 25;
    o->key64_sweeps = 1;
    o->plan = 0;
    o->first_pass_big = 1;
}

gs_status gs_onesweep_create(gs_onesweep** out, uint32_t max_keys, gs_mode mode, uint32_t value_bytes) {
    return gs_onesweep_create_ex(out, max_keys, mode, value_bytes, nullptr);
}

gs_status gs_onesweep_create_ex(gs_onesweep** out, uint32_t max_keys, gs_mode mode, uint32_t value_bytes, const gs_onesweep_options* options) {
    if (!out) return GS_ERR_ARG;
    *out = nullptr;
    gs_onesweep_options o;
    gs_onesweep_options_default(&o);
    if (options) {
        if (options->struct_size != sizeof(gs_onesweep_options)) return GS_ERR_ARG;  // (one layout so far)
        o = *options;
    }
    const int max_plan = 2;
    if (o.rank_mode < -1 || o.rank_mode > 1 || o.position_chains < 0 || o.position_chains > 2 || o.plan < 0 || o.plan > max_plan ||
        (o.key64_sweeps != 1 && o.key64_sweeps != 2) || o.position_chains_min_log2 < 20 || o.position_chains_min_log2 > 30)
        return GS_ERR_ARG;
    int shape_pick = -1;
    if (o.shape_threads || o.shape_keys_per_thread) {
        for (int i = 0; i < g_num_shapes; ++i)
            if ((uint32_t)g_shapes[i].threads == o.shape_threads && (uint32_t)g_shapes[i].kpt == o.shape_keys_per_thread) shape_pick = i;
        if (shape_pick < 0) return GS_ERR_ARG;
    }
    if (max_keys == 0 || max_keys > GS_MAX_KEYS) return GS_ERR_SIZE;
    if (mode == GS_MODE_KEYS_ONLY) {
        if (value_bytes != 0) return GS_ERR_MODE;
    } else if (mode == GS_MODE_PAIRS) {
        if (value_bytes != 4 && value_bytes != 8) return GS_ERR_MODE;
    } else {
        return GS_ERR_ARG;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return GS_ERR_NO_DEVICE;
    gs_onesweep* h = new (std::nothrow) gs_onesweep();
    if (!h) return GS_ERR_ARG;
    h->max_keys = max_keys;
    h->mode = mode;
    h->value_bytes = value_bytes;
    h->shape = (mode == GS_MODE_PAIRS && value_bytes == 4) ? 1 : 0;
    h->shape_auto = 1;
    if (shape_pick >= 0) { h->shape = shape_pick; h->shape_auto = 0; }
    h->small_path = o.small_path ? 1 : 0;
    h->pos_chains = o.position_chains;
    h->pos_min_keys = o.position_chains_min_log2 == 25 ? (1u << 25) + 1u : 1u << o.position_chains_min_log2;
    h->key64_sweeps = o.key64_sweeps;
    h->skip_passes = o.skip_passes ? 1 : 0;
    h->mid_path = o.mid_path ? 1 : 0;
    h->hist_blocks_opt = o.hist_blocks;
    h->first_pass_big = o.first_pass_big ? 1 : 0;
#if defined(GS_TUNING) || GS_EXP
    h->debug_flags = o.debug_flags;
#else
    h->debug_flags = 0;  // debug bits belong to tuning / experiment builds (bit 30 — skip the bucket-local sort, tools/hy_bringup.py — would hand back keys ordered on their top 16 bits only)
#endif
    h->profiling = 0;
    h->ev_valid = false;
    h->profile_pending = false;
    h->slab = nullptr;
    h->pinned = nullptr;
    h->trace_buf = nullptr;
    h->msd_keys = nullptr;
    h->last_n = h->last_tile = h->last_tile0 = h->last_p0 = h->last_np = h->last_dyn = h->last_desc_stride = h->last_pos_tile = 0;
    h->hist_dirty = false;
    h->plan = o.plan;
    h->hy_min_keys = HY_MIN_KEYS_DEFAULT;
    h->hy_tab = nullptr;
    h->hy_grid = hy_grid_for_device();
    h->last_hy = 0;
    h->exp_keep_desc = false;
    h->msd_n = h->msd_grid = 0;
    h->msd_kt = GS_KEY_UINT32;
    h->slab_words = slab_words_for(max_keys);
    // Tile ranking: the returning-LDS-atomic path needs same-address lanes of one
    // wave-instruction served in ascending lane order.  Probe the device once per
    // process; fall back to the ballot multi-split if a single lane disagrees.
    h->rank_mode = o.rank_mode >= 0 ? o.rank_mode : (lds_atomic_order_ok() ? 1 : 0);
    h->partials = nullptr;
    hipError_t e = hipMalloc(&h->slab, h->slab_words * sizeof(uint32_t));
    // the two-level plan's tables (0.8 MiB) and its histogram slices (hy_grid x 129 KiB: 33 MiB on 256 CUs): only for handles the default
    // routing can send there — keys-only and pairs handles that hold a sort of the plan's size — or that ask for plan 2 (tests, tools:
    // wherever the position-chain plan, its fall-back, runs: from 2^20 keys); gs_onesweep_set_plan(2) allocates them on demand otherwise
    const bool hy_handle = o.plan != 1 && (o.plan == 2 ? max_keys > (1u << 20) : max_keys >= (mode == GS_MODE_PAIRS ? HY_MIN_PAIRS_DEFAULT : HY_MIN_KEYS_DEFAULT));
    h->partials_words = (size_t)hist_blocks_cap(max_keys, o.hist_blocks) * gs::HIST_TABLE_WORDS;
    if (hy_handle && (size_t)h->hy_grid * gs::HY_SLICE_WORDS > h->partials_words) h->partials_words = (size_t)h->hy_grid * gs::HY_SLICE_WORDS;
    if (e == hipSuccess) e = hipMalloc(&h->partials, h->partials_words * sizeof(uint32_t));
    if (e == hipSuccess && hy_handle) e = hipMalloc(&h->hy_tab, gs::HYT_WORDS * sizeof(uint32_t));
    // counters/status/info start defined: gs_onesweep_check() may run before any tiled sort (single-tile path)
    if (e == hipSuccess) e = hipMemset(h->slab, 0, SLAB_DESC * sizeof(uint32_t));
    if (e == hipSuccess) e = hipHostMalloc(&h->pinned, (4 * gs::NCH * gs::RADIX + 8) * sizeof(uint32_t), hipHostMallocDefault);
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        if (h->slab) (void)hipFree(h->slab);
        if (h->partials) (void)hipFree(h->partials);
        if (h->hy_tab) (void)hipFree(h->hy_tab);
        delete h;
        return GS_ERR_HIP;
    }
    *out = h;
    return GS_OK;
}

gs_status gs_onesweep_destroy(gs_onesweep* h) {
    if (!h) return GS_ERR_ARG;
    if (h->ev_valid)
        for (auto& e : h->ev) (void)hipEventDestroy(e);
    if (h->pinned) (void)hipHostFree(h->pinned);
    if (h->slab) (void)hipFree(h->slab);
    if (h->partials) (void)hipFree(h->partials);
    if (h->hy_tab) (void)hipFree(h->hy_tab);
    delete h;
    return GS_OK;
}

gs_status gs_debug_read_slab(gs_onesweep* h, uint32_t first_word, uint32_t count, uint32_t* h_out, void* stream) {
    // bit 31 of first_word: the workgroups' table slices (hist partials) instead of the slab
    const bool part = (first_word >> 31) != 0u;
    first_word &= 0x7fffffffu;
    const size_t limit = part ? (h ? h->partials_words : 0) : (h ? h->slab_words : 0);
    if (!h || !h_out || (size_t)first_word + count > limit) return GS_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    GS_HIP(hipMemcpyAsync(h_out, (part ? h->partials : h->slab) + first_word, (size_t)count * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    GS_HIP(hipStreamSynchronize(s));
    return GS_OK;
}

gs_status gs_onesweep_set_plan(gs_onesweep* h, int plan) {
    if (!h || plan < 0) return GS_ERR_ARG;
    if (plan > 2) return GS_ERR_ARG;
    if (plan == 2 && !h->hy_tab) {
        // the handle was created without the plan's tables (below the plan's default size, or plan 1): allocate them now.  The caller
        // must not have a sort of this handle in flight (as for every setter): the slices are re-allocated.
        if (h->max_keys <= (1u << 20)) return GS_ERR_MODE;  // (the plan's fall-back, the position-chain plan, starts above 2^20 keys)
        const size_t need = (size_t)h->hy_grid * gs::HY_SLICE_WORDS;
        if (need > h->partials_words) {
            uint32_t* p = nullptr;
            GS_HIP(hipDeviceSynchronize());
            GS_HIP(hipMalloc(&p, need * sizeof(uint32_t)));
            (void)hipFree(h->partials);
            h->partials = p;
            h->partials_words = need;
        }
        GS_HIP(hipMalloc(&h->hy_tab, gs::HYT_WORDS * sizeof(uint32_t)));
    }
    h->plan = plan;
    return GS_OK;
}

gs_status gs_onesweep_last_plan(gs_onesweep* h, uint32_t* plan, uint32_t* largest_bucket, void* stream) {
    if (!h || !plan) return GS_ERR_ARG;
    *plan = 0;
    if (largest_bucket) *largest_bucket = 0;
    if (!h->last_hy) return GS_OK;  // the last sort was not offered the two-level plan (size, mode, options): LSD passes, or a one- / two-launch route
    hipStream_t s = static_cast<hipStream_t>(stream);
    GS_HIP(hipMemcpyAsync(h->pinned, h->slab + gs::SLAB_HY, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    GS_HIP(hipStreamSynchronize(s));
    *plan = h->pinned[gs::HY_VALID] ? 1u : 0u;
    if (largest_bucket) *largest_bucket = h->pinned[gs::HY_MAXBUCKET];
    return GS_OK;
}

gs_status gs_onesweep_set_shape(gs_onesweep* h, uint32_t threads, uint32_t keys_per_thread) {
    if (!h) return GS_ERR_ARG;
    for (int i = 0; i < g_num_shapes; ++i)
        if ((uint32_t)g_shapes[i].threads == threads && (uint32_t)g_shapes[i].kpt == keys_per_thread) {
            h->shape = i;
            h->shape_auto = 0;
            return GS_OK;
        }
    return GS_ERR_ARG;
}

gs_status gs_debug_set_trace(gs_onesweep* h, void* d_buf) {  // experiment builds: 4 passes x grid x 8 words
    if (!h) return GS_ERR_ARG;
    h->trace_buf = d_buf;
    return GS_OK;
}

gs_status gs_onesweep_set_small_path(gs_onesweep* h, int on) {
    if (!h) return GS_ERR_ARG;
    h->small_path = on ? 1 : 0;
    return GS_OK;
}

gs_status gs_onesweep_set_mid_path(gs_onesweep* h, int on) {
    if (!h) return GS_ERR_ARG;
    h->mid_path = on ? 1 : 0;
    return GS_OK;
}

gs_status gs_onesweep_set_skip_passes(gs_onesweep* h, int on) {
    if (!h) return GS_ERR_ARG;
    h->skip_passes = on ? 1 : 0;
    return GS_OK;
}

gs_status gs_onesweep_set_rank_mode(gs_onesweep* h, int mode) {
    if (!h || mode < 0 || mode > 1) return GS_ERR_ARG;
    h->rank_mode = mode;
    return GS_OK;
}

int gs_onesweep_get_rank_mode(gs_onesweep* h) { return h ? h->rank_mode : -1; }

gs_status gs_selftest_lds_atomic_order(uint32_t iters, uint32_t seed, uint64_t* h_failures, void* stream) {
    if (!h_failures || iters == 0) return GS_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    uint32_t* d = nullptr;
    GS_HIP(hipMalloc(&d, sizeof(uint32_t)));
    gs_status ret = GS_OK;
    uint32_t h32 = 0;
    if (hipMemsetAsync(d, 0, sizeof(uint32_t), s) != hipSuccess) ret = GS_ERR_HIP;
    if (ret == GS_OK) {
        hipLaunchKernelGGL(gs::lds_atomic_order_probe, dim3(256 * 4), dim3(512), 0, s, seed, iters, d);
        if (hipMemcpyAsync(&h32, d, sizeof(uint32_t), hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess)
            ret = GS_ERR_HIP;
    }
    (void)hipFree(d);
    *h_failures = h32;
    return ret;
}

gs_status gs_selftest_wave_primitives(uint32_t seed, uint32_t waves, uint32_t* d_out, void* stream) {
    if (!d_out || waves == 0 || (waves & 3u) != 0u || waves > (1u << 20)) return GS_ERR_ARG;
    hipLaunchKernelGGL(gs::wave_primitives_kernel, dim3(waves / 4u), dim3(256), 0, static_cast<hipStream_t>(stream), seed, d_out);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

#ifdef GS_TUNING
// Tuning aid: global access pattern of a DigitBinningPass without ranking or look-back (memory floor of the tile shape);
// threads == 0: plain streaming copies (kpt 0 / 1 / 2 = default / nt loads / nt loads and stores) and a read-only sweep (kpt 3).
gs_status gs_debug_copy_floor(const void* d_in, void* d_out, uint32_t n, uint32_t threads, uint32_t kpt, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint32_t tiles = threads ? n / (threads * (kpt ? kpt : 1u)) : 1u;
    if (!tiles) return GS_ERR_SIZE;
    const uint32_t* in = static_cast<const uint32_t*>(d_in);
    uint32_t* out = static_cast<uint32_t*>(d_out);
    if (threads == 512 && kpt == 32) hipLaunchKernelGGL((gs::copy_floor_kernel<512, 32>), dim3(tiles), dim3(512), 0, s, in, out, n);
    else if (threads == 512 && kpt == 16) hipLaunchKernelGGL((gs::copy_floor_kernel<512, 16>), dim3(tiles), dim3(512), 0, s, in, out, n);
    else if (threads == 1024 && kpt == 16) hipLaunchKernelGGL((gs::copy_floor_kernel<1024, 16>), dim3(tiles), dim3(1024), 0, s, in, out, n);
    else if (threads == 256 && kpt == 32) hipLaunchKernelGGL((gs::copy_floor_kernel<256, 32>), dim3(tiles), dim3(256), 0, s, in, out, n);
    else if (threads == 0) {  // calibration copies: kpt = 0/1/2 copy policy, 3 = read-only sweep
        const uint32_t nvec = n / 4, grid = 256 * 8;
        const gs::u32x4* vi = static_cast<const gs::u32x4*>(d_in);
        gs::u32x4* vo = static_cast<gs::u32x4*>(d_out);
        if (kpt == 0) hipLaunchKernelGGL(gs::copy_x4_kernel<0>, dim3(grid), dim3(256), 0, s, vi, vo, nvec);
        else if (kpt == 1) hipLaunchKernelGGL(gs::copy_x4_kernel<1>, dim3(grid), dim3(256), 0, s, vi, vo, nvec);
        else if (kpt == 2) hipLaunchKernelGGL(gs::copy_x4_kernel<2>, dim3(grid), dim3(256), 0, s, vi, vo, nvec);
        else if (kpt == 3) hipLaunchKernelGGL(gs::read_x4_kernel, dim3(grid), dim3(256), 0, s, vi, out, nvec);
        // round 4: the best shapes tools/r04_probe.hip found — 4: read, four nt loads in flight, two workgroups per CU; 5: copy, four nt
        // loads in flight, one-shot grid of n / 8192 workgroups; 6: copy, four plain loads in flight, one workgroup per CU; 7: hipMemcpyAsync
        else if (kpt == 4) hipLaunchKernelGGL((gs::read_xu_kernel<4, true>), dim3(256 * 2), dim3(256), 0, s, vi, out, (size_t)nvec);
        else if (kpt == 5) hipLaunchKernelGGL((gs::copy_xu_kernel<4, true>), dim3(nvec / (256 * 4) / 2 ? nvec / (256 * 4) / 2 : 1), dim3(256), 0, s, vi, vo, (size_t)nvec);
        else if (kpt == 6) hipLaunchKernelGGL((gs::copy_xu_kernel<4, false>), dim3(256), dim3(256), 0, s, vi, vo, (size_t)nvec);
        else if (kpt == 7) GS_HIP(hipMemcpyAsync(d_out, d_in, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
        else return GS_ERR_ARG;
    } else return GS_ERR_ARG;
    GS_HIP(hipGetLastError());
    return GS_OK;
}
#endif

uint32_t gs_onesweep_get_partition_size(gs_onesweep* h) {
    return h ? (uint32_t)g_shapes[h->shape].threads * g_shapes[h->shape].kpt : 0;
}

gs_status gs_onesweep_sort_keys(gs_onesweep* h, void* d_keys, void* d_alt, uint32_t n, gs_key_type kt, gs_order order,
                                void* stream) {
    gs_status st = check_common(h, d_keys, d_alt, n, kt, order);
    if (st != GS_OK) return st;
    return sort_impl(h, d_keys, nullptr, d_alt, nullptr, n, kt, order, static_cast<hipStream_t>(stream), 0);
}

gs_status gs_onesweep_sort_pairs(gs_onesweep* h, void* d_keys, void* d_vals, void* d_alt_keys, void* d_alt_vals,
                                 uint32_t n, gs_key_type kt, gs_order order, void* stream) {
    gs_status st = check_common(h, d_keys, d_alt_keys, n, kt, order);
    if (st != GS_OK) return st;
    if (h->mode != GS_MODE_PAIRS) return GS_ERR_MODE;
    if (!d_vals || !d_alt_vals || misaligned(d_vals) || misaligned(d_alt_vals)) return GS_ERR_ARG;
    return sort_impl(h, d_keys, d_vals, d_alt_keys, d_alt_vals, n, kt, order, static_cast<hipStream_t>(stream),
                     h->value_bytes);
}

gs_status gs_onesweep_check(gs_onesweep* h, void* stream) {
    if (!h) return GS_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    GS_HIP(hipMemcpyAsync(h->pinned, h->slab + SLAB_STATUS, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    GS_HIP(hipStreamSynchronize(s));
    return h->pinned[0] == gs::STATUS_OK ? GS_OK : GS_ERR_TIMEOUT;
}

#if (GS_EXP & 1024)
gs_status gs_debug_read_status_words(gs_onesweep* h, uint32_t out[32], void* stream) {  // experiment builds only
    if (!h || !out) return GS_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    GS_HIP(hipMemcpyAsync(h->pinned, h->slab + SLAB_STATUS, 32 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    GS_HIP(hipStreamSynchronize(s));
    memcpy(out, h->pinned, 32 * sizeof(uint32_t));
    return GS_OK;
}
#endif

gs_status gs_debug_poke_status(gs_onesweep* h, uint32_t word, void* stream) {  // tests: forge the device status word
    if (!h) return GS_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    h->pinned[0] = word;
    GS_HIP(hipMemcpyAsync(h->slab + SLAB_STATUS, h->pinned, sizeof(uint32_t), hipMemcpyHostToDevice, s));
    GS_HIP(hipStreamSynchronize(s));
    return GS_OK;
}

gs_status gs_debug_check_state(gs_onesweep* h, uint64_t report[8], void* stream) {
    if (!h || !report) return GS_ERR_ARG;
    for (int i = 0; i < 8; ++i) report[i] = 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (h->last_tile == 0) return GS_OK;  // single-tile sort or nothing yet: there is no scan state to check
    unsigned long long* d = nullptr;
    GS_HIP(hipMalloc(&d, 8 * sizeof(unsigned long long)));
    gs_status ret = GS_OK;
    if (hipMemsetAsync(d, 0, 8 * sizeof(unsigned long long), s) != hipSuccess) ret = GS_ERR_HIP;
    if (ret == GS_OK) {
        hipLaunchKernelGGL(gs::check_state_kernel, dim3(h->last_hy ? gs::CHMAX : gs::MAXCH, h->last_np), dim3(256), 0, s, h->slab, h->last_desc_stride,
                           h->last_tile, 0u, h->last_dyn, d, h->last_pos_tile, h->last_tile0);
        if (hipMemcpyAsync(report, d, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess)
            ret = GS_ERR_HIP;
    }
    (void)hipFree(d);
    return ret;
}

gs_status gs_onesweep_global_histogram(gs_onesweep* h, const void* d_keys, uint32_t n, gs_key_type kt, uint32_t* h_hist,
                                       void* stream) {
    if (!h || !d_keys || !h_hist || misaligned(d_keys) || (int)kt < 0 || (int)kt > 2) return GS_ERR_ARG;
    if (n == 0 || n > h->max_keys) return GS_ERR_SIZE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    PassPlan plan;
    gs_status st = prologue(h, d_keys, n, kt, s, 0, 4, &plan);
    if (st != GS_OK) return st;
    const size_t words = 4 * (size_t)gs::NCH * gs::RADIX;
    GS_HIP(hipMemcpyAsync(h->pinned, h->slab + SLAB_HIST, words * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    GS_HIP(zero_hist(h, s));  // no pass follows: hand HIST back zeroed
    h->hist_dirty = false;
    GS_HIP(hipStreamSynchronize(s));
    for (uint32_t q = 0; q < 4; ++q)  // digit totals = joint histogram summed over chains
        for (uint32_t d = 0; d < gs::RADIX; ++d) {
            uint32_t g = 0;
            for (uint32_t x = 0; x < gs::NCH; ++x) g += h->pinned[gs::hist_index(q, d, x)];
            h_hist[q * gs::RADIX + d] = g;
        }
    return GS_OK;
}

gs_status gs_onesweep_scan(gs_onesweep* h, const void* d_keys, uint32_t n, gs_key_type kt, uint32_t* h_rows, void* stream) {
    if (!h || !d_keys || !h_rows || misaligned(d_keys) || (int)kt < 0 || (int)kt > 2) return GS_ERR_ARG;
    if (n == 0 || n > h->max_keys) return GS_ERR_SIZE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    PassPlan plan;
    gs_status st = prologue(h, d_keys, n, kt, s, 0, 4, &plan);
    if (st != GS_OK) return st;
    // chain 0 of every pass starts at row 0 of the pass's descriptor region: its seed row holds the digit starts themselves
    for (uint32_t q = 0; q < 4; ++q)
        GS_HIP(hipMemcpyAsync(h->pinned + q * gs::RADIX, h->slab + SLAB_DESC + (size_t)q * plan.desc_stride, gs::RADIX * sizeof(uint32_t),
                              hipMemcpyDeviceToHost, s));
    GS_HIP(zero_hist(h, s));  // no pass follows: hand HIST back zeroed
    h->hist_dirty = false;
    GS_HIP(hipStreamSynchronize(s));
    memcpy(h_rows, h->pinned, 4 * gs::RADIX * sizeof(uint32_t));
    return GS_OK;
}

gs_status gs_onesweep_digit_pass(gs_onesweep* h, const void* d_keys_in, void* d_keys_out, const void* d_vals_in,
                                 void* d_vals_out, uint32_t n, uint32_t pass, gs_key_type kt, int reverse_index,
                                 void* stream) {
    gs_status st = check_common(h, d_keys_in, d_keys_out, n, kt, GS_ORDER_ASCENDING);
    if (st != GS_OK) return st;
    if (pass > (is_key64(kt) ? 7u : 3u)) return GS_ERR_ARG;
    uint32_t vb = 0;
    if (d_vals_in || d_vals_out) {
        if (h->mode != GS_MODE_PAIRS) return GS_ERR_MODE;
        if (!d_vals_in || !d_vals_out || misaligned(d_vals_in) || misaligned(d_vals_out)) return GS_ERR_ARG;
        vb = h->value_bytes;
    }
    int shape = h->shape;
    if (is_key64(kt) && !g_shapes[shape].fn[h->rank_mode][vb_index(vb)][kt]) shape = MID_SHAPE;
    const Shape& sh = g_shapes[shape];
    BinLauncher fn = sh.fn[h->rank_mode][vb_index(vb)][kt];
    if (!fn) return GS_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    PassPlan plan;
    st = prologue(h, d_keys_in, n, kt, s, pass & 3u, 1, &plan, 0, shape, pass >> 2);  // a stand-alone pass: position segments on ANY input
    if (st != GS_OK) return st;
    fn(s, plan.grid, const_cast<uint32_t*>(static_cast<const uint32_t*>(d_keys_in)), static_cast<uint32_t*>(d_keys_out),
       const_cast<void*>(d_vals_in), d_vals_out,
       h->slab + SLAB_DESC, h->slab + SLAB_COUNTERS, h->slab + SLAB_INFO, h->slab + gs::SLAB_HSUB, h->slab + SLAB_STATUS, n, pass * 8,
       (reverse_index ? gs::BM_REVERSE : 0u) | gs::BM_ZERO_HIST);
    GS_HIP(hipGetLastError());
    h->hist_dirty = false;
    if (h->profiling)  // slot 3 = this pass, slots 4..6 = 0
        for (int e = 4; e <= 7; ++e) GS_HIP(hipEventRecord(h->ev[e], s));
    h->profile_pending = h->profiling != 0;
    return GS_OK;
}

// ---- multi-GPU MSD split in two steps that share ONE histogram + scan of the shard -------------
gs_status gs_onesweep_msd_prepare(gs_onesweep* h, const void* d_keys, uint32_t n, gs_key_type kt, uint32_t* h_hist256,
                                  void* stream) {
    if (!h || !d_keys || !h_hist256 || misaligned(d_keys) || (int)kt < 0 || (int)kt > 2) return GS_ERR_ARG;
    if (n == 0 || n > h->max_keys || n > GS_MAX_KEYS) return GS_ERR_SIZE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    PassPlan plan;
    gs_status st = prologue(h, d_keys, n, kt, s, 3, 1, &plan);  // top byte, position chains
    if (st != GS_OK) return st;
    const size_t words = (size_t)gs::NCH * gs::RADIX;
    GS_HIP(hipMemcpyAsync(h->pinned, h->slab + SLAB_HIST, words * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    GS_HIP(zero_hist(h, s));  // msd_partition may never be called
    h->hist_dirty = false;
    GS_HIP(hipStreamSynchronize(s));
    for (uint32_t d = 0; d < gs::RADIX; ++d) {
        uint32_t g = 0;
        for (uint32_t x = 0; x < gs::NCH; ++x) g += h->pinned[gs::hist_index(0, d, x)];
        h_hist256[d] = g;
    }
    h->msd_keys = d_keys;
    h->msd_n = n;
    h->msd_kt = kt;
    h->msd_grid = plan.grid;
    return GS_OK;
}

gs_status gs_onesweep_msd_partition(gs_onesweep* h, const void* d_keys_in, void* d_keys_out, const void* d_vals_in,
                                    void* d_vals_out, uint32_t n, void* stream) {
    if (!h || h->msd_keys == nullptr || h->msd_keys != d_keys_in || h->msd_n != n) return GS_ERR_ARG;  // needs its prepare
    gs_status st = check_common(h, d_keys_in, d_keys_out, n, h->msd_kt, GS_ORDER_ASCENDING);
    if (st != GS_OK) return st;
    uint32_t vb = 0;
    if (d_vals_in || d_vals_out) {
        if (h->mode != GS_MODE_PAIRS) return GS_ERR_MODE;
        if (!d_vals_in || !d_vals_out || misaligned(d_vals_in) || misaligned(d_vals_out)) return GS_ERR_ARG;
        vb = h->value_bytes;
    }
    BinLauncher fn = g_shapes[h->shape].fn[h->rank_mode][vb_index(vb)][h->msd_kt];
    if (!fn) return GS_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    fn(s, h->msd_grid, const_cast<uint32_t*>(static_cast<const uint32_t*>(d_keys_in)), static_cast<uint32_t*>(d_keys_out),
       const_cast<void*>(d_vals_in), d_vals_out,
       h->slab + SLAB_DESC, h->slab + SLAB_COUNTERS, h->slab + SLAB_INFO, h->slab + gs::SLAB_HSUB, h->slab + SLAB_STATUS, n, 24, gs::BM_ZERO_HIST);
    GS_HIP(hipGetLastError());
    h->msd_keys = nullptr;  // the scan state is consumed
    h->profile_pending = false;
    return GS_OK;
}

gs_status gs_onesweep_set_profiling(gs_onesweep* h, int enabled) {
    if (!h) return GS_ERR_ARG;
    if (enabled && !h->ev_valid) {
        for (auto& e : h->ev) GS_HIP(hipEventCreate(&e));
        h->ev_valid = true;
    }
    h->profiling = enabled ? 1 : 0;
    h->profile_pending = false;
    return GS_OK;
}

gs_status gs_onesweep_get_profile(gs_onesweep* h, float ms[GS_PROFILE_SLOTS]) {
    if (!h || !ms) return GS_ERR_ARG;
    if (!h->profile_pending) return GS_ERR_ARG;
    GS_HIP(hipEventSynchronize(h->ev[7]));
    for (int i = 0; i < 7; ++i) GS_HIP(hipEventElapsedTime(&ms[i], h->ev[i], h->ev[i + 1]));
    GS_HIP(hipEventElapsedTime(&ms[7], h->ev[0], h->ev[7]));
    return GS_OK;
}

gs_status gs_init_random(void* d_keys, void* d_vals, uint32_t value_bytes, uint32_t and_count, uint32_t seed, uint32_t n,
                         void* stream) {
    if (!d_keys || n == 0) return n == 0 ? GS_ERR_SIZE : GS_ERR_ARG;
    if (and_count > 31) return GS_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    uint32_t* k = static_cast<uint32_t*>(d_keys);
    if (!d_vals || value_bytes == 0)
        hipLaunchKernelGGL(gs::init_random_kernel<0>, dim3(256), dim3(256), 0, s, k, nullptr, and_count, seed, n);
    else if (value_bytes == 4)
        hipLaunchKernelGGL(gs::init_random_kernel<4>, dim3(256), dim3(256), 0, s, k, d_vals, and_count, seed, n);
    else if (value_bytes == 8)
        hipLaunchKernelGGL(gs::init_random_kernel<8>, dim3(256), dim3(256), 0, s, k, d_vals, and_count, seed, n);
    else
        return GS_ERR_MODE;
    GS_HIP(hipGetLastError());
    return GS_OK;
}

gs_status gs_validate(const void* d_keys, const void* d_vals, uint32_t value_bytes, uint32_t n, gs_key_type kt,
                      gs_order order, uint32_t* h_err_count, void* stream) {
    if (!d_keys || !h_err_count || (int)kt < 0 || (int)kt > 5) return GS_ERR_ARG;
    if (n == 0) return GS_ERR_SIZE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    uint32_t* d_err = nullptr;
    GS_HIP(hipMalloc(&d_err, sizeof(uint32_t)));
    gs_status ret = GS_OK;
    do {
        if (hipMemsetAsync(d_err, 0, sizeof(uint32_t), s) != hipSuccess) { ret = GS_ERR_HIP; break; }
        const uint32_t blocks = div_up(n, 256 * 16) < 2048 ? div_up(n, 256 * 16) : 2048;
        const uint32_t* k = static_cast<const uint32_t*>(d_keys);
        const int desc = order == GS_ORDER_DESCENDING;
        if (is_key64(kt))  // 64-bit keys: the keys' order only
            hipLaunchKernelGGL(gs::validate64_kernel, dim3(blocks), dim3(256), 0, s, static_cast<const uint2*>(d_keys), n, (int)kt, desc, d_err);
        else if (!d_vals || value_bytes == 0)
            hipLaunchKernelGGL(gs::validate_kernel<0>, dim3(blocks), dim3(256), 0, s, k, nullptr, n, (int)kt, desc, d_err);
        else if (value_bytes == 4)
            hipLaunchKernelGGL(gs::validate_kernel<4>, dim3(blocks), dim3(256), 0, s, k, d_vals, n, (int)kt, desc, d_err);
        else if (value_bytes == 8)
            hipLaunchKernelGGL(gs::validate_kernel<8>, dim3(blocks), dim3(256), 0, s, k, d_vals, n, (int)kt, desc, d_err);
        else { ret = GS_ERR_MODE; break; }
        if (hipMemcpyAsync(h_err_count, d_err, sizeof(uint32_t), hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess)
            ret = GS_ERR_HIP;
    } while (0);
    (void)hipFree(d_err);
    return ret;
}

gs_status gs_msd_splitters_n(const uint64_t* hist, uint32_t nbins, uint32_t world, uint32_t* first_bin) {
    if (!hist || !first_bin || world == 0 || nbins == 0 || world > nbins) return GS_ERR_ARG;
    uint64_t total = 0;
    for (uint32_t b = 0; b < nbins; ++b) total += hist[b];
    // Rank r starts at the first bin whose exclusive prefix reaches ceil(r * total / world):
    // equal-count buckets at bin granularity.
    first_bin[0] = 0;
    uint64_t excl = 0;
    uint32_t b = 0;
    for (uint32_t r = 1; r < world; ++r) {
        const uint64_t target = (total * r + world - 1) / world;
        while (b < nbins && excl < target) excl += hist[b++];
        first_bin[r] = b;
    }
    first_bin[world] = nbins;
    return GS_OK;
}

gs_status gs_msd_splitters(const uint64_t hist256[256], uint32_t world, uint32_t* first_bin) {
    if (world > 256) return GS_ERR_ARG;
    return gs_msd_splitters_n(hist256, 256, world, first_bin);
}

// 12-bit prefix histogram of a shard: (top byte, top nibble of the byte below) = the joint histogram the sort's
// own GlobalHistogram kernel counts for the last pass (chain = group of the previous digit), bin = d3*16 + (d2>>4).
gs_status gs_onesweep_msd_fine_histogram(gs_onesweep* h, const void* d_keys, uint32_t n, gs_key_type kt,
                                         uint32_t* h_hist4096, void* stream) {
    if (gs::NCH != 16) return GS_ERR_ARG;  // the fine MSD histogram is the 16-chain joint histogram (tuning builds with other chain counts)
    if (!h || !d_keys || !h_hist4096 || misaligned(d_keys) || (int)kt < 0 || (int)kt > 2) return GS_ERR_ARG;
    if (n == 0 || n > h->max_keys || n > GS_MAX_KEYS) return GS_ERR_SIZE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    PassPlan plan;
    gs_status st = prologue(h, d_keys, n, kt, s, 2, 2, &plan);  // bytes 2 and 3: row 1 = H(d3, group of d2)
    if (st != GS_OK) return st;
    const size_t words = 2 * (size_t)gs::NCH * gs::RADIX;
    GS_HIP(hipMemcpyAsync(h->pinned, h->slab + SLAB_HIST, words * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    GS_HIP(zero_hist(h, s));  // no pass follows: hand HIST back zeroed
    h->hist_dirty = false;
    if (h->profiling)  // slots 0..2 (clear, histogram, scan) are this call's; the pass slots read 0
        for (int e = 4; e <= 7; ++e) GS_HIP(hipEventRecord(h->ev[e], s));
    h->profile_pending = h->profiling != 0;
    GS_HIP(hipStreamSynchronize(s));
    for (uint32_t d = 0; d < gs::RADIX; ++d)
        for (uint32_t x = 0; x < gs::NCH; ++x) h_hist4096[d * gs::NCH + x] = h->pinned[gs::hist_index(1, d, x)];
    return GS_OK;
}

}  // extern "C"

#include "gpusort_mgpu.hpp"
