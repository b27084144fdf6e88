// onesweep_kernels.hpp — gfx950 (CDNA4, wave64) device code of the OneSweep
// 8-bit LSD radix sort.  Written for MI355X only.
//
// Behavioural spec (what, not how): reference b0nes164/GPUSorting
//   GlobalHistogram      GPUSortingCUDA/Sort/OneSweep.cu:44-123
//   Scan                 GPUSortingCUDA/Sort/OneSweep.cu:125-162
//   DigitBinningPass*    GPUSortingCUDA/Sort/OneSweep.cu:164-344, 346-600
//   key transforms       GPUSortingD3D12/Shaders/SortCommon.hlsl:134-154
//   descending rule      GPUSortingD3D12/Shaders/SortCommon.hlsl:594-597,645-656
//   InitRandom/Validate  GPUSortingCUDA/UtilityKernels.cuh:53-117, 402-479
//
// MI355X-first design decisions (measurements: profiles/r01_*.txt):
//   * MULTI-CHAIN chained scan.  256 CUs start ~40 tiles per microsecond while one
//     dependent descriptor read costs ~1.7 us under streaming load; a single
//     chain then makes every tile walk ~30 rows (the walk obeys W = lambda*L^2/2b).
//     Each pass is therefore split into GS_NCHAINS independent chains.  Pass p>=1
//     reads an array that is already partitioned by digit p-1, so chain x = a group
//     of digit-(p-1) values and its digit-p counts are a JOINT histogram the
//     upfront GlobalHistogram kernel counts in the same single read; the first
//     pass uses position segments.  Chain bases are therefore known before the
//     pass starts and the per-chain arrival rate drops by GS_NCHAINS.
//   * ranking inside a tile: one returning LDS atomic per key on a wave-private
//     counter (the LDS serves same-address lanes of one instruction in lane
//     order — probed on the device before use), with the 64-lane ballot
//     multi-split as the guaranteed fallback.
//   * descriptors are single dwords {count:30, flag:2} accessed with relaxed
//     agent-scope atomics (sc1): the data IS the flag, no fences.
//   * everything is decided on the device, nothing needs a host round trip: the Scan kernel plans the passes
//     (identity passes are dropped in pairs, a skewed pass ranks with wave-aggregated adds, a skewed SORT runs every pass on
//     position chains whose bases the pass before counts while it scatters), and a look-back that waits too long
//     recounts the missing tile itself, so no workgroup depends on another's progress for more than a bounded time.
//   * a sort on the LSD plan is 7 launches: GlobalHistogram (which also clears the scan state), the sum of its workgroups' tables,
//     Scan, 4 x DigitBinningPass.  Large sorts are offered the TWO-LEVEL plan (hybrid_kernels.hpp: the same DigitBinningPass on bytes
//     3 and 2 — told its digit and chain count by the info block — then bucket-local LDS sorts of the low 16 bits: 28 B/key instead of
//     36), chosen per sort on the device.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gs {

constexpr uint32_t RADIX = 256;
constexpr uint32_t FLAG_NOT_READY = 0;  // tile has published nothing yet
constexpr uint32_t FLAG_REDUCTION = 1;  // count<<2 = this tile's digit count
constexpr uint32_t FLAG_INCLUSIVE = 2;  // count<<2 = count of this and all earlier tiles of the chain (+ chain base)
constexpr uint32_t FLAG_POISON = 3;     // the tile gave up its look-back (bounded spin expired): successors give up too
constexpr uint32_t FLAG_MASK = 3;

constexpr uint32_t STATUS_OK = 0;
constexpr uint32_t STATUS_TIMEOUT = 4;  // == GS_ERR_TIMEOUT

// Bound for every look-back spin (polls, each >= ~0.5 us with the sleep): ~1 s.
#ifndef GS_SPIN_LIMIT
#define GS_SPIN_LIMIT (1u << 21)
#endif
constexpr uint32_t SPIN_LIMIT = GS_SPIN_LIMIT;

#ifndef GS_NCHAINS
#define GS_NCHAINS 16  // independent chained scans per pass (power of two, <= 32)
#endif
constexpr uint32_t NCH = GS_NCHAINS;
static_assert(NCH >= 1 && NCH <= 32 && (NCH & (NCH - 1)) == 0, "GS_NCHAINS must be a power of two <= 32");

// Ablation / instrumented / fault-injection builds (-DGS_EXP=<flags>, tools/ and tests/test_gpu_fault.py) hook into
// the kernels through the GS_ABL_* / GS_TRACE* macros below; their code lives in onesweep_ablation.hpp and is not
// part of the product translation unit (GS_EXP == 0: every hook is empty or a compile-time false).
#ifndef GS_EXP
#define GS_EXP 0
#endif
#if GS_EXP
#include "onesweep_ablation.hpp"
#else
#define GS_FAULT_TILE(chain, tile) false       // fault injection: this tile never publishes its descriptor
#define GS_TRACE_SETUP() do { } while (0)      // per-tile phase timestamps
#define GS_TRACE(slot) do { } while (0)
#define GS_TRACE_TRIP() do { } while (0)
#define GS_TRACE_END(chain) do { } while (0)
#define GS_ABL_HIST_STREAM_ONLY(t) do { } while (0)   // histogram kernel: stream the keys, count nothing
#define GS_ABL_LOOKBACK_SKIPPED false          // no look-back wait
#define GS_ABL_ASSUME_PREV() do { } while (0)  // no wait, positions extrapolated from this tile's own counts
#define GS_ABL_GENERIC_SCATTER false           // force the generic (masked) scatter loops
#define GS_ABL_OUT_INDEX(o, i) do { } while (0)  // rewrite an output index (sequential / wrapped)
#define GS_ABL_REPLAY false                    // descriptors of an identical earlier run are still there: no REDUCTION publish
#define GS_ABL_EARLY_ROW(v) do { } while (0)   // request the predecessor's row before the key loads
#define GS_ABL_EARLY_USE(v) do { } while (0)   // ... and take it in the look-back if it is INCLUSIVE
#define GS_ABL_CLOCKS_BEGIN() do { } while (0)  // histogram kernel: shader clock against the 100 MHz wall clock
#define GS_ABL_CLOCKS_END() do { } while (0)
#define GS_HIST_STAMP(i) do { } while (0)       // histogram kernel: phase time stamps of its first and last workgroup
#define GS_HIST_STAMPS_DECL() do { } while (0)
#define GS_HIST_STAMPS_OUT() do { } while (0)
#define GS_ABL_COUNT_LDS 0                     // extra LDS of the next-digit counting experiment
#define GS_ABL_COUNT_NEXT(kb, o) do { } while (0)
#endif
// (measured and dropped: issuing the first look-back read before the staging phase, -4 %: the early read mostly
//  returns a not-yet-final row and the wait moves in front of staging)
#ifndef GS_FALLBACK
#define GS_FALLBACK 1  // a look-back that waited FALLBACK_SPINS polls on one row recounts that tile's digits itself
                       // (whole workgroup, from the pass input) and goes on: no tile ever depends on another
                       // workgroup's progress for more than a bounded time (reference: SweepCommon.hlsl:297-425,
                       // EmulatedDeadlocking.cu:159-267).  0 = bounded spin -> POISON -> GS_ERR_TIMEOUT only.
#endif
#ifndef GS_FALLBACK_SPINS
#define GS_FALLBACK_SPINS (1u << 12)  // ~ms: orders of magnitude above any healthy wait, false triggers only cost work
#endif
constexpr uint32_t FALLBACK_SPINS = GS_FALLBACK_SPINS;

#ifndef GS_FUSED_PAIRS
#define GS_FUSED_PAIRS 1  // (key, u32 value) pairs staged and scattered together (BinCfg::FUSED)
#endif
#ifndef GS_WALK_ROWS
#define GS_WALK_ROWS 1  // descriptor rows per round trip of the look-back walk.  Measured in round 2 (profiles/
                        // r02_ab_early_lookback_rows.txt): 4 rows per trip +3 %; 4 / 8 / 16 rows requested BEFORE the
                        // staging phase and consumed after it +3 / +5 / +6 % — an INCLUSIVE row is further back than that
                        // when the request is issued, so the walk repeats the reads and the chip only moved more bytes
#endif
// Chains of a pass: chain c < NCH = position segment of the pass's input (first pass of every sort; every pass of a sort
// planned on position chains, PF_POS) or group of the previous digit's values (NCH consecutive values per chain).  Every
// chain is one contiguous range of the pass's input.
constexpr uint32_t MAXCH = NCH;
// Chain SLOTS of an info block / a pass's ticket row.  The LSD plans use NCH of them; the second pass of the two-level plan
// (hybrid_kernels.hpp) runs on 256 chains — one per value of the top byte, whose bucket its input is already partitioned into.
constexpr uint32_t CHMAX = 256;
static_assert(CHMAX >= MAXCH && (CHMAX & (CHMAX - 1)) == 0, "chain slots");

// per-pass info block (uint32 words), written by scan_kernel (or hy_scan_kernel)
constexpr uint32_t I_START = 0;                // seg_start[CHMAX]
constexpr uint32_t I_END = CHMAX;              // seg_end[CHMAX]
constexpr uint32_t I_ROW = 2 * CHMAX;          // first descriptor row of each chain
constexpr uint32_t PASS_FLAGS = 3 * CHMAX;     // PF_* bits
constexpr uint32_t I_NCH = PASS_FLAGS + 1;     // chains in use (NCH, or CHMAX: a power of two)
constexpr uint32_t I_NEXT_SHIFT = PASS_FLAGS + 2;  // PF_POS: bit position of the digit of the next pass that runs (this pass counts it per
                                                   // output position segment while it scatters), ~0 = nothing to count
constexpr uint32_t I_SEGLOG = PASS_FLAGS + 3;      // PF_POS: log2 of the position segments of the passes behind the first one
constexpr uint32_t I_DSTRIDE = PASS_FLAGS + 5;     // words of one pass's descriptor region (launches with mode bit 9 zero the two regions behind their own)
constexpr uint32_t I_SHIFT = PASS_FLAGS + 4;       // bit position of this pass's digit (launches with mode bit 7 take it from here: the plan is the device's)
constexpr uint32_t I_MODE = PASS_FLAGS + 6;        // PF_SKEW passes: the most frequent value of this pass's digit
constexpr uint32_t INFO_STRIDE = ((PASS_FLAGS + 7 + 31) / 32) * 32;
constexpr uint32_t PF_SKEW = 1;    // some digit holds > n/16 keys (GS_SKEW_SHIFT): rank with wave-aggregated adds
constexpr uint32_t PF_SKIP = 2;     // every key has the same digit AND the pass is one of an even number of such
                                    // passes: the pass is the identity permutation, its workgroups exit at once
constexpr uint32_t PF_SRC_ALT = 4;  // an odd number of earlier passes ran: this pass reads alt and writes keys
constexpr uint32_t PF_LAST = 8;     // last pass that runs: applies the descending index reversal
constexpr uint32_t PF_POS = 16;     // the sort runs on position chains in EVERY pass (skewed keys: digit-group chains would be
                                    // as uneven as the digit values).  Behind the first pass nobody knows the chains' bases
                                    // upfront: each pass counts the next pass's digit per output position segment while it
                                    // scatters (CNEXT), every workgroup of the next pass derives digit starts / skew / mode digit
                                    // from those counts when it starts, and tile 0 of each chain seeds the chain's row 0

// ---- state slab layout (uint32 words), shared by host and kernels -------------
//  COUNTERS  tile tickets, [pass][chain]                       (reference m_index)
//  STATUS    device status word
//  INFO      per pass: the info block above                    (written by scan_kernel)
//  HIST      joint histograms H[pass][chain][digit]            (reference m_globalHistogram)
//  HSUB      CNEXT[pass q][segment x][digit d]: keys in position segment x of pass q's INPUT whose digit q is d — counted by the
//            pass that wrote that input (sorts planned on position chains, PF_POS); rows of NCH + 1 x 256 words per pass
//  DESC      descriptors: pass q at DESC + q*desc_stride, rows of 256 words
constexpr uint32_t SLAB_COUNTERS = 0;
constexpr uint32_t COUNTER_STRIDE = 32;  // one 128-byte line per ticket counter: chains do not share a line
constexpr uint32_t COUNTERS_PER_PASS = CHMAX + 8;
static_assert(COUNTERS_PER_PASS >= CHMAX, "ticket counters");
constexpr uint32_t MAX_PASSES = 8;  // 64-bit keys: one GlobalHistogram + Scan plans all eight passes (32-bit keys use the first four slots)
constexpr uint32_t SLAB_STATUS = MAX_PASSES * COUNTERS_PER_PASS * COUNTER_STRIDE;
constexpr uint32_t SLAB_INFO = SLAB_STATUS + 32;
constexpr uint32_t SLAB_HIST = SLAB_INFO + MAX_PASSES * INFO_STRIDE + 32;
// behind the joint tables: what else the histogram kernel tells the Scan kernel (zero between calls, like the tables)
constexpr uint32_t HIST_TABLE_WORDS = MAX_PASSES * NCH * RADIX;
constexpr uint32_t HX_SKEW = 0;  // a workgroup found the digit groups of its keys uneven and stopped counting the joint tables
constexpr uint32_t HX_OR = 1;    // OR of the digit words of all keys (sortable form) ...
constexpr uint32_t HX_NAND = 2;  // ... and of their complements: a bit set in both varies; a byte clear in their AND is constant — how a
                                 // sort planned on position chains (its joint tables are incomplete) still finds its identity passes
constexpr uint32_t HX_HY = 3;    // two-level plan (hybrid_kernels.hpp): 1 = hy_scan_kernel found it valid and planned it — scan_kernel leaves the info blocks alone
constexpr uint32_t HX_HY_BAD = 4;  // ... a histogram workgroup's packed 16-bit counters overflowed: the joint table is void
constexpr uint32_t HIST_WORDS = HIST_TABLE_WORDS + 32;
constexpr uint32_t SLAB_HSUB = SLAB_HIST + HIST_WORDS;
constexpr uint32_t HSUB_STRIDE = (NCH + 1) * RADIX;  // per pass
// MID: scratch of the two-launch sort of mid-size inputs (mid_kernels.hpp): plan epoch, route flag, bucket table, per-tile
// claim / flag words, two count tables of MID_MAX_TILES rows
constexpr uint32_t SLAB_MID = SLAB_HSUB + 4 * HSUB_STRIDE;
constexpr uint32_t SLAB_MID_WORDS = 2048 + 2 * 256 * RADIX;
// HY: the two-level plan's words (hybrid_kernels.hpp): valid flag, largest bucket
constexpr uint32_t SLAB_HY = SLAB_MID + SLAB_MID_WORDS;
constexpr uint32_t SLAB_HY_WORDS = 32;
constexpr uint32_t SLAB_DESC = SLAB_HY + SLAB_HY_WORDS;
static_assert(SLAB_HIST % 4 == 0 && SLAB_HSUB % 4 == 0 && SLAB_MID % 4 == 0 && SLAB_HY % 4 == 0 && SLAB_DESC % 4 == 0, "regions are cleared with 16-byte stores");
#ifndef GS_GHIST_THREADS
#define GS_GHIST_THREADS 1024
#endif
#ifndef GS_HIST_SKEW_LANES
#define GS_HIST_SKEW_LANES 8  // lanes sharing the first lane's bin that switch a byte's counting to wave-aggregated adds
#endif
#ifndef GS_HIST_PROBE
#define GS_HIST_PROBE 1  // 0 (ablation): no skew probe, plain adds only
#endif
#ifndef GS_GHIST_WAVES_PER_SIMD
#define GS_GHIST_WAVES_PER_SIMD (GS_GHIST_THREADS / 256)  // one workgroup per CU
#endif
#ifndef GS_HIST_UNROLL
#define GS_HIST_UNROLL 4
#endif
#ifndef GS_HIST_NT
#define GS_HIST_NT 1  // non-temporal key loads in the histogram kernel: 0.296 -> 0.250 ms at 2^28 (the pass that follows pays 0.008 ms
                     // of it back: fewer of its first reads hit the memory-side cache), profiles/r03_ab_hist_nt_loads.txt.  Measured
                     // with it and not kept: the next work item's loads in flight while this one is counted (no change: the kernel
                     // does not wait for its loads)
#endif
#ifndef GS_SKEW_SHIFT
#define GS_SKEW_SHIFT 4  // a pass is ranked with wave-aggregated adds (PF_SKEW, mode digit) when some digit holds more than n >> GS_SKEW_SHIFT keys.
                         // Rounds 1-2: 3.  Entropy preset 2 (two ANDs: digit 0 holds 10 %) fell between — plain LDS adds with a 10 % digit:
                         // 4 takes 3 % off its keys-only sort and 6 % off its (u32, u64) pairs, nothing changes elsewhere
                         // (profiles/r03_ab_skew_threshold.txt)
#endif
#ifndef GS_POS_KPT
#define GS_POS_KPT 32  // keys per thread of the counting position-chain passes (512 threads): the full tile, with packed counters
#endif
#ifndef GS_POSV_KPT
#define GS_POSV_KPT 24  // the same for pairs (32-bit counters: packing them costs these passes more than the larger tile returns)
#endif
#ifndef GS_POSV8_LAST_SMALL
#define GS_POSV8_LAST_SMALL 1  // (u32, u64) pairs: the last position-chain pass runs on the smaller tile as well
#endif
#ifndef GS_POS_SHARE
#define GS_POS_SHARE 3u  // a digit group holding more than GS_POS_SHARE / 16 of a workgroup's first 16 384 keys (even: 1 / 16) sends the
                         // sort to the position-chain kernels
#endif
#ifndef GS_HIST_REPLICAS
#define GS_HIST_REPLICAS 1  // pass-0 digit counts on 32 lane-private, bank-conflict-free replicas (see global_histogram_kernel)
#endif
constexpr uint32_t HIST_FOLD_CHUNKS = 256;  // replicas are folded at least this often (16-bit counters, 128 keys per replica per chunk)
constexpr uint32_t HIST_CHUNK = 4 * GS_GHIST_THREADS;  // keys per histogram work item (4 per thread); position segments are multiples of it

enum : int { KEY_U32 = 0, KEY_I32 = 1, KEY_F32 = 2, KEY_U64 = 3, KEY_I64 = 4, KEY_F64 = 5 };
// 64-bit keys (8 passes; the reference has 32-bit keys only) are sorted by eight passes of the SAME
// machinery — the digit of pass p is byte p & 3 of word p >> 2, both words travel as one 8-byte element — planned by one
// GlobalHistogram sweep (eight joint tables) and one Scan (MAX_PASSES info blocks, ticket rows and descriptor regions).
// KW = 32-bit words per key.
template <int KT>
struct KeyWords { static constexpr int value = KT >= KEY_U64 ? 2 : 1; };

template <int KT>
__device__ __forceinline__ uint32_t to_bits(uint32_t u) {
    if constexpr (KT == KEY_I32) return u ^ 0x80000000u;
    if constexpr (KT == KEY_F32) return u ^ ((uint32_t)(-(int32_t)(u >> 31)) | 0x80000000u);
    return u;
}
template <int KT>
__device__ __forceinline__ uint32_t from_bits(uint32_t u) {
    if constexpr (KT == KEY_I32) return u ^ 0x80000000u;
    if constexpr (KT == KEY_F32) return u ^ (((u >> 31) - 1u) | 0x80000000u);
    return u;
}

__device__ __forceinline__ uint32_t ld_agent(uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(uint32_t* p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// streaming accesses of the key/value arrays (every element is touched once per pass)
#ifndef GS_NT
#define GS_NT 1  // bit 0: non-temporal key loads in keys-only sorts (pass 0: -7 %; pairs: +10 %, so not there),
                 // bit 1: non-temporal stores (-40 %: the scatter needs L2 write-combining) — measured, r01_sweep_v21
#endif
template <bool NT, class T>
__device__ __forceinline__ T ld_stream(const T* p) {
    if constexpr (NT && (GS_NT & 1)) return __builtin_nontemporal_load(p);
    else return *p;
}
template <class T>
__device__ __forceinline__ void st_stream(T* p, T v) {
#if (GS_NT & 2)
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

// 64-bit keys: native (lo, hi) words -> radix-sortable words and back (the 32-bit rules of SortCommon.hlsl:134-154
// applied to the 64-bit pattern: signed flips the sign bit, floating point flips all bits of negatives too)
template <int KT>
__device__ __forceinline__ uint2 to_bits2(uint2 k) {
    if constexpr (KT == KEY_I64) k.y ^= 0x80000000u;
    if constexpr (KT == KEY_F64) {
        const uint32_t m = (uint32_t)(-(int32_t)(k.y >> 31));
        k.x ^= m;
        k.y ^= m | 0x80000000u;
    }
    return k;
}
template <int KT>
__device__ __forceinline__ uint2 from_bits2(uint2 k) {
    if constexpr (KT == KEY_I64) k.y ^= 0x80000000u;
    if constexpr (KT == KEY_F64) {
        const uint32_t m = (k.y >> 31) - 1u;
        k.x ^= m;
        k.y ^= m | 0x80000000u;
    }
    return k;
}

typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ __forceinline__ uint2 ld_stream(const uint2* p) {
    const u32x2_t v = ld_stream<NT>(reinterpret_cast<const u32x2_t*>(p));
    return uint2{v.x, v.y};
}
__device__ __forceinline__ void st_stream(uint2* p, uint2 v) {
    st_stream(reinterpret_cast<u32x2_t*>(p), u32x2_t{v.x, v.y});
}

template <int N>
struct IntTag { static constexpr int value = N; };
// block placement: the rare paths (steal, partial tiles, skew, fallback) tripled the
// kernel's code; keeping the common path contiguous keeps it in the instruction cache
#define GS_LIKELY(x) __builtin_expect(!!(x), 1)
#define GS_UNLIKELY(x) __builtin_expect(!!(x), 0)

// tiles of a chain [s0, s1): its tile grid starts at s0 rounded down to 64 keys (256-byte aligned wave loads)
__host__ __device__ __forceinline__ uint32_t chain_tiles(uint32_t s0, uint32_t s1, uint32_t tile_keys) {
    return s1 != s0 ? (s1 - (s0 & ~63u) + tile_keys - 1u) / tile_keys : 0u;
}

// inclusive scan across the 64 lanes of a wave
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (lane >= (uint32_t)d) v += t;
    }
    return v;
}

// the same scan on the VALU's data-parallel primitives: four row shifts inside each 16-lane row, then the two
// row broadcasts (lane 15 -> next row of each row pair, lane 31 -> the upper half).  No LDS round trips (the
// shuffle form costs six dependent ds_bpermute, ~0.3 us in the per-tile critical path).
__device__ __forceinline__ uint32_t wave_inclusive_scan_dpp(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);  // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2, 3
    return v;
}

__device__ __forceinline__ uint32_t wave_reduce_sum(uint32_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
    return v;
}

// ---------------------------------------------------------------------------
// GlobalHistogram: ONE sweep over the keys, `np` joint histograms
//   H[q](d, x), q = 0..np-1 (digit byte p0+q), d = digit, x = chain:
//   q == 0 : x = position segment (key index / seg_len0; seg_len0 % HIST_CHUNK == 0); stored [x][d]
//            (x is constant per workgroup, so [d][x] would put every add on two LDS banks)
//   q >= 1 : x = group of the PREVIOUS digit = its top log2(NCH) bits; stored [d][x], which makes the
//            bin ONE bit-field of the key: the 8 + log2(NCH) contiguous bits ending at the top of byte p0+q.
// 16-byte loads; one LDS histogram per workgroup (ds_add_u32); one global atomic
// per non-empty bin per workgroup.  grid-stride over HIST_CHUNK-key chunks.
// ---------------------------------------------------------------------------
constexpr int GHIST_THREADS = GS_GHIST_THREADS;
constexpr uint32_t LOG_NCH = NCH == 1 ? 0 : NCH == 2 ? 1 : NCH == 4 ? 2 : NCH == 8 ? 3 : NCH == 16 ? 4 : 5;

// index of joint-histogram bin (pass q, digit d, chain x)
__host__ __device__ constexpr uint32_t hist_index(uint32_t q, uint32_t d, uint32_t x) {
    return q == 0 ? x * RADIX + d : (q * RADIX + d) * NCH + x;
}

// 64-bit keys: np <= 4: `word` selects the 32-bit word the np digits are taken from (stand-alone passes, the multi-GPU split);
// np == 8 (the sort): ONE sweep counts all eight tables — the digits of the low word as above, byte 4's table joint with the
// low word's top bits (the bin is the 12-bit field at bit 28 of the 64-bit key), bytes 5..7 inside the high word — so
// the sort reads its keys once for the histogram instead of once per word (8 x 4096 bins: 128 KiB of LDS).  A work item is
// still HIST_CHUNK keys — two 16-byte loads per thread instead of one.
template <int KT>
__global__ __launch_bounds__(GHIST_THREADS, GS_GHIST_WAVES_PER_SIMD) void global_histogram_kernel(const uint32_t* __restrict__ keys,
                                                                         uint32_t* slab, size_t slab_used_words,
                                                                         uint32_t n, uint32_t seg_len0, uint32_t p0,
                                                                         uint32_t np, uint32_t word,
                                                                         uint32_t allow_pos /*1: the sort may run on position
                                                                         chains (scan_kernel, PF_POS) — uneven digit groups end the
                                                                         counting of the joint tables*/,
                                                                         uint32_t* partials /*[gridDim][HIST_TABLE_WORDS]: every
                                                                         workgroup's tables, summed by hist_reduce_kernel*/) {
    constexpr int KW = KeyWords<KT>::value;
    constexpr uint32_t NQ = KW == 2 ? MAX_PASSES : 4;  // tables a workgroup can count
    __shared__ __attribute__((aligned(16))) uint32_t s_h[NQ * NCH * RADIX];  // (read as uint4 for the slice store)
    __shared__ uint32_t s_uneven;  // a digit group of this workgroup's first work item holds more than GS_POS_SHARE of its keys
#if GS_HIST_REPLICAS
    // Pass-0 digit counts on 32 lane-private replicas, 16-bit counters packed two per dword: dword (d >> 1) * 32 +
    // (lane & 31) lies in bank lane & 31, so a wave's add never meets a bank conflict — whatever the keys are: the
    // 256-bin table of one position segment was the most expensive of the four (9.7 clk per wave-add against 7.5
    // for the 4096-bin joint tables and 4.2 conflict-free, profiles/r02_lds_microbench.txt: 64 random lanes on
    // 256 bins collide on ADDRESSES, which serialises atomics), and a constant low byte — every lane on one
    // counter, 128 clk — costs nothing here.  Folded into s_h when the workgroup's segment changes, every
    // HIST_FOLD_CHUNKS chunks (a replica sees 128 keys per chunk: 16-bit counters hold 511 chunks), and at the end.
    __shared__ __attribute__((aligned(16))) uint32_t s_r[RADIX / 2 * 32];
#endif
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    uint32_t* hist = slab + SLAB_HIST;
    GS_ABL_CLOCKS_BEGIN();
    GS_HIST_STAMPS_DECL();
    GS_HIST_STAMP(0);
    // This kernel is also the sort's CLEAR (reference: ClearMemory, OneSweepDispatcher.cuh:301-309): it zeroes the
    // scan state nobody reads before it ends — ticket counters, status, info, slice counts, descriptors — as
    // 16-byte grid-stride stores next to its read stream; a separate memset was one more launch (5 us of a 50 us
    // sort at mid sizes).  The HIST region is zero whenever no call is in flight: the first DigitBinningPass
    // launched after the Scan re-zeroes it (mode bit 2).
    {
        uint4* a = reinterpret_cast<uint4*>(slab);
        const size_t na = SLAB_HIST / 4, b0 = SLAB_HSUB / 4, nb = slab_used_words / 4;  // all multiples of 4 words
        const size_t stride = (size_t)gridDim.x * GHIST_THREADS;
        const uint4 z = {0u, 0u, 0u, 0u};
        for (size_t i = (size_t)blockIdx.x * GHIST_THREADS + tid; i < na; i += stride) a[i] = z;
        for (size_t i = b0 + (size_t)blockIdx.x * GHIST_THREADS + tid; i < nb; i += stride) a[i] = z;
    }
    const uint32_t bins = np * NCH * RADIX;
    for (uint32_t i = tid; i < bins; i += GHIST_THREADS) s_h[i] = 0;
    if (tid == 0) s_uneven = 0;
    bool joint_off = (allow_pos & 2u) != 0u;  // uniform: the joint tables are given up (see the probe behind the first work item; bit 1: from the start)
#if GS_HIST_REPLICAS
    for (uint32_t i = tid; i < RADIX / 2 * 32; i += GHIST_THREADS) s_r[i] = 0;
    uint32_t cur_x0 = 0xffffffffu, since_fold = 0;  // uniform: segment the replicas are counting for, chunks since the last fold
    // Eight threads share the 32 replicas of a counter pair (one 16-byte read each, conflict-free), sum them with three
    // butterfly steps, and the first of them owns the two bins they go to.  (One LDS add per non-empty replica word had the
    // 32 replicas of a pair meet on ONE address: 128 wave-adds of 32 serial steps each, 9 us per fold — half of this
    // kernel's fixed cost at mid sizes and 3 % of it at 2^28, profiles/r03_mid_route_v3_and_hist_fold.txt.)
    auto fold = [&](uint32_t x) {
        __syncthreads();
        if (x != 0xffffffffu) {
            for (uint32_t i = tid; i < RADIX / 2 * 8; i += GHIST_THREADS) {  // (one round with 1024 threads)
                const uint4 v = reinterpret_cast<const uint4*>(s_r)[i];
                reinterpret_cast<uint4*>(s_r)[i] = uint4{0u, 0u, 0u, 0u};
                uint32_t lo = (v.x & 0xffffu) + (v.y & 0xffffu) + (v.z & 0xffffu) + (v.w & 0xffffu);
                uint32_t hi = (v.x >> 16) + (v.y >> 16) + (v.z >> 16) + (v.w >> 16);
#pragma unroll
                for (int m = 1; m < 8; m <<= 1) {
                    lo += __shfl_xor(lo, m, 64);
                    hi += __shfl_xor(hi, m, 64);
                }
                if ((i & 7u) == 0u) {
                    s_h[hist_index(0, (i >> 3) * 2u, x)] += lo;
                    s_h[hist_index(0, (i >> 3) * 2u + 1u, x)] += hi;
                }
            }
        }
        __syncthreads();
    };
#endif
    __syncthreads();
    GS_HIST_STAMP(1);

    const uint32_t shift0 = p0 * 8u;
    const bool both = KW == 2 && np > 4u;  // uniform: all eight tables of a 64-bit key in this sweep (p0 = 0)
    // b: the word the first four digits come from, hi: the key's high word (both only)
    auto bin_of = [&](uint32_t b, uint32_t hi, uint32_t q, uint32_t x0) -> uint32_t {
        if (q == 0) return hist_index(0, (b >> shift0) & 255u, x0);
        if (q < 4) return q * (RADIX * NCH) + __builtin_amdgcn_ubfe(b, shift0 + 8u * q - LOG_NCH, 8u + LOG_NCH);
        if (q == 4) return 4u * (RADIX * NCH) + (((hi & 255u) << LOG_NCH) | (b >> (32u - LOG_NCH)));
        return q * (RADIX * NCH) + __builtin_amdgcn_ubfe(hi, 8u * (q - 4u) - LOG_NCH, 8u + LOG_NCH);
    };

    uint32_t k_or = 0, k_nand = 0;  // OR of this thread's digit words / of their complements (free: the kernel waits for its LDS adds)
    uint32_t skew_mode = 0;  // bit q (wave-uniform): a dominant bin was seen for byte q; cleared when it fades
    uint32_t sticky[NQ];     // wave-uniform guess of that bin
#pragma unroll
    for (uint32_t q = 0; q < NQ; ++q) sticky[q] = 0xffffffffu;
    // One work item = HIST_UNROLL consecutive chunks; all their 16-byte loads are issued before the
    // first is consumed (one load per thread in flight left the kernel latency-bound at 3.1 TB/s).
    constexpr uint32_t HIST_UNROLL = GS_HIST_UNROLL;
    // JOINT = 0: the joint tables are given up (joint_off) — only the first digit is counted
    // t: four keys' digit words, sortable form; th: their high words (both)
    auto process = [&](auto joint_tag, const uint4 t, const uint4 th, const uint32_t x0, const bool probe) {
            constexpr bool JOINT = decltype(joint_tag)::value != 0;
            GS_ABL_HIST_STREAM_ONLY(t);
            const uint32_t b[4] = {t.x, t.y, t.z, t.w};
            k_or |= t.x | t.y | t.z | t.w;
            k_nand |= ~(t.x & t.y & t.z & t.w);
            const uint32_t bh[4] = {th.x, th.y, th.z, th.w};
#if GS_HIST_REPLICAS
            if (x0 != cur_x0 || since_fold >= HIST_FOLD_CHUNKS) {  // uniform
                fold(cur_x0);
                cur_x0 = x0;
                since_fold = 0;
            }
            ++since_fold;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t d = (b[j] >> shift0) & 255u;
                atomicAdd(&s_r[(d >> 1) * 32u + (lane & 31u)], 1u << ((d & 1u) * 16u));
            }
#endif
            // Skew (Thearling-Smith presets, constant bytes): same-address LDS atomics serialise per lane.  Cheap probe on
            // the first key of a work item's first chunk: do >= 8 lanes share the first lane's bin?  (On every chunk the
            // probe cost the uniform case 10 % of the kernel, profiles/r02_ab_hist_variants.txt; skew does not come and go
            // chunk by chunk.)  The probe runs BEFORE the adds so that a wave that has seen no dominant bin does every add of
            // the chunk in one basic block: the address arithmetic of one table overlaps the LDS adds of the previous one
            // (with uniform branches between the tables they could not: profiles/r04_hist_kernel_variants.txt).
            if (GS_HIST_PROBE && probe) {
#pragma unroll
                for (uint32_t q = GS_HIST_REPLICAS ? 1 : 0; q < NQ; ++q)
                    if (q < np && (JOINT || q == 0)) {
                        const uint32_t bin0 = bin_of(b[0], bh[0], q, x0);
                        const uint32_t b0 = __builtin_amdgcn_readfirstlane(bin0);
                        const uint32_t pc = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(bin0 == b0));
                        if (pc >= GS_HIST_SKEW_LANES) {
                            skew_mode |= 1u << q;
                            if (pc >= 24 || sticky[q] == 0xffffffffu) sticky[q] = b0;  // (re)learn the dominant bin
                        }
                    }
            }
            if (skew_mode == 0u) {  // uniform
#pragma unroll
                for (uint32_t q = GS_HIST_REPLICAS ? 1 : 0; q < NQ; ++q)
                    if (q < np && (JOINT || q == 0)) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) atomicAdd(&s_h[bin_of(b[j], bh[j], q, x0)], 1u);
                    }
                return;
            }
#pragma unroll
            for (uint32_t q = GS_HIST_REPLICAS ? 1 : 0; q < NQ; ++q) {
                if (q < np && (JOINT || q == 0)) {
                    uint32_t bin[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) bin[j] = bin_of(b[j], bh[j], q, x0);
                    if (skew_mode & (1u << q)) {
                        // lanes holding the remembered dominant bin are counted with ONE add of their
                        // popcount (by their first lane); all other lanes add individually
                        // (measured and not kept: counting those lanes on the scalar unit only — a running popcount,
                        //  added when the guess changes — skewed presets -2 %, uniform +4 %: the cost under skew is the
                        //  conflicts among the OTHER hot bins, profiles/r02_skew_rank_fixed_mode.txt)
                        const uint32_t b0 = __builtin_amdgcn_readfirstlane(bin[0]);
                        const uint32_t pc = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(bin[0] == b0));
                        // (round 5: relearn BEFORE the adds — on sorted input every chunk lies under a new dominant bin, and relearning
                        //  behind the adds put 64 lanes x 4 keys on one counter one by one first: 0.88 ms for this kernel on sorted keys)
                        if (pc >= 24 && b0 != sticky[q]) sticky[q] = b0;
                        uint32_t hit = 0;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const unsigned long long m = __builtin_amdgcn_ballot_w64(bin[j] == sticky[q]);
                            hit += (uint32_t)__popcll(m);
                            if (bin[j] != sticky[q]) atomicAdd(&s_h[bin[j]], 1u);
                            else if (__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)) == 0u)
                                atomicAdd(&s_h[sticky[q]], (uint32_t)__popcll(m));
                        }
                        if (hit < 32) {  // the guess covers < 1/8 of the lanes: relearn, or leave skew mode
                            sticky[q] = b0;
                            if (pc < 8) skew_mode &= ~(1u << q);
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) atomicAdd(&s_h[bin[j]], 1u);
                    }
                }
            }
    };
    // the four keys of thread tid in chunk c, reduced to their digit words (radix-sortable form)
    auto word_of = [&](uint32_t lo, uint32_t hi) {
        const uint2 b = to_bits2<KT>(uint2{lo, hi});
        return word ? b.y : b.x;
    };
    typedef uint32_t hv4 __attribute__((ext_vector_type(4)));
    auto ld16 = [](const uint4* q) -> uint4 {
#if GS_HIST_NT
        const hv4 v = __builtin_nontemporal_load(reinterpret_cast<const hv4*>(q));
        return uint4{v.x, v.y, v.z, v.w};
#else
        return *q;
#endif
    };
    struct Chunk { uint4 w, hi; };  // the digit words of a thread's four keys (and, 64-bit keys counted in one sweep, their high words)
    auto load_chunk = [&](uint32_t c) -> Chunk {
        if constexpr (KW == 2) {
            const uint4* p = reinterpret_cast<const uint4*>(keys) + (size_t)c * (HIST_CHUNK / 2);
            const uint4 a = ld16(p + tid), b2 = ld16(p + tid + GHIST_THREADS);
            if (both) {
                const uint2 k0 = to_bits2<KT>(uint2{a.x, a.y}), k1 = to_bits2<KT>(uint2{a.z, a.w});
                const uint2 k2 = to_bits2<KT>(uint2{b2.x, b2.y}), k3 = to_bits2<KT>(uint2{b2.z, b2.w});
                return Chunk{uint4{k0.x, k1.x, k2.x, k3.x}, uint4{k0.y, k1.y, k2.y, k3.y}};
            }
            return Chunk{uint4{word_of(a.x, a.y), word_of(a.z, a.w), word_of(b2.x, b2.y), word_of(b2.z, b2.w)}, uint4{0u, 0u, 0u, 0u}};
        } else {
            const uint4 a = ld16(reinterpret_cast<const uint4*>(keys + (size_t)c * HIST_CHUNK) + tid);
            return Chunk{uint4{to_bits<KT>(a.x), to_bits<KT>(a.y), to_bits<KT>(a.z), to_bits<KT>(a.w)}, uint4{0u, 0u, 0u, 0u}};
        }
    };
    const uint32_t nchunks_all = (n + HIST_CHUNK - 1) / HIST_CHUNK;
#if GS_HIST_REPLICAS
    // every workgroup takes ONE contiguous range of chunks: it stays inside a position segment (the replicas
    // count for one segment at a time) and streams 1/gridDim of the array
    const uint32_t per_wg = (nchunks_all + gridDim.x - 1) / gridDim.x;
    const uint32_t c_first = blockIdx.x * per_wg, c_step = HIST_UNROLL;
    const uint32_t nchunks = c_first + per_wg < nchunks_all ? c_first + per_wg : nchunks_all;
#else
    const uint32_t c_first = blockIdx.x * HIST_UNROLL, c_step = gridDim.x * HIST_UNROLL, nchunks = nchunks_all;
#endif
    uint32_t c0 = c_first;
    for (; c0 < nchunks; c0 += c_step) {
        if (c0 + HIST_UNROLL <= nchunks && (unsigned long long)(c0 + HIST_UNROLL) * HIST_CHUNK <= n) {
            // common case: HIST_UNROLL full chunks — UNCONDITIONAL loads (conditional ones get an
            // s_waitcnt vmcnt(0) each from the compiler and end up one at a time in flight)
            Chunk t[HIST_UNROLL];
#pragma unroll
            for (uint32_t u = 0; u < HIST_UNROLL; ++u) t[u] = load_chunk(c0 + u);
            if (GS_LIKELY(!joint_off)) {
#pragma unroll
                for (uint32_t u = 0; u < HIST_UNROLL; ++u) process(IntTag<1>{}, t[u].w, t[u].hi, (c0 + u) * HIST_CHUNK / seg_len0, u == 0);
            } else {
#pragma unroll
                for (uint32_t u = 0; u < HIST_UNROLL; ++u) process(IntTag<0>{}, t[u].w, t[u].hi, (c0 + u) * HIST_CHUNK / seg_len0, u == 0);
            }
            if (c0 == c_first) GS_HIST_STAMP(2);
            if (allow_pos && c0 == c_first && !joint_off) {
                // Are the digit groups even?  The chains of passes 1..3 are the NCH groups of the previous digit's values:
                // with skewed keys (Thearling-Smith presets 2..5: group 0 holds 32 .. 88 % of them) one chain gets most
                // tiles and commits in order, and the joint tables themselves cost 2-3x (same-address LDS adds).  The
                // first work item (16 384 keys) is the sample: a group above GS_POS_SHARE / 16 of it ends the joint
                // counting HERE and tells the Scan kernel to plan the sort on position chains (HX_SKEW); the tables
                // of other workgroups that go on counting are simply not used.
                __syncthreads();
                if (tid < 3 * NCH) {
                    const uint32_t q = 1u + tid / NCH, x = tid % NCH;
                    uint32_t c = 0;
                    for (uint32_t d = 0; d < RADIX; ++d) c += s_h[hist_index(q, d, x)];
                    // (a group holding EVERY key of the sample: a constant byte — its pass is dropped, not a crowded chain)
                    if (q < np && c > (HIST_UNROLL * HIST_CHUNK / 16u) * GS_POS_SHARE && c != HIST_UNROLL * HIST_CHUNK) s_uneven = 1u;
                }
                __syncthreads();
                joint_off = joint_off || s_uneven != 0u;
                GS_HIST_STAMP(3);
            }
            continue;
        }
        Chunk t[HIST_UNROLL];
#pragma unroll
        for (uint32_t u = 0; u < HIST_UNROLL; ++u) {
            const uint32_t base = (c0 + u) * HIST_CHUNK;
            if (c0 + u < nchunks && base + HIST_CHUNK <= n) t[u] = load_chunk(c0 + u);
        }
#pragma unroll
        for (uint32_t u = 0; u < HIST_UNROLL; ++u) {
            const uint32_t base = (c0 + u) * HIST_CHUNK;
            if (c0 + u < nchunks) {
                const uint32_t x0 = base / seg_len0;  // uniform: a whole chunk lies in one position segment
                if (base + HIST_CHUNK <= n) {
                    if (joint_off) process(IntTag<0>{}, t[u].w, t[u].hi, x0, true);
                    else process(IntTag<1>{}, t[u].w, t[u].hi, x0, true);
                } else {
                    for (uint32_t i = base + tid; i < n; i += GHIST_THREADS) {
                        uint32_t kb, kh = 0;
                        if constexpr (KW == 2) {
                            const uint2 k2 = to_bits2<KT>(uint2{keys[2 * (size_t)i], keys[2 * (size_t)i + 1]});
                            kb = (both || !word) ? k2.x : k2.y;
                            kh = k2.y;
                        } else {
                            kb = to_bits<KT>(keys[i]);
                        }
                        k_or |= kb;
                        k_nand |= ~kb;
                        for (uint32_t q = 0; q < np; ++q)
                            if (!(joint_off && q >= 1)) atomicAdd(&s_h[bin_of(kb, kh, q, x0)], 1u);
                    }
                }
            }
        }
    }
    GS_HIST_STAMP(4);
#if GS_HIST_REPLICAS
    fold(cur_x0);
#else
    __syncthreads();
#endif
    GS_HIST_STAMP(5);
    // The workgroup's tables go out as they are — coalesced plain stores into its own slice — and hist_reduce_kernel sums the
    // slices.  (One global atomic per non-empty bin cost 256 workgroups x ~14 000 device-scope atomics on the same 16 384 words:
    // 20-40 us of every sort from 2^23 keys up, half of this kernel at 2^23, profiles/r03_mid_size_routes.txt.)
    {
        uint32_t* mine = partials + (size_t)blockIdx.x * HIST_TABLE_WORDS;
        for (uint32_t i = tid; i < bins / 4u; i += GHIST_THREADS)
            reinterpret_cast<uint4*>(mine)[i] = reinterpret_cast<const uint4*>(s_h)[i];
    }
    if (tid == 0 && joint_off) atomicOr(&hist[HIST_TABLE_WORDS + HX_SKEW], 1u);
    if (allow_pos) {
        // (only a sort that may end up on position chains needs them.  ONE pair of atomics per workgroup: same-address device atomics
        //  run at ~100 per microsecond — a pair per wave, 8192 of them, cost this kernel 0.075 ms, profiles/r04_hist_orand_atomics.txt)
#pragma unroll
        for (int dd = 32; dd > 0; dd >>= 1) {
            k_or |= __shfl_xor(k_or, dd, 64);
            k_nand |= __shfl_xor(k_nand, dd, 64);
        }
        __syncthreads();  // (s_h is stored: its first words serve as scratch)
        if (tid < 2) s_h[tid] = 0;
        __syncthreads();
        if (lane == 0) {
            atomicOr(&s_h[0], k_or);
            atomicOr(&s_h[1], k_nand);
        }
        __syncthreads();
        if (tid == 0) {
            atomicOr(&hist[HIST_TABLE_WORDS + HX_OR], s_h[0]);
            atomicOr(&hist[HIST_TABLE_WORDS + HX_NAND], s_h[1]);
        }
    }
    GS_HIST_STAMP(6);
    GS_ABL_CLOCKS_END();
    GS_HIST_STAMPS_OUT();
}

// Sum of the histogram workgroups' tables: hist[i] = sum over workgroups of partials[w][i].  A workgroup of 256 threads owns
// 64 consecutive bins; thread (g, b) = (tid / 64, tid % 64) sums every fourth slice of bin b — 256-byte wave loads, up to
// 8 in flight, at most nblocks / 4 of them in a row — and the four partial sums meet in LDS.  (One thread per bin, 256 loads
// in a row, was latency-bound: 30 us.)  The HIST region is overwritten, not accumulated into.
// (Measured and not kept: the Scan run by the LAST workgroup of this kernel to store its sums — sc1 stores, a ticket, the four
//  passes one after the other on one CU — 22 us instead of 5 + 5 for the two kernels, profiles/r03_mid_size_timeline.txt.)
__global__ __launch_bounds__(256) void hist_reduce_kernel(const uint32_t* __restrict__ partials, uint32_t nblocks, uint32_t bins,
                                                           uint32_t* __restrict__ hist) {
    __shared__ uint32_t s_part[4][64];
    const uint32_t b = threadIdx.x & 63u, g = threadIdx.x >> 6;
    const uint32_t i = blockIdx.x * 64u + b;
    uint32_t acc = 0;
    if (i < bins) {
        uint32_t w = g;
        for (; w + 28u < nblocks; w += 32u) {
            uint32_t v[8];
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j) v[j] = partials[(size_t)(w + 4u * j) * HIST_TABLE_WORDS + i];
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j) acc += v[j];
        }
        for (; w < nblocks; w += 4u) acc += partials[(size_t)w * HIST_TABLE_WORDS + i];
    }
    s_part[g][b] = acc;
    __syncthreads();
    if (g == 0 && i < bins) hist[i] = s_part[0][b] + s_part[1][b] + s_part[2][b] + s_part[3][b];
}

// ---------------------------------------------------------------------------
// Scan: per pass q (one workgroup each, 256 threads = digits)
//   G_q[d]      = sum_x H[q][d][x]                 digit totals
//   dstart[d]   = exclusive prefix of G_q          global start of digit d's run
//   seg_start[] : q == 0 -> x*seg_len0 ; q >= 1 -> starts of the digit-(q-1) groups
//   row_base[]  : first descriptor row of chain x  (chain x owns tiles_x + 1 rows)
//   row 0 of chain x seeded INCLUSIVE with dstart[d] + sum_{x'<x} H[q][d][x']
// ---------------------------------------------------------------------------
// NPT: joint tables that may hold counts (4: 32-bit keys and every stand-alone use; 8: the sort of 64-bit keys); grid = passes planned
template <int NPT>
__global__ __launch_bounds__(256) void scan_kernel(const uint32_t* hist, uint32_t* desc, uint32_t* info,
                                                    uint32_t desc_stride /*words per pass*/, uint32_t n,
                                                    uint32_t seg_len0, uint32_t tile_keys,
                                                    uint32_t plan /*bit0 descending, bit1 full 4-pass sort: may skip identity
                                                                    passes, bit2 (with bit1): position chains in every pass allowed*/,
                                                    uint32_t tile_keys_pos /*tile of the position-chain kernels (bit2); bit 31: of their last pass (3) as well*/,
                                                    uint32_t tile_keys0 /*tile of the plan's first pass (mid sizes: the first pass runs on the
                                                                          larger tile, its position segments are whole tiles)*/) {
    __shared__ uint32_t s_wtot[2][4];
    __shared__ uint32_t s_cum[RADIX + 1];
    __shared__ uint32_t s_start[MAXCH], s_end[MAXCH], s_rowbase[MAXCH + 1];
    __shared__ uint32_t s_triv;
    __shared__ unsigned long long s_mode;     // most frequent value of this digit, if it holds more than 1/16 of the keys
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, q = blockIdx.x;
    uint32_t* my_info = info + q * INFO_STRIDE;
    uint32_t* my_desc = desc + (size_t)q * desc_stride;

    // Everything this workgroup needs from global memory, in ONE round trip: thread d's row of every joint
    // histogram (rows of passes that were not counted are zero).  The kernel is a serial step of every sort;
    // with the loads strung out behind each other it took 11 us.
    uint32_t hq[NCH], g_all[NPT], g = 0, gprev = 0;
    const uint32_t np = gridDim.x;  // passes of this plan
    if (hist[HIST_TABLE_WORDS + HX_HY] != 0u) return;  // (uniform) the sort runs on the two-level plan: hy_scan_kernel has written every info block
    const uint32_t hx_skew = hist[HIST_TABLE_WORDS + HX_SKEW];
    {
        uint32_t h[NPT][NCH];
#pragma unroll
        for (uint32_t qq = 0; qq < NPT; ++qq)
#pragma unroll
            for (uint32_t x = 0; x < NCH; ++x) h[qq][x] = hist[hist_index(qq, tid, x)];
#pragma unroll
        for (uint32_t x = 0; x < NCH; ++x) hq[x] = 0;
#pragma unroll
        for (uint32_t qq = 0; qq < NPT; ++qq) {
            g_all[qq] = 0;
#pragma unroll
            for (uint32_t x = 0; x < NCH; ++x) g_all[qq] += h[qq][x];
            if (qq == q) {
                g = g_all[qq];
#pragma unroll
                for (uint32_t x = 0; x < NCH; ++x) hq[x] = h[qq][x];
            }
            if (qq + 1 == q) gprev = g_all[qq];
        }
    }
    // Position chains in every pass (PF_POS): some workgroup of the histogram kernel found the digit groups of its keys
    // uneven and stopped counting the joint tables — they are incomplete, only the first digit's position histogram
    // (hq of pass 0) and the OR / AND of all keys are whole.  Uniform: every workgroup reads the same words.
    const bool pos = (plan & 6u) == 6u && hx_skew != 0u;
    if (tid == 0) { s_triv = 0; s_mode = 0; }
    __syncthreads();

    // ---- which passes run (full sorts only).  A pass whose digit is the same for every key is the identity
    // permutation; such passes are dropped in PAIRS, so the result still lands in the caller's buffer with no
    // extra copy and no host round trip: every workgroup of a dropped pass exits on its flag word, every other
    // pass learns from its flags which buffer it reads.  A descending sort keeps one pass to do the reversal.
    // (Position chains: the digit totals behind the first digit are not known here; the OR / AND of all keys, which the histogram
    //  kernel accumulates on the side, say which bytes are constant — see below.)
    if ((plan & 2u) && !pos) {
#pragma unroll
        for (uint32_t qq = 0; qq < NPT; ++qq)
            if (qq < np && g_all[qq] == n) atomicOr(&s_triv, 1u << qq);
    }
    if ((plan & 2u) && pos && NPT == 4 && tid == 0) {
        // position chains: the digit totals behind the first digit are incomplete — the OR / AND of all keys say which BYTES are
        // constant.  The first pass always runs (its chains are the histogram's position segments); passes 1..3 are dropped in pairs.
        const uint32_t varying = hist[HIST_TABLE_WORDS + HX_OR] & hist[HIST_TABLE_WORDS + HX_NAND];
        uint32_t t = 0;
        for (uint32_t qq = 1; qq < np; ++qq)
            if (((varying >> (8u * qq)) & 255u) == 0u) t |= 1u << qq;
        if (t) atomicOr(&s_triv, t);
    }
    const bool counted = !pos || q == 0;  // this pass's digit totals g and chain rows hq are complete
    if (counted && g >= (n >> GS_SKEW_SHIFT) + 1u) atomicMax(&s_mode, ((unsigned long long)g << 8) | (255u - tid));  // (ties: the smaller digit)
    // digit scans of this pass's totals and (for the segment starts) of the previous digit's totals
    const uint32_t incl = wave_inclusive_scan(g, lane);
    const uint32_t incl_prev = wave_inclusive_scan(gprev, lane);
    if (lane == 63) { s_wtot[0][wave] = incl; s_wtot[1][wave] = incl_prev; }
    __syncthreads();
    uint32_t base = 0, base_prev = 0;
    for (uint32_t w = 0; w < wave; ++w) { base += s_wtot[0][w]; base_prev += s_wtot[1][w]; }
    uint32_t skip = 0;  // uniform: passes that are dropped
    if (plan & 2u) {
        const uint32_t triv = s_triv;
        uint32_t drop = (uint32_t)__popc(triv) & ~1u;
        if ((plan & 1u) && drop == np) drop = np - 2u;  // (a descending sort keeps one pair: its second pass reverses)
        for (uint32_t qq = 0; qq < np && drop; ++qq)
            if ((triv >> qq) & 1u) { skip |= 1u << qq; --drop; }
    }
    const uint32_t run_mask = ~skip & ((1u << np) - 1u);
    if ((plan & 2u) && tid == 0) {
        uint32_t f = 0;
        if ((skip >> q) & 1u) f |= PF_SKIP;
        if (__popc(run_mask & ((1u << q) - 1u)) & 1) f |= PF_SRC_ALT;
        if (run_mask && q == 31u - (uint32_t)__clz(run_mask)) f |= PF_LAST;
        if (pos) f |= PF_POS;
        if (f) atomicOr(&my_info[PASS_FLAGS], f);
    }
    // position segments behind the first pass (PF_POS): a power of two, so the pass in front finds a key's segment
    // with one shift of its output index
    const uint32_t seglog = 32u - (uint32_t)__clz((((n + NCH - 1u) / NCH) - 1u) | 1u);

    // segment starts: position segments (first pass: multiples of the histogram's chunk; PF_POS: powers of two);
    // otherwise starts of the digit-(q-1) groups
    if (q == 0) {
        if (tid <= NCH) {
            const unsigned long long s = (unsigned long long)tid * seg_len0;
            s_cum[tid] = s < n ? (uint32_t)s : n;
        }
    } else if (pos) {
        if (tid <= NCH) {
            const unsigned long long s = (unsigned long long)tid << seglog;
            s_cum[tid] = s < n ? (uint32_t)s : n;
        }
    } else {
        s_cum[tid + 1] = base_prev + incl_prev;  // keys with previous digit <= tid
        if (tid == 0) s_cum[0] = 0;
        __syncthreads();
        uint32_t v = 0;
        if (tid <= NCH) v = s_cum[tid * (RADIX / NCH)];
        __syncthreads();
        if (tid <= NCH) s_cum[tid] = v;  // compact: s_cum[x] = start of digit group x
    }
    __syncthreads();
    if (tid < MAXCH) {
        s_start[tid] = s_cum[tid];
        s_end[tid] = s_cum[tid + 1];
    }
    __syncthreads();
    // first descriptor row of every chain: a wave-level scan over the chains' row counts
    if (wave == 0) {
        uint32_t rows = 0;
        // (PF_POS: the passes that count for a successor run on the smaller tile, the last one on the full-size tile)
        if (lane < MAXCH)
            rows = chain_tiles(s_start[lane], s_end[lane], (pos && (q != 3u || (tile_keys_pos >> 31))) ? (tile_keys_pos & 0x7fffffffu) : (q == 0u ? tile_keys0 : tile_keys)) + 1u;
        const uint32_t rincl = wave_inclusive_scan(rows, lane);
        if (lane <= MAXCH) s_rowbase[lane] = rincl - rows;
    }
    __syncthreads();
    if (tid < MAXCH) {
        my_info[I_START + tid] = s_start[tid];
        my_info[I_END + tid] = s_end[tid];
        my_info[I_ROW + tid] = s_rowbase[tid];
    }
    if (tid == 0) {
        my_info[I_NCH] = NCH;
        // the pass behind this one that runs: this pass counts ITS digit per output segment (PF_POS)
        const uint32_t later = run_mask & ~((2u << q) - 1u);
        my_info[I_NEXT_SHIFT] = (pos && later) ? 8u * (uint32_t)__builtin_ctz(later) : 0xffffffffu;
        my_info[I_SEGLOG] = seglog;
        my_info[I_SHIFT] = 8u * q;
        my_info[I_DSTRIDE] = desc_stride;
        my_info[I_MODE] = s_mode ? 255u - (uint32_t)(s_mode & 255u) : 0xffffffffu;
    }
    if (!counted) return;  // PF_POS behind the first pass: the pass's own workgroups derive skew flag, mode digit and seeds

    // skew flag for the pass: some digit holds more than 1/16 of the keys -> tiles rank with
    // wave-aggregated adds (a dominant digit would serialise 64 lanes on one LDS counter)
    const unsigned long long skewed = __builtin_amdgcn_ballot_w64(g >= (n >> GS_SKEW_SHIFT) + 1u);
    if (lane == 0 && skewed) atomicOr(&my_info[PASS_FLAGS], PF_SKEW);
    // digit starts and chain bases
    uint32_t run = base + incl - g;  // dstart[tid]
#pragma unroll
    for (uint32_t x = 0; x < NCH; ++x) {
        my_desc[(size_t)s_rowbase[x] * RADIX + tid] = (run << 2) | FLAG_INCLUSIVE;
        run += hq[x];
    }
}

// ---------------------------------------------------------------------------
// DigitBinningPass: one stable 8-bit partition pass; one tile of THREADS*KPT
// keys per workgroup; chained-scan decoupled look-back inside the tile's chain.
//   VB = value bytes (0 keys-only, 4, 8), KT = key type, RANK = 0 ballots / 1 LDS atomic
// Tile-local order == array order (keys are loaded wave-striped: wave w owns
// 64*KPT consecutive keys, round i = 64 consecutive keys), which is what makes
// each pass stable.
// ---------------------------------------------------------------------------
template <int VB>
struct ValT { using type = uint32_t; };
template <>
struct ValT<8> { using type = uint64_t; };

template <int THREADS, int KPT, int VB, int KW = 1, int VR = 1, int POS = 0>
struct BinCfg {
    static constexpr int WAVES = THREADS / 64;
    static constexpr int TILE = THREADS * KPT;
    // 4-byte values travel WITH their keys: loaded up front, staged as 8-byte (key, value) slots, scattered in
    // the same loop — no second staging round, no saved positions/digits, two barriers fewer per tile
    // (8-byte values the same way need a 512 x 24 tile, 12 288 pairs x 12 B = 144 KiB in two LDS arrays: measured
    //  5.975 vs 6.014 ms, not worth a shape of its own; the code path stays generic in VB)
    static constexpr bool FUSED = GS_FUSED_PAIRS && VB == 4 && KW == 1 && POS == 0;  // (the position-chain forms keep their LDS for the count table)
    // VR = 2: 8-byte values of 4-byte keys go through the stage in TWO rounds of TILE / 2 values: the stage stays at the
    // 4 bytes per key the keys need, and a 16 384-pair tile leaves room for a second workgroup on the CU.  Measured
    // (profiles/r02_ab_value_rounds.txt, 2^28 (u32, u64) pairs): uniform keys +5 % per pass (two predicated staging
    // rounds, two more barriers), skewed keys -9 .. -13 % (the second workgroup covers the look-back waits of the
    // crowded chain) — so both forms are compiled and the pass's PF_SKEW flag picks one on the device (mode bits 4, 5).
    static constexpr int VROUNDS = (VR == 2 && VB == 8 && KW == 1 && !FUSED) ? 2 : 1;
    static constexpr int STAGE_BYTES = FUSED ? TILE * (4 + VB) : (VROUNDS == 2 ? TILE * 4 : TILE * ((VB == 8 || KW == 2) ? 8 : 4));
    // POS = 1, the position-chain form of the pass (sorts planned with PF_POS): persistent workgroups that count the next
    // pass's digit per output position segment while they scatter — table [NCH][256] kept for the workgroup's whole life —
    // and derive the pass's digit starts from the counts of the pass before (256 words + 8)
    // POS = 2: the same without the counting and its table — the last pass of such a sort, on full-size tiles
    // (POS = 1 keeps its counters PACKED, two 16-bit counts per word — the next-digit table 8 KiB, the per-wave rank counters
    //  4 KiB — so that a full 512 x 32 tile leaves room for a second workgroup on the CU: 79.2 KiB.  Round 3 counted in 32-bit
    //  words on 512 x 24 tiles and paid 0.08 ms per pass for the smaller tile, profiles/r04_pos_packed_counters.txt.)
    //  Only where the 32-bit counters do not fit beside a second workgroup: the packed form costs ~2 vector instructions per key.)
    static constexpr bool PACKED = POS == 1 && STAGE_BYTES + WAVES * RADIX * 4 + 2 * RADIX * 4 + 64 + NCH * RADIX * 4 + RADIX * 4 + 128 > 80 * 1024;
    static constexpr int WHIST_BYTES = WAVES * RADIX * (PACKED ? 2 : 4);
    static constexpr int POS_BYTES = POS == 1 ? (NCH * RADIX * (PACKED ? 2 : 4) + RADIX * 4 + 128) : POS == 2 ? (RADIX * 4 + 32) : 0;
    static constexpr int LDS_BYTES = STAGE_BYTES + WHIST_BYTES + 2 * RADIX * 4 + 64 + POS_BYTES + GS_ABL_COUNT_LDS;
    // residency we ask the register allocator for: as many workgroups per CU as
    // LDS (160 KiB) and the 2048-thread limit admit, so that one workgroup's
    // look-back wait is covered by its neighbours' work
    static constexpr int BPC_LDS = (160 * 1024) / LDS_BYTES;
    static constexpr int BPC_THR = 2048 / THREADS;
    static constexpr int BPC_RAW = BPC_LDS < BPC_THR ? BPC_LDS : BPC_THR;
    static constexpr int BPC = BPC_RAW < 1 ? 1 : BPC_RAW;
    static constexpr int WAVES_PER_SIMD_RAW = BPC * THREADS / 256;
    // never ask for fewer registers than the unrolled tile needs (~1.5 regs/key + temps)
    static constexpr int VGPR_NEED = KPT * ((VB == 8 ? 3 : 2) + (KW == 2 ? 1 : 0)) + 32;
    static constexpr int WAVES_PER_SIMD_CAP = 512 / VGPR_NEED < 1 ? 1 : 512 / VGPR_NEED;
    static constexpr int WAVES_PER_SIMD =
        WAVES_PER_SIMD_RAW < WAVES_PER_SIMD_CAP ? WAVES_PER_SIMD_RAW : WAVES_PER_SIMD_CAP;
};

// The pass as a device function: one tile (PERSIST = false: one workgroup per tile, the grid covers the tiles) or tile after
// tile until every chain is claimed (PERSIST = true).  s_raw: Cfg::LDS_BYTES of LDS.
// Mode word of a DigitBinningPass launch: what the HOST knows about the launch (everything the DEVICE decided is in the pass's flag
// word PF_* of its info block).  One launch carries at most the bits of its family: plain passes BM_REVERSE / BM_PLANNED /
// BM_ZERO_HIST (+ BM_IF_SKEW / BM_IF_EVEN for the two forms of the 8-byte-value pass); launches of a sort that may run on position
// chains add BM_FORMS; the two launches the two-level plan shares with the LSD plan add BM_INFO_SHIFT / BM_INFO_CHAINS / BM_ZERO_DESC23.
constexpr uint32_t BM_REVERSE = 1;        // descending: reversed output index — with BM_PLANNED only on the plan's last pass (PF_LAST)
constexpr uint32_t BM_PLANNED = 2;        // part of a full sort: the flag word decides whether the pass runs (PF_SKIP) and which buffer it reads (PF_SRC_ALT)
constexpr uint32_t BM_ZERO_HIST = 4;      // first pass launched after the Scan: hands the HIST region back zeroed (see global_histogram_kernel)
constexpr uint32_t BM_IF_SKEW = 16;       // one of two forms of the pass: works only if the pass is flagged PF_SKEW ...
constexpr uint32_t BM_IF_EVEN = 32;       // ... only if it is not
constexpr uint32_t BM_FORMS = 64;         // the pass is also launched in its position-chain form: this launch works only if PF_POS matches its POS
constexpr uint32_t BM_INFO_SHIFT = 128;   // the digit's bit position comes from the info block (I_SHIFT), not from shift_full
constexpr uint32_t BM_INFO_CHAINS = 256;  // the chain count comes from the info block BEFORE the first ticket (I_NCH may be CHMAX: the two-level plan's second pass)
constexpr uint32_t BM_ZERO_DESC23 = 512;  // LSD pass 1 of a sort that was offered the two-level plan: zeroes the descriptor regions of passes 2 and 3 if the LSD plan runs
// (experiment builds, GS_EXP & 1024 / 2048, reuse bits 256 .. 2048 as run-time ablation switches of plain LSD launches: onesweep_ablation.hpp)

// ---------------------------------------------------------------------------
// binning_body, phase "load": the tile's keys, wave-striped (lane l of wave w holds keys tile_base + 64 KPT w + 64 i + l: coalesced
// 256 B per wave instruction, 512 B for 64-bit keys), in radix-sortable form.  key[] is the word that holds the pass's digit, key2[]
// (64-bit keys) the other one, which just travels along.  A partial tile [lo, hi) masks the slots outside.
// ---------------------------------------------------------------------------
template <int KPT, int VB, int KT>
__device__ __forceinline__ void bin_load_keys(const uint32_t* __restrict__ keys_in, const uint32_t my_base, const uint32_t lo, const uint32_t hi,
                                               const bool full, const bool hi_word, uint32_t (&key)[KPT],
                                               uint32_t (&key2)[KeyWords<KT>::value == 2 ? KPT : 1]) {
    constexpr int KW = KeyWords<KT>::value;
    if constexpr (KW == 2) {
        const uint2* kin2 = reinterpret_cast<const uint2*>(keys_in);
        uint2 raw[KPT];
        if (GS_LIKELY(full)) {
#pragma unroll
            for (int i = 0; i < KPT; ++i) raw[i] = ld_stream<VB == 0>(kin2 + my_base + i * 64u);
        } else {  // unconditional loads on a clamped index, masked below (see the 32-bit form)
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                const uint32_t idx = my_base + i * 64u;
                raw[i] = ld_stream<VB == 0>(kin2 + (idx < lo ? lo : (idx >= hi ? hi - 1u : idx)));
            }
        }
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const uint32_t idx = my_base + i * 64u;
            const uint2 b = to_bits2<KT>(raw[i]);
            key[i] = hi_word ? b.y : b.x;
            key2[i] = hi_word ? b.x : b.y;
            if (!full) key[i] = idx < lo ? 0u : (idx >= hi ? 0xffffffffu : key[i]);
        }
    } else if (GS_LIKELY(full)) {
#pragma unroll
        for (int i = 0; i < KPT; ++i) key[i] = to_bits<KT>(ld_stream<VB == 0>(keys_in + my_base + i * 64u));
    } else {
        // Masked slots become dummy keys that are never written: in FRONT of the segment
        // all-zero bits (digit 0: being first in array order they open the digit-0 run, stage
        // slots [0, head)), BEHIND it all-one bits (digit 255: they close the last run).
        // The loads are UNCONDITIONAL on a clamped index and masked afterwards: guarded loads are issued one
        // at a time (a wait after each), which made every partial tile ~20 us — the whole pass at mid sizes,
        // where each chain is one or two partial tiles.
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const uint32_t idx = my_base + i * 64u;
            key[i] = ld_stream<VB == 0>(keys_in + (idx < lo ? lo : (idx >= hi ? hi - 1u : idx)));
        }
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const uint32_t idx = my_base + i * 64u;
            key[i] = idx < lo ? 0u : (idx >= hi ? 0xffffffffu : to_bits<KT>(key[i]));
        }
    }
}

// ---------------------------------------------------------------------------
// binning_body, phase "rank": every key's rank among the keys of its digit inside this wave, two 16-bit ranks per register of offp[]
// (later: tile-local positions); the wave's counters `whist` end up holding its digit counts.  RANK 0: the reference's ballot
// multi-split (OneSweep.cu:207-253) for 64 lanes.  RANK 1: one returning LDS add per key — with the forms for crowded waves
// (presorted input), partial tiles (the dummies behind the segment are not ranked) and skewed passes (the pass's most frequent digit
// `skew_digit` never touches the LDS).  PK: two waves share a counter word, one 16-bit half each (`wsh` = this wave's shift).
// ---------------------------------------------------------------------------
template <int KPT, int RANK, bool PK>
__device__ __forceinline__ void bin_rank_keys(const uint32_t (&key)[KPT], uint32_t (&offp)[KPT / 2], const uint32_t shift, uint32_t* __restrict__ whist,
                                               const uint32_t wsh, const uint32_t pflags, const bool full, const uint32_t my_base, const uint32_t hi,
                                               const uint32_t skew_digit, const uint32_t lane) {
    if constexpr (RANK == 0) {
        // Wave-level multi-split with 64-lane ballots: each lane finds its peers
        // (lanes holding the same digit) with 8 ballots, ranks itself among them
        // with mbcnt, and the LAST peer bumps the wave's private LDS counter.  LDS
        // operations of one wave execute in issue order, so the plain read of round
        // i+1 sees the write of round i; the asm clobber only pins the compiler.
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const uint32_t d = (key[i] >> shift) & 255u;
            uint32_t acc_lo = 0, acc_hi = 0;  // bit l set <=> lane l's digit differs from mine
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t B = (uint32_t)__builtin_amdgcn_sbfe((int32_t)key[i], shift + k, 1);  // 0 or ~0
                const unsigned long long b = __builtin_amdgcn_ballot_w64(B != 0u);
                acc_lo = __builtin_amdgcn_bitop3_b32(acc_lo, (uint32_t)b, B, 0xF6);  // acc | (b ^ B)
                acc_hi = __builtin_amdgcn_bitop3_b32(acc_hi, (uint32_t)(b >> 32), B, 0xF6);
            }
            const uint32_t plo = ~acc_lo, phi = ~acc_hi;  // peers: lanes with my digit
            const uint32_t below = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
            const uint32_t total = __popc(plo) + __popc(phi);
            const uint32_t pre = whist[d];                    // same value for all peers (LDS broadcast)
            if (below == total - 1u) whist[d] = pre + total;  // last peer bumps the wave's counter
            asm volatile("" ::: "memory");
            offp[i >> 1] |= (pre + below) << (16 * (i & 1));
        }
    } else {
        // One returning LDS atomic per key on the wave-private counter.  Correct only
        // where the LDS hands same-address lanes of ONE wave-instruction their
        // results in ascending lane order; gs_selftest_lds_atomic_order() probes
        // exactly that on the device before this path is ever selected.
      {
        // one returning LDS add on the wave's counter of digit d
        auto rank_add = [&](uint32_t d) -> uint32_t {
            const uint32_t r = __hip_atomic_fetch_add(&whist[d], 1u << wsh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return PK ? (r >> wsh) & 0xffffu : r;
        };
        // Crowded waves: presorted or clustered input puts the SAME digit in most lanes of a wave (a sorted tile holds one or two
        // values of every byte above its own span) although no digit dominates the pass as a whole (PF_SKEW is clear) — and 64
        // lanes on one LDS counter are served one after the other: 1.10 ms per pass on sorted keys against 0.43 (profiles/
        // r05_sorted_inputs.txt).  One probe per wave and tile (its first and its last key round): a wave that finds >= 16 lanes on
        // the first lane's digit ranks the whole tile with the first lane's digit aggregated — those lanes take ballot ranks on top of
        // ONE add of their count by their first lane, every other lane adds for itself in the same instruction (different
        // counters: nothing changes for them, and ranks stay in lane order, i.e. stable).
        bool crowded = false;  // wave-uniform
        if ((pflags & PF_SKEW) == 0u && full) {
            const uint32_t da = (key[0] >> shift) & 255u, db = (key[KPT - 1] >> shift) & 255u;
            crowded = __popcll(__builtin_amdgcn_ballot_w64(da == (uint32_t)__builtin_amdgcn_readfirstlane((int)da))) >= 16 ||
                      __popcll(__builtin_amdgcn_ballot_w64(db == (uint32_t)__builtin_amdgcn_readfirstlane((int)db))) >= 16;
        }
        if (GS_LIKELY((pflags & PF_SKEW) == 0u && full && !crowded)) {  // uniform per pass (set by scan_kernel)
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                const uint32_t d = (key[i] >> shift) & 255u;
                const uint32_t r = rank_add(d);
                offp[i >> 1] |= r << (16 * (i & 1));
            }
        } else if ((pflags & PF_SKEW) == 0u && full) {
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                const uint32_t d = (key[i] >> shift) & 255u;
                const uint32_t f = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
                const unsigned long long m = __builtin_amdgcn_ballot_w64(d == f);  // (lane 0 is in it)
                const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                const bool in_m = d == f;
                uint32_t r = 0;
                if (!in_m || below == 0u) {
                    const uint32_t inc = in_m ? (uint32_t)__popcll(m) : 1u;
                    const uint32_t w = __hip_atomic_fetch_add(&whist[d], inc << wsh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    r = PK ? (w >> wsh) & 0xffffu : w;
                }
                const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)r, (int)__builtin_ctzll(m));
                offp[i >> 1] |= (in_m ? base + below : r) << (16 * (i & 1));
            }
        } else if ((pflags & PF_SKEW) == 0u) {
            // Partial tile: the dummies behind the segment take no part at all (mask_tail below).  Ranked like
            // keys they would put up to 64 lanes x KPT rounds on the single counter of digit 255 — ~12 us, on
            // the last tile of every chain: the tail of every pass, and most of a pass at mid sizes.
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                if (my_base + i * 64u < hi) {
                    const uint32_t d = (key[i] >> shift) & 255u;
                    const uint32_t r = rank_add(d);
                    offp[i >> 1] |= r << (16 * (i & 1));
                }
            }
        } else {
            // Skewed pass: the keys holding the pass's MOST FREQUENT digit value (known to scan_kernel from the
            // histogram: I_MODE, the same for every wave of the pass) never touch the LDS — their rank is the wave's
            // running count of such keys (a scalar) plus mbcnt of one ballot, and the counter of that digit, which
            // nobody else adds to, is written once at the end.  All other lanes add 1 for themselves in the same
            // round.  Nothing depends on a returned value, so the atomics of a chunk are issued back to back.
            // (The earlier form learned the dominant digit per wave while ranking — ballots, a relearn branch and a
            // leader election per key: load + rank 6.6 us per 16 384-key tile at entropy preset 3 against 3.3 us now and
            // 2.4 us for uniform keys, profiles/r02_skew_rank_fixed_mode.txt.)
            constexpr int SKEW_CHUNK = KPT % 8 == 0 ? 8 : 4;
            static_assert(KPT % SKEW_CHUNK == 0, "KPT must be a multiple of the skew chunk");
            const uint32_t sd = skew_digit;  // wave-uniform (scan_kernel's I_MODE, or the position-chain plan's mode digit)
            uint32_t run = 0;  // wave-uniform: keys of digit sd in this wave so far
#pragma unroll
            for (int c = 0; c < KPT; c += SKEW_CHUNK) {
                uint32_t ret[SKEW_CHUNK];
#pragma unroll
                for (int j = 0; j < SKEW_CHUNK; ++j) {
                    const uint32_t d = (key[c + j] >> shift) & 255u;
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(d == sd);
                    ret[j] = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, run));
                    run += (uint32_t)__popcll(m);
                    // (PK: the counter's WORD comes back; the wave's half is taken out below, outside the branch — inside, every
                    //  add waited for its own return before the next one was issued.  The other lanes' ranks go through the same
                    //  extraction: they are shifted into the half here.)
                    if constexpr (PK) ret[j] <<= wsh;
                    if (d != sd) ret[j] = __hip_atomic_fetch_add(&whist[d], 1u << wsh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
#pragma unroll
                for (int j = 0; j < SKEW_CHUNK; ++j) {
                    const int i = c + j;
                    if constexpr (PK) ret[j] = (ret[j] >> wsh) & 0xffffu;
                    if (i & 1) offp[i >> 1] |= ret[j] << 16; else offp[i >> 1] |= ret[j];
                }
            }
            if (lane == 0 && sd < RADIX) {
                if constexpr (PK) atomicAdd(&whist[sd], run << wsh);  // (nobody of this wave added to its half; the other half is the neighbour wave's)
                else whist[sd] = run;
            }
        }
      }
    }
}

// ---------------------------------------------------------------------------
// binning_body, phase "claim", second half: the workgroup's own chain is fully claimed — take a tile of ANOTHER chain that still has
// unclaimed tiles.  Leaves the chain in s_misc[0] and the ticket in s_misc[1] (~0: every chain is fully claimed); the caller reads
// them behind a barrier.  `chain` = the chain just found exhausted (the linear search starts behind it).
// ---------------------------------------------------------------------------
template <uint32_t THREADS, uint32_t TILE>
__device__ __forceinline__ void bin_steal_tile(const uint32_t nch, const uint32_t chain, const uint32_t* __restrict__ info, uint32_t* __restrict__ counters,
                                                uint32_t* __restrict__ s_misc) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    // Steal: wave 0 looks at ALL chains in one parallel round trip (lane x = chain x); a serial
    // scan with dependent sc1 loads cost ~22 us per exhausted workgroup and stretched every pass's tail.
    __syncthreads();
    if (nch > 64u) {
        // More chains than a wave has lanes (the two-level plan's second pass: CHMAX chains, THREADS >= CHMAX): thread x looks at
        // chain x; the open chain nearest to a per-workgroup starting point (Fibonacci hashing of the workgroup id: stealers
        // spread over the open chains) is tried with one ticket, until a ticket holds or no chain is open.
        const uint32_t start = ((blockIdx.x * 0x9E3779B1u) >> 16) & (nch - 1u);
        for (;;) {
            if (tid == 0) s_misc[0] = 0xffffffffu;
            __syncthreads();
            if (tid < nch) {
                const uint32_t tx = chain_tiles(info[I_START + tid], info[I_END + tid], TILE);
                if (ld_agent(&counters[tid * COUNTER_STRIDE]) < tx) atomicMin(&s_misc[0], (((tid - start) & (nch - 1u)) << 16) | tid);
            }
            __syncthreads();
            const uint32_t cand = uni(s_misc[0]);
            if (cand == 0xffffffffu) {  // every chain is fully claimed
                if (tid == 0) s_misc[1] = 0xffffffffu;
                break;
            }
            const uint32_t x = cand & 0xffffu;
            if (tid == 0) {
                const uint32_t t = atomicAdd(&counters[x * COUNTER_STRIDE], 1u);
                s_misc[1] = t < chain_tiles(info[I_START + x], info[I_END + x], TILE) ? t : 0xffffffffu;
            }
            __syncthreads();
            if (uni(s_misc[1]) != 0xffffffffu) {
                if (tid == 0) s_misc[0] = x;
                break;
            }
        }
    } else if (wave == 0) {
        uint32_t tiles_x = 0;
        bool open = false;
        if (lane < nch) {
            tiles_x = chain_tiles(info[I_START + lane], info[I_END + lane], TILE);
            open = ld_agent(&counters[lane * COUNTER_STRIDE]) < tiles_x;
        }
        unsigned long long m = __builtin_amdgcn_ballot_w64(open);
        uint32_t got_x = 0, got_t = 0xffffffffu;
        // First try: a chain drawn in proportion to the chains' tile counts (Fibonacci hashing of the
        // workgroup id: consecutive ids spread evenly over the cumulative tile range).  With skewed
        // digit groups most workgroups land here — their own chain is tiny — and the big chains still
        // get their workgroups interleaved, in one atomic instead of a scan of the open chains.
        if (m) {
            const uint32_t incl = wave_inclusive_scan(tiles_x, lane);
            const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
            const uint32_t v = (uint32_t)(((unsigned long long)(blockIdx.x * 0x9E3779B1u) * total) >> 32);
            const unsigned long long ge = __builtin_amdgcn_ballot_w64(lane < nch && incl > v);
            const uint32_t x = ge ? (uint32_t)__builtin_ctzll(ge) : 0u;
            if ((m >> x) & 1ull) {
                uint32_t t = 0;
                if (lane == 0) t = atomicAdd(&counters[x * COUNTER_STRIDE], 1u);
                t = __builtin_amdgcn_readfirstlane(t);
                if (t < __builtin_amdgcn_readlane(tiles_x, x)) { got_x = x; got_t = t; m = 0; }
                else m &= ~(1ull << x);
            }
        }
        while (m) {  // wave-uniform: try the open chains one by one, starting after our own
            const unsigned long long above = m & ~((2ull << chain) - 1ull);  // chain < NCH <= 32
            const uint32_t x = (uint32_t)__builtin_ctzll(above ? above : m);
            uint32_t t = 0;
            if (lane == 0) t = atomicAdd(&counters[x * COUNTER_STRIDE], 1u);
            t = __builtin_amdgcn_readfirstlane(t);
            const uint32_t tx = __builtin_amdgcn_readlane(tiles_x, x);
            if (t < tx) { got_x = x; got_t = t; break; }
            m &= ~(1ull << x);
        }
        if (lane == 0) { s_misc[0] = got_x; s_misc[1] = got_t; }
    }
}

// ---------------------------------------------------------------------------
// binning_body, phase "values": the tile's values in the keys' wave-striped arrangement — at the start of the tile for the fused
// forms, late (the key registers are dead by then and the loads fly while the keys are scattered) for the others.  Clamped,
// unconditional loads on a partial tile (see the key loads); slots outside [lo, hi) are never written out.
// ---------------------------------------------------------------------------
template <int KPT, typename V>
__device__ __forceinline__ void bin_load_values(const V* __restrict__ vals_in, const uint32_t my_base, const uint32_t lo, const uint32_t hi, const bool full,
                                                 V (&val)[KPT]) {
    if (GS_LIKELY(full)) {
#pragma unroll
        for (int i = 0; i < KPT; ++i) val[i] = ld_stream<false>(vals_in + my_base + i * 64u);
    } else {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const uint32_t idx = my_base + i * 64u;
            val[i] = ld_stream<false>(vals_in + (idx < lo ? lo : (idx >= hi ? hi - 1u : idx)));
        }
    }
}

// ---------------------------------------------------------------------------
// binning_body, phase "count-next" (POS == 1), once per workgroup: the next-pass digit this workgroup does NOT count while it scatters
// (its counts are recovered at the flush from the keys written per segment).  Every wave looks at 64 staged keys of its own (`mine` =
// this lane's), eight candidate lanes each; a digit that at least 8 of the 64 share is a find, the most frequent find wins
// (s_pos[5] = (popcount << 8) | digit) and becomes s_pos[4]; without a find the choice is left to the next tile.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void bin_elect_left_out_digit(const uint32_t* __restrict__ mine, const uint32_t next_shift, uint32_t* __restrict__ s_pos) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    __syncthreads();  // (everybody has read s_pos[4])
    {
        const uint32_t d0 = (*mine >> next_shift) & 255u;
        uint32_t best = 0;  // (popcount << 8) | digit
#pragma unroll
        for (int c = 0; c < 64; c += 8) {
            const uint32_t cand = (uint32_t)__builtin_amdgcn_readlane((int)d0, c);
            const uint32_t pc = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(d0 == cand));
            best = ((pc << 8) | cand) > best ? ((pc << 8) | cand) : best;
        }
        if (lane == 0 && (best >> 8) >= 8u) atomicMax(&s_pos[5], best);
    }
    __syncthreads();
    if (tid == 0 && s_pos[5] != 0u) s_pos[4] = s_pos[5] & 255u;
    __syncthreads();
}

// ---------------------------------------------------------------------------
// binning_body, phase "entry": what a LAUNCH owes the sort whatever its tiles do — BM_ZERO_HIST: the first pass launched after the Scan
// hands the HIST region back zeroed (see global_histogram_kernel); BM_ZERO_DESC23: LSD pass 1 of a sort that was offered the two-level
// plan zeroes the descriptor regions of passes 2 and 3 if the LSD plan runs — and the licences: a pass launched in several forms
// (BM_IF_SKEW / BM_IF_EVEN: the two forms of the 8-byte-value pass; BM_FORMS: plain and position-chain form) works in exactly one of
// them, the one the pass's flag word (PF_SKEW, PF_POS: decided on the device) names.  Returns false for the others.
// ---------------------------------------------------------------------------
template <uint32_t THREADS, int POS>
__device__ __forceinline__ bool bin_entry(const uint32_t mode, const uint32_t* __restrict__ info, uint32_t* __restrict__ desc, uint32_t* __restrict__ hsub) {
    const uint32_t tid = threadIdx.x;
    if (mode & BM_ZERO_HIST) {  // first pass launched after the Scan: hand the HIST region back zeroed (see global_histogram_kernel)
        // (grid-stride: a small grid of a 256-thread tuning shape does not cover the 8200 16-byte words with one store per thread)
        for (uint32_t i = blockIdx.x * THREADS + tid; i < HIST_WORDS / 4; i += gridDim.x * THREADS)
            reinterpret_cast<uint4*>(hsub - HIST_WORDS)[i] = uint4{0u, 0u, 0u, 0u};
    }
    if (mode & BM_ZERO_DESC23) {
        // A sort that was offered the two-level plan and runs on the LSD passes after all (PF_POS is set): the histogram kernel zeroed
        // only the descriptor regions both plans use (passes 0 and 1); this launch — LSD pass 1 — zeroes the regions of passes 2 and 3
        // beside its own work (they lie behind its own region; nobody touches them before pass 2 starts).
        if (__builtin_amdgcn_readfirstlane((int)info[PASS_FLAGS]) & (int)PF_POS) {
            const uint32_t stride = (uint32_t)__builtin_amdgcn_readfirstlane((int)info[I_DSTRIDE]);
            uint4* z = reinterpret_cast<uint4*>(desc + stride);
            for (uint32_t i = blockIdx.x * THREADS + tid; i < stride / 2u; i += gridDim.x * THREADS) z[i] = uint4{0u, 0u, 0u, 0u};  // 2 x stride words
        }
    }
    if (mode & (BM_IF_SKEW | BM_IF_EVEN)) {  // one of two launches of this pass: the flag word says which one works (before any ticket is drawn)
        const bool skewed = (__builtin_amdgcn_readfirstlane((int)info[PASS_FLAGS]) & (int)PF_SKEW) != 0;
        if (skewed != ((mode & BM_IF_SKEW) != 0u)) return false;
    }
    if (mode & BM_FORMS) {  // the pass is launched in several forms: the plan says whether the position-chain form or the others work
        const bool planned_pos = (__builtin_amdgcn_readfirstlane((int)info[PASS_FLAGS]) & (int)PF_POS) != 0;
        if (planned_pos != (POS != 0)) return false;
    }
    return true;
}

// MAP of binning_body.  The phases with a narrow interface are functions of their own above (bin_entry, bin_steal_tile, bin_load_keys,
// bin_load_values, bin_rank_keys, bin_elect_left_out_digit: round 6); what is left in the body works on the tile's state all at once —
// key[KPT], the packed ranks offp[], the per-digit prefixes and bases in LDS, the chain's descriptor rows — and would pass ~30 of them
// by reference.  The compile-time switches prune each instantiation to the phases its kernel needs.  Phases, in order:
//   entry      bin_entry: BM_ZERO_HIST / BM_ZERO_DESC23 zeroing; BM_IF_* / BM_FORMS: does THIS form of the pass work?
//   POS setup  (POS != 0) digit starts / skew / mode digit from CNEXT of the pass before; the workgroup's next-digit table
//   [tile loop, PERSIST: until every chain is claimed]
//   claim      ticket on the workgroup's chain (BM_INFO_CHAINS: chain groups of a CHMAX-chain pass), else bin_steal_tile
//   load       bin_load_keys (KW == 2: 64-bit keys carry their other word along); partial tiles masked; fused forms: bin_load_values
//   rank       bin_rank_keys — RANK 0: eight-ballot multi-split; RANK 1: one returning LDS add per key (crowded waves, partial tiles, skew)
//   reduce     per-digit exclusive prefix over waves, tile total, REDUCTION row published, digit scan
//   stage      keys to LDS in digit order (VR == 2: 8-byte values in two staging rounds)
//   look-back  one digit per thread walks the chain's rows back to an INCLUSIVE one; GS_FALLBACK: recount a row nobody published
//   values     (VB != 0, not fused) bin_load_values: fetched late, staged behind the keys
//   count-next (POS == 1) bin_elect_left_out_digit once; the next pass's digit counted per output position segment while scattering
//   scatter    stage slot i -> global base of its digit + i (reverse: the descending rule on the plan's last pass), coalesced runs
//   exit       (POS == 1) CNEXT flush; trace / fault hooks of experiment builds
template <int THREADS, int KPT, int VB, int KT, int RANK, int VR, int POS, bool PERSIST>
__device__ __forceinline__ void binning_body(
    unsigned char* s_raw,
    uint32_t* keys_a, uint32_t* keys_b, void* vals_a, void* vals_b,  // the pass reads a and writes b, unless ...
    uint32_t* desc,          // this pass: rows of 256 descriptor words; chain x starts at row_base[x]
    uint32_t* counters,      // this pass: one ticket counter per chain
    const uint32_t* info,    // this pass: the info block written by scan_kernel
    uint32_t* hsub,          // SLAB_HSUB = CNEXT (PF_POS sorts): read [this pass], added to [the next pass that runs]
    uint32_t* status, uint32_t n, uint32_t shift_full /*bit position of the digit in the key: 0..24, 64-bit keys 0..56*/,
    uint32_t mode /*BM_* bits, above*/) {
    constexpr int KW = KeyWords<KT>::value;
    using Cfg = BinCfg<THREADS, KPT, VB, KW, VR, POS>;
    if (mode & BM_INFO_SHIFT) shift_full = (uint32_t)__builtin_amdgcn_readfirstlane((int)info[I_SHIFT]);
    // Chain a workgroup asks first: blockIdx modulo NCH, in chain GROUP `group`.  The LSD plans have one group (NCH chains).  A pass on
    // CHMAX chains (the two-level plan's second pass: one chain per top-byte bucket) is walked group by group — NCH chains at a time,
    // each by the 1 / NCH of the workgroups that share its lane, moving on to chain + NCH when it is fully claimed — so that at any
    // time ~NCH chains are live with 32 workgroups each, exactly the LSD passes' picture: every (chain, digit) write cursor is fed by
    // a whole row of neighbouring tiles (with all 256 chains live at once, two workgroups each, the pass wrote through 65 536 cursors
    // with two tiles behind each and ran at 0.66 ms instead of 0.47: DRAM pages served 512 bytes per activation, profiles/r05_*).
#ifndef GS_HY_GROUP_CHAINS
#define GS_HY_GROUP_CHAINS NCH  // chains of a CHMAX-chain pass that are live at a time (a power of two >= NCH)
#endif
    const uint32_t nch_info = (mode & BM_INFO_CHAINS) ? (uint32_t)__builtin_amdgcn_readfirstlane((int)info[I_NCH]) : NCH;
    const uint32_t gchains = nch_info > NCH ? (uint32_t)GS_HY_GROUP_CHAINS : NCH;  // uniform
    const uint32_t ngroups = (mode & BM_INFO_CHAINS) ? nch_info / gchains : 1u;
    uint32_t group = 0;  // uniform; persistent workgroups keep it across their tiles
    static_assert(!POS || (KW == 1 && RANK == 1 && (VB == 0 || VB == 4 || (VB == 8 && VR == 2))),
                  "the position-chain forms exist for 32-bit keys, keys-only, with 4-byte values (staged behind the keys) or with 8-byte values (two staging rounds), LDS-atomic ranking");
    using V = typename ValT<VB>::type;
    constexpr int WAVES = Cfg::WAVES;
    constexpr uint32_t TILE = Cfg::TILE;
    // the digit's position inside the word that holds it; 64-bit keys: hi_word says which word that is.  key[] is
    // always that word (radix-sortable form), key2[] the other one, which just travels along.
    const uint32_t shift = shift_full & 31u;
    const bool hi_word = KW == 2 && shift_full >= 32u;
    static_assert(THREADS >= 256 && THREADS % 64 == 0, "need >= 256 threads");
    static_assert(KPT % 4 == 0 && TILE <= 65536, "offsets are packed 2 x 16 bit, digits 4 x 8 bit");
    auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };

    static_assert(POS == 0 || PERSIST, "the position-chain forms keep state across their tiles");
    // packed counters: two waves share a counter word and only the returning-atomic ranking adds to its own half (the ballot
    // ranking stores whole words); the overflow guard of the packed next-digit table (0xC000 + TILE <= 0xFFFF) holds up to 16 384 keys
    static_assert(!(Cfg::PACKED && RANK == 0), "packed per-wave counters need the LDS-atomic ranking");
    static_assert(!Cfg::PACKED || Cfg::TILE <= 16384, "packed next-digit counters: a tile adds at most 16 384 to a 16-bit counter flushed at 0xC000");
    uint32_t* s_stage = reinterpret_cast<uint32_t*>(s_raw);
    uint32_t* s_whist = reinterpret_cast<uint32_t*>(s_raw + Cfg::STAGE_BYTES);
    constexpr bool PK = Cfg::PACKED;             // two 16-bit counters per word: per-wave rank counters and next-digit table
    uint32_t* s_dpre = s_whist + Cfg::WHIST_BYTES / 4;  // tile-local exclusive digit prefix
    uint32_t* s_gbase = s_dpre + RADIX;          // global base of digit run minus s_dpre
    uint32_t* s_misc = s_gbase + RADIX;          // [0] chain, [1] ticket (~0 = none), [4..7] wave totals of the digit scan,
                                                 // [9..14] the pass's flag/plan words
    uint32_t* s_cnt = s_misc + 16;               // POS: [NCH][256] keys written to position segment x whose next digit is d, 16 bits each
    uint32_t* s_dstart = s_cnt + (POS == 1 ? NCH * RADIX / (PK ? 2 : 1) : 0);  // POS: digit starts of this pass (from the counts of the pass before)
    uint32_t* s_pos = s_dstart + (POS ? RADIX : 0);        // POS: [0] some digit holds > n/16 keys, [2..3] (count << 8 | 255 - digit) max;
                                                           // POS == 1: [4] the next digit this workgroup does NOT count (see count_next), [5] its election, [8..23] keys written per output segment
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;

    if (!bin_entry<(uint32_t)THREADS, POS>(mode, info, desc, hsub)) return;  // launch-level duties; is this form of the pass the one that works?
    // ---- POS: what a workgroup of a position-chain pass sets up once.  Behind the first pass nobody knows the pass's digit
    // counts upfront: CNEXT[this pass][segment][digit], gathered by the pass that wrote this pass's input, is all there
    // is — every workgroup derives the digit starts (s_dstart), whether the pass is skewed and its most frequent digit
    // from it (16 coalesced loads per digit thread); tile 0 of each chain turns it into the chain's row 0.
    bool pos_derived = false, pos_skew = false;  // uniform
    uint32_t pos_mode = 0xffffffffu, cnt_guess = 0xffffffffu;
    const uint32_t* cn_in = hsub + (shift_full >> 3) * HSUB_STRIDE;
    if constexpr (POS != 0) {
        if constexpr (POS == 1) {
            for (uint32_t i = tid; i < NCH * RADIX / (PK ? 2 : 1); i += THREADS) s_cnt[i] = 0;
            if (tid < 32u && tid >= 4u) s_pos[tid] = tid == 4u ? 0xfffffffeu : 0u;  // [4]: not chosen yet
        }
        pos_derived = shift_full != 0u;  // (the first pass of a PF_POS sort is never dropped: its chains come from the Scan kernel)
        if (pos_derived) {
            uint32_t G = 0, incl = 0;
            if (tid < 4) s_pos[tid] = 0;
            if (tid < RADIX) {
#pragma unroll
                for (uint32_t x = 0; x < NCH; ++x) G += cn_in[x * RADIX + tid];
                incl = wave_inclusive_scan_dpp(G);
                if (lane == 63) s_misc[4 + wave] = incl;
            }
            __syncthreads();
            if (tid < RADIX) {
                uint32_t wbase = 0;
                for (uint32_t w = 0; w < wave; ++w) wbase += s_misc[4 + w];
                s_dstart[tid] = wbase + incl - G;
                if (G >= (n >> GS_SKEW_SHIFT) + 1u) {
                    s_pos[0] = 1u;
                    atomicMax(reinterpret_cast<unsigned long long*>(s_pos + 2), ((unsigned long long)G << 8) | (255u - tid));
                }
            }
            __syncthreads();
            pos_skew = uni(s_pos[0]) != 0u;
            pos_mode = pos_skew ? 255u - (uni(s_pos[2]) & 255u) : 0xffffffffu;
        }
    }
    // (POS == 1) the digit position this workgroup counts for, ~0: none — the same word every tile reads below as next_shift
    const uint32_t guard_ns = (POS == 1 && (mode & BM_PLANNED)) ? uni(info[I_NEXT_SHIFT]) : 0xffffffffu;
    GS_TRACE_SETUP();
    // (measured and not kept, PERSIST: the next tile's ticket drawn while this tile is scattered — +0.5 .. 1 %: a tile claimed
    //  4 us before its workgroup starts on it publishes its counts 4 us late for the tiles behind it,
    //  profiles/r03_ab_ticket_ahead.txt)
#pragma unroll 1
    for (;;) {  // PERSIST: one tile after the other until every chain is claimed; otherwise ONE tile per workgroup
    if constexpr (PERSIST) __syncthreads();  // the last tile's readers of the stage and of s_misc are through
    if constexpr (POS == 1 && PK) {
        // Overflow guard of the packed next-digit table: a tile adds at most TILE = 16 384 to a counter, so a counter at or above
        // 0xC000 goes to CNEXT now (and out of the segment's written-keys total, which the left-out digit is recovered from).
        // Nobody adds to the table between the barrier above and this tile's scatter.  Rare: a workgroup sees ~32 tiles, and its
        // most frequent next digit is not counted at all.
        if (guard_ns != 0xffffffffu) {  // uniform
            for (uint32_t i = tid; i < NCH * RADIX / 2; i += THREADS) {
                uint32_t w = s_cnt[i];
                if (GS_UNLIKELY((w & (w << 1) & 0x80008000u) != 0u)) {
                    uint32_t* cn_out = hsub + (guard_ns >> 3) * HSUB_STRIDE;
#pragma unroll
                    for (uint32_t h = 0; h < 2; ++h) {
                        const uint32_t v = (w >> (16u * h)) & 0xffffu;
                        if (v >= 0xC000u) {
                            atomicAdd(&cn_out[2u * i + h], v);
                            atomicSub(&s_pos[8u + (i >> 7)], v);
                            w &= ~(0xffffu << (16u * h));
                        }
                    }
                    s_cnt[i] = w;
                }
            }
        }
    }
    for (uint32_t i = tid; i < Cfg::WHIST_BYTES / 4; i += THREADS) s_whist[i] = 0;
    GS_TRACE(0);
    // ---- claim a tile.  Fast path: ONE returning atomic on the ticket counter of
    // chain blockIdx % NCH (each counter has its own cache line).  Ticket order inside
    // a chain is the start order, so every predecessor of a claimed tile is running.
    // Only when that chain is already fully claimed does thread 0 try the others. ----
    uint32_t chain = (blockIdx.x & (gchains - 1u)) + group * gchains;
    // geometry of the fast-path chain: requested before the ticket is (scalar loads that depend on blockIdx only), so
    // their round trip runs beside the ticket atomic's instead of after the barrier
    const uint32_t seg_start_f = info[I_START + chain], seg_end_f = info[I_END + chain], row_f = info[I_ROW + chain];
    if (tid == 0) {
        s_misc[2] = 0u;  // set by the look-back if it has to give up
        s_misc[8] = 0u;  // row a stuck look-back asks the workgroup to recount (GS_FALLBACK)
        s_misc[1] = atomicAdd(&counters[chain * COUNTER_STRIDE], 1u);
    }
    // An earlier pass of this sort gave up (status word set): its output is incomplete, so positions derived
    // from the upfront histograms no longer bound this pass's writes — do nothing.  Read by another wave, in
    // flight together with the ticket atomic, so it adds no latency.  Same for the pass's flag and plan words.
    if (tid == 64) s_misc[3] = ld_agent(status);
    if (tid >= 128 && tid < 135) s_misc[9 + (tid - 128)] = info[PASS_FLAGS + (tid - 128)];  // flags, nch, next_shift, seglog, -, -, mode
    __syncthreads();
    // Everything below that is the same for the whole workgroup is made SCALAR explicitly (values read from LDS
    // or through a VGPR index are vector registers to the compiler: pointers selected by them cost two VGPRs
    // each and a 64-bit vector add per access, and every branch on them is an exec-mask branch).
    if (uni(s_misc[3]) != STATUS_OK) break;
    const uint32_t pflags = uni(s_misc[9]) | (pos_skew ? PF_SKEW : 0u);
    if ((mode & BM_PLANNED) && (pflags & PF_SKIP)) break;  // identity pass of a full sort
#ifdef GS_STATIC_IO  // A/B aid: the pass always reads a and writes b (run with skip_passes = 0)
    const bool swapped = false;
#else
    const bool swapped = (mode & BM_PLANNED) && (pflags & PF_SRC_ALT);
#endif
    const uint32_t* keys_in = swapped ? keys_b : keys_a;
    uint32_t* keys_out = swapped ? keys_a : keys_b;
    const void* vals_in_ = swapped ? vals_b : vals_a;
    void* vals_out_ = swapped ? vals_a : vals_b;
    const bool reverse = (mode & BM_REVERSE) && (!(mode & BM_PLANNED) || (pflags & PF_LAST));
    const uint32_t nch = uni(s_misc[10]);                                 // chains of this pass
    // POS: bit position of the next running pass's digit (~0: nothing to count) and log2 of its position segments
    const uint32_t next_shift = (POS == 1 && (mode & BM_PLANNED)) ? uni(s_misc[11]) : 0xffffffffu, seglog = uni(s_misc[12]);
    uint32_t tile = uni(s_misc[1]);
    // A chain's tile grid starts at its segment start rounded DOWN to 64 keys, so every
    // wave-load is 256-byte aligned; keys in front of the segment are masked like the tail.
    uint32_t seg_start = uni(seg_start_f), seg_end = uni(seg_end_f), row0 = uni(row_f);
    if (GS_UNLIKELY(tile >= chain_tiles(seg_start, seg_end, TILE))) {  // uniform
        if constexpr (PERSIST) {
            if (group + 1u < ngroups) {  // this lane's chain of the group is fully claimed: on to the next group's
                ++group;
                continue;
            }
        }
        bin_steal_tile<(uint32_t)THREADS, TILE>(nch, chain, info, counters, s_misc);
        __syncthreads();
        chain = uni(s_misc[0]);
        tile = uni(s_misc[1]);
        if (tile == 0xffffffffu) break;  // every chain is fully claimed
        seg_start = uni(info[I_START + chain]);
        seg_end = uni(info[I_END + chain]);
        row0 = uni(info[I_ROW + chain]);
    }
    const uint32_t tile_base = (seg_start & ~63u) + tile * TILE;
    const uint32_t lo = tile_base > seg_start ? tile_base : seg_start;  // valid keys: [lo, hi)
    const uint32_t hi = (seg_end - tile_base < TILE) ? seg_end : tile_base + TILE;
    const uint32_t count = hi - lo;
    const uint32_t head = lo - tile_base;  // masked keys in front (first tile of a chain only)
    const bool full = (count == TILE);
    uint32_t* cdesc = desc + (size_t)row0 * RADIX;  // row 0 of this chain
    // POS behind the first pass: tile 0 of a chain seeds the chain's row 0 — the digit's start plus the chains in front — long
    // before a successor can walk that far (a walk that gets there first polls the zero word like any row not yet there)
    if constexpr (POS != 0) {
        if (GS_UNLIKELY(pos_derived && tile == 0u && tid < RADIX)) {
            uint32_t seed = s_dstart[tid];
            for (uint32_t x = 0; x < chain; ++x) seed += cn_in[x * RADIX + tid];
            st_agent(&cdesc[tid], (seed << 2) | FLAG_INCLUSIVE);
        }
    }
    GS_TRACE(1);
    uint32_t early_row = 0;
    GS_ABL_EARLY_ROW(early_row);

    // ---- load (wave-striped, coalesced 256 B per wave-instruction; 64-bit keys: 512 B) ----
    uint32_t key[KPT];
    uint32_t key2[KW == 2 ? KPT : 1];
    const uint32_t my_base = tile_base + wave * (64u * KPT) + lane;
    bin_load_keys<KPT, VB, KT>(keys_in, my_base, lo, hi, full, hi_word, key, key2);

    V val[VB != 0 ? KPT : 1];
    if constexpr (Cfg::FUSED) bin_load_values<KPT, V>(reinterpret_cast<const V*>(vals_in_), my_base, lo, hi, full, val);  // the values come along from the start

    // ---- rank every key among the keys of its digit inside this wave ----
    // offp[] holds two 16-bit ranks (later: tile-local positions) per register.
    // PK: waves 2k and 2k + 1 share the 256 words of a counter block, one half each (the most a wave counts is 64 x KPT <= 2048
    // and a stage slot is < TILE <= 65536, so a half never carries into the other).  The half is a property of the WAVE: its shift
    // and its increment are scalars, and no two digits of one wave meet on a word.
    // (Measured and not kept: the ranking and staging code once per half, picked by a uniform branch on the wave's parity, so that the
    //  shift is a compile-time constant — 34 spilled registers and 0.55 -> 0.58 ms per counting pass, profiles/r04_pos_packed_counters.txt.)
    const uint32_t wsh = PK ? (uint32_t)__builtin_amdgcn_readfirstlane((int)((wave & 1u) * 16u)) : 0u;
    uint32_t* whist = s_whist + (PK ? wave >> 1 : wave) * RADIX;
    uint32_t offp[KPT / 2];
#pragma unroll
    for (int i = 0; i < KPT / 2; ++i) offp[i] = 0;
    bin_rank_keys<KPT, RANK, PK>(key, offp, shift, whist, wsh, pflags, full, my_base, hi,
                                 ((pflags & PF_SKEW) != 0u) ? (pos_derived ? pos_mode : uni(s_misc[15])) : 0u, lane);
    GS_TRACE(2);
    __syncthreads();

    // ---- per-digit: exclusive prefix over waves, tile total, publish, digit scan ----
    uint32_t tile_total = 0, scan_incl = 0, dpre = 0, dummies = 0;
    // the tile's trailing dummies are not ranked (and later not staged) in the plain LDS-atomic ranking path
    const bool tail_unranked = RANK == 1 && !full && (pflags & PF_SKEW) == 0u;
    if (tid < RADIX) {
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            if constexpr (PK) {  // (one word holds this digit's counts of waves w and w + 1)
                if ((w & 1) == 0) {
                    const uint32_t c2 = s_whist[(w >> 1) * RADIX + tid];
                    const uint32_t c_even = c2 & 0xffffu, c_odd = c2 >> 16;
                    s_whist[(w >> 1) * RADIX + tid] = run | ((run + c_even) << 16);
                    run += c_even + c_odd;
                }
            } else {
                const uint32_t c = s_whist[w * RADIX + tid];
                s_whist[w * RADIX + tid] = run;
                run += c;
            }
        }
        // published counts are the tile's REAL keys: without the dummies in front (digit 0) and, where they were
        // ranked at all (ballot ranking, skewed passes), without the dummies behind the segment (digit 255) — the
        // same numbers a fallback recount of this tile produces
        dummies = (tid == 0 ? head : 0u) + ((tid == RADIX - 1 && !full && !tail_unranked) ? TILE - head - count : 0u);
        tile_total = run - dummies;
        if (!GS_FAULT_TILE(chain, tile) && !GS_ABL_REPLAY)
            st_agent(&cdesc[(size_t)(tile + 1u) * RADIX + tid], (tile_total << 2) | FLAG_REDUCTION);
        scan_incl = wave_inclusive_scan_dpp(run);
        if (lane == 63) s_misc[4 + wave] = scan_incl;
    }
    __syncthreads();
    if (tid < RADIX) {
        uint32_t wbase = 0;
        for (uint32_t w = 0; w < wave; ++w) wbase += s_misc[4 + w];
        dpre = wbase + scan_incl - (tile_total + dummies);  // stage offset of the run (dummies included)
        s_dpre[tid] = dpre;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            if constexpr (PK) { if ((w & 1) == 0) s_whist[(w >> 1) * RADIX + tid] += dpre * 0x10001u; }  // (both halves stay stage slots: < TILE <= 65536)
            else s_whist[w * RADIX + tid] += dpre;
        }
    }
    __syncthreads();

    GS_TRACE(3);
    // ---- stage keys in LDS, sorted by digit (stable) ----
    // mask_tail: the tile's trailing dummies were not ranked (above) and are not staged
    const bool mask_tail = tail_unranked;
    // fused pairs: slot = (key, value); 4-byte values as one 8-byte LDS word, 8-byte values in a second array
    auto stage_pair = [&](uint32_t slot, uint32_t k, V v) {
        if constexpr (VB == 4) {
            reinterpret_cast<uint2*>(s_raw)[slot] = uint2{k, (uint32_t)v};
        } else {
            s_stage[slot] = k;
            reinterpret_cast<V*>(s_raw + TILE * 4)[slot] = v;
        }
    };
    if (GS_LIKELY(!mask_tail)) {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const uint32_t d = (key[i] >> shift) & 255u;
            const uint32_t lpos = ((offp[i >> 1] >> (16 * (i & 1))) & 0xffffu) + (PK ? (whist[d] >> wsh) & 0xffffu : whist[d]);
            if constexpr (Cfg::FUSED) {
                stage_pair(lpos, key[i], val[i]);
            } else {
                if constexpr (KW == 2) reinterpret_cast<uint2*>(s_raw)[lpos] = uint2{key[i], key2[i]};  // (digit word, other word)
                else s_stage[lpos] = key[i];
                if constexpr (VB != 0) {  // values follow the same positions later
                    if ((i & 1) == 0) offp[i >> 1] = (offp[i >> 1] & 0xffff0000u) | lpos;
                    else offp[i >> 1] = (offp[i >> 1] & 0x0000ffffu) | (lpos << 16);
                }
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const uint32_t d = (key[i] >> shift) & 255u;
            const uint32_t lpos = ((offp[i >> 1] >> (16 * (i & 1))) & 0xffffu) + (PK ? (whist[d] >> wsh) & 0xffffu : whist[d]);
            if constexpr (Cfg::FUSED) {
                if (my_base + i * 64u < hi) stage_pair(lpos, key[i], val[i]);
            } else {
                if (my_base + i * 64u < hi) {
                    if constexpr (KW == 2) reinterpret_cast<uint2*>(s_raw)[lpos] = uint2{key[i], key2[i]};
                    else s_stage[lpos] = key[i];
                }
                if constexpr (VB != 0) {
                    if ((i & 1) == 0) offp[i >> 1] = (offp[i >> 1] & 0xffff0000u) | lpos;
                    else offp[i >> 1] = (offp[i >> 1] & 0x0000ffffu) | (lpos << 16);
                }
            }
        }
    }

    // ---- decoupled look-back inside the chain: one digit per thread ----
    // Row k holds tile k-1's descriptor; row 0 = chain base (INCLUSIVE), so every walk ends at row 0 at the
    // latest.  One row per round trip (walks are ~3 rows with 16 chains).  Measured and not kept: more rows per trip,
    // rows requested before the staging phase, rows through the scalar data path, larger batches in crowded chains
    // (profiles/r02_ab_early_lookback_rows.txt, r02_ab_scalar_lookback.txt).
    uint32_t prev = 0, spins = 0;
    int32_t k = (int32_t)tile;
    bool done = GS_ABL_LOOKBACK_SKIPPED, poisoned = false, finished = tid >= RADIX;
    auto walk = [&](auto nb_tag) {
        constexpr int NB = decltype(nb_tag)::value;
        while (!done) {
            GS_TRACE_TRIP();
            uint32_t v[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int32_t r = k - j < 0 ? 0 : k - j;
                v[j] = ld_agent(&cdesc[(size_t)r * RADIX + tid]);
            }
            bool stalled = false;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if (!done && !stalled) {
                    const uint32_t f = v[j] & FLAG_MASK;
                    if (f == FLAG_INCLUSIVE) { prev += v[j] >> 2; done = true; }
                    else if (f == FLAG_REDUCTION) { prev += v[j] >> 2; --k; }
                    else if (f == FLAG_POISON) { poisoned = true; done = true; }  // a predecessor gave up
                    else stalled = true;  // row k (>= 1: row 0 is seeded before the pass starts) is not there yet
                }
            }
            if (stalled) {
                __builtin_amdgcn_s_sleep(1);
                ++spins;
                if (GS_FALLBACK && k > 0 && spins > FALLBACK_SPINS) {  // (row 0 of a PF_POS chain behind the first pass is seeded by its tile 0)
                    atomicMax(&s_misc[8], (uint32_t)k);  // ask the workgroup to recount tile k-1
                    return;
                }
                if (spins > SPIN_LIMIT || ((spins & 1023u) == 0 && ld_agent(status) != STATUS_OK)) {
                    poisoned = true;  // give up: nothing hangs, nothing is written with a wrong prefix
                    done = true;
                }
            }
        }
    };
    GS_TRACE(4);
    GS_ABL_ASSUME_PREV();
    GS_ABL_EARLY_USE(early_row);
    for (;;) {
        if (!finished) {
            // (measured in round 3 and not kept: four rows per trip in short sorts — 2^20 .. 2^25 keys, where the tiles of a chain
            //  run in step and the walks are ~5 trips: no change, the wait is for the predecessor, profiles/r03_mid_size_timeline.txt)
            walk(IntTag<GS_WALK_ROWS>{});
            if (done) {
                finished = true;
                if (poisoned) {
                    st_agent(status, STATUS_TIMEOUT);
                    s_misc[2] = 1u;  // this tile must not scatter
                }
                if (!GS_FAULT_TILE(chain, tile))
                    st_agent(&cdesc[(size_t)(tile + 1u) * RADIX + tid],
                             poisoned ? FLAG_POISON : (((prev + tile_total) << 2) | FLAG_INCLUSIVE));
                s_gbase[tid] = prev - dpre - (tid == 0 ? head : 0u);  // digit 0's real keys start `head` slots into its run
            }
        }
        __syncthreads();
        if (!GS_FALLBACK) break;
        const uint32_t fb_row = uni(s_misc[8]);  // uniform
        if (GS_LIKELY(fb_row == 0u)) break;
        // ---- fallback: some digit's walk waited FALLBACK_SPINS polls on row fb_row.  The whole workgroup
        // recounts that tile's digits from the pass input (which nobody writes during the pass), offers the
        // counts to everyone as REDUCTION descriptors (compare-and-swap on NOT_READY: whatever the owner
        // publishes later is the same count or its inclusive form), and the stuck walks go on below that row.
        // The per-wave counters are dead once the stage is written (the barrier above), so they are the scratch.
        {
            uint32_t* s_fb = s_whist;
            const uint32_t fbase = (seg_start & ~63u) + (fb_row - 1u) * TILE;
            const uint32_t flo = fbase > seg_start ? fbase : seg_start;
            const uint32_t fhi = (seg_end - fbase < TILE) ? seg_end : fbase + TILE;
            for (uint32_t i = tid; i < RADIX; i += THREADS) s_fb[i] = 0;
            __syncthreads();
            // every thread has read the request (fb_row above) before it is withdrawn; the next request can only
            // be posted after the barrier that ends this block
            if (tid == 0) s_misc[8] = 0u;
            for (uint32_t idx = flo + tid; idx < fhi; idx += THREADS) {
                uint32_t w;
                if constexpr (KW == 2) {
                    const uint2 b = to_bits2<KT>(reinterpret_cast<const uint2*>(keys_in)[idx]);
                    w = hi_word ? b.y : b.x;
                } else {
                    w = to_bits<KT>(keys_in[idx]);
                }
                atomicAdd(&s_fb[(w >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (!finished) {
                if (k == (int32_t)fb_row) {
                    const uint32_t c = s_fb[tid];
                    atomicCAS(&cdesc[(size_t)fb_row * RADIX + tid], 0u, (c << 2) | FLAG_REDUCTION);
                    prev += c;
                    --k;
                }
                spins = 0;
            }
            __syncthreads();
        }
    }
    GS_TRACE(5);
    GS_TRACE_END(chain);
    if (uni(s_misc[2]) != 0u) break;  // look-back gave up (timeout or poisoned predecessor): write nothing
    if constexpr (POS == 1) {
        // keys this tile writes per output position segment, from the digit threads' run geometry (ONE LDS add per run, two for the
        // rare run across a segment boundary): what count_next leaves out — the workgroup's hot next digit — is recovered from these
        // totals when the table is flushed
        if (next_shift != 0xffffffffu && tid < RADIX && tile_total != 0u) {
            const uint32_t x0 = prev >> seglog, x1 = (prev + tile_total - 1u) >> seglog;
            if (x0 == x1) {
                atomicAdd(&s_pos[8u + x0], tile_total);
            } else {  // (a run is shorter than a segment: two segments at most)
                const uint32_t first = ((x0 + 1u) << seglog) - prev;
                atomicAdd(&s_pos[8u + x0], first);
                atomicAdd(&s_pos[8u + x1], tile_total - first);
            }
        }
    }

    // ---- (pairs) fetch this tile's values now: the key registers are dead, and the loads fly
    // while the keys are scattered ----
    if constexpr (VB != 0 && !Cfg::FUSED) bin_load_values<KPT, V>(reinterpret_cast<const V*>(vals_in_), my_base, lo, hi, full, val);

    // ---- POS: count the next pass's digit per output position segment while scattering.  One LDS add per key on the
    // workgroup's table [segment][digit] (flushed once, when the workgroup runs out of tiles); the stage is ordered by
    // THIS digit, so the lanes of one scatter instruction write one or two runs — one segment, with few exceptions — and
    // their next digits are what collides.  The workgroup therefore leaves ONE next digit out — its guess of the most frequent
    // one, fixed at its first tile — and recovers that digit's counts at the flush from the number of keys it wrote to each segment
    // (known per run from the digit threads: s_pos[8..]).  (Round 3, first form: the lanes holding a per-wave guess counted with ONE
    // add of a ballot's popcount — exact, but two ballots, a readlane and a popcount per key: 153 M VALU + 97 M SALU instructions per
    // counting pass against 59 M + 29 M for the plain pass, profiles/r03_rocprofv3_pmc_sq_insts.txt.  Plain adds cost 0.23 / 0.57 ms per pass
    // at entropy presets 3 / 5 against 0.01 ms for uniform keys, profiles/r03_next_digit_count_cost.txt).
    auto count_next = [&](uint32_t kb, uint32_t o, bool valid) {
        const uint32_t dn = (kb >> (next_shift & 31u)) & 255u;
        if constexpr (PK) { if (valid && dn != cnt_guess) atomicAdd(&s_cnt[((o >> seglog) << 7) + (dn >> 1)], 1u << ((dn & 1u) * 16u)); }
        else { if (valid && dn != cnt_guess) atomicAdd(&s_cnt[((o >> seglog) << 8) + dn], 1u); }
#ifdef GS_ABL_CNT  // ablation (timing only, results stay exact): bit 0: the address arithmetic once more; bit 1: the LDS add once more (adds 0)
        if (GS_ABL_CNT & 1) {
            uint32_t kb2 = kb, o2 = o;
            asm volatile("" : "+v"(kb2), "+v"(o2));
            const uint32_t dn2 = (kb2 >> (next_shift & 31u)) & 255u;
            uint32_t idx2 = ((o2 >> seglog) << 7) + (dn2 >> 1), inc2 = 1u << ((dn2 & 1u) * 16u);
            if (valid && dn2 != cnt_guess) asm volatile("" :: "v"(idx2), "v"(inc2));
        }
        if (GS_ABL_CNT & 2) {
            if constexpr (PK) { if (valid && dn != cnt_guess) atomicAdd(&s_cnt[((o >> seglog) << 7) + (dn >> 1)], 0u); }
        }
#endif
    };
    const bool counting = POS == 1 && next_shift != 0xffffffffu;  // uniform
    if constexpr (POS == 1) {
        if (counting) {
            // the digit this workgroup leaves out, chosen ONCE: a next digit that at least 8 of 64 staged keys share (every wave looks at 64
            // of its own, eight candidate lanes each; the most frequent find wins); a tile without such a digit leaves the choice to the next one
            if (uni(s_pos[4]) == 0xfffffffeu) bin_elect_left_out_digit(s_stage + wave * (TILE / WAVES) + lane, next_shift & 31u, s_pos);
            cnt_guess = uni(s_pos[4]);
            if (cnt_guess == 0xfffffffeu) cnt_guess = 0xffffffffu;  // (no digit left out in this tile: nothing to recover for it)
        }
    }

    // ---- scatter runs to global memory (slot i of the stage -> s_gbase[digit] + i) ----
    // descending (last pass): index n-1-o == (o ^ ~0) + n, folded into two uniform operands so the straight-line
    // scatter serves both orders
    const uint32_t rev_xor = reverse ? 0xffffffffu : 0u, rev_add = reverse ? n : 0u;
    uint32_t digs[VB != 0 ? KPT / 4 : 1];  // digit of stage slot tid + j*THREADS, 4 per register (value phase)
    if constexpr (VB != 0) {
#pragma unroll
        for (int j = 0; j < KPT / 4; ++j) digs[j] = 0;
    }
    if constexpr (Cfg::FUSED) {
        V* vals_out = reinterpret_cast<V*>(vals_out_);
        auto load_pair = [&](uint32_t slot, uint32_t& k, V& v) {
            if constexpr (VB == 4) {
                const uint2 kv = reinterpret_cast<const uint2*>(s_raw)[slot];
                k = kv.x;
                v = kv.y;
            } else {
                k = s_stage[slot];
                v = reinterpret_cast<const V*>(s_raw + TILE * 4)[slot];
            }
        };
        if (GS_LIKELY(full) && !GS_ABL_GENERIC_SCATTER) {
            uint32_t kk[KPT];
            V vv[KPT];
#pragma unroll
            for (int j = 0; j < KPT; ++j) load_pair(tid + j * THREADS, kk[j], vv[j]);
#pragma unroll
            for (int j = 0; j < KPT; ++j) {
                const uint32_t o = ((s_gbase[(kk[j] >> shift) & 255u] + tid + j * THREADS) ^ rev_xor) + rev_add;
                st_stream(keys_out + o, from_bits<KT>(kk[j]));
                st_stream(vals_out + o, vv[j]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < KPT; ++j) {
                const uint32_t i = tid + j * THREADS;
                uint32_t k;
                V v;
                load_pair(i, k, v);
                uint32_t o = s_gbase[(k >> shift) & 255u] + i;
                if (reverse) o = n - 1u - o;
                GS_ABL_OUT_INDEX(o, i);
                if (full || (i >= head && i < head + count)) {
                    st_stream(keys_out + o, from_bits<KT>(k));
                    st_stream(vals_out + o, v);
                }
            }
        }
    } else if (GS_LIKELY(full) && !GS_ABL_GENERIC_SCATTER) {
        // the common case as straight-line code: all stage reads first, then the base look-ups, then the stores
        // (with the masks and the reversal in the loop every key got its own branches and LDS round trips).
        // Measured in round 3 and not kept: a RUN-aligned scatter — every wave walks the runs of its 32 digits, one store
        // instruction per 256-byte-aligned window of the output, so that no instruction ends inside a 64-byte chunk (the
        // slot-ordered loop cuts every run once more than its misalignment does: +0.94 of 4 L2 write requests per run,
        // DESIGN.md 0).  Exact, and 0.47 -> 0.58 ms per pass: ~65 dependent (readlane, LDS read, store) steps per wave
        // instead of 32 + 32 independent ones (profiles/r03_ab_run_aligned_scatter.txt).
        if constexpr (KW == 2) {
            uint2 kb[KPT];
#pragma unroll
            for (int j = 0; j < KPT; ++j) kb[j] = reinterpret_cast<const uint2*>(s_raw)[tid + j * THREADS];
#pragma unroll
            for (int j = 0; j < KPT; ++j) {
                const uint32_t d = (kb[j].x >> shift) & 255u;
                const uint2 nat = from_bits2<KT>(hi_word ? uint2{kb[j].y, kb[j].x} : kb[j]);
                st_stream(reinterpret_cast<uint2*>(keys_out) + (((s_gbase[d] + tid + j * THREADS) ^ rev_xor) + rev_add), nat);
                if constexpr (VB != 0) digs[j >> 2] |= d << (8 * (j & 3));
            }
        } else {
            uint32_t kb[KPT];
#pragma unroll
            for (int j = 0; j < KPT; ++j) kb[j] = s_stage[tid + j * THREADS];
#pragma unroll
            for (int j = 0; j < KPT; ++j) {
                const uint32_t d = (kb[j] >> shift) & 255u;
                const uint32_t opos = s_gbase[d] + tid + j * THREADS;
                st_stream(keys_out + ((opos ^ rev_xor) + rev_add), from_bits<KT>(kb[j]));
                GS_ABL_COUNT_NEXT(kb[j], opos);
                if constexpr (POS == 1) { if (counting) count_next(kb[j], opos, true); }
                if constexpr (VB != 0) digs[j >> 2] |= d << (8 * (j & 3));
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const uint32_t i = tid + j * THREADS;
            uint2 kb2 = {0u, 0u};
            if constexpr (KW == 2) kb2 = reinterpret_cast<const uint2*>(s_raw)[i];
            const uint32_t kb = KW == 2 ? kb2.x : s_stage[i];
            const uint32_t d = (kb >> shift) & 255u;
            uint32_t o = s_gbase[d] + i;
            if constexpr (POS == 1) { if (counting) count_next(kb, o, full || (i >= head && i < head + count)); }
            if (reverse) o = n - 1u - o;
            GS_ABL_OUT_INDEX(o, i);
            if (full || (i >= head && i < head + count)) {
                if constexpr (KW == 2)
                    st_stream(reinterpret_cast<uint2*>(keys_out) + o, from_bits2<KT>(hi_word ? uint2{kb2.y, kb2.x} : kb2));
                else
                    st_stream(keys_out + o, from_bits<KT>(kb));
            }
            if constexpr (VB != 0) digs[j >> 2] |= d << (8 * (j & 3));
        }
    }
    GS_TRACE(6);

    if constexpr (VB != 0 && !Cfg::FUSED) {
        V* vals_out = reinterpret_cast<V*>(vals_out_);
        V* s_vstage = reinterpret_cast<V*>(s_raw);
        constexpr int VRN = Cfg::VROUNDS;
        constexpr uint32_t HALF = TILE / VRN;  // stage slots per round
#pragma unroll
        for (int h = 0; h < VRN; ++h) {
            __syncthreads();  // everyone is done reading the stage (the keys; the previous round's values)
            if (GS_LIKELY(!mask_tail)) {
#pragma unroll
                for (int i = 0; i < KPT; ++i) {
                    const uint32_t pos = (offp[i >> 1] >> (16 * (i & 1))) & 0xffffu;
                    if (VRN == 1 || pos / HALF == (uint32_t)h) s_vstage[pos - h * HALF] = val[i];
                }
            } else {
#pragma unroll
                for (int i = 0; i < KPT; ++i) {
                    const uint32_t pos = (offp[i >> 1] >> (16 * (i & 1))) & 0xffffu;
                    if (my_base + i * 64u < hi && (VRN == 1 || pos / HALF == (uint32_t)h)) s_vstage[pos - h * HALF] = val[i];
                }
            }
            __syncthreads();
            constexpr int J0 = 0, JN = KPT / VRN;  // this round's stage slots: tid + (h * JN + j) * THREADS
            if (GS_LIKELY(full) && !GS_ABL_GENERIC_SCATTER) {
                // (batches of 8 stage reads: with two rounds the other round's values are still in registers)
                constexpr int VBATCH = VRN == 2 ? (JN % 8 == 0 ? 8 : 4) : JN;
                static_assert(JN % VBATCH == 0, "value batches must tile a staging round");
#pragma unroll
                for (int j0 = J0; j0 < JN; j0 += VBATCH) {
                    V vv[VBATCH];
#pragma unroll
                    for (int j = 0; j < VBATCH; ++j) vv[j] = s_vstage[tid + (j0 + j) * THREADS];
#pragma unroll
                    for (int j = 0; j < VBATCH; ++j) {
                        const int jj = h * JN + j0 + j;
                        st_stream(vals_out + (((s_gbase[(digs[jj >> 2] >> (8 * (jj & 3))) & 255u] + tid + jj * THREADS) ^ rev_xor) + rev_add), vv[j]);
                    }
                }
            } else {
#pragma unroll
                for (int j = J0; j < JN; ++j) {
                    const int jj = h * JN + j;
                    const uint32_t i = tid + jj * THREADS;
                    uint32_t o = s_gbase[(digs[jj >> 2] >> (8 * (jj & 3))) & 255u] + i;
                    if (reverse) o = n - 1u - o;
                    GS_ABL_OUT_INDEX(o, i);
                    if (full || (i >= head && i < head + count)) st_stream(vals_out + o, s_vstage[tid + j * THREADS]);
                }
            }
        }
    }
    if constexpr (!PERSIST) break;
    }  // tiles
    if constexpr (POS == 1) {
        // hand the counts to the next pass that runs: CNEXT[that pass][segment][digit] += this workgroup's table
        const uint32_t ns = uni(info[I_NEXT_SHIFT]);
        if ((mode & BM_PLANNED) && ns != 0xffffffffu && !(uni(info[PASS_FLAGS]) & PF_SKIP)) {
            __syncthreads();
            const uint32_t left_out = uni(s_pos[4]);
            uint32_t* cn_out = hsub + (ns >> 3) * HSUB_STRIDE;
          if constexpr (PK) {
            if (left_out < RADIX) {  // the digit count_next skipped: keys written to the segment minus everything that was counted
                const uint32_t lo_word = left_out >> 1, lo_keep = 0xffff0000u >> ((left_out & 1u) * 16u);  // its word in a row, the OTHER half
                for (uint32_t x = wave; x < NCH; x += WAVES) {
                    uint32_t sum = 0;
                    for (uint32_t c = lane; c < RADIX / 2; c += 64u) {
                        uint32_t w = s_cnt[x * (RADIX / 2) + c];
                        if (c == lo_word) w &= lo_keep;  // (what the tiles in front of the choice counted for it is part of the total)
                        sum += (w & 0xffffu) + (w >> 16);
                    }
                    sum = wave_reduce_sum(sum);
                    if (lane == 0) {
                        const uint32_t v = s_pos[8u + x] - sum;
                        if (v != 0u) atomicAdd(&cn_out[x * RADIX + left_out], v);
                        s_cnt[x * (RADIX / 2) + lo_word] &= lo_keep;
                    }
                }
                __syncthreads();
            }
            // (Round 4, measured and not kept: one copy of the table per XCD — workgroup b adds to copy b % 8, the last workgroup of the
            //  pass to arrive sums the copies — against the same-address contention of these atomics: the counting passes got SLOWER,
            //  0.566 -> 0.588 ms at entropy preset 3, profiles/r04_pos_flush_copies.txt: non-returning atomics do not hold the
            //  workgroups up, the closing sum by one workgroup does.)
            for (uint32_t i = tid; i < NCH * RADIX / 2; i += THREADS) {
                const uint32_t w = s_cnt[i];
                if ((w & 0xffffu) != 0u) atomicAdd(&cn_out[2u * i], w & 0xffffu);
                if ((w >> 16) != 0u) atomicAdd(&cn_out[2u * i + 1u], w >> 16);
            }
          } else {
            if (left_out < RADIX) {  // the digit count_next skipped: keys written to the segment minus everything that was counted
                for (uint32_t x = wave; x < NCH; x += WAVES) {
                    uint32_t sum = 0;
                    for (uint32_t d = lane; d < RADIX; d += 64u) sum += d == left_out ? 0u : s_cnt[x * RADIX + d];
                    sum = wave_reduce_sum(sum);
                    if (lane == 0) s_cnt[x * RADIX + left_out] = s_pos[8u + x] - sum;
                }
                __syncthreads();
            }
            for (uint32_t i = tid; i < NCH * RADIX; i += THREADS) {
                const uint32_t v = s_cnt[i];
                if (v != 0u) atomicAdd(&cn_out[i], v);
            }
          }
        }
    }
}

// One workgroup per tile: every key / value type, every compiled shape.
template <int THREADS, int KPT, int VB, int KT, int RANK, int VR = 1>
__global__ __launch_bounds__(THREADS, (BinCfg<THREADS, KPT, VB, KeyWords<KT>::value, VR>::WAVES_PER_SIMD)) void digit_binning_kernel(
    uint32_t* keys_a, uint32_t* keys_b, void* vals_a, void* vals_b, uint32_t* desc, uint32_t* counters, const uint32_t* info,
    uint32_t* hsub, uint32_t* status, uint32_t n, uint32_t shift_full, uint32_t mode) {
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[BinCfg<THREADS, KPT, VB, KeyWords<KT>::value, VR>::LDS_BYTES];
    binning_body<THREADS, KPT, VB, KT, RANK, VR, 0, false>(s_raw, keys_a, keys_b, vals_a, vals_b, desc, counters, info, hsub, status, n,
                                                           shift_full, mode);
}

// The plain form as PERSISTENT workgroups (as many as fit a CU): the two DigitBinningPasses of the two-level plan for pairs — its
// second pass walks 256 chains group by group, which a workgroup does across its tiles (binning_body, `group`).
template <int THREADS, int KPT, int VB, int KT, int RANK, int VR = 1>
__global__ __launch_bounds__(THREADS, (BinCfg<THREADS, KPT, VB, KeyWords<KT>::value, VR>::WAVES_PER_SIMD)) void digit_binning_persist_kernel(
    uint32_t* keys_a, uint32_t* keys_b, void* vals_a, void* vals_b, uint32_t* desc, uint32_t* counters, const uint32_t* info,
    uint32_t* hsub, uint32_t* status, uint32_t n, uint32_t shift_full, uint32_t mode) {
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[BinCfg<THREADS, KPT, VB, KeyWords<KT>::value, VR>::LDS_BYTES];
    binning_body<THREADS, KPT, VB, KT, RANK, VR, 0, true>(s_raw, keys_a, keys_b, vals_a, vals_b, desc, counters, info, hsub, status, n,
                                                          shift_full, mode);
}

// Keys-only sorts of 32-bit keys that the Scan kernel MAY plan on position chains (PF_POS, decided on the device from what the
// histogram kernel saw): ONE launch per pass serves both plans — persistent workgroups, two per CU, that run the plain form of
// the pass (512 x 32 tiles, chains as planned) or, under PF_POS, its position-chain form on the same tiles: with the (packed) next-digit
// table while a later pass needs the counts (LAST = false), without it in the last pass (LAST = true).  (As two
// launches per pass, one of them exiting on the flag, every pass paid a second kernel boundary: +0.01 ms, profiles/r03_pos_*.)
template <int KT, bool LAST>
__global__ __launch_bounds__(512, 4) void digit_binning_dual_kernel(
    uint32_t* keys_a, uint32_t* keys_b, void* vals_a, void* vals_b, uint32_t* desc, uint32_t* counters, const uint32_t* info,
    uint32_t* hsub, uint32_t* status, uint32_t n, uint32_t shift_full, uint32_t mode) {
    constexpr int LDS_PLAIN = BinCfg<512, 32, 0, 1, 1, 0>::LDS_BYTES;
    constexpr int LDS_POS = LAST ? BinCfg<512, 32, 0, 1, 1, 2>::LDS_BYTES : BinCfg<512, GS_POS_KPT, 0, 1, 1, 1>::LDS_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[LDS_PLAIN > LDS_POS ? LDS_PLAIN : LDS_POS];
    static_assert(sizeof(s_raw) * 2 <= 160 * 1024, "two workgroups per CU");
    if ((__builtin_amdgcn_readfirstlane((int)info[PASS_FLAGS]) & (int)PF_POS) == 0) {
        binning_body<512, 32, 0, KT, 1, 1, 0, true>(s_raw, keys_a, keys_b, vals_a, vals_b, desc, counters, info, hsub, status, n, shift_full, mode);
    } else if constexpr (LAST) {
        binning_body<512, 32, 0, KT, 1, 1, 2, true>(s_raw, keys_a, keys_b, vals_a, vals_b, desc, counters, info, hsub, status, n, shift_full, mode);
    } else {
        binning_body<512, GS_POS_KPT, 0, KT, 1, 1, 1, true>(s_raw, keys_a, keys_b, vals_a, vals_b, desc, counters, info, hsub, status, n, shift_full, mode);
    }
}

// (u32 key, u64 value) pairs that the Scan kernel may plan on position chains: the position-chain form as a launch of its own
// (persistent workgroups, values staged in two rounds through the keys' stage) beside the two plain forms, each of which exits
// on the plan's flags (mode bit 6) — one kernel holding all three would need the one-round form's 140 KiB of LDS.
// VB = 4 (round 3, late): the same for (u32, u32) pairs — their plain form stages key and value together (128 KiB: no room for the
// count table), this one stages the values behind the keys through the keys' stage.
template <int VB, int KT, bool LAST>
__global__ __launch_bounds__(512, 4) void digit_binning_posv_kernel(
    uint32_t* keys_a, uint32_t* keys_b, void* vals_a, void* vals_b, uint32_t* desc, uint32_t* counters, const uint32_t* info,
    uint32_t* hsub, uint32_t* status, uint32_t n, uint32_t shift_full, uint32_t mode) {
    constexpr int KPT = (LAST && !(VB == 8 && GS_POSV8_LAST_SMALL)) ? 32 : GS_POSV_KPT, POS = LAST ? 2 : 1, VR = VB == 8 ? 2 : 1;
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[BinCfg<512, KPT, VB, 1, VR, POS>::LDS_BYTES];
    static_assert(sizeof(s_raw) * 2 <= 160 * 1024, "two workgroups per CU");
    binning_body<512, KPT, VB, KT, 1, VR, POS, true>(s_raw, keys_a, keys_b, vals_a, vals_b, desc, counters, info, hsub, status, n, shift_full, mode);
}

// ---------------------------------------------------------------------------
// Debug: post-sort invariants of the scan state (reference: ValidateInitialOneSweepState and the index/flag
// checks around it, GPUSortingCUDA/UtilityKernels.cuh:482-502).  One workgroup per (pass, chain), one digit per
// thread: walks the chain's descriptor rows 0..tiles and counts rows that are not INCLUSIVE, rows whose inclusive
// count went down, chains whose ticket counter stayed below their tile count; adds (last row - row 0) into the
// pass's key total.  Block (0,0) also counts non-zero words of the HIST region (which must be zero whenever no
// call is in flight).  report: [0] rows not INCLUSIVE, [1] non-monotone rows, [2] chains with too few tickets,
// [3] non-zero HIST words, [4 + pass] keys accounted for by the pass's descriptors.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void check_state_kernel(const uint32_t* slab, uint32_t desc_stride, uint32_t tile_keys,
                                                           uint32_t p0, uint32_t dyn, unsigned long long* report,
                                                           uint32_t tile_keys_pos /*tile of a sort planned with PF_POS*/,
                                                           uint32_t tile_keys0 /*tile of the plan's first pass*/) {
    const uint32_t tid = threadIdx.x, q = blockIdx.y, chain = blockIdx.x;
    if (q == 0 && chain == 0) {
        uint32_t nz = 0;
        for (uint32_t i = tid; i < HIST_WORDS; i += 256) nz += slab[SLAB_HIST + i] != 0u;
        if (nz) atomicAdd(&report[3], (unsigned long long)nz);
    }
    const uint32_t* info = slab + SLAB_INFO + q * INFO_STRIDE;
    if (dyn && (info[PASS_FLAGS] & PF_SKIP)) return;  // an identity pass that was dropped: nothing ran
    if (chain >= info[I_NCH]) return;
    const uint32_t tiles = chain_tiles(info[I_START + chain], info[I_END + chain],
                                       ((info[PASS_FLAGS] & PF_POS) && (q != 3u || (tile_keys_pos >> 31))) ? (tile_keys_pos & 0x7fffffffu) : (q == 0u ? tile_keys0 : tile_keys));
    if (tiles == 0) return;
    const uint32_t* rows = slab + SLAB_DESC + (size_t)q * desc_stride + (size_t)info[I_ROW + chain] * RADIX;
    if (tid == 0 && slab[SLAB_COUNTERS + ((p0 + q) * COUNTERS_PER_PASS + chain) * COUNTER_STRIDE] < tiles)
        atomicAdd(&report[2], 1ull);
    uint32_t bad_flag = 0, bad_mono = 0;
    const uint32_t first = rows[tid];
    uint32_t prev = first >> 2;
    bad_flag += (first & FLAG_MASK) != FLAG_INCLUSIVE;
    for (uint32_t r = 1; r <= tiles; ++r) {
        const uint32_t v = rows[(size_t)r * RADIX + tid];
        bad_flag += (v & FLAG_MASK) != FLAG_INCLUSIVE;
        bad_mono += (v >> 2) < prev;
        prev = v >> 2;
    }
    if (bad_flag) atomicAdd(&report[0], (unsigned long long)bad_flag);
    if (bad_mono) atomicAdd(&report[1], (unsigned long long)bad_mono);
    atomicAdd(&report[4 + (q & 3u)], (unsigned long long)(prev - (first >> 2)));  // (64-bit keys: passes q and q + 4 share a slot)
}

// ---------------------------------------------------------------------------
// Single-tile sort (n <= SMALL_TILE): ONE workgroup does all four passes in LDS — one
// launch instead of clear + histogram + scan + 4 passes (at small n the multi-kernel
// path is ~105 us of nearly empty launches at idle clocks).  Same tile machinery as
// the DigitBinningPass: wave-striped order, per-wave ranking, prefix over waves and
// digits, stage sorted by digit; the staged order is the next pass's array order.
// Descending = final index reversal (SortCommon.hlsl:594-597).  Slots >= n hold
// all-one dummy keys that stay behind every real key in every pass.
// ---------------------------------------------------------------------------
// Shapes: 512 x 16 (8192 slots, every value type), 1024 x 16 (16 384: keys-only and 4-byte values) and
// 1024 x 32 (32 768: keys-only) — as far as 160 KiB of LDS go.
constexpr uint32_t SMALL_TILE = 512 * 16;         // 8192: the size every mode can sort in one workgroup
constexpr uint32_t SMALL_TILE_MAX = 1024 * 32;    // largest single-tile sort (keys-only)

template <int SMALL_THREADS, int SMALL_KPT, int VB, int KT, int RANK>
__global__ __launch_bounds__(SMALL_THREADS) void small_sort_kernel(uint32_t* keys, void* vals_, uint32_t n,
                                                                   uint32_t descending, uint32_t* status) {
    using V = typename ValT<VB>::type;
    constexpr int KW = KeyWords<KT>::value;  // 64-bit keys: eight passes, both words staged
    constexpr int KPT = SMALL_KPT, WAVES = SMALL_THREADS / 64;
    constexpr uint32_t SMALL_TILE = SMALL_THREADS * SMALL_KPT;
    static_assert(SMALL_TILE * (4 * KW + (VB == 8 ? 8 : VB)) + WAVES * RADIX * 4 + 64 <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(16))) uint32_t s_stage[SMALL_TILE];
    __shared__ __attribute__((aligned(16))) uint32_t s_stage2[KW == 2 ? SMALL_TILE : 1];
    __shared__ __attribute__((aligned(16))) V s_vstage[VB != 0 ? SMALL_TILE : 1];
    __shared__ uint32_t s_whist[WAVES * RADIX];
    __shared__ uint32_t s_wtot[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t my_base = wave * (64u * KPT) + lane;
    uint32_t* whist = s_whist + wave * RADIX;
    if (tid == 0) st_agent(status, STATUS_OK);  // this sort cannot time out; an earlier call's verdict is not this call's

    uint32_t key[KPT];                      // low word (the only one for 32-bit keys)
    uint32_t keyh[KW == 2 ? KPT : 1];       // high word
    V val[VB != 0 ? KPT : 1];
    // unconditional loads on a clamped index, masked afterwards (guarded loads are issued one at a time)
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const uint32_t idx = my_base + i * 64u;
        const uint32_t ci = idx < n ? idx : n - 1u;
        if constexpr (KW == 2) {
            const uint2 k = reinterpret_cast<const uint2*>(keys)[ci];
            key[i] = k.x;
            keyh[i] = k.y;
        } else {
            key[i] = keys[ci];
        }
        if constexpr (VB != 0) val[i] = reinterpret_cast<const V*>(vals_)[ci];
    }
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const bool valid = my_base + i * 64u < n;
        if constexpr (KW == 2) {
            const uint2 b = to_bits2<KT>(uint2{key[i], keyh[i]});
            key[i] = valid ? b.x : 0xffffffffu;
            keyh[i] = valid ? b.y : 0xffffffffu;
        } else {
            key[i] = valid ? to_bits<KT>(key[i]) : 0xffffffffu;
        }
    }

#pragma unroll 1
    for (uint32_t shift_full = 0; shift_full < 32u * KW; shift_full += 8) {
        const uint32_t shift = shift_full & 31u;
        const bool hi_word = KW == 2 && shift_full >= 32u;
        auto dword = [&](int i) { return KW == 2 && hi_word ? keyh[KW == 2 ? i : 0] : key[i]; };  // the word holding this pass's digit
        for (uint32_t i = tid; i < WAVES * RADIX; i += SMALL_THREADS) s_whist[i] = 0;
        __syncthreads();
        uint32_t off[KPT];
        if constexpr (RANK == 0) {
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                const uint32_t w = dword(i);
                const uint32_t d = (w >> shift) & 255u;
                uint32_t acc_lo = 0, acc_hi = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint32_t B = (uint32_t)__builtin_amdgcn_sbfe((int32_t)w, shift + k, 1);
                    const unsigned long long b = __builtin_amdgcn_ballot_w64(B != 0u);
                    acc_lo = __builtin_amdgcn_bitop3_b32(acc_lo, (uint32_t)b, B, 0xF6);
                    acc_hi = __builtin_amdgcn_bitop3_b32(acc_hi, (uint32_t)(b >> 32), B, 0xF6);
                }
                const uint32_t plo = ~acc_lo, phi = ~acc_hi;
                const uint32_t below = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
                const uint32_t total = __popc(plo) + __popc(phi);
                const uint32_t pre = whist[d];
                if (below == total - 1u) whist[d] = pre + total;
                asm volatile("" ::: "memory");
                off[i] = pre + below;
            }
        } else {
            // slots >= n take no part (the dummies would all meet on the counter of digit 255, 64 lanes deep,
            // in every pass); they stay behind the n real keys, so validity is a property of the slot
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                const uint32_t d = (dword(i) >> shift) & 255u;
                off[i] = 0;
                if (my_base + i * 64u < n)
                    off[i] = __hip_atomic_fetch_add(&whist[d], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        __syncthreads();
        uint32_t run = 0, scan_incl = 0;
        if (tid < RADIX) {
#pragma unroll
            for (int w = 0; w < WAVES; ++w) {
                const uint32_t c = s_whist[w * RADIX + tid];
                s_whist[w * RADIX + tid] = run;
                run += c;
            }
            scan_incl = wave_inclusive_scan(run, lane);
            if (lane == 63) s_wtot[wave] = scan_incl;
        }
        __syncthreads();
        if (tid < RADIX) {
            uint32_t wbase = 0;
            for (uint32_t w = 0; w < wave; ++w) wbase += s_wtot[w];
            const uint32_t dpre = wbase + scan_incl - run;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) s_whist[w * RADIX + tid] += dpre;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const uint32_t lpos = off[i] + s_whist[wave * RADIX + ((dword(i) >> shift) & 255u)];
            if (RANK == 0 || my_base + i * 64u < n) {
                s_stage[lpos] = key[i];
                if constexpr (KW == 2) s_stage2[lpos] = keyh[i];
                if constexpr (VB != 0) s_vstage[lpos] = val[i];
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            key[i] = s_stage[my_base + i * 64u];
            if constexpr (KW == 2) keyh[i] = s_stage2[my_base + i * 64u];
            if constexpr (VB != 0) val[i] = s_vstage[my_base + i * 64u];
        }
        // the next pass starts with a barrier (after zeroing whist) before anything writes the stage
    }
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const uint32_t idx = my_base + i * 64u;
        if (idx < n) {
            const uint32_t o = descending ? n - 1u - idx : idx;
            if constexpr (KW == 2) reinterpret_cast<uint2*>(keys)[o] = from_bits2<KT>(uint2{key[i], keyh[i]});
            else keys[o] = from_bits<KT>(key[i]);
            if constexpr (VB != 0) reinterpret_cast<V*>(vals_)[o] = val[i];
        }
    }
}

// ---------------------------------------------------------------------------
// Hardware probe for RANK 1: does a returning LDS atomic hand out its results
// in ascending lane order among the lanes of ONE wave-instruction that hit the
// same address?  Every wave draws pseudo-random digits (several skews), issues
// back-to-back ds_add_rtn on a private 256-counter table and compares each
// returned value with the exact stable rank computed with ballots.
// ---------------------------------------------------------------------------
// Self-test of the wave-level primitives the kernels are built from — the wave64 counterparts of the reference's warp primitives
// (GPUSortingCUDA/Utils.cuh:22-126: getLaneId / getLaneMaskLt by PTX, Inclusive / ExclusiveWarpScan by __shfl_up_sync,
// WarpReduceSum; the 32-lane ballot multi-split of OneSweep.cu:207-253).  Every lane of every wave takes the word
// x = wave_primitive_input(seed, wave * 64 + lane) and writes, per wave, eight rows of 64 words:
//   0 x   1 wave_inclusive_scan(x & 0xffff) (shuffles)   2 wave_inclusive_scan_dpp(x & 0xffff)   3 wave_reduce_sum(x & 0xffff), lane 0's
//   4 / 5 low / high word of the 64-bit ballot of (x & 1)   6 mbcnt of that ballot = set lanes BELOW this lane (the lanemask_lt popcount)
//   7 rank of the lane among the lanes that hold the same low byte, by the eight-ballot multi-split of binning_body's RANK == 0 path
// tests/test_gpu_parity.py recomputes all of it on the host.
// ---------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint32_t wave_primitive_input(uint32_t seed, uint32_t i) {
    uint32_t h = (seed ^ (i * 2654435761u)) * 2246822519u;
    h ^= h >> 13;
    h *= 3266489917u;
    return h ^ (h >> 16);
}
__global__ __launch_bounds__(256) void wave_primitives_kernel(uint32_t seed, uint32_t* __restrict__ out) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = blockIdx.x * 4u + (tid >> 6);
    const uint32_t x = wave_primitive_input(seed, wave * 64u + lane);
    uint32_t* row = out + (size_t)wave * 8u * 64u;
    row[lane] = x;
    row[64 + lane] = wave_inclusive_scan(x & 0xffffu, lane);
    row[128 + lane] = wave_inclusive_scan_dpp(x & 0xffffu);
    row[192 + lane] = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_reduce_sum(x & 0xffffu));
    const unsigned long long b = __builtin_amdgcn_ballot_w64((x & 1u) != 0u);
    row[256 + lane] = (uint32_t)b;
    row[320 + lane] = (uint32_t)(b >> 32);
    row[384 + lane] = __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
    uint32_t acc_lo = 0, acc_hi = 0;  // bit l set <=> lane l's low byte differs from mine
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t B = (uint32_t)__builtin_amdgcn_sbfe((int32_t)x, k, 1);  // 0 or ~0
        const unsigned long long bb = __builtin_amdgcn_ballot_w64(B != 0u);
        acc_lo = __builtin_amdgcn_bitop3_b32(acc_lo, (uint32_t)bb, B, 0xF6);  // acc | (b ^ B)
        acc_hi = __builtin_amdgcn_bitop3_b32(acc_hi, (uint32_t)(bb >> 32), B, 0xF6);
    }
    row[448 + lane] = __builtin_amdgcn_mbcnt_hi(~acc_hi, __builtin_amdgcn_mbcnt_lo(~acc_lo, 0u));
}

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(512) void lds_atomic_order_probe(uint32_t seed, uint32_t iters, uint32_t* failures) {
    __shared__ uint32_t s_cnt[8][RADIX];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t* cnt = s_cnt[wave];
    uint32_t x = (blockIdx.x * 512u + tid) * 2654435761u + seed * 40503u + 12345u;
    uint32_t bad = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        for (uint32_t j = lane; j < RADIX; j += 64) cnt[j] = 0;
        const uint32_t mode = (it + blockIdx.x) & 7u;  // 0: uniform, ..., 7: all equal
#pragma unroll 1
        for (int r = 0; r < 8; ++r) {
            x ^= x << 13; x ^= x >> 17; x ^= x << 5;
            uint32_t d = x >> 24;
            if (mode == 1) d &= 0x0fu;
            if (mode == 2) d &= 0x03u;
            if (mode == 3) d = (d & 1u) ? 7u : (d >> 1);
            if (mode == 4) d &= x >> 16;
            if (mode == 5) d = (lane >> 1) ^ (d & 1u);
            if (mode == 6) d = (d < 250u) ? 42u : d;
            if (mode == 7) d = 0;
            // exact: number of lower lanes with my digit + how often the digit occurred in earlier rounds
            uint32_t acc_lo = 0, acc_hi = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t B = 0u - ((d >> k) & 1u);
                const unsigned long long b = __builtin_amdgcn_ballot_w64(B != 0u);
                acc_lo |= (uint32_t)b ^ B;
                acc_hi |= (uint32_t)(b >> 32) ^ B;
            }
            const uint32_t below = __builtin_amdgcn_mbcnt_hi(~acc_hi, __builtin_amdgcn_mbcnt_lo(~acc_lo, 0u));
            const uint32_t before = cnt[d];  // in-order LDS: sees all earlier rounds
            asm volatile("" ::: "memory");
            const uint32_t got = __hip_atomic_fetch_add(&cnt[d], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("" ::: "memory");
            bad += (got != before + below);
        }
    }
    bad = wave_reduce_sum(bad);
    if (lane == 0 && bad) atomicAdd(failures, bad);
}

#ifdef GS_TUNING  // calibration kernels: in the tuning build (libgpusort_tuning.so: tools/, bench.py's box_floor) only
// Memory-floor kernel for tuning: the binning pass's exact global access shape
// (wave-striped dword loads of a tile, coalesced dword stores) and one LDS
// round trip, with no ranking and no look-back.
template <int THREADS, int KPT>
__global__ __launch_bounds__(THREADS) void copy_floor_kernel(const uint32_t* in, uint32_t* out, uint32_t n) {
    __shared__ uint32_t s_stage[THREADS * KPT];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t tile_base = blockIdx.x * (THREADS * KPT);
    if (tile_base + THREADS * KPT > n) return;
    uint32_t key[KPT];
    const uint32_t my_base = tile_base + wave * (64u * KPT) + lane;
#pragma unroll
    for (int i = 0; i < KPT; ++i) key[i] = in[my_base + i * 64u];
#pragma unroll
    for (int i = 0; i < KPT; ++i) s_stage[wave * (64u * KPT) + i * 64u + lane] = key[i];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < KPT; ++j) out[tile_base + tid + j * THREADS] = s_stage[tid + j * THREADS];
}

// Plain streaming copies for calibration of the box's achievable HBM rate: 16-byte grid-stride
// (mode 0: default policy, 1: non-temporal loads, 2: non-temporal loads and stores), and a read-only sweep.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void copy_x4_kernel(const u32x4* in, u32x4* out, uint32_t nvec) {
    const uint32_t stride = gridDim.x * 256u;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < nvec; i += stride) {
        u32x4 v;
        if constexpr (MODE >= 1) v = __builtin_nontemporal_load(in + i); else v = in[i];
        if constexpr (MODE >= 2) __builtin_nontemporal_store(v, out + i); else out[i] = v;
    }
}
// The same with U 16-byte loads in flight per thread (round 4: what the box streams at best — tools/r04_probe.hip found the read
// sweep at 7.1 TB/s with four non-temporal loads in flight and two workgroups per CU, the copy at 6.0 TB/s as a one-shot grid)
template <int U, bool NT>
__global__ __launch_bounds__(256) void read_xu_kernel(const u32x4* __restrict__ in, uint32_t* sink, size_t nvec) {
    const size_t stride = (size_t)gridDim.x * 256 * U;
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i + (U - 1) * 256 < nvec; i += stride) {
        u32x4 t[U];
#pragma unroll
        for (int u = 0; u < U; ++u) t[u] = NT ? __builtin_nontemporal_load(in + i + u * 256) : in[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= t[u];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}
template <int U, bool NTL>
__global__ __launch_bounds__(256) void copy_xu_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t nvec) {
    const size_t stride = (size_t)gridDim.x * 256 * U;
    for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i + (U - 1) * 256 < nvec; i += stride) {
        u32x4 t[U];
#pragma unroll
        for (int u = 0; u < U; ++u) t[u] = NTL ? __builtin_nontemporal_load(in + i + u * 256) : in[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) out[i + u * 256] = t[u];
    }
}
__global__ __launch_bounds__(256) void read_x4_kernel(const u32x4* in, uint32_t* sink, uint32_t nvec) {
    const uint32_t stride = gridDim.x * 256u;
    uint32_t acc = 0;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < nvec; i += stride) {
        const u32x4 v = in[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;  // keeps the loads alive
}
#endif  // GS_TUNING

// ---------------------------------------------------------------------------
// Fixtures: InitRandom and Validate.
// ---------------------------------------------------------------------------
// Same 65536-virtual-thread structure as the reference's <<<256,256>>> launch so
// the generated array is defined element-for-element (UtilityKernels.cuh:53-117).
template <int VB>
__global__ __launch_bounds__(256) void init_random_kernel(uint32_t* keys, void* vals_, uint32_t and_count,
                                                           uint32_t seed, uint32_t n) {
    const uint32_t idx = threadIdx.x + blockDim.x * blockIdx.x;
    uint32_t z1 = (idx << 2) * seed;
    uint32_t z2 = ((idx << 2) + 1u) * seed;
    uint32_t z3 = ((idx << 2) + 2u) * seed;
    uint32_t z4 = ((idx << 2) + 3u) * seed;
    auto step = [&]() {
        z1 = ((z1 & 4294967294u) << 12) ^ (((z1 << 13) ^ z1) >> 19);
        z2 = ((z2 & 4294967288u) << 4) ^ (((z2 << 2) ^ z2) >> 25);
        z3 = ((z3 & 4294967280u) << 17) ^ (((z3 << 3) ^ z3) >> 11);
        z4 = z4 * 1664525u + 1013904223u;
    };
    step();
    for (uint64_t i = idx; i < n; i += 65536u) {
        uint32_t t = 0xffffffffu;
        for (uint32_t k = 0; k <= and_count; ++k) {
            step();
            t &= z1 ^ z2 ^ z3 ^ z4;
        }
        keys[i] = t;
        if constexpr (VB == 4) reinterpret_cast<uint32_t*>(vals_)[i] = t;
        if constexpr (VB == 8) reinterpret_cast<uint64_t*>(vals_)[i] = t;
    }
}

// Counts adjacent inversions in keys (and values), order/type aware.
template <int VB>
__global__ __launch_bounds__(256) void validate_kernel(const uint32_t* keys, const void* vals_, uint32_t n,
                                                        int key_type, int descending, uint32_t* err) {
    auto bits = [&](uint32_t u) {
        return key_type == KEY_I32 ? to_bits<KEY_I32>(u) : key_type == KEY_F32 ? to_bits<KEY_F32>(u) : u;
    };
    uint32_t bad = 0;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint64_t i = blockIdx.x * blockDim.x + threadIdx.x; i + 1 < n; i += stride) {
        const uint32_t a = bits(keys[i]), b = bits(keys[i + 1]);
        bad += descending ? (a < b) : (a > b);
        if constexpr (VB == 4) {
            const uint32_t* v = reinterpret_cast<const uint32_t*>(vals_);
            const uint32_t x = bits(v[i]), y = bits(v[i + 1]);
            bad += descending ? (x < y) : (x > y);
        }
        if constexpr (VB == 8) {
            const uint64_t* v = reinterpret_cast<const uint64_t*>(vals_);
            bad += descending ? (v[i] < v[i + 1]) : (v[i] > v[i + 1]);
        }
    }
    bad = wave_reduce_sum(bad);
    if ((threadIdx.x & 63u) == 0 && bad) atomicAdd(err, bad);
}

// 64-bit keys: adjacent inversions of the keys only (the payload convention value = key of the 32-bit fixtures has no
// 64-bit counterpart in the reference)
__global__ __launch_bounds__(256) void validate64_kernel(const uint2* keys, uint32_t n, int key_type, int descending, uint32_t* err) {
    auto bits = [&](uint2 k) -> unsigned long long {
        const uint2 b = key_type == KEY_I64 ? to_bits2<KEY_I64>(k) : key_type == KEY_F64 ? to_bits2<KEY_F64>(k) : k;
        return ((unsigned long long)b.y << 32) | b.x;
    };
    uint32_t bad = 0;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint64_t i = blockIdx.x * blockDim.x + threadIdx.x; i + 1 < n; i += stride) {
        const unsigned long long a = bits(keys[i]), b = bits(keys[i + 1]);
        bad += descending ? (a < b) : (a > b);
    }
    bad = wave_reduce_sum(bad);
    if ((threadIdx.x & 63u) == 0 && bad) atomicAdd(err, bad);
}

}  // namespace gs
