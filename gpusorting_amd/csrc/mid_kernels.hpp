// mid_kernels.hpp — the TWO-launch sort of mid-size inputs (single-tile limit < n <= 2^20; keys-only and 4-byte
// values up to 2^21, keys-only up to 2^22, with larger tiles).
//
// SURVEY.md 8f N1 / the reference's BenchmarkOneSweep size sweep (GPUSortingD3D12/Tests.h:392-393,415-416): between
// 2^14 and 2^20 keys the six-launch pipeline is a ~50 us plateau of launch and tile latencies (profiles/
// r01_size_and_entropy_sweep_v3.txt) — four global passes of >= 9 us each whatever the size.  At these sizes the keys
// of one top-byte value fit ONE workgroup's LDS, so the sort is done as
//   K1 mid_msd_kernel     one MSD pass: every workgroup (one tile each — 8192 keys, 16 384 above 2^20; at most 128 tiles, 256
//                         above 2^21) ranks its tile by the TOP byte, publishes its 256 counts, waits until the count
//                         table is complete, derives every bucket's start and its own offsets from it (the whole workgroup
//                         reads the table: 16-byte sc1 loads), and scatters its tile into the alt buffer (stable);
//   K2 bucket_sort_kernel one workgroup per top-byte bucket sorts it on the remaining 24 bits entirely in LDS (three
//                         stable passes, as the single-tile sort; the bucket spread evenly over the workgroup's waves) and
//                         writes it to its final place in the key buffer.
// Two global passes instead of four (20 B/key instead of 36), two launches instead of six.  Same result as the LSD
// sort: (stable by top byte) o (stable sort of each bucket by the low 24 bits) == stable sort by the whole key.
// If a bucket would not fit a workgroup (skewed top byte) every workgroup sees that in the SAME count table and K1
// runs the four LSD passes itself, phase by phase (about the cost of the six-launch path, no extra launch, no host
// decision); K2 then finds the route flag and exits.
//
// FORWARD PROGRESS WITHOUT RESIDENCY (round 3; reference: the look-back-with-fallback of SweepCommon.hlsl:297-425 and the
// fault emulation of EmulatedDeadlocking.cu:36-37,159-267).  The wait for the count table is not a grid barrier: a tile
// is WORK that a workgroup CLAIMS (atomicMax of the call's epoch on the tile's claim word), normally its own
// (tile = blockIdx).  A workgroup that has waited ADOPT_SPINS polls for a tile's counts looks at the claim word: a tile
// nobody has claimed — its workgroup has not been dispatched: another stream holds the CUs, the device is partitioned,
// more tiles than resident workgroups — is claimed and processed by the waiter (counts now, scatter after its own).  A
// workgroup that finally starts and finds its tile claimed exits.  Every wait is therefore for a tile whose owner is
// RUNNING, whatever the occupancy; no timeout is needed for progress (the bounded spin stays as a last resort and
// reports GS_ERR_TIMEOUT; K2 then writes nothing).  Nothing in the scratch region has to be cleared between calls: claim
// words, row flags and the plan's epoch word carry the call's epoch — a DEVICE word (MID_CTR + 1) that the last workgroup of
// K2 to finish advances, so that a captured HIP graph replays with fresh epochs.
// Cross-workgroup data inside K1: the count table — write-through (sc1) stores, drained, then the row's flag word;
// readers poll the flags with sc1 loads and read the table with sc1 loads (MI355X_MICROARCH.md, valid forms).  On the
// LSD route the keys themselves cross workgroups between passes: written and read with sc1 accesses as well.
#pragma once
#include "onesweep_kernels.hpp"

namespace gs {

// Three classes by the bucket K2's workgroup can hold (a top-byte bucket must fit it); K1's tiles:
//   K2 512 x 16 =  8 192 keys  n <= 2^20   every value width                                      K1 <= 128 tiles of  8 192
//   K2 512 x 32 = 16 384 keys  n <= 2^21   keys-only and 4-byte values (stage 64 + 64 KiB)        K1 <= 128 tiles of 16 384
//   K2 1024 x 32 = 32 768 keys n <= 2^22   keys-only (stage 128 KiB)                              K1 <= 256 tiles of 16 384
//   K2 1024 x 34 = 34 816 keys n <= 2^23   keys-only (stage 136 KiB: what 160 KiB of LDS hold)    K1 <= 256 tiles of 32 768 (one per CU)
//   K2 512 x 34  = 17 408 pairs n <= 2^22  4-byte values (stage 68 + 68 KiB)                      K1 <= 256 tiles of 16 384
//      (round 5: a top-byte bucket of 2^23 uniform keys is 32 768 +- 181 keys — 6 % of slack; anything less even runs K1's LSD route)
constexpr uint32_t MID_THREADS = 512, MID_KPT = 16, MID_TILE = MID_THREADS * MID_KPT;  // the smallest shape
constexpr uint32_t MID_MAX_TILES = 256;                                                // (the smallest shape stops at 128: one launch wave of half the CUs)
constexpr uint32_t MID_MAX_KEYS = 128 * MID_TILE;                                      // 2^20: limit of the smallest shape
// scratch words in the handle's slab (SLAB_MID)
constexpr uint32_t MID_EPOCH = 0;                   // epoch of the call whose plan (route, bucket table) is below — K2's licence
constexpr uint32_t MID_CTR = 8;                     // epoch of the last completed call (advanced by K2's last workgroup)
constexpr uint32_t MID_DONE = 12;                   // K2 workgroups of the running call that have finished
constexpr uint32_t MID_RESET = 16;                  // epoch of the call whose first workgroup has reset the status word
constexpr uint32_t MID_ROUTE = 32;                  // 0 = MSD route (K2 sorts the buckets), 1 = K1 did the LSD passes
constexpr uint32_t MID_BSTART = 64;                 // bucket starts [256]
constexpr uint32_t MID_BCOUNT = MID_BSTART + RADIX; // bucket counts [256]
constexpr uint32_t MID_CLAIM = MID_BCOUNT + RADIX;  // [MID_MAX_TILES] epoch of the call in which the tile was claimed
constexpr uint32_t MID_AFLAG = MID_CLAIM + MID_MAX_TILES;  // [..] tag of the tile's last published count row
constexpr uint32_t MID_BFLAG = MID_AFLAG + MID_MAX_TILES;  // [..] tag of the tile's last finished scatter (LSD route)
constexpr uint32_t MID_TABLE = 2048;                // two count tables [2][MID_MAX_TILES][256]
constexpr uint32_t MID_WORDS = MID_TABLE + 2 * MID_MAX_TILES * RADIX;
static_assert(MID_BFLAG + MID_MAX_TILES <= MID_TABLE, "mid scratch layout");
static_assert(MID_WORDS <= SLAB_MID_WORDS, "SLAB_MID is too small");
// a tag orders the phases of all calls on a handle: (epoch, step), step 0 = MSD counts, 1 + 2 p = counts of LSD pass p,
// 2 + 2 p = its scatter; epoch < 2^26: K2's last workgroup then zeroes claim words and flags and restarts the count
__host__ __device__ constexpr uint32_t mid_tag(uint32_t epoch, uint32_t step) { return epoch * 16u + step + 1u; }

#ifndef GS_MID_ADOPT_SPINS
#define GS_MID_ADOPT_SPINS 64u  // polls (~1-2 us each) on a missing count row before the waiter looks at the tile's claim word
#endif
// experiment builds (GS_EXP & 4096): phase stamps (10 ns ticks) of workgroup 7 of K1 / of K2 into the status words 16.. / 24..
// (tools/r03_mid_phases.py)
#if (GS_EXP & 4096)
#define GS_MID_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 7) status[16 + (i)] = (uint32_t)wall_clock64(); } while (0)
#define GS_MID_STAMP2(i) do { if (threadIdx.x == 0 && blockIdx.x == 7) const_cast<uint32_t*>(status)[24 + (i)] = (uint32_t)wall_clock64(); } while (0)
#else
#define GS_MID_STAMP(i) do { } while (0)
#define GS_MID_STAMP2(i) do { } while (0)
#endif
#ifndef GS_FAULT_MID_SILENT
#define GS_FAULT_MID_SILENT(block) false  // fault injection: this workgroup of K1 claims its tile and then never publishes anything
#endif
#ifndef GS_FAULT_MID_ABSENT
#define GS_FAULT_MID_ABSENT(block) false  // fault injection: this workgroup of K1 behaves as if it had never been dispatched
#endif

__device__ __forceinline__ uint32_t ld_sc1(const uint32_t* p) {
    return __hip_atomic_load(const_cast<uint32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld_sc1(const unsigned long long* p) {
    return __hip_atomic_load(const_cast<unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
typedef uint32_t tv4 __attribute__((ext_vector_type(4)));
template <int VB>
struct SC1T { using type = uint32_t; };
template <>
struct SC1T<8> { using type = unsigned long long; };

// Rank the tile's keys among the keys of their digit inside their wave (see digit_binning_kernel): off = the ranks (below
// 64 * KPT <= 2048: two per register, which keeps the 32-keys-per-thread shapes out of scratch), the per-wave counters keep the counts.  RANK 1: slots >= count take no part; RANK 0 (ballots): the all-ones dummy
// keys behind `count` rank last in digit 255.
// kpt_eff (uniform, <= KPT): slots per thread in use — K2 spreads a bucket that fills only part of the tile over ALL waves.
template <int RANK, int KPT>
__device__ __forceinline__ void mid_rank(const uint32_t (&key)[KPT], uint32_t shift, uint32_t my_base, uint32_t count,
                                         uint32_t* whist, uint32_t (&off)[KPT / 2], uint32_t kpt_eff = KPT) {
#pragma unroll
    for (int i = 0; i < KPT / 2; ++i) off[i] = 0;
    if constexpr (RANK == 0) {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            if ((uint32_t)i >= kpt_eff) continue;  // uniform
            const uint32_t d = (key[i] >> shift) & 255u;
            uint32_t acc_lo = 0, acc_hi = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t B = (uint32_t)__builtin_amdgcn_sbfe((int32_t)key[i], shift + k, 1);
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
            off[i >> 1] |= (pre + below) << (16 * (i & 1));
        }
    } else {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            if ((uint32_t)i >= kpt_eff) continue;  // uniform
            const uint32_t d = (key[i] >> shift) & 255u;
            if (my_base + i * 64u < count)
                off[i >> 1] |= __hip_atomic_fetch_add(&whist[d], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) << (16 * (i & 1));
        }
    }
}

// ---------------------------------------------------------------------------
// K1
// ---------------------------------------------------------------------------
// BUCKET_CAP: keys K2's workgroup can hold (its tile) — a larger bucket sends the sort down the LSD route
template <int VB, int KT, int RANK, int THREADS_ = (int)MID_THREADS, int KPT_ = (int)MID_KPT, int BUCKET_CAP = THREADS_ * KPT_>
__global__ __launch_bounds__(THREADS_) void mid_msd_kernel(uint32_t* keys, uint32_t* alt, void* vals_, void* valt_, uint32_t* scratch,
                                                           uint32_t* status, uint32_t n, uint32_t descending) {
    using V = typename ValT<VB>::type;
    using VA = typename SC1T<VB>::type;
    const uint32_t epoch = (uint32_t)__builtin_amdgcn_readfirstlane((int)scratch[MID_CTR]) + 1u;  // constant while K1 runs
    constexpr int KPT = KPT_, WAVES = THREADS_ / 64;
    constexpr uint32_t THREADS = THREADS_, TILE = THREADS_ * KPT_;
    __shared__ __attribute__((aligned(16))) uint32_t s_stage[TILE];
    __shared__ __attribute__((aligned(16))) V s_vstage[VB != 0 ? TILE : 1];
    __shared__ uint32_t s_whist[WAVES * RADIX];
    __shared__ uint32_t s_dpre[RADIX], s_gbase[RADIX], s_gsum[RADIX], s_front[RADIX], s_wtot[4], s_max;
    __shared__ uint32_t s_owned[MID_MAX_TILES];  // tiles this workgroup has claimed: its own first, then the adopted ones
    __shared__ uint32_t s_ctl[4];                // [0] claim result / adopt result, [1] first tile whose flag is missing, [2] failed
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t tiles = gridDim.x;
    const uint32_t my_base = wave * (64u * KPT) + lane;
    uint32_t* whist = s_whist + wave * RADIX;
    uint32_t* claim = scratch + MID_CLAIM;
    uint32_t* aflag = scratch + MID_AFLAG;
    uint32_t* bflag = scratch + MID_BFLAG;
    if (GS_FAULT_MID_ABSENT(blockIdx.x)) return;
    GS_MID_STAMP(0);

    uint32_t key[KPT];
    V val[VB != 0 ? KPT : 1];
    uint32_t off[KPT / 2];
    uint32_t cur_tile = blockIdx.x, cur_base = 0, cur_count = 0;  // the tile in the registers
    auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };  // workgroup-uniform values stay scalar
    auto load_tile = [&](uint32_t t, const uint32_t* kin, const void* vin, bool sc1) {
        cur_tile = uni(t);
        cur_base = t * TILE;
        cur_count = n - cur_base < TILE ? n - cur_base : TILE;  // valid slots [0, count)
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const uint32_t slot = my_base + i * 64u;
            const uint32_t ci = cur_base + (slot < cur_count ? slot : cur_count - 1u);
            if (sc1) {
                key[i] = ld_sc1(&kin[ci]);
                if constexpr (VB != 0) val[i] = (V)ld_sc1(&reinterpret_cast<const VA*>(vin)[ci]);
            } else {
                key[i] = kin[ci];
                if constexpr (VB != 0) val[i] = reinterpret_cast<const V*>(vin)[ci];
            }
        }
#pragma unroll
        for (int i = 0; i < KPT; ++i) key[i] = my_base + i * 64u < cur_count ? to_bits<KT>(key[i]) : 0xffffffffu;
    };

    // ---- claim this workgroup's own tile; its keys are requested meanwhile ----
    if (tid == 0) {
        s_ctl[0] = atomicMax(&claim[blockIdx.x], epoch) < epoch ? 1u : 0u;
        s_ctl[2] = 0u;
        s_owned[0] = blockIdx.x;
    }
    // a timeout of an earlier call is not this call's: the first workgroup of the call to get here clears the status word
    // (long before any spin of this call can expire); another wave, so that the two atomics fly together
    if (tid == 64 && atomicMax(&scratch[MID_RESET], epoch) < epoch) st_agent(status, STATUS_OK);
    load_tile(blockIdx.x, keys, vals_, false);
    __syncthreads();
    if (uni(s_ctl[0]) == 0u) return;  // somebody adopted the tile while this workgroup was waiting to be dispatched
    if (GS_FAULT_MID_SILENT(blockIdx.x)) return;  // (fault build: the tile is claimed — nobody can adopt it — and stays silent)
    uint32_t n_owned = 1;        // uniform
    bool own_in_regs = true;     // uniform: the registers still hold the own tile as loaded above

    // rank the tile in the registers on the digit at `shift`, publish its counts as row cur_tile of `table` (flag = tag)
    uint32_t run = 0, scan_incl = 0;  // of this thread's digit: the tile's count (dummies included), its inclusive scan
    auto rank_and_publish = [&](uint32_t shift, uint32_t* table, uint32_t tag, bool publish) {
        for (uint32_t i = tid; i < WAVES * RADIX; i += THREADS) s_whist[i] = 0;
        if (tid == 0) s_max = 0;
        __syncthreads();
        mid_rank<RANK, KPT>(key, shift, my_base, cur_count, whist, off);
        __syncthreads();
        if (tid < RADIX) {
            run = 0;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) {
                const uint32_t c = s_whist[w * RADIX + tid];
                s_whist[w * RADIX + tid] = run;
                run += c;
            }
            const uint32_t mine = run - ((RANK == 0 && tid == RADIX - 1) ? TILE - cur_count : 0u);  // real keys only
            if (publish) st_sc1(&table[cur_tile * RADIX + tid], mine);
            scan_incl = wave_inclusive_scan_dpp(run);
            if (lane == 63) s_wtot[wave] = scan_incl;
        }
        if (publish) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through stores are out before anyone is told
            __syncthreads();
            if (tid == 0) st_sc1(&aflag[cur_tile], tag);
        } else {
            __syncthreads();
        }
    };
    // wait until every tile's flag word has reached `tag`.  adopt: a tile nobody has claimed is taken over (its counts are
    // published by this workgroup at once, its scatter follows this workgroup's own).  false = gave up (status word set).
    auto wait_flags = [&](const uint32_t* flags, uint32_t tag, bool adopt, uint32_t shift, uint32_t* table) -> bool {
        uint32_t spins = 0, since_adopt = 0;  // uniform: polls without progress (bounded by SPIN_LIMIT in every mode), polls since the last adoption attempt
        if (tid == 0) s_ctl[1] = 0xffffffffu;
        __syncthreads();
        for (;;) {
            // lane t watches tile t's flag: a few polls on its own (the common case ends here: one barrier), then the
            // workgroup looks at who is still missing
            if (tid < tiles) {
                bool ready = false;
                for (uint32_t i = 0; i < 8u && !ready; ++i) {
                    ready = (int32_t)(ld_sc1(&flags[tid]) - tag) >= 0;
                    if (!ready) __builtin_amdgcn_s_sleep(1);
                }
                if (!ready) atomicMin(&s_ctl[1], tid);
            }
            __syncthreads();
            const uint32_t missing = uni(s_ctl[1]);
            if (missing == 0xffffffffu) return true;
            __syncthreads();  // everybody has read the verdict
            if (tid == 0) s_ctl[1] = 0xffffffffu;
            __syncthreads();
            spins += 8u;
            since_adopt += 8u;
            if (adopt && since_adopt >= GS_MID_ADOPT_SPINS) {
                // (the adoption cadence has its own counter: `spins` keeps growing while attempts FAIL — a tile that was claimed but whose
                //  counts never appear must run into SPIN_LIMIT below like every other wait, not spin for ever)
                since_adopt = 0;
                if (tid == 0) s_ctl[0] = atomicMax(&claim[missing], epoch) < epoch ? 1u : 0u;
                __syncthreads();
                if (uni(s_ctl[0]) != 0u) {  // nobody had it: ours now
                    spins = 0;  // progress
                    if (tid == 0) s_owned[n_owned] = missing;
                    ++n_owned;
                    own_in_regs = false;
                    load_tile(missing, keys, vals_, false);
                    rank_and_publish(shift, table, tag, true);
                }
            }
            if (spins > SPIN_LIMIT) {
                if (tid == 0) { st_agent(status, STATUS_TIMEOUT); s_ctl[2] = 1u; }
                __syncthreads();
                return false;
            }
        }
    };
    // global bases of the tile in the registers from the complete table: leaves s_gbase[d] = global position of stage slot 0
    // of digit d's run minus its stage offset, s_whist[w][d] += run offset, and returns G (all tiles' count of the digit)
    auto bases = [&](uint32_t* table) -> uint32_t {
        // every tile's count of every digit: all of them for the digit's start, the tiles in front for this tile's offset.
        // The whole workgroup reads the table: lane l of wave w takes digits 4 l .. 4 l + 3 of rows w, w + WAVES, .. with
        // 16-byte sc1 loads, NB in flight (one thread per digit walking the rows eight at a time took tiles / 8 round
        // trips of ~1 us: 16 of the 44 us of this kernel at 2^22 keys, profiles/r03_mid_size_timeline.txt); rows behind
        // the table's end read as zero (buffer range check).
        if (tid < RADIX) { s_gsum[tid] = 0u; s_front[tid] = 0u; }
        __syncthreads();
        {
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(table, 0, (int)(tiles * RADIX * 4u), 0x00020000);
            uint32_t g4[4] = {0u, 0u, 0u, 0u}, f4[4] = {0u, 0u, 0u, 0u};
            constexpr uint32_t NB = (VB != 0 && KPT >= 32) ? 2u : 8u;  // loads in flight (the 32-pairs-per-thread shape has no registers to spare: 4 spill 100 B per lane)
            for (uint32_t t0 = wave; t0 < tiles; t0 += NB * WAVES) {  // (uniform bounds)
                tv4 c[NB];
#pragma unroll
                for (uint32_t j = 0; j < NB; ++j)
                    c[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(((t0 + j * WAVES) * RADIX + lane * 4u) * 4u), 0, 16 /*sc1*/);
#pragma unroll
                for (uint32_t j = 0; j < NB; ++j) {
                    const bool in_front = t0 + j * WAVES < cur_tile;  // uniform
                    g4[0] += c[j].x; g4[1] += c[j].y; g4[2] += c[j].z; g4[3] += c[j].w;
                    if (in_front) { f4[0] += c[j].x; f4[1] += c[j].y; f4[2] += c[j].z; f4[3] += c[j].w; }
                }
            }
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) {
                if (g4[k]) atomicAdd(&s_gsum[lane * 4u + k], g4[k]);
                if (f4[k]) atomicAdd(&s_front[lane * 4u + k], f4[k]);
            }
        }
        __syncthreads();
        uint32_t G = 0;
        if (tid < RADIX) {
            uint32_t wbase = 0;
            for (uint32_t w = 0; w < wave; ++w) wbase += s_wtot[w];
            const uint32_t dpre = wbase + scan_incl - run;  // stage offset of the digit's run (dummies included)
            s_dpre[tid] = dpre;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) s_whist[w * RADIX + tid] += dpre;
            G = s_gsum[tid];
            atomicMax(&s_max, G);
            scan_incl = wave_inclusive_scan_dpp(G);
            s_gbase[tid] = s_front[tid] - dpre;  // + digit start, below
        }
        __syncthreads();
        if (tid < RADIX && lane == 63) s_wtot[wave] = scan_incl;
        __syncthreads();
        if (tid < RADIX) {
            uint32_t wbase = 0;
            for (uint32_t w = 0; w < wave; ++w) wbase += s_wtot[w];
            s_gbase[tid] += wbase + scan_incl - G;  // digit start = keys with a smaller digit, in all tiles
        }
        __syncthreads();
        return G;
    };
    auto stage = [&](uint32_t shift) {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const uint32_t lpos = ((off[i >> 1] >> (16 * (i & 1))) & 0xffffu) + s_whist[wave * RADIX + ((key[i] >> shift) & 255u)];
            if (RANK == 0 || my_base + i * 64u < cur_count) {
                s_stage[lpos] = key[i];
                if constexpr (VB != 0) s_vstage[lpos] = val[i];
            }
        }
        __syncthreads();
    };
    auto scatter = [&](uint32_t shift, uint32_t* kout, void* vout, bool sc1, bool reverse) {
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const uint32_t i = tid + j * THREADS;
            if (i < cur_count) {
                const uint32_t kb = s_stage[i];
                uint32_t o = s_gbase[(kb >> shift) & 255u] + i;
                if (reverse) o = n - 1u - o;
                if (sc1) {
                    st_sc1(&kout[o], from_bits<KT>(kb));
                    if constexpr (VB != 0) st_sc1(&reinterpret_cast<VA*>(vout)[o], (VA)s_vstage[i]);
                } else {
                    kout[o] = from_bits<KT>(kb);
                    if constexpr (VB != 0) reinterpret_cast<V*>(vout)[o] = s_vstage[i];
                }
            }
        }
    };

    // ---- MSD step on the top byte ----
    uint32_t* table0 = scratch + MID_TABLE;
    uint32_t* table1 = table0 + MID_MAX_TILES * RADIX;
    GS_MID_STAMP(1);  // own tile loaded
    rank_and_publish(24, table0, mid_tag(epoch, 0), true);
    GS_MID_STAMP(2);  // ranked, counts published
    if (!wait_flags(aflag, mid_tag(epoch, 0), true, 24, table0)) return;
    GS_MID_STAMP(3);  // every row there
    if (!own_in_regs) {  // an adoption used the registers: back to the own tile
        load_tile(blockIdx.x, keys, vals_, false);
        rank_and_publish(24, table0, 0u, false);
    }
    const uint32_t G = bases(table0);
    GS_MID_STAMP(4);  // bases
    const bool lsd_route = uni(s_max) > (uint32_t)BUCKET_CAP;  // a bucket K2 could not hold; the same table everywhere: the same decision everywhere
    __syncthreads();                      // everybody has read s_max before the next ranking resets it
    // K2's plan, written by whoever owns tile 0 (its own workgroup, or the one that adopted it) while that tile is at hand
    auto write_plan = [&]() {  // the registers / LDS hold tile 0
        if (tid < RADIX) {
            scratch[MID_BSTART + tid] = s_gbase[tid] + s_dpre[tid];  // tile 0 has no tile in front: its base IS the digit start
            scratch[MID_BCOUNT + tid] = G;
            if (tid == 0) {
                scratch[MID_ROUTE] = lsd_route ? 1u : 0u;
                scratch[MID_EPOCH] = epoch;
            }
        }
    };
    if (!lsd_route) {
        for (uint32_t e = 0; e < n_owned; ++e) {  // uniform
            if (e != 0) {
                __syncthreads();
                load_tile(s_owned[e], keys, vals_, false);
                rank_and_publish(24, table0, 0u, false);
                bases(table0);
            }
            if (cur_tile == 0u) write_plan();
            stage(24);
            GS_MID_STAMP(5);  // staged
            scatter(24, alt, valt_, false, false);
            GS_MID_STAMP(6);  // scatter issued
        }
        return;
    }
    for (uint32_t e = 0; e < n_owned; ++e)  // K2 only has to learn that there is nothing for it to do
        if (s_owned[e] == 0u && tid == 0) {
            scratch[MID_ROUTE] = 1u;
            scratch[MID_EPOCH] = epoch;
        }

    // ---- LSD route: the four passes here, phase by phase; keys cross workgroups through sc1 accesses ----
    uint32_t* kbuf[2] = {keys, alt};
    void* vbuf[2] = {vals_, valt_};
#pragma unroll 1
    for (uint32_t p = 0; p < 4; ++p) {
        const uint32_t shift = p * 8u;
        uint32_t* table = (p & 1u) ? table0 : table1;
        const uint32_t tag_a = mid_tag(epoch, 1u + 2u * p), tag_b = mid_tag(epoch, 2u + 2u * p);
        // counts of every owned tile (its input: what pass p-1 wrote)
        for (uint32_t e = 0; e < n_owned; ++e) {
            __syncthreads();
            load_tile(s_owned[e], kbuf[p & 1u], vbuf[p & 1u], p != 0);
            rank_and_publish(shift, table, tag_a, true);
        }
        if (!wait_flags(aflag, tag_a, false, shift, table)) return;
        // scatter of every owned tile (the last one counted is still in the registers)
        for (uint32_t e = n_owned; e-- > 0;) {
            if (e + 1 != n_owned) {
                __syncthreads();
                load_tile(s_owned[e], kbuf[p & 1u], vbuf[p & 1u], p != 0);
                rank_and_publish(shift, table, 0u, false);
            }
            bases(table);
            stage(shift);
            scatter(shift, kbuf[(p + 1u) & 1u], vbuf[(p + 1u) & 1u], true, descending && p == 3);
            if (p != 3) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) st_sc1(&bflag[cur_tile], tag_b);
            }
        }
        if (p != 3 && !wait_flags(bflag, tag_b, false, shift, table)) return;  // everybody has written pass p's output before anybody reads it
    }
}

// ---------------------------------------------------------------------------
// K2: one workgroup per top-byte bucket; the low 24 bits in three stable LDS passes
// ---------------------------------------------------------------------------
template <int VB, int KT, int RANK, int THREADS_ = (int)MID_THREADS, int KPT_ = (int)MID_KPT>
__global__ __launch_bounds__(THREADS_) void bucket_sort_kernel(uint32_t* keys, const uint32_t* alt, void* vals_, const void* valt_,
                                                               uint32_t* scratch, const uint32_t* status, uint32_t n,
                                                               uint32_t descending) {
    using V = typename ValT<VB>::type;
    constexpr int KPT = KPT_, WAVES = THREADS_ / 64;
    constexpr uint32_t THREADS = THREADS_, TILE = THREADS_ * KPT_;
    __shared__ __attribute__((aligned(16))) uint32_t s_stage[TILE];
    __shared__ __attribute__((aligned(16))) V s_vstage[VB != 0 ? TILE : 1];
    __shared__ uint32_t s_whist[WAVES * RADIX];
    __shared__ uint32_t s_wtot[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t epoch = (uint32_t)__builtin_amdgcn_readfirstlane((int)scratch[MID_CTR]) + 1u;  // (advanced when ALL of K2 is through)
    // K2's licence: the plan below is THIS call's (epoch) and no workgroup of K1 gave up (status).  Otherwise the bucket
    // table may be an earlier call's — of another n — and nothing may be written.
    GS_MID_STAMP2(0);
    bool work = scratch[MID_EPOCH] == epoch && *status == STATUS_OK && scratch[MID_ROUTE] == 0u;  // (route 1: K1 ran the LSD passes itself)
    const uint32_t start = scratch[MID_BSTART + blockIdx.x], count = scratch[MID_BCOUNT + blockIdx.x];
    if (count == 0u || count > TILE || start > n || count > n - start) work = false;  // (the last three cannot happen with a valid plan)
    if (work) {
    // A bucket rarely fills the tile (n / 256 keys on average, the tile holds the class's worst case): its keys are spread
    // over ALL waves, kpt slots per thread, instead of filling the first waves with KPT each — the ranking phases take
    // as long as the busiest wave.  Slot order = wave-major, then i, then lane, as before: stable.
    const uint32_t kpt = (uint32_t)__builtin_amdgcn_readfirstlane((int)((count + THREADS - 1u) / THREADS));  // uniform, 1 .. KPT
    const uint32_t my_base = wave * (64u * kpt) + lane;
    uint32_t* whist = s_whist + wave * RADIX;
    uint32_t key[KPT];
    V val[VB != 0 ? KPT : 1];
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        if ((uint32_t)i >= kpt) continue;  // uniform
        const uint32_t slot = my_base + i * 64u;
        const uint32_t ci = start + (slot < count ? slot : count - 1u);
        key[i] = alt[ci];
        if constexpr (VB != 0) val[i] = reinterpret_cast<const V*>(valt_)[ci];
    }
#pragma unroll
    for (int i = 0; i < KPT; ++i)
        if ((uint32_t)i < kpt) key[i] = my_base + i * 64u < count ? to_bits<KT>(key[i]) : 0xffffffffu;
    GS_MID_STAMP2(1);  // bucket loaded
    // a bucket of 64 keys or fewer could stop earlier; the three passes on LDS cost a few microseconds at any size
#pragma unroll 1
    for (uint32_t shift = 0; shift < 24; shift += 8) {
        GS_MID_STAMP2(2 + (shift >> 3));  // LDS pass starts
        for (uint32_t i = tid; i < WAVES * RADIX; i += THREADS) s_whist[i] = 0;
        __syncthreads();
        uint32_t off[KPT / 2];
        mid_rank<RANK, KPT>(key, shift, my_base, count, whist, off, kpt);
        __syncthreads();
        uint32_t run = 0, scan_incl = 0;
        if (tid < RADIX) {
#pragma unroll
            for (int w = 0; w < WAVES; ++w) {
                const uint32_t c = s_whist[w * RADIX + tid];
                s_whist[w * RADIX + tid] = run;
                run += c;
            }
            scan_incl = wave_inclusive_scan_dpp(run);
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
            if ((uint32_t)i >= kpt) continue;
            const uint32_t lpos = ((off[i >> 1] >> (16 * (i & 1))) & 0xffffu) + s_whist[wave * RADIX + ((key[i] >> shift) & 255u)];
            if (RANK == 0 || my_base + i * 64u < count) {
                s_stage[lpos] = key[i];
                if constexpr (VB != 0) s_vstage[lpos] = val[i];
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            if ((uint32_t)i >= kpt) continue;
            key[i] = s_stage[my_base + i * 64u];
            if constexpr (VB != 0) val[i] = s_vstage[my_base + i * 64u];
        }
    }
    GS_MID_STAMP2(5);  // three passes done
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const uint32_t slot = my_base + i * 64u;
        if ((uint32_t)i < kpt && slot < count) {
            const uint32_t idx = start + slot;
            const uint32_t o = descending ? n - 1u - idx : idx;
            keys[o] = from_bits<KT>(key[i]);
            if constexpr (VB != 0) reinterpret_cast<V*>(vals_)[o] = val[i];
        }
    }
    GS_MID_STAMP2(6);  // stores issued
    }  // work
    // the call is over when every workgroup of K2 has read the epoch: the last one to finish advances it
    __syncthreads();
    if (tid == 0 && atomicAdd(&scratch[MID_DONE], 1u) == gridDim.x - 1u) {
        scratch[MID_DONE] = 0u;
        if (epoch >= (1u << 26)) {  // long before tags could wrap: back to the state of a fresh handle
            for (uint32_t i = 0; i < 3u * MID_MAX_TILES; ++i) scratch[MID_CLAIM + i] = 0u;
            scratch[MID_RESET] = 0u;
            scratch[MID_EPOCH] = 0u;
            __threadfence();
            scratch[MID_CTR] = 0u;
        } else {
            scratch[MID_CTR] = epoch;
        }
    }
}

}  // namespace gs
