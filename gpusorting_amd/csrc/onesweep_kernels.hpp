// onesweep_kernels.hpp — gfx950 (CDNA4, wave64) device code of the OneSweep
// 8-bit LSD radix sort.  Written for MI355X only: 64-lane ballots, LDS-staged
// tiles, chained-scan decoupled look-back on agent-scope relaxed atomics (the
// descriptor word carries flag+count, so the data IS the flag — no fences).
//
// Behavioural spec (what, not how): reference b0nes164/GPUSorting
//   GlobalHistogram      GPUSortingCUDA/Sort/OneSweep.cu:44-123
//   Scan                 GPUSortingCUDA/Sort/OneSweep.cu:125-162
//   DigitBinningPass*    GPUSortingCUDA/Sort/OneSweep.cu:164-344, 346-600
//   key transforms       GPUSortingD3D12/Shaders/SortCommon.hlsl:134-154
//   descending rule      GPUSortingD3D12/Shaders/SortCommon.hlsl:594-597,645-656
//   InitRandom/Validate  GPUSortingCUDA/UtilityKernels.cuh:53-117, 402-479
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gs {

constexpr uint32_t RADIX = 256;
constexpr uint32_t FLAG_NOT_READY = 0;  // tile has published nothing yet
constexpr uint32_t FLAG_REDUCTION = 1;  // count<<2 = this tile's digit count
constexpr uint32_t FLAG_INCLUSIVE = 2;  // count<<2 = count of this and all earlier tiles (+ global digit start)
constexpr uint32_t FLAG_MASK = 3;

constexpr uint32_t STATUS_OK = 0;
constexpr uint32_t STATUS_TIMEOUT = 4;  // == GS_ERR_TIMEOUT

// Bound for every look-back spin (polls, each >= ~0.5 us with the sleep): ~1 s.
constexpr uint32_t SPIN_LIMIT = 1u << 21;

enum : int { KEY_U32 = 0, KEY_I32 = 1, KEY_F32 = 2 };

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

// 16-byte write-through (sc1) accesses to the tile-descriptor rows through a
// buffer resource: one wave moves a whole 256-digit row (1 KiB) per instruction.
// A dword sc1 access is its own fabric transaction (~6x the cost per byte), which
// is what made per-digit dword descriptors cost a third of every pass.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t desc_rsrc(uint32_t* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(base, 0, bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 ld_row16(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16 /* sc1 */);
}
__device__ __forceinline__ void st_row16(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, u32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(v, r, byte_off, 0, 16 /* sc1 */);
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

__device__ __forceinline__ uint32_t wave_reduce_sum(uint32_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
    return v;
}

// ---------------------------------------------------------------------------
// GlobalHistogram: one sweep over the keys, four 256-bin digit histograms.
// 16-byte loads, per-wave LDS histograms (ds_add_u32), one global atomic per
// non-empty bin per block.
// ---------------------------------------------------------------------------
constexpr int GHIST_THREADS = 256;
constexpr int GHIST_WAVES = GHIST_THREADS / 64;

template <int KT>
__global__ __launch_bounds__(GHIST_THREADS) void global_histogram_kernel(
    const uint32_t* __restrict__ keys, uint32_t* ghist, uint32_t n) {
    __shared__ uint32_t s_h[GHIST_WAVES][4 * RADIX];
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < GHIST_WAVES * 4 * RADIX; i += GHIST_THREADS) (&s_h[0][0])[i] = 0;
    __syncthreads();

    uint32_t* h = s_h[tid >> 6];
    const uint32_t nvec = n >> 2;
    const uint4* kv = reinterpret_cast<const uint4*>(keys);
    const uint32_t stride = gridDim.x * GHIST_THREADS;
    for (uint32_t i = blockIdx.x * GHIST_THREADS + tid; i < nvec; i += stride) {
        const uint4 t = kv[i];
        const uint32_t k[4] = {to_bits<KT>(t.x), to_bits<KT>(t.y), to_bits<KT>(t.z), to_bits<KT>(t.w)};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            atomicAdd(&h[k[j] & 255u], 1u);
            atomicAdd(&h[256u + ((k[j] >> 8) & 255u)], 1u);
            atomicAdd(&h[512u + ((k[j] >> 16) & 255u)], 1u);
            atomicAdd(&h[768u + (k[j] >> 24)], 1u);
        }
    }
    // scalar tail (n not a multiple of 4): handled by block 0
    if (blockIdx.x == 0) {
        for (uint32_t i = (nvec << 2) + tid; i < n; i += GHIST_THREADS) {
            const uint32_t k = to_bits<KT>(keys[i]);
            atomicAdd(&h[k & 255u], 1u);
            atomicAdd(&h[256u + ((k >> 8) & 255u)], 1u);
            atomicAdd(&h[512u + ((k >> 16) & 255u)], 1u);
            atomicAdd(&h[768u + (k >> 24)], 1u);
        }
    }
    __syncthreads();
    for (uint32_t b = tid; b < 4 * RADIX; b += GHIST_THREADS) {
        uint32_t s = 0;
#pragma unroll
        for (int w = 0; w < GHIST_WAVES; ++w) s += s_h[w][b];
        if (s) atomicAdd(&ghist[b], s);
    }
}

// ---------------------------------------------------------------------------
// Scan: exclusive prefix of each 256-bin row; seeds descriptor row 0 of pass p
// as INCLUSIVE (so every look-back terminates at row 0 at worst).
// grid = 4 (one block per pass), block = 256.
// ---------------------------------------------------------------------------
#ifndef GS_LB
#define GS_LB 1  // 1 = delegated look-back (scanner waves), 0 = classic per-tile walk
#endif
constexpr uint32_t SLICE_DIGITS_ = 8;  // == SLICE_DIGITS below (RADIX / NSCAN_WAVES)
__global__ __launch_bounds__(256) void scan_kernel(const uint32_t* ghist, uint32_t* desc,
                                                    uint32_t desc_stride /*words per pass*/, uint32_t slice_major) {
    __shared__ uint32_t s_wtot[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t c = ghist[blockIdx.x * RADIX + tid];
    const uint32_t incl = wave_inclusive_scan(c, lane);
    if (lane == 63) s_wtot[wave] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t w = 0; w < wave; ++w) base += s_wtot[w];
    // slice-major exclusive-prefix array (delegated look-back): slice tid/8, row 0, digit tid%8, rows = stride/512
    const uint32_t rows = desc_stride / (2u * RADIX);
    const size_t word = slice_major ? (size_t)(tid / SLICE_DIGITS_) * (rows * SLICE_DIGITS_) + (tid % SLICE_DIGITS_) : tid;
    desc[(size_t)blockIdx.x * desc_stride + word] = ((base + incl - c) << 2) | FLAG_INCLUSIVE;
}

// ---------------------------------------------------------------------------
// DigitBinningPass: one stable 8-bit partition pass with chained-scan
// decoupled look-back.  One tile of THREADS*KPT keys per workgroup.
//   VB  = value bytes (0 keys-only, 4, 8)
//   KT  = key type
// Tile-local order == array order (keys are loaded wave-striped: wave w owns
// 64*KPT consecutive keys, round i = 64 consecutive keys), which is what makes
// each pass stable.
// ---------------------------------------------------------------------------
template <int VB>
struct ValT { using type = uint32_t; };
template <>
struct ValT<8> { using type = uint64_t; };

template <int THREADS, int KPT, int VB>
struct BinCfg {
    static constexpr int WAVES = THREADS / 64;
    static constexpr int TILE = THREADS * KPT;
    static constexpr int STAGE_BYTES = TILE * (VB == 8 ? 8 : 4);
    static constexpr int LDS_BYTES = STAGE_BYTES + WAVES * RADIX * 4 + 3 * RADIX * 4 + 64;
    // residency we ask the register allocator for: as many workgroups per CU as
    // LDS (160 KiB) and the 2048-thread limit admit, so that one workgroup's
    // look-back wait is covered by its neighbours' work
    static constexpr int BPC_LDS = (160 * 1024) / LDS_BYTES;
    static constexpr int BPC_THR = 2048 / THREADS;
    static constexpr int BPC_RAW = BPC_LDS < BPC_THR ? BPC_LDS : BPC_THR;
    static constexpr int BPC = BPC_RAW < 1 ? 1 : BPC_RAW;
    static constexpr int WAVES_PER_SIMD_RAW = BPC * THREADS / 256;
    // never ask for fewer registers than the unrolled tile needs (~1.5 regs/key + temps)
    static constexpr int VGPR_NEED = KPT * (VB == 8 ? 3 : 2) + 32;
    static constexpr int WAVES_PER_SIMD_CAP = 512 / VGPR_NEED < 1 ? 1 : 512 / VGPR_NEED;
    static constexpr int WAVES_PER_SIMD =
        WAVES_PER_SIMD_RAW < WAVES_PER_SIMD_CAP ? WAVES_PER_SIMD_RAW : WAVES_PER_SIMD_CAP;
};

#ifndef GS_EXP
#define GS_EXP 0  // experiment flags (ablation builds only): 1 = no look-back wait
#endif
// GS_EXP & 2: per-tile phase timestamps (10 ns ticks) into the buffer whose
// address the host stored at status[4..5]; 8 words per (pass, tile).
#if (GS_EXP & 2)
#define GS_TRACE(slot) do { if (tid == TRACE_TID) trace[(slot)] = (uint32_t)wall_clock64(); } while (0)
#else
#define GS_TRACE(slot) do { } while (0)
#endif
#ifndef GS_LOOKBACK_BATCH
#define GS_LOOKBACK_BATCH 4  // descriptor rows fetched per look-back round trip
#endif

#ifndef GS_LB
#define GS_LB 1  // 1 = delegated look-back (scanner waves), 0 = classic per-tile walk
#endif
// Delegated look-back.  A per-tile serial walk cannot keep up on this chip: ~40
// tiles start per microsecond across 256 CUs while one dependent descriptor read
// costs ~1.7 us under streaming load, so every tile walks ~30 rows (measured:
// 14 us of a 24 us tile, profiles/r01_tile_phase_trace_classic_lookback.txt).
// Instead the first workgroups of a pass (by ticket) are scanners: NSCAN_WAVES
// scanner waves, wave q owns digits 8q..8q+7.  Descriptors are stored SLICE-MAJOR
// (slice q = [rows][8 digits], 32 bytes per row) so a scanner's view of 32
// consecutive tiles is ONE contiguous 1 KiB: it streams the REDUCTION slices in
// tile order with GS_SCAN_GROUPS x 32 rows in flight, carries the running sum in
// registers (no store->load dependency between steps) and publishes every
// tile's exclusive prefix.  A tile scatters its 256 counts into the 32 slices
// with one store instruction and later polls its own 32 pieces.
constexpr uint32_t NSCAN_WAVES = 32;
constexpr uint32_t SLICE_DIGITS = RADIX / NSCAN_WAVES;  // 8 digits = 32 bytes per row
#ifndef GS_SCAN_GROUPS
#define GS_SCAN_GROUPS 4
#endif

// byte offset of lane l's 16-byte chunk (digits 4l..4l+3) of row `row` in a slice-major array of `rows` rows
__device__ __forceinline__ uint32_t slice_off(uint32_t rows, uint32_t row, uint32_t lane) {
    return (lane >> 1) * (rows * 32u) + row * 32u + (lane & 1u) * 16u;
}

__device__ __forceinline__ void scanner_wave(uint32_t q, __amdgpu_buffer_rsrc_t excl, __amdgpu_buffer_rsrc_t red,
                                             uint32_t tiles, uint32_t* status, uint32_t lane) {
    const uint32_t rows = tiles + 1u;
    const uint32_t r = lane >> 1, c = lane & 1u;  // lane = (row within a 32-row group, half of the 32-byte piece)
    const uint32_t base = q * (rows * 32u) + c * 16u;
    u32x4 carry = ld_row16(excl, base) >> 2;  // row 0: global digit start, seeded by scan_kernel
    uint32_t t0 = 0, spins = 0;
    while (t0 < tiles) {
        u32x4 v[GS_SCAN_GROUPS];
#pragma unroll
        for (int g = 0; g < GS_SCAN_GROUPS; ++g) {
            const uint32_t row = t0 + g * 32u + r;
            v[g] = ld_row16(red, base + (row < tiles ? row : 0u) * 32u);
            if (row >= tiles) v[g] = u32x4{0, 0, 0, 0};
        }
        asm volatile("" ::: "memory");  // slices are re-read on every trip
        bool stop = false;
        uint32_t advanced = 0;
#pragma unroll
        for (int g = 0; g < GS_SCAN_GROUPS; ++g) {
            if (!stop) {
                const bool ok = ((v[g].x & FLAG_MASK) != 0u) && ((v[g].y & FLAG_MASK) != 0u) &&
                                ((v[g].z & FLAG_MASK) != 0u) && ((v[g].w & FLAG_MASK) != 0u);
                const unsigned long long m = __builtin_amdgcn_ballot_w64(ok);
                const unsigned long long both = m & (m >> 1) & 0x5555555555555555ull;  // bit 2r: row r complete
                const unsigned long long miss = ~both & 0x5555555555555555ull;
                const uint32_t nready = miss ? (uint32_t)(__builtin_ctzll(miss) >> 1) : 32u;  // leading complete rows
                u32x4 x = (r < nready) ? (v[g] >> 2) : u32x4{0, 0, 0, 0};
#pragma unroll
                for (int d = 2; d < 64; d <<= 1) {  // inclusive sum over rows (lane stride 2)
                    u32x4 y;
                    y.x = __shfl_up(x.x, d, 64); y.y = __shfl_up(x.y, d, 64);
                    y.z = __shfl_up(x.z, d, 64); y.w = __shfl_up(x.w, d, 64);
                    if (lane >= (uint32_t)d) x += y;
                }
                if (r < nready)
                    st_row16(excl, base + (t0 + advanced + r + 1u) * 32u, ((carry + x) << 2) | FLAG_INCLUSIVE);
                if (nready) {
                    const int src = (int)((nready - 1u) * 2u + c);
                    u32x4 t;
                    t.x = __shfl(x.x, src, 64); t.y = __shfl(x.y, src, 64);
                    t.z = __shfl(x.z, src, 64); t.w = __shfl(x.w, src, 64);
                    carry += t;
                }
                advanced += nready;
                if (nready < 32u) stop = true;
            }
        }
        t0 += advanced;
        if (advanced < GS_SCAN_GROUPS * 32u && t0 < tiles) {
            if (advanced) __builtin_amdgcn_s_sleep(2);  // do not hammer rows that are still being written
            else __builtin_amdgcn_s_sleep(8);
            if (!advanced && (++spins > SPIN_LIMIT || ((spins & 1023u) == 0 && ld_agent(status) != STATUS_OK))) {
                st_agent(status, STATUS_TIMEOUT);
                return;
            }
        }
    }
}

template <int THREADS, int KPT, int VB, int KT, int RANK>
__global__ __launch_bounds__(THREADS, (BinCfg<THREADS, KPT, VB>::WAVES_PER_SIMD)) void digit_binning_kernel(
    const uint32_t* keys_in, uint32_t* keys_out, const void* vals_in_, void* vals_out_,
    uint32_t* desc,          // this pass: (tiles+1) rows of 256 descriptor words
    uint32_t* tile_counter,  // this pass's ticket counter
    uint32_t* status, uint32_t n, uint32_t shift, uint32_t reverse) {
    using Cfg = BinCfg<THREADS, KPT, VB>;
    using V = typename ValT<VB>::type;
    constexpr int WAVES = Cfg::WAVES;
    constexpr uint32_t TILE = Cfg::TILE;
    static_assert(THREADS >= 256 && THREADS % 64 == 0, "need >= 256 threads");
    static_assert(KPT % 4 == 0 && TILE <= 65536, "offsets are packed 2 x 16 bit, digits 4 x 8 bit");

    __shared__ __attribute__((aligned(16))) unsigned char s_raw[Cfg::LDS_BYTES];
    uint32_t* s_stage = reinterpret_cast<uint32_t*>(s_raw);
    uint32_t* s_whist = reinterpret_cast<uint32_t*>(s_raw + Cfg::STAGE_BYTES);
    uint32_t* s_dpre = s_whist + WAVES * RADIX;  // tile-local exclusive digit prefix
    uint32_t* s_gbase = s_dpre + RADIX;          // global base of digit run minus s_dpre
    uint32_t* s_tot = s_gbase + RADIX;           // this tile's digit counts
    uint32_t* s_misc = s_tot + RADIX;            // [0] tile id, [4..7] wave totals of the digit scan
    constexpr uint32_t LBW = WAVES - 1;          // the wave that owns the descriptor row (4 digits per lane)

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;

    for (uint32_t i = tid; i < WAVES * RADIX; i += THREADS) s_whist[i] = 0;
    if (tid == 0) s_misc[0] = atomicAdd(tile_counter, 1u);  // dynamic tile id: predecessors have started
    __syncthreads();
#if GS_LB
    // two slice-major arrays of (tiles+1) rows: exclusive prefixes (row t = prefix of tile t), then REDUCTIONs
    const uint32_t ntiles = (n + TILE - 1) / TILE;
    const uint32_t nrows = ntiles + 1u;
    const __amdgpu_buffer_rsrc_t xrsrc = desc_rsrc(desc, nrows * (RADIX * 4u));
    const __amdgpu_buffer_rsrc_t rrsrc = desc_rsrc(desc + (size_t)nrows * RADIX, nrows * (RADIX * 4u));
    constexpr uint32_t NSCAN = NSCAN_WAVES / WAVES;  // scanner workgroups: the first tickets of the pass
    if (s_misc[0] < NSCAN) {
        scanner_wave(s_misc[0] * WAVES + wave, xrsrc, rrsrc, ntiles, status, lane);
        return;
    }
    const uint32_t tile = s_misc[0] - NSCAN;
#else
    const uint32_t tile = s_misc[0];
#endif
#if (GS_EXP & 2)
    constexpr uint32_t TRACE_TID = (WAVES - 1) * 64;  // lane 0 of the look-back wave
    uint32_t* trace = reinterpret_cast<uint32_t*>(((unsigned long long)status[5] << 32) | status[4]) +
                      ((size_t)(shift >> 3) * ((n + TILE - 1) / TILE) + tile) * 8;
    uint32_t trace_trips = 0, trace_rows = 0;
#endif
    GS_TRACE(0);
    const uint32_t tile_base = tile * TILE;
    const uint32_t count = (n - tile_base < TILE) ? (n - tile_base) : TILE;
    const bool full = (count == TILE);

    // ---- load (wave-striped, coalesced 256 B per wave-instruction) ----
    uint32_t key[KPT];
    const uint32_t my_base = tile_base + wave * (64u * KPT) + lane;
    if (full) {
#pragma unroll
        for (int i = 0; i < KPT; ++i) key[i] = to_bits<KT>(keys_in[my_base + i * 64u]);
    } else {
        // dummy keys with the highest digit sort last inside the tile and are never written
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const uint32_t idx = my_base + i * 64u;
            key[i] = idx < n ? to_bits<KT>(keys_in[idx]) : 0xffffffffu;
        }
    }

    // ---- rank every key among the keys of its digit inside this wave ----
    // offp[] holds two 16-bit ranks (later: tile-local positions) per register.
    uint32_t* whist = s_whist + wave * RADIX;
    uint32_t offp[KPT / 2];
#pragma unroll
    for (int i = 0; i < KPT / 2; ++i) offp[i] = 0;
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
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const uint32_t d = (key[i] >> shift) & 255u;
            const uint32_t r = __hip_atomic_fetch_add(&whist[d], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            offp[i >> 1] |= r << (16 * (i & 1));
        }
    }
    GS_TRACE(1);
    __syncthreads();

    // ---- per-digit: exclusive prefix over waves, tile total, digit scan ----
    uint32_t tile_total = 0, scan_incl = 0;
    if (tid < RADIX) {
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const uint32_t c = s_whist[w * RADIX + tid];
            s_whist[w * RADIX + tid] = run;
            run += c;
        }
        tile_total = run;
        s_tot[tid] = run;
        scan_incl = wave_inclusive_scan(run, lane);
        if (lane == 63) s_misc[4 + wave] = scan_incl;
    }
    __syncthreads();
    if (tid < RADIX) {
        uint32_t wbase = 0;
        for (uint32_t w = 0; w < wave; ++w) wbase += s_misc[4 + w];
        const uint32_t dpre = wbase + scan_incl - tile_total;
        s_dpre[tid] = dpre;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) s_whist[w * RADIX + tid] += dpre;
    }
    // publish this tile's digit counts (REDUCTION): one 1 KiB row, one store instruction
#if !GS_LB
    const __amdgpu_buffer_rsrc_t drsrc = desc_rsrc(desc, ((n + TILE - 1) / TILE + 1u) * (RADIX * 4u));
#endif
    u32x4 tot4 = {0, 0, 0, 0};
    if (wave == LBW) {
        tot4 = *reinterpret_cast<const u32x4*>(s_tot + 4u * lane);
#if GS_LB
        st_row16(rrsrc, slice_off(nrows, tile, lane), (tot4 << 2) | FLAG_REDUCTION);
#else
        st_row16(drsrc, (tile + 1u) * (RADIX * 4u) + lane * 16u, (tot4 << 2) | FLAG_REDUCTION);
#endif
    }
    GS_TRACE(2);
    __syncthreads();

    // ---- stage keys in LDS, sorted by digit (stable) ----
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const uint32_t d = (key[i] >> shift) & 255u;
        const uint32_t lpos = ((offp[i >> 1] >> (16 * (i & 1))) & 0xffffu) + s_whist[wave * RADIX + d];
        s_stage[lpos] = key[i];
        if constexpr (VB != 0) {  // values follow the same positions later
            if ((i & 1) == 0) offp[i >> 1] = (offp[i >> 1] & 0xffff0000u) | lpos;
            else offp[i >> 1] = (offp[i >> 1] & 0x0000ffffu) | (lpos << 16);
        }
    }

#if GS_LB
    // ---- delegated look-back: poll this tile's exclusive-prefix row (lane l: digits 4l..4l+3) ----
    if (wave == LBW) {
        u32x4 e = {0, 0, 0, 0};
        uint32_t spins = 0;
        GS_TRACE(3);
        while (!(GS_EXP & 1)) {
#if (GS_EXP & 2)
            ++trace_trips;
#endif
            e = ld_row16(xrsrc, slice_off(nrows, tile, lane));
            asm volatile("" ::: "memory");
            if (((e.x & FLAG_MASK) != 0u) && ((e.y & FLAG_MASK) != 0u) && ((e.z & FLAG_MASK) != 0u) &&
                ((e.w & FLAG_MASK) != 0u))
                break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_LIMIT || ((spins & 1023u) == 0 && ld_agent(status) != STATUS_OK)) {
                st_agent(status, STATUS_TIMEOUT);  // give up: result is invalid, but nothing hangs
                break;
            }
        }
        const u32x4 dp = *reinterpret_cast<const u32x4*>(s_dpre + 4u * lane);
        *reinterpret_cast<u32x4*>(s_gbase + 4u * lane) = (e >> 2) - dp;
        GS_TRACE(4);
#if (GS_EXP & 2)
        if (tid == TRACE_TID) { trace[6] = trace_trips; trace[7] = trace_rows | (__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) << 16); }
#endif
    }
    __syncthreads();

#else
    // ---- decoupled look-back by ONE wave: lane l owns digits 4l..4l+3 ----
    // Row k holds tile k-1's descriptor; row 0 = global digit start (INCLUSIVE), so
    // every walk ends at row 0 at the latest.  GS_LOOKBACK_BATCH rows per round trip.
    if (wave == LBW) {
        u32x4 prev = {0, 0, 0, 0};
        uint32_t pending = (GS_EXP & 1) ? 0u : 0xfu;  // digits (bit c) still walking
        int32_t k = (int32_t)tile;
        uint32_t spins = 0;
        GS_TRACE(3);
        while (pending) {
#if (GS_EXP & 2)
            ++trace_trips;
#endif
            u32x4 v[GS_LOOKBACK_BATCH];
#pragma unroll
            for (int j = 0; j < GS_LOOKBACK_BATCH; ++j) {
                const int32_t r = k - j < 0 ? 0 : k - j;
                v[j] = ld_row16(drsrc, (uint32_t)r * (RADIX * 4u) + lane * 16u);
            }
            asm volatile("" ::: "memory");  // the rows are re-read on every trip
            bool stalled = false;
#pragma unroll
            for (int j = 0; j < GS_LOOKBACK_BATCH; ++j) {
                if (pending && !stalled) {
                    bool ready = true;
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if ((pending >> c) & 1u) ready = ready && ((v[j][c] & FLAG_MASK) != FLAG_NOT_READY);
                    if (!ready) {
                        stalled = true;
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if ((pending >> c) & 1u) {
                                prev[c] += v[j][c] >> 2;
                                if ((v[j][c] & FLAG_MASK) == FLAG_INCLUSIVE) pending &= ~(1u << c);
                            }
                        --k;
#if (GS_EXP & 2)
                        ++trace_rows;
#endif
                    }
                }
            }
            if (stalled) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > SPIN_LIMIT || ((spins & 1023u) == 0 && ld_agent(status) != STATUS_OK)) {
                    st_agent(status, STATUS_TIMEOUT);  // give up: result is invalid, but nothing hangs
                    pending = 0;
                }
            }
        }
        st_row16(drsrc, (tile + 1u) * (RADIX * 4u) + lane * 16u, ((prev + tot4) << 2) | FLAG_INCLUSIVE);
        const u32x4 dp = *reinterpret_cast<const u32x4*>(s_dpre + 4u * lane);
        *reinterpret_cast<u32x4*>(s_gbase + 4u * lane) = prev - dp;
        GS_TRACE(4);
#if (GS_EXP & 2)
        if (tid == TRACE_TID) { trace[6] = trace_trips; trace[7] = trace_rows | (__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) << 16); }
#endif
    }
    __syncthreads();

#endif
    // ---- scatter runs to global memory (slot i of the stage -> s_gbase[digit] + i) ----
    uint32_t digs[KPT / 4];  // digit of stage slot tid + j*THREADS, 4 per register (value phase)
#pragma unroll
    for (int j = 0; j < KPT / 4; ++j) digs[j] = 0;
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const uint32_t i = tid + j * THREADS;
        const uint32_t kb = s_stage[i];
        const uint32_t d = (kb >> shift) & 255u;
        uint32_t o = s_gbase[d] + i;
        if (reverse) o = n - 1u - o;
        if (GS_EXP & 1) o = (tile_base + i) % n;  // ablation: positions are meaningless without the look-back
        if (full || i < count) keys_out[o] = from_bits<KT>(kb);
        if constexpr (VB != 0) digs[j >> 2] |= d << (8 * (j & 3));
    }

    GS_TRACE(5);
    if constexpr (VB != 0) {
        const V* vals_in = reinterpret_cast<const V*>(vals_in_);
        V* vals_out = reinterpret_cast<V*>(vals_out_);
        V* s_vstage = reinterpret_cast<V*>(s_raw);
        __syncthreads();  // everyone is done reading the key stage
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const uint32_t idx = my_base + i * 64u;
            const V val = (full || idx < n) ? vals_in[idx] : V(0);
            s_vstage[(offp[i >> 1] >> (16 * (i & 1))) & 0xffffu] = val;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const uint32_t i = tid + j * THREADS;
            uint32_t o = s_gbase[(digs[j >> 2] >> (8 * (j & 3))) & 255u] + i;
            if (reverse) o = n - 1u - o;
            if (GS_EXP & 1) o = (tile_base + i) % n;
            if (full || i < count) vals_out[o] = s_vstage[i];
        }
    }
}

// ---------------------------------------------------------------------------
// DigitBinningPass, persistent + software-pipelined form (the default).
//
// Every chained scan must wait for its slowest predecessor, and on this chip the
// spread of REDUCTION publish times alone is ~6 us at p99 (profiles/
// r01_tile_phase_trace_*.txt), so the answer is latency TOLERANCE: more tiles in
// flight per CU at the same footprint.  A workgroup holds TWO tiles:
//   tile A  staged in LDS (sorted by digit), counts published, waiting for its prefix
//   tile B  keys in registers: loaded, ranked, counts published while A waits
// Per iteration:  rank B -> publish B -> [look-back A -> scatter A] -> stage B -> fetch C.
// Descriptors: row-major, one dword per (tile, digit), thread d of the first 256
// threads owns digit d across the whole pipeline (its running state stays in registers).
// Forward progress: tickets are taken in order; a tile's counts are published before
// its workgroup waits for anything that depends on a LARGER tile index.
// ---------------------------------------------------------------------------
template <int THREADS, int KPT, int VB, int KT, int RANK>
__global__ __launch_bounds__(THREADS, (BinCfg<THREADS, KPT, VB>::WAVES_PER_SIMD)) void digit_binning_persistent(
    const uint32_t* keys_in, uint32_t* keys_out, const void* vals_in_, void* vals_out_,
    uint32_t* desc,          // this pass: (tiles+1) rows of 256 descriptor words, row-major
    uint32_t* tile_counter,  // this pass's ticket counter
    uint32_t* status, uint32_t n, uint32_t shift, uint32_t reverse) {
    using Cfg = BinCfg<THREADS, KPT, VB>;
    using V = typename ValT<VB>::type;
    constexpr int WAVES = Cfg::WAVES;
    constexpr uint32_t TILE = Cfg::TILE;
    static_assert(THREADS >= 256 && THREADS % 64 == 0, "need >= 256 threads");
    static_assert(KPT % 4 == 0 && TILE <= 65536, "offsets are packed 2 x 16 bit, digits 4 x 8 bit");

    __shared__ __attribute__((aligned(16))) unsigned char s_raw[Cfg::LDS_BYTES];
    uint32_t* s_stage = reinterpret_cast<uint32_t*>(s_raw);
    uint32_t* s_whist = reinterpret_cast<uint32_t*>(s_raw + Cfg::STAGE_BYTES);
    uint32_t* s_gbase = s_whist + WAVES * RADIX;  // tile A: global base of digit run minus local run start
    uint32_t* s_misc = s_gbase + 3 * RADIX;       // [0] next ticket, [4..7] wave totals of the digit scan

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t ntiles = (n + TILE - 1) / TILE;
    uint32_t* whist = s_whist + wave * RADIX;

    for (uint32_t i = tid; i < WAVES * RADIX; i += THREADS) s_whist[i] = 0;
    if (tid == 0) s_misc[0] = atomicAdd(tile_counter, 1u);
    __syncthreads();
    uint32_t tileB = s_misc[0];
    if (tileB >= ntiles) return;

    // per-thread pipeline state
    uint32_t key[KPT];      // tile B
    uint32_t offp[KPT / 2]; // tile B ranks -> tile-local positions (2 x 16 bit)
    uint32_t tileA = 0, totA = 0, dpreA = 0;  // tile A (totA/dpreA: digit `tid`, threads < 256)
    bool haveA = false;
    uint32_t posA[VB != 0 ? KPT / 2 : 1];  // tile A tile-local positions, kept for the value phase

    auto load_keys = [&](uint32_t tile) {
        const uint32_t base = tile * TILE + wave * (64u * KPT) + lane;
        if (tile * TILE + TILE <= n) {
#pragma unroll
            for (int i = 0; i < KPT; ++i) key[i] = to_bits<KT>(keys_in[base + i * 64u]);
        } else {
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                const uint32_t idx = base + i * 64u;
                key[i] = idx < n ? to_bits<KT>(keys_in[idx]) : 0xffffffffu;  // dummies sort last, never written
            }
        }
    };
    load_keys(tileB);

    while (true) {
        // ---- (a) rank tile B inside each wave ----
#pragma unroll
        for (int i = 0; i < KPT / 2; ++i) offp[i] = 0;
        if constexpr (RANK == 0) {
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
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
                offp[i >> 1] |= (pre + below) << (16 * (i & 1));
            }
        } else {
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                const uint32_t d = (key[i] >> shift) & 255u;
                const uint32_t r = __hip_atomic_fetch_add(&whist[d], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                offp[i >> 1] |= r << (16 * (i & 1));
            }
        }
        __syncthreads();

        // ---- (b) tile B: per-digit prefix over waves, publish counts, digit scan, fold ----
        uint32_t totB = 0, scan_incl = 0, dpreB = 0;
        if (tid < RADIX) {
            uint32_t run = 0;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) {
                const uint32_t c = s_whist[w * RADIX + tid];
                s_whist[w * RADIX + tid] = run;
                run += c;
            }
            totB = run;
            st_agent(&desc[(size_t)(tileB + 1u) * RADIX + tid], (run << 2) | FLAG_REDUCTION);
            scan_incl = wave_inclusive_scan(run, lane);
            if (lane == 63) s_misc[4 + wave] = scan_incl;
        }
        __syncthreads();
        if (tid < RADIX) {
            uint32_t wbase = 0;
            for (uint32_t w = 0; w < wave; ++w) wbase += s_misc[4 + w];
            dpreB = wbase + scan_incl - totB;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) s_whist[w * RADIX + tid] += dpreB;
        }

        // ---- (c) tile A: decoupled look-back (digit `tid`), then (d) scatter it ----
        if (haveA) {
            if (tid < RADIX) {
                uint32_t prev = 0;
                int32_t k = (int32_t)tileA;  // row k = tile k-1; row 0 = global digit start (INCLUSIVE)
                uint32_t spins = 0;
                bool done = (GS_EXP & 1) != 0;
                while (!done) {
                    uint32_t v[GS_LOOKBACK_BATCH];
#pragma unroll
                    for (int j = 0; j < GS_LOOKBACK_BATCH; ++j) {
                        const int32_t r = k - j < 0 ? 0 : k - j;
                        v[j] = ld_agent(&desc[(size_t)r * RADIX + tid]);
                    }
                    bool stalled = false;
#pragma unroll
                    for (int j = 0; j < GS_LOOKBACK_BATCH; ++j) {
                        if (!done && !stalled) {
                            const uint32_t f = v[j] & FLAG_MASK;
                            if (f == FLAG_INCLUSIVE) { prev += v[j] >> 2; done = true; }
                            else if (f == FLAG_REDUCTION) { prev += v[j] >> 2; --k; }
                            else stalled = true;
                        }
                    }
                    if (stalled) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > SPIN_LIMIT || ((spins & 1023u) == 0 && ld_agent(status) != STATUS_OK)) {
                            st_agent(status, STATUS_TIMEOUT);  // give up: result invalid, nothing hangs
                            done = true;
                        }
                    }
                }
                st_agent(&desc[(size_t)(tileA + 1u) * RADIX + tid], ((prev + totA) << 2) | FLAG_INCLUSIVE);
                s_gbase[tid] = prev - dpreA;
            }
            __syncthreads();
            const uint32_t baseA = tileA * TILE;
            const uint32_t countA = (n - baseA < TILE) ? (n - baseA) : TILE;
            const bool fullA = countA == TILE;
            uint32_t digs[VB != 0 ? KPT / 4 : 1];
            if constexpr (VB != 0) {
#pragma unroll
                for (int j = 0; j < KPT / 4; ++j) digs[j] = 0;
            }
#pragma unroll
            for (int j = 0; j < KPT; ++j) {
                const uint32_t i = tid + j * THREADS;
                const uint32_t kb = s_stage[i];
                const uint32_t d = (kb >> shift) & 255u;
                uint32_t o = s_gbase[d] + i;
                if (reverse) o = n - 1u - o;
                if (GS_EXP & 1) o = (baseA + i) % n;
                if (fullA || i < countA) keys_out[o] = from_bits<KT>(kb);
                if constexpr (VB != 0) digs[j >> 2] |= d << (8 * (j & 3));
            }
            if constexpr (VB != 0) {
                const V* vals_in = reinterpret_cast<const V*>(vals_in_);
                V* vals_out = reinterpret_cast<V*>(vals_out_);
                V* s_vstage = reinterpret_cast<V*>(s_raw);
                const uint32_t vb = baseA + wave * (64u * KPT) + lane;
                __syncthreads();  // key stage fully read
#pragma unroll
                for (int i = 0; i < KPT; ++i) {
                    const uint32_t idx = vb + i * 64u;
                    const V val = (fullA || idx < n) ? vals_in[idx] : V(0);
                    s_vstage[(posA[i >> 1] >> (16 * (i & 1))) & 0xffffu] = val;
                }
                __syncthreads();
#pragma unroll
                for (int j = 0; j < KPT; ++j) {
                    const uint32_t i = tid + j * THREADS;
                    uint32_t o = s_gbase[(digs[j >> 2] >> (8 * (j & 3))) & 255u] + i;
                    if (reverse) o = n - 1u - o;
                    if (GS_EXP & 1) o = (baseA + i) % n;
                    if (fullA || i < countA) vals_out[o] = s_vstage[i];
                }
            }
        }
        __syncthreads();  // stage free; folded whist of tile B visible

        // ---- (e) stage tile B in LDS sorted by digit; it becomes tile A ----
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const uint32_t d = (key[i] >> shift) & 255u;
            const uint32_t lpos = ((offp[i >> 1] >> (16 * (i & 1))) & 0xffffu) + s_whist[wave * RADIX + d];
            s_stage[lpos] = key[i];
            if constexpr (VB != 0) {
                if ((i & 1) == 0) posA[i >> 1] = lpos;
                else posA[i >> 1] |= lpos << 16;
            }
        }
        tileA = tileB; totA = totB; dpreA = dpreB; haveA = true;
        if (tid == 0) s_misc[0] = atomicAdd(tile_counter, 1u);
        __syncthreads();  // whist reads done, stage written, ticket visible
        tileB = s_misc[0];
        for (uint32_t i = tid; i < WAVES * RADIX; i += THREADS) s_whist[i] = 0;
        if (tileB >= ntiles) break;
        load_keys(tileB);
        __syncthreads();  // whist zeroed before the next rank  (also: nobody re-reads s_misc[0] late)
    }

    // ---- drain: the last tile A of this workgroup ----
    if (tid < RADIX) {
        uint32_t prev = 0;
        int32_t k = (int32_t)tileA;
        uint32_t spins = 0;
        bool done = (GS_EXP & 1) != 0;
        while (!done) {
            const uint32_t v = ld_agent(&desc[(size_t)k * RADIX + tid]);
            const uint32_t f = v & FLAG_MASK;
            if (f == FLAG_INCLUSIVE) { prev += v >> 2; done = true; }
            else if (f == FLAG_REDUCTION) { prev += v >> 2; --k; }
            else {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > SPIN_LIMIT || ((spins & 1023u) == 0 && ld_agent(status) != STATUS_OK)) {
                    st_agent(status, STATUS_TIMEOUT);
                    done = true;
                }
            }
        }
        st_agent(&desc[(size_t)(tileA + 1u) * RADIX + tid], ((prev + totA) << 2) | FLAG_INCLUSIVE);
        s_gbase[tid] = prev - dpreA;
    }
    __syncthreads();
    {
        const uint32_t baseA = tileA * TILE;
        const uint32_t countA = (n - baseA < TILE) ? (n - baseA) : TILE;
        uint32_t digs[VB != 0 ? KPT / 4 : 1];
        if constexpr (VB != 0) {
#pragma unroll
            for (int j = 0; j < KPT / 4; ++j) digs[j] = 0;
        }
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const uint32_t i = tid + j * THREADS;
            const uint32_t kb = s_stage[i];
            const uint32_t d = (kb >> shift) & 255u;
            uint32_t o = s_gbase[d] + i;
            if (reverse) o = n - 1u - o;
            if (GS_EXP & 1) o = (baseA + i) % n;
            if (i < countA) keys_out[o] = from_bits<KT>(kb);
            if constexpr (VB != 0) digs[j >> 2] |= d << (8 * (j & 3));
        }
        if constexpr (VB != 0) {
            const V* vals_in = reinterpret_cast<const V*>(vals_in_);
            V* vals_out = reinterpret_cast<V*>(vals_out_);
            V* s_vstage = reinterpret_cast<V*>(s_raw);
            const uint32_t vb = baseA + wave * (64u * KPT) + lane;
            __syncthreads();
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                const uint32_t idx = vb + i * 64u;
                const V val = idx < n ? vals_in[idx] : V(0);
                s_vstage[(posA[i >> 1] >> (16 * (i & 1))) & 0xffffu] = val;
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < KPT; ++j) {
                const uint32_t i = tid + j * THREADS;
                uint32_t o = s_gbase[(digs[j >> 2] >> (8 * (j & 3))) & 255u] + i;
                if (reverse) o = n - 1u - o;
                if (GS_EXP & 1) o = (baseA + i) % n;
                if (i < countA) vals_out[o] = s_vstage[i];
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Hardware probe for RANK 1: does a returning LDS atomic hand out its results
// in ascending lane order among the lanes of ONE wave-instruction that hit the
// same address?  Every wave draws pseudo-random digits (several skews), issues
// ROUNDS back-to-back ds_add_rtn on a private 256-counter table and compares
// each returned value with the exact stable rank computed with ballots.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(512) void lds_atomic_order_probe(uint32_t seed, uint32_t iters, uint32_t* failures) {
    __shared__ uint32_t s_cnt[8][RADIX];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t* cnt = s_cnt[wave];
    uint32_t x = (blockIdx.x * 512u + tid) * 2654435761u + seed * 40503u + 12345u;
    uint32_t bad = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        for (uint32_t j = lane; j < RADIX; j += 64) cnt[j] = 0;
        uint32_t shadow[4] = {0, 0, 0, 0};  // exact counters, digit d lives in lane d&63, slot d>>6 (kept via ballots below)
        (void)shadow;
        const uint32_t mode = (it + blockIdx.x) & 7u;  // 0: uniform, ..., 7: nearly all equal
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

}  // namespace gs
