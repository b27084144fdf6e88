/*
 * oracle/ref_onesweep.cpp — TEST INFRASTRUCTURE ONLY: runs the REFERENCE's own OneSweep kernels on the CPU.
 *
 * GPUSortingCUDA/Sort/OneSweep.cu (GlobalHistogram, Scan, DigitBinningPassKeysOnly, DigitBinningPassPairs) is
 * compiled FROM WHERE IT LIES under /root/reference (nothing is copied) and executed by the SIMT emulator in
 * oracle/shim/ (32-lane warps, fibers, warp collectives, block barriers; blocks in blockIdx order).
 *
 * What is NOT the reference's code here:
 *   - the warp primitives of GPUSortingCUDA/Utils.cuh:22-143.  They are built on inline PTX (%laneid,
 *     %lanemask_*), which no host compiler assembles, so that header is parsed under other names (dead code) and
 *     the same functions are provided below, written with the same shuffle sequences the reference uses;
 *   - the host-side launch sequence: OneSweepDispatcher.cuh:301-363 uses cudaMemset and <<< >>>; it is restated in
 *     run() below line for line (clear, GlobalHistogram<<<ceil(n/65536),128>>>, Scan<<<4,256>>>,
 *     4 x DigitBinningPass<<<ceil(n/7680),512>>> with ping-pong buffers).  The pass-histogram allocations get ONE
 *     extra row: the reference's last tile publishes its descriptor to row `tiles`, past the end of its own
 *     tiles*256 allocation when size == maxSize (OneSweep.cu:266,316).
 *
 * Output: oracle/_ref/libref_onesweep.so (git-ignored; built only where /root/reference exists).
 */
#define GS_SIMT_EMU 1
#include "cuda_runtime.h"  // oracle/shim -> simt_emu.h

// 1) the reference's warp primitives, parsed but dead (inline PTX inside)
#define getLaneId ref_ptx_getLaneId
#define getLaneMaskLt ref_ptx_getLaneMaskLt
#define getLaneMaskGt ref_ptx_getLaneMaskGt
#define getLaneMaskGe ref_ptx_getLaneMaskGe
#define InclusiveWarpScan ref_ptx_InclusiveWarpScan
#define ActiveInclusiveWarpScan ref_ptx_ActiveInclusiveWarpScan
#define InclusiveWarpScanCircularShift ref_ptx_InclusiveWarpScanCircularShift
#define ActiveInclusiveWarpScanCircularShift ref_ptx_ActiveInclusiveWarpScanCircularShift
#define ExclusiveWarpScan ref_ptx_ExclusiveWarpScan
#define ActiveExclusiveWarpScan ref_ptx_ActiveExclusiveWarpScan
#define WarpReduceSum ref_ptx_WarpReduceSum
#include "Utils.cuh"  // -I/root/reference/GPUSortingCUDA; OneSweep.cuh's "../Utils.cuh" is the same file (#pragma once)
#undef getLaneId
#undef getLaneMaskLt
#undef getLaneMaskGt
#undef getLaneMaskGe
#undef InclusiveWarpScan
#undef ActiveInclusiveWarpScan
#undef InclusiveWarpScanCircularShift
#undef ActiveInclusiveWarpScanCircularShift
#undef ExclusiveWarpScan
#undef ActiveExclusiveWarpScan
#undef WarpReduceSum

// 2) the same primitives for the emulator (semantics of Utils.cuh:22-143, 32 lanes)
static inline uint32_t getLaneId() { return gs_emu::cur->lane; }
static inline unsigned getLaneMaskLt() { return (1u << gs_emu::cur->lane) - 1u; }
static inline unsigned getLaneMaskGt() { return gs_emu::cur->lane == 31 ? 0u : ~((2u << gs_emu::cur->lane) - 1u); }
static inline unsigned getLaneMaskGe() { return ~((1u << gs_emu::cur->lane) - 1u); }
static inline uint32_t gs_scan_up(unsigned mask, uint32_t val) {  // Kogge-Stone over shfl_up, as Utils.cuh:51-60
    for (int i = 1; i <= 16; i <<= 1) {
        const uint32_t t = __shfl_up_sync(mask, val, i, 32);
        if (getLaneId() >= (uint32_t)i) val += t;
    }
    return val;
}
static inline uint32_t InclusiveWarpScan(uint32_t v) { return gs_scan_up(0xffffffffu, v); }
static inline uint32_t ActiveInclusiveWarpScan(uint32_t v) { return gs_scan_up(__activemask(), v); }
static inline uint32_t InclusiveWarpScanCircularShift(uint32_t v) {  // Utils.cuh:76-86: lane l gets lane l-1's sum, lane 0 the total
    v = gs_scan_up(0xffffffffu, v);
    return __shfl_sync(0xffffffffu, v, (int)(getLaneId() + LANE_MASK & LANE_MASK));
}
static inline uint32_t ActiveInclusiveWarpScanCircularShift(uint32_t v) {
    const unsigned m = __activemask();
    v = gs_scan_up(m, v);
    return __shfl_sync(m, v, (int)(getLaneId() + LANE_MASK & LANE_MASK));
}
static inline uint32_t ExclusiveWarpScan(uint32_t v) {  // Utils.cuh:101-112
    v = gs_scan_up(0xffffffffu, v);
    const uint32_t t = __shfl_up_sync(0xffffffffu, v, 1, 32);
    return getLaneId() ? t : 0;
}
static inline uint32_t ActiveExclusiveWarpScan(uint32_t v) {  // Utils.cuh:114-126
    const unsigned m = __activemask();
    v = gs_scan_up(m, v);
    const uint32_t t = __shfl_up_sync(m, v, 1, 32);
    return getLaneId() ? t : 0;
}
static inline uint32_t WarpReduceSum(uint32_t v) {
    for (int m = 16; m; m >>= 1) v += __shfl_xor_sync(0xffffffffu, v, m, LANE_COUNT);
    return v;
}

// 3) the reference's kernels
#include "Sort/OneSweep.cu"
#include "UtilityKernels.cuh"  // Validate (:402-479): the reference's pass criterion

namespace {
inline uint32_t div_round_up(uint32_t x, uint32_t y) { return (x + y - 1) / y; }  // OneSweepDispatcher.cuh:296-299

/* OneSweepDispatcher.cuh:301-363 (ClearMemory + DispatchKernelsKeysOnly / DispatchKernelsPairs) */
void run(uint32_t* sort, uint32_t* payload, uint32_t size, uint32_t* out_global_hist, uint32_t* out_after_pass,
         uint32_t* out_payload_after_pass) {
    const uint32_t k_radix = 256, k_passes = 4, k_partitionSize = 7680, k_globalHistPartitionSize = 65536;
    const uint32_t k_globalHistThreads = 128, k_binningThreads = 512;
    const uint32_t globalHistThreadBlocks = div_round_up(size, k_globalHistPartitionSize);
    const uint32_t binningThreadBlocks = div_round_up(size, k_partitionSize);
    std::vector<uint32_t> alt(size), altPayload(payload ? size : 0), index(k_passes, 0), globalHistogram(k_radix * k_passes, 0);
    std::vector<uint32_t> passHist[4];
    for (auto& h : passHist) h.assign((size_t)k_radix * (binningThreadBlocks + 1), 0);  // +1 row: see the header comment
    uint32_t* k[2] = {sort, alt.data()};
    uint32_t* v[2] = {payload, payload ? altPayload.data() : nullptr};

    gs_emu::launch(globalHistThreadBlocks, k_globalHistThreads, [&] { OneSweep::GlobalHistogram(sort, globalHistogram.data(), size); });
    if (out_global_hist) memcpy(out_global_hist, globalHistogram.data(), sizeof(uint32_t) * k_radix * k_passes);
    gs_emu::launch(k_passes, k_radix, [&] {
        OneSweep::Scan(globalHistogram.data(), passHist[0].data(), passHist[1].data(), passHist[2].data(), passHist[3].data());
    });
    for (uint32_t p = 0; p < k_passes; ++p) {
        uint32_t* in = k[p & 1];
        uint32_t* out = k[(p + 1) & 1];
        if (!payload) {
            gs_emu::launch(binningThreadBlocks, k_binningThreads,
                           [&] { OneSweep::DigitBinningPassKeysOnly(in, out, passHist[p].data(), index.data(), size, p * 8); });
        } else {
            uint32_t* vin = v[p & 1];
            uint32_t* vout = v[(p + 1) & 1];
            gs_emu::launch(binningThreadBlocks, k_binningThreads, [&] {
                OneSweep::DigitBinningPassPairs(in, vin, out, vout, passHist[p].data(), index.data(), size, p * 8);
            });
            if (out_payload_after_pass) memcpy(out_payload_after_pass + (size_t)p * size, vout, sizeof(uint32_t) * size);
        }
        if (out_after_pass) memcpy(out_after_pass + (size_t)p * size, out, sizeof(uint32_t) * size);
    }
    // four passes: the result is back in sort / payload (OneSweepDispatcher.cuh:325-335)
}
}  // namespace

extern "C" {
/* sorts keys[size] in place with the reference's kernels; optional outputs: the 4x256 global histogram after
 * GlobalHistogram, and the key buffer after each of the 4 passes (4*size words) */
void ref_onesweep_sort_keys(uint32_t* keys, uint32_t size, uint32_t* out_global_hist, uint32_t* out_after_pass) {
    run(keys, nullptr, size, out_global_hist, out_after_pass, nullptr);
}
void ref_onesweep_sort_pairs(uint32_t* keys, uint32_t* payload, uint32_t size, uint32_t* out_global_hist,
                             uint32_t* out_after_pass, uint32_t* out_payload_after_pass) {
    run(keys, payload, size, out_global_hist, out_after_pass, out_payload_after_pass);
}
/* OneSweepDispatcher.cuh:365-391 (DispatchValidateKeys / DispatchValidatePairs): Validate<<<ceil(n/4096), 256>>>;
 * returns the reference's error count (the dispatcher reports success iff it is 0) */
uint32_t ref_validate(uint32_t* keys, uint32_t* payload_or_null, uint32_t size) {
    uint32_t err = 0;
    const uint32_t blocks = div_round_up(size, 4096);
    if (payload_or_null) gs_emu::launch(blocks, 256, [&] { Validate(keys, payload_or_null, &err, size); });
    else gs_emu::launch(blocks, 256, [&] { Validate(keys, &err, size); });
    return err;
}
const char* ref_onesweep_source() {
    return "GPUSortingCUDA/Sort/OneSweep.cu:44-600 executed on the CPU by oracle/shim/simt_emu (warp primitives of "
           "Utils.cuh restated; launch sequence of OneSweepDispatcher.cuh:301-363 restated)";
}
}
