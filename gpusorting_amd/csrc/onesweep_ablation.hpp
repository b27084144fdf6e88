// onesweep_ablation.hpp — ablation, tracing and fault-injection hooks of the OneSweep kernels.  NOT part of the
// product build: onesweep_kernels.hpp includes this file only when GS_EXP != 0 (tools/README.md "experiment
// builds", tests/test_gpu_fault.py).  GS_EXP flags:
//   1   no look-back wait, sequential output positions (memory floor of the tile machinery)
//   2   per-tile phase timestamps (10 ns ticks, lane 0 of wave 0) into the buffer whose address the host stored in
//       the slab at STATUS+8; 8 words per (pass, block) — tools/trace_tiles.py
//   4   histogram kernel streams the keys and counts nothing
//   8   fault injection (cf. the reference's EmulatedDeadlocking.cu:36-37,339-345): tile 5 of chain 3 never publishes
//       its descriptor, as if its workgroup had stalled.  With the fallback (default) its successors recount it and
//       the sort is exact; with -DGS_FALLBACK=0 every later tile of that chain runs into the bounded spin, the sort
//       still finishes, and gs_onesweep_check says GS_ERR_TIMEOUT.
//   256 no look-back wait with the real scatter shape: every earlier tile of the chain is assumed to hold this
//       tile's digit counts (positions approximate, wrapped into range)
#pragma once

#define GS_FAULT_TILE(chain, tile) (((GS_EXP)&8) && (chain) == 3u && (tile) == 5u)
// ... and in the two-launch mid-size sort every fourth workgroup of K1 behaves as if it had never been dispatched (the
// others adopt its tile)
#define GS_FAULT_MID_ABSENT(block) (((GS_EXP)&8) && ((block)&3u) == 1u)
// ... and, in the build WITHOUT the fallback (-DGS_FALLBACK=0: the one that must report GS_ERR_TIMEOUT), workgroup 2 of K1 claims its
// tile and never publishes its counts: nobody can adopt a claimed tile, so every waiter must run into its bounded spin
#define GS_FAULT_MID_SILENT(block) (((GS_EXP)&8) && !GS_FALLBACK && (block) == 2u)

#if (GS_EXP & 2)
#define GS_TRACE_SETUP()                                                                                                  \
    uint32_t* trace = reinterpret_cast<uint32_t*>(((unsigned long long)status[9] << 32) | status[8]) +                    \
                      ((size_t)(shift >> 3) * gridDim.x + blockIdx.x) * 8;                                                \
    uint32_t trace_trips = 0
#define GS_TRACE(slot) do { if (tid == 0) trace[(slot)] = (uint32_t)wall_clock64(); } while (0)
#define GS_TRACE_TRIP() ++trace_trips
#define GS_TRACE_END(chain) do { if (tid == 0) trace[7] = trace_trips | ((chain) << 16) | (1u << 31); } while (0)
#else
#define GS_TRACE_SETUP() do { } while (0)
#define GS_TRACE(slot) do { } while (0)
#define GS_TRACE_TRIP() do { } while (0)
#define GS_TRACE_END(chain) do { } while (0)
#endif

#if (GS_EXP & 4)
#define GS_ABL_HIST_STREAM_ONLY(t) do { asm volatile("" ::"v"((t).x), "v"((t).y), "v"((t).z), "v"((t).w)); return; } while (0)
#else
#define GS_ABL_HIST_STREAM_ONLY(t) do { } while (0)
#endif

#define GS_ABL_LOOKBACK_SKIPPED (((GS_EXP)&1) != 0)
#define GS_ABL_GENERIC_SCATTER (((GS_EXP)&(257 | 1024)) != 0)
// 2048: what counting the NEXT digit per output position segment would cost inside the pass: one LDS add per
//       key in the scatter loop, table [16 segments][256] behind the kernel's own LDS (never flushed: cost only); runtime
//       mode bit 2048 switches it on, so the build's occupancy is the same with and without
#if (GS_EXP & 2048)
#define GS_ABL_COUNT_LDS 16384
#define GS_ABL_COUNT_NEXT(kb, o) do { if (mode & 2048u) atomicAdd(reinterpret_cast<uint32_t*>(s_raw + Cfg::LDS_BYTES - 16384) + \
        ((((o) >> (32u - __builtin_clz((n - 1u) >> 4))) & 15u) << 8) + (((kb) >> ((shift + 8u) & 31u)) & 255u), 1u); } while (0)
#else
#define GS_ABL_COUNT_LDS 0
#define GS_ABL_COUNT_NEXT(kb, o) do { } while (0)
#endif
// 1024: the look-back / scatter decomposition of round 3 (tools/r03_ablate.py); the variants are RUNTIME bits of the
//       kernel's mode word (gs_onesweep_options::debug_flags), so one build serves every combination:
//         mode 256   replay: the descriptors of an identical earlier sort are still in the slab (the histogram kernel
//                    does not clear them) and no tile publishes REDUCTION, so every look-back finds its predecessor's
//                    INCLUSIVE row in its first read — a look-back of exactly one round trip, exact positions
//         mode 512   the predecessor's row is requested before the key loads and consumed in the look-back (with 256:
//                    a look-back that never waits)
//         mode 1024  sequential output positions (the real look-back still runs)
#define GS_ABL_REPLAY (((GS_EXP)&1024) && (mode & 256u))
#define GS_ABL_EARLY_ROW(v) do { if (((GS_EXP)&1024) && (mode & 512u) && tid < RADIX) (v) = ld_agent(&cdesc[(size_t)tile * RADIX + tid]); } while (0)
#define GS_ABL_EARLY_USE(v) do { if (((GS_EXP)&1024) && (mode & 512u) && !finished && ((v) & FLAG_MASK) == FLAG_INCLUSIVE) { prev = (v) >> 2; done = true; } } while (0)
#define GS_ABL_CLOCKS_BEGIN() const unsigned long long abl_c0_ = (unsigned long long)clock64(), abl_w0_ = (unsigned long long)wall_clock64()
#define GS_ABL_CLOCKS_END() do { if (((GS_EXP)&1024) && blockIdx.x == 0 && threadIdx.x == 0) { \
        unsigned long long* c_ = reinterpret_cast<unsigned long long*>(slab + SLAB_STATUS + 16); \
        c_[0] = abl_c0_; c_[1] = abl_w0_; c_[2] = (unsigned long long)clock64(); c_[3] = (unsigned long long)wall_clock64(); } } while (0)
// 4096: phase stamps of the histogram kernel (10 ns ticks of the 100 MHz counter, low words): the first and the last workgroup
//       write eight each into the slab at STATUS+16 / STATUS+24 — tools/r03_hist_phases.py (build with 1024 too: the reader)
#if (GS_EXP & 4096)
#define GS_HIST_STAMP(i) do { if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) abl_stamp_[(i)] = (uint32_t)wall_clock64(); } while (0)
#define GS_HIST_STAMPS_DECL() uint32_t abl_stamp_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define GS_HIST_STAMPS_OUT() do { if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) { \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); abl_stamp_[7] = (uint32_t)wall_clock64(); \
        for (int i_ = 0; i_ < 8; ++i_) slab[SLAB_STATUS + (blockIdx.x == 0 ? 16 : 24) + i_] = abl_stamp_[i_]; } } while (0)
#else
#define GS_HIST_STAMP(i) do { } while (0)
#define GS_HIST_STAMPS_DECL() do { } while (0)
#define GS_HIST_STAMPS_OUT() do { } while (0)
#endif
#if (GS_EXP & 256)
#define GS_ABL_ASSUME_PREV() do { if (!finished) { prev = (ld_agent(&cdesc[tid]) >> 2) + tile * tile_total; done = true; } } while (0)
#else
#define GS_ABL_ASSUME_PREV() do { } while (0)
#endif
// positions are meaningless without the look-back: sequential (1) or wrapped into the array (256)
#define GS_ABL_OUT_INDEX(o, i)                                     \
    do {                                                           \
        if ((GS_EXP)&1) (o) = (tile_base + (i)) % n;               \
        if ((GS_EXP)&256) (o) = (o) % n;                           \
        if (((GS_EXP)&1024) && (mode & 1024u)) (o) = tile_base + (i); \
    } while (0)
