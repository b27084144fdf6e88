// rocprim_compare — comparator only (the AMD analogue of the reference's CubDispatcher,
// GPUSortingCUDA/Sort/CubDispatcher.cuh:105-404): times rocprim::radix_sort_keys / _pairs on the
// same generator, size and protocol as BatchTimingKeysOnly/Pairs.  NOT part of the product.
// Usage: rocprim_compare [log2_size=28] [batch=20]
//        rocprim_compare sweep [batch=5]        — ONE JSON line: rocPRIM and this library side by side, same box, same inputs, HIP
//        events around each sort: sizes 2^16 .. 2^28 at entropy preset 1 and the five presets at 2^28, keys-only and (u32, u32)
//        pairs (the reference compares itself with the vendor sort over sizes and presets with the same BatchTiming arguments:
//        CubDispatcher.cuh:105-404, GPUSortingCUDA.cu:24-33, README.md:9-37).  bench.py carries the line as more.comparator.
//        rocprim_compare check [log2_size=24]   — sorts the same input with rocPRIM and with this library
//        (keys; pairs with value = original index, both sorts are stable) and compares the outputs element for
//        element on the device.
#include <hip/hip_runtime.h>
#include <cstring>
#include <string.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "gpusort.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void iota_kernel(uint32_t* v, uint32_t n) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) v[i] = i;
}
__global__ void diff_kernel(const uint32_t* a, const uint32_t* b, uint32_t n, unsigned long long* count) {
    unsigned long long bad = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) bad += a[i] != b[i];
    if (bad) atomicAdd(count, bad);
}

// Same input through both sorters; returns 0 when every element agrees.
static int check_identical(uint32_t lg) {
    const uint32_t n = (1u << lg) + 12345u;  // not a multiple of any tile
    uint32_t *keys, *kalt, *kin, *kout, *vals, *valt, *vin, *vout;
    unsigned long long* d_bad;
    for (uint32_t** p : {&keys, &kalt, &kin, &kout, &vals, &valt, &vin, &vout}) CK(hipMalloc(p, (size_t)n * 4));
    CK(hipMalloc(&d_bad, 8));
    int rc = 0;
    for (int pairs = 0; pairs < 2; ++pairs) {
        if (gs_init_random(kin, nullptr, 0u, pairs ? 2u : 0u, 77u + pairs, n, nullptr) != GS_OK) return 2;  // pairs: preset 3 (duplicates)
        hipLaunchKernelGGL(iota_kernel, dim3(1024), dim3(256), 0, 0, vin, n);
        CK(hipMemcpy(keys, kin, (size_t)n * 4, hipMemcpyDeviceToDevice));
        CK(hipMemcpy(vals, vin, (size_t)n * 4, hipMemcpyDeviceToDevice));
        size_t tmp_bytes = 0;
        void* tmp = nullptr;
        if (pairs) CK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, kin, kout, vin, vout, n));
        else CK(rocprim::radix_sort_keys(nullptr, tmp_bytes, kin, kout, n));
        CK(hipMalloc(&tmp, tmp_bytes));
        if (pairs) CK(rocprim::radix_sort_pairs(tmp, tmp_bytes, kin, kout, vin, vout, n));
        else CK(rocprim::radix_sort_keys(tmp, tmp_bytes, kin, kout, n));
        gs_onesweep* h = nullptr;
        if (gs_onesweep_create(&h, n, pairs ? GS_MODE_PAIRS : GS_MODE_KEYS_ONLY, pairs ? 4u : 0u) != GS_OK) return 3;
        const gs_status st = pairs ? gs_onesweep_sort_pairs(h, keys, vals, kalt, valt, n, GS_KEY_UINT32, GS_ORDER_ASCENDING, nullptr)
                                   : gs_onesweep_sort_keys(h, keys, kalt, n, GS_KEY_UINT32, GS_ORDER_ASCENDING, nullptr);
        if (st != GS_OK || gs_onesweep_check(h, nullptr) != GS_OK) return 4;
        unsigned long long bad_k = 0, bad_v = 0;
        CK(hipMemset(d_bad, 0, 8));
        hipLaunchKernelGGL(diff_kernel, dim3(1024), dim3(256), 0, 0, keys, kout, n, d_bad);
        CK(hipMemcpy(&bad_k, d_bad, 8, hipMemcpyDeviceToHost));
        if (pairs) {
            CK(hipMemset(d_bad, 0, 8));
            hipLaunchKernelGGL(diff_kernel, dim3(1024), dim3(256), 0, 0, vals, vout, n, d_bad);
            CK(hipMemcpy(&bad_v, d_bad, 8, hipMemcpyDeviceToHost));
        }
        printf("check %s n=%u: gpusort vs rocprim  key mismatches=%llu  value mismatches=%llu  identical=%s\n", pairs ? "pairs" : "keys ", n,
               bad_k, bad_v, (bad_k | bad_v) == 0 ? "yes" : "NO");
        rc |= (bad_k | bad_v) != 0;
        gs_onesweep_destroy(h);
        CK(hipFree(tmp));
    }
    return rc;
}

// rocPRIM and this library over sizes and entropy presets, one JSON line
static int sweep(uint32_t batch) {
    const uint32_t nmax = 1u << 28;
    uint32_t *keys, *kout, *vals, *vout, *kwork, *kalt, *vwork, *valt;
    for (uint32_t** p : {&keys, &kout, &vals, &vout, &kwork, &kalt, &vwork, &valt}) CK(hipMalloc(p, (size_t)nmax * 4));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    struct Point { uint32_t lg, preset; };
    const Point pts[] = {{16, 0}, {18, 0}, {20, 0}, {22, 0}, {24, 0}, {26, 0}, {28, 0}, {28, 1}, {28, 2}, {28, 3}, {28, 4}};
    printf("{\"tool\": \"rocprim_compare sweep\", \"comparator\": \"rocprim::radix_sort_keys / radix_sort_pairs (ROCm's device radix sort)\", "
           "\"protocol\": \"InitRandom seed 10 + i on the device, HIP events around each sort, mean of %u sorts after one warm-up; both sorters on the same inputs\", \"rows\": [", batch);
    bool first = true;
    for (int pairs = 0; pairs < 2; ++pairs) {
        gs_onesweep* h = nullptr;
        if (gs_onesweep_create(&h, nmax, pairs ? GS_MODE_PAIRS : GS_MODE_KEYS_ONLY, pairs ? 4u : 0u) != GS_OK) return 3;
        for (const Point& pt : pts) {
            const uint32_t n = 1u << pt.lg;
            size_t tmp_bytes = 0;
            void* tmp = nullptr;
            if (pairs) CK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys, kout, vals, vout, n));
            else CK(rocprim::radix_sort_keys(nullptr, tmp_bytes, keys, kout, n));
            CK(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
            float t_roc = 0.f, t_gs = 0.f;
            uint32_t err_roc = 1, err_gs = 1;
            for (uint32_t i = 0; i <= batch; ++i) {
                if (gs_init_random(keys, pairs ? vals : nullptr, pairs ? 4u : 0u, pt.preset, i + 10, n, nullptr) != GS_OK) return 2;
                CK(hipMemcpyAsync(kwork, keys, (size_t)n * 4, hipMemcpyDeviceToDevice, nullptr));
                if (pairs) CK(hipMemcpyAsync(vwork, vals, (size_t)n * 4, hipMemcpyDeviceToDevice, nullptr));
                CK(hipDeviceSynchronize());
                float ms = 0.f;
                CK(hipEventRecord(a));
                if (pairs) CK(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, kout, vals, vout, n));
                else CK(rocprim::radix_sort_keys(tmp, tmp_bytes, keys, kout, n));
                CK(hipEventRecord(b));
                CK(hipEventSynchronize(b));
                CK(hipEventElapsedTime(&ms, a, b));
                if (i) t_roc += ms;
                CK(hipEventRecord(a));
                const gs_status st = pairs ? gs_onesweep_sort_pairs(h, kwork, vwork, kalt, valt, n, GS_KEY_UINT32, GS_ORDER_ASCENDING, nullptr)
                                           : gs_onesweep_sort_keys(h, kwork, kalt, n, GS_KEY_UINT32, GS_ORDER_ASCENDING, nullptr);
                if (st != GS_OK) return 4;
                CK(hipEventRecord(b));
                CK(hipEventSynchronize(b));
                CK(hipEventElapsedTime(&ms, a, b));
                if (i) t_gs += ms;
            }
            gs_validate(kout, pairs ? vout : nullptr, pairs ? 4u : 0u, n, GS_KEY_UINT32, GS_ORDER_ASCENDING, &err_roc, nullptr);
            gs_validate(kwork, pairs ? vwork : nullptr, pairs ? 4u : 0u, n, GS_KEY_UINT32, GS_ORDER_ASCENDING, &err_gs, nullptr);
            printf("%s{\"mode\": \"%s\", \"log2_keys\": %u, \"entropy_preset\": %u, \"rocprim_ms\": %.4f, \"gpusort_ms\": %.4f, "
                   "\"rocprim_GKeys_per_s\": %.2f, \"gpusort_GKeys_per_s\": %.2f, \"speedup\": %.3f, \"both_sorted\": %s}",
                   first ? "" : ", ", pairs ? "pairs_u32" : "keys", pt.lg, pt.preset + 1, t_roc / batch, t_gs / batch, n / (t_roc / batch) / 1e6,
                   n / (t_gs / batch) / 1e6, t_roc / t_gs, (err_roc == 0 && err_gs == 0) ? "true" : "false");
            first = false;
            CK(hipFree(tmp));
        }
        gs_onesweep_destroy(h);
    }
    printf("]}\n");
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "check")) return check_identical(argc > 2 ? (uint32_t)atoi(argv[2]) : 24u);
    if (argc > 1 && !strcmp(argv[1], "sweep")) return sweep(argc > 2 ? (uint32_t)atoi(argv[2]) : 5u);
    const uint32_t lg = argc > 1 ? (uint32_t)atoi(argv[1]) : 28u;
    const uint32_t batch = argc > 2 ? (uint32_t)atoi(argv[2]) : 20u;
    const uint32_t n = 1u << lg;
    uint32_t *keys, *kout, *vals, *vout;
    CK(hipMalloc(&keys, (size_t)n * 4)); CK(hipMalloc(&kout, (size_t)n * 4));
    CK(hipMalloc(&vals, (size_t)n * 4)); CK(hipMalloc(&vout, (size_t)n * 4));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int pairs = 0; pairs < 2; ++pairs) {
        size_t tmp_bytes = 0;
        void* tmp = nullptr;
        if (pairs) CK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys, kout, vals, vout, n));
        else CK(rocprim::radix_sort_keys(nullptr, tmp_bytes, keys, kout, n));
        CK(hipMalloc(&tmp, tmp_bytes));
        float total = 0.f;
        for (uint32_t i = 0; i <= batch; ++i) {
            if (gs_init_random(keys, pairs ? vals : nullptr, pairs ? 4u : 0u, 0, i + 10, n, nullptr) != GS_OK) return 2;
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(a));
            if (pairs) CK(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, kout, vals, vout, n));
            else CK(rocprim::radix_sort_keys(tmp, tmp_bytes, keys, kout, n));
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, a, b));
            if (i) total += ms;
        }
        uint32_t err = 1;
        gs_validate(kout, pairs ? vout : nullptr, pairs ? 4u : 0u, n, GS_KEY_UINT32, GS_ORDER_ASCENDING, &err, nullptr);
        printf("rocprim::radix_sort_%s  n=2^%u  batch=%u  temp=%zu MiB  %.3f ms/sort  %E keys/sec  sorted=%s\n",
               pairs ? "pairs" : "keys ", lg, batch, tmp_bytes >> 20, total / batch, n / (total / 1000.f) * batch,
               err == 0 ? "yes" : "NO");
        CK(hipFree(tmp));
    }
    return 0;
}
