// rocprim_compare — comparator only (the AMD analogue of the reference's CubDispatcher,
// GPUSortingCUDA/Sort/CubDispatcher.cuh:105-404): times rocprim::radix_sort_keys / _pairs on the
// same generator, size and protocol as BatchTimingKeysOnly/Pairs.  NOT part of the product.
// Usage: rocprim_compare [log2_size=28] [batch=20]
#include <hip/hip_runtime.h>
#include <cstring>
#include <string.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "gpusort.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
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
