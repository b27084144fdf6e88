// gpusorting_main — the OneSweep part of the reference's main()
// (GPUSortingCUDA/GPUSortingCUDA.cu:16-58) against the MI355X-native library.
// Usage: gpusorting_main [log2_size=28] [batch=100]
#include <stdlib.h>

#include "gpusort/OneSweepDispatcher.hpp"

int main(int argc, char** argv) {
    const uint32_t lg = argc > 1 ? (uint32_t)atoi(argv[1]) : 28u;
    const uint32_t batch = argc > 2 ? (uint32_t)atoi(argv[2]) : 100u;
    printf("-----------------BEGINNING KEYS TESTS-----------------\n\n");
    OneSweepDispatcher* oneSweep = new OneSweepDispatcher(true, 1u << lg);
    oneSweep->TestAllKeysOnly();
    oneSweep->BatchTimingKeysOnly(1u << lg, batch, 10, ENTROPY_PRESET_1);
    bool healthy = oneSweep->Healthy();
    delete oneSweep;

    printf("----------------BEGINNING PAIRS TESTS----------------\n\n");
    oneSweep = new OneSweepDispatcher(false, 1u << lg);
    oneSweep->TestAllPairs();
    oneSweep->BatchTimingPairs(1u << lg, batch, 10, ENTROPY_PRESET_1);
    healthy = healthy && oneSweep->Healthy();
    delete oneSweep;
    return healthy ? 0 : 1;
}
