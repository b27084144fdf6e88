# gpusort-mi355x — build the C-ABI library (gfx950 only) and the CPU oracle.
HIPCC ?= hipcc
HIPFLAGS ?= --offload-arch=gfx950 -O3 -std=c++17 -fPIC
LIB := gpusorting_amd/lib/libgpusort.so
SRC := gpusorting_amd/csrc/gpusort_capi.hip
HDR := gpusorting_amd/csrc/onesweep_kernels.hpp gpusorting_amd/csrc/onesweep_ablation.hpp gpusorting_amd/csrc/mid_kernels.hpp gpusorting_amd/csrc/hybrid_kernels.hpp gpusorting_amd/csrc/msd_kernels.hpp gpusorting_amd/csrc/gpusort_mgpu.hpp include/gpusort.h

all: $(LIB) gpusorting_amd/lib/libgpusort_fault.so gpusorting_amd/lib/libgpusort_fault_nofallback.so gpusorting_amd/lib/libgpusort_tuning.so oracle tools
$(LIB): $(SRC) $(HDR)
	@mkdir -p gpusorting_amd/lib
	$(HIPCC) $(HIPFLAGS) -shared $(SRC) -o $@
gpusorting_amd/lib/libgpusort_fault.so: $(SRC) $(HDR)
	$(HIPCC) $(HIPFLAGS) -shared -DGS_EXP=8 -DGS_FALLBACK_SPINS=4096 -DGS_MID_ADOPT_SPINS=4 $(SRC) -o $@
gpusorting_amd/lib/libgpusort_fault_nofallback.so: $(SRC) $(HDR)
	$(HIPCC) $(HIPFLAGS) -shared -DGS_EXP=8 -DGS_FALLBACK=0 -DGS_SPIN_LIMIT=4096 $(SRC) -o $@
# calibration kernels + tuning tile shapes (u32 keys-only kernels only: a 10 s compile); tools/ and bench.py's box_floor block
gpusorting_amd/lib/libgpusort_tuning.so: $(SRC) $(HDR)
	$(HIPCC) $(HIPFLAGS) -shared -DGS_MINIMAL -DGS_TUNING $(SRC) -o $@
oracle:
	$(MAKE) -C oracle
tools: build/gpusorting_main build/gpusorting_d3d12_main build/rocprim_compare build/mgpu_main
build/mgpu_main: tools/mgpu_main.cpp include/gpusort.h $(LIB)
	@mkdir -p build
	$(HIPCC) -O2 -std=c++17 -Iinclude tools/mgpu_main.cpp -Lgpusorting_amd/lib -lgpusort -lpthread -Wl,-rpath,'$$ORIGIN/../gpusorting_amd/lib' -o $@
build/gpusorting_d3d12_main: tools/gpusorting_d3d12_main.cpp include/gpusort/GPUSortBase.hpp $(LIB)
	@mkdir -p build
	$(HIPCC) -O2 -std=c++17 -Iinclude tools/gpusorting_d3d12_main.cpp -Lgpusorting_amd/lib -lgpusort -Wl,-rpath,'$$ORIGIN/../gpusorting_amd/lib' -o $@
build/gpusorting_main: tools/gpusorting_main.cpp include/gpusort/OneSweepDispatcher.hpp $(LIB)
	@mkdir -p build
	$(HIPCC) -O2 -std=c++17 -Iinclude tools/gpusorting_main.cpp -Lgpusorting_amd/lib -lgpusort -Wl,-rpath,'$$ORIGIN/../gpusorting_amd/lib' -o $@
build/rocprim_compare: tools/rocprim_compare.cpp $(LIB)
	@mkdir -p build
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/rocprim_compare.cpp -Lgpusorting_amd/lib -lgpusort -Wl,-rpath,'$$ORIGIN/../gpusorting_amd/lib' -o $@
clean:
	rm -rf build $(LIB); $(MAKE) -C oracle clean
.PHONY: all oracle tools clean
