# Top-level build: the product (libmi_denoise.so, gfx950 only) and the checker (oracle/).
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
CSRC    := image_denoising_filter_amd/csrc
LIB     := image_denoising_filter_amd/libmi_denoise.so
SRCS    := $(CSRC)/capi.cpp $(CSRC)/hostcopy.cpp $(CSRC)/markers.cpp $(CSRC)/recording.cpp $(CSRC)/pointwise.hip $(CSRC)/bilateral.hip $(CSRC)/nlm.hip $(CSRC)/nlm_small.hip $(CSRC)/nlm_rt.hip $(CSRC)/nlm_rt4.hip $(CSRC)/pipeline.cpp $(CSRC)/sharded.cpp \
           $(CSRC)/codec/png.cpp $(CSRC)/codec/exr.cpp $(CSRC)/codec/piz.cpp $(CSRC)/codec/image_capi.cpp
OBJS    := $(patsubst $(CSRC)/%,build/%.o,$(SRCS))
HIPFLAGS := -x hip --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function -Iinclude

CLI     := image_denoising_filter_amd/mi_denoise

all: $(LIB) $(CLI) oracle standin

# the drop-in command-line driver (host C++ only; links the C-ABI library next to it)
$(CLI): $(CSRC)/cli/mi_denoise.cpp include/mi_denoise.h $(LIB)
	g++ -std=c++17 -O2 -fopenmp -Wall -o $@ $(CSRC)/cli/mi_denoise.cpp -Limage_denoising_filter_amd -lmi_denoise -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,/opt/rocm/lib

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -lz -ldl

# per-file extras: the NLM kernels gain 2-6 % from LLVM's max-ILP scheduling strategy (A/B on MI355X: 3074 vs 3011
# Mpixel/s batched, 2882 vs 2710 single frame); the bilateral kernels lose 8 % with it, so it stays off there.
EXTRA_nlm.hip := -mllvm -amdgpu-sched-strategy=max-ilp
EXTRA_nlm_rt.hip := $(EXTRA_nlm.hip)
# small launches of the tuned windows (a lone frame, two, three) are fastest under the iterative-ILP strategy instead: nlm_small.hip
EXTRA_nlm_small.hip := -mllvm -amdgpu-sched-strategy=iterative-ilp
EXTRA_nlm_rt4.hip := $(EXTRA_nlm.hip)

build/%.o: $(CSRC)/% $(CSRC)/common.hpp $(CSRC)/nlm_strip.hpp $(CSRC)/codec/image_io.hpp include/mi_denoise.h
	@mkdir -p $(dir $@)
	$(HIPCC) $(HIPFLAGS) $(EXTRA_$(notdir $<)) -c $< -o $@

oracle:
	$(MAKE) -C oracle

# test-only stand-in for librccl: lets the multi-rank halo path run with N ranks on ONE device (tests/test_gpu_sharded_multirank.py)
standin:
	$(MAKE) -C tests/standin_rccl

clean:
	rm -rf build $(LIB) $(CLI); $(MAKE) -C oracle clean; $(MAKE) -C tests/standin_rccl clean
.PHONY: all oracle standin clean
