# Builds the product library (HIP, gfx950 only) and, through oracle/Makefile, the test oracle.
#
#   make lib      -> cuda-pathtracer_amd/libptmi.so
#   make oracle   -> oracle/libpt_oracle.so (+ oracle/_ref/*.so where /root/reference exists)
#   make examples -> examples/pt_headless
#   make asan     -> tools/asan/asan_oracle, tools/asan/asan_host (sanitizer builds of the host code)
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
PKG      := cuda-pathtracer_amd
CSRC     := $(PKG)/csrc
# -ffp-contract=off: the trace loop is bit-reproducible against the host oracle only without
# fused multiply-add; division and sqrt stay IEEE (hipcc's default
# -fhip-fp32-correctly-rounded-divide-sqrt is left on), denormals are kept.
# -fno-slp-vectorize: packed fp32 (v_pk_*_f32) issues at half rate on gfx950, so SLP-packing
# the scalar vector math only adds register shuffles (measured +14..30 % without it).
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize \
            -fhip-fp32-correctly-rounded-divide-sqrt -Iinclude -I$(CSRC) -Wall -Wno-unused-function

SRCS := $(CSRC)/pt_kernels.hip $(CSRC)/pt_wavefront.hip $(CSRC)/pt_nee.hip $(CSRC)/pt_api.cpp $(CSRC)/pt_scene.cpp \
        $(CSRC)/pt_bvh.cpp $(CSRC)/pt_checkpoint.cpp
HDRS := include/ptmi.h $(CSRC)/pt_checkpoint.hpp $(CSRC)/pt_kernels.hpp $(CSRC)/pt_device.hpp $(CSRC)/pt_trace.hpp $(CSRC)/pt_bvh.hpp $(CSRC)/pt_nee.hpp

all: lib oracle

lib: $(PKG)/libptmi.so

$(PKG)/libptmi.so: $(SRCS) $(HDRS) Makefile
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(SRCS)

# Same library with the BVH traversal counters (pt_accel_stats) and the megakernel's divergence
# counters (pt_lane_stats) compiled in; measurement only.
stats: tools/stats/libptmi.so
tools/stats/libptmi.so: $(SRCS) $(HDRS) Makefile
	mkdir -p tools/stats
	$(HIPCC) $(HIPFLAGS) -DPT_BVH_STATS -DPT_LANE_STATS -shared -o $@ $(SRCS)

oracle:
	$(MAKE) -C oracle all
	if [ -d /root/reference ]; then $(MAKE) -C oracle ref; fi

examples: examples/pt_headless tools/pt_bench

tools/pt_bench: tools/pt_bench.cpp $(PKG)/libptmi.so
	$(HIPCC) -O2 -std=c++17 -Iinclude -o $@ tools/pt_bench.cpp -L$(PKG) -lptmi -Wl,-rpath,'$$ORIGIN/../$(PKG)'

examples/pt_headless: examples/main.cpp $(PKG)/host/pathtracer.hpp $(PKG)/host/multi_gpu.hpp $(PKG)/host/shard_plan.hpp $(PKG)/host/scene_file.hpp $(PKG)/libptmi.so
	$(HIPCC) -O2 -std=c++17 -ffp-contract=off -Iinclude -I$(PKG)/host -o $@ examples/main.cpp \
	    -L$(PKG) -lptmi -L/opt/rocm/lib -lrccl -lpthread -Wl,-rpath,'$$ORIGIN/../$(PKG)'

# Host-side code under AddressSanitizer + UBSan (no device code, runs without a GPU): the CPU
# oracle, and the library's scene parser / built-in scenes / camera functions / BVH builder /
# checkpoint files.  tests/test_sanitizers.py runs both drivers.
SANFLAGS := -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined -ffp-contract=off
asan: tools/asan/asan_oracle tools/asan/asan_host tools/asan/plain_oracle
# the oracle driver without sanitizers, as the product's tests build the oracle: its output is what the sanitized one must print
tools/asan/plain_oracle: tools/asan/asan_oracle.c oracle/pt_oracle.c oracle/pt_oracle.h oracle/pt_oracle_math.h oracle/pt_oracle_nee.h
	gcc -std=c11 -O2 -ffp-contract=off -fno-fast-math -Ioracle -o $@ tools/asan/asan_oracle.c oracle/pt_oracle.c -lm -lpthread
tools/asan/asan_oracle: tools/asan/asan_oracle.c oracle/pt_oracle.c oracle/pt_oracle.h oracle/pt_oracle_math.h oracle/pt_oracle_nee.h
	gcc -std=c11 $(SANFLAGS) -Wall -Wextra -Ioracle -o $@ tools/asan/asan_oracle.c oracle/pt_oracle.c -lm -lpthread
tools/asan/asan_host: tools/asan/asan_host.cpp $(CSRC)/pt_bvh.cpp $(CSRC)/pt_scene.cpp $(CSRC)/pt_checkpoint.cpp \
                      $(CSRC)/pt_bvh.hpp $(CSRC)/pt_checkpoint.hpp $(PKG)/host/scene_file.hpp $(PKG)/host/pathtracer.hpp $(PKG)/host/shard_plan.hpp include/ptmi.h
	g++ -std=c++17 $(SANFLAGS) -Wall -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude -I$(CSRC) -I$(PKG)/host -o $@ \
	    tools/asan/asan_host.cpp $(CSRC)/pt_bvh.cpp $(CSRC)/pt_scene.cpp $(CSRC)/pt_checkpoint.cpp -lpthread

clean:
	rm -f $(PKG)/libptmi.so examples/pt_headless tools/asan/asan_oracle tools/asan/asan_host tools/asan/plain_oracle
	$(MAKE) -C oracle clean

.PHONY: all lib stats oracle examples asan clean
