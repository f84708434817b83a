# Builds the three shared libraries in-tree:
#   raytracer_amd/lib/librt_hip.so   HIP backend (C ABI of include/rt_hip.h), gfx950 only
#   raytracer_amd/lib/librt_host.so  C++ host layer (RayScene/RayEngine/...), links the backend
#   oracle/liboracle.so              CPU checker (test infrastructure)
HIPCC ?= /opt/rocm/bin/hipcc
CXX ?= g++
ARCH ?= gfx950

LIBDIR := raytracer_amd/lib
CSRC := raytracer_amd/csrc
HOST := raytracer_amd/host

# -ffp-contract=off + correctly rounded divide/sqrt: required by rt_math.h's bit-exact contract.
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
            -fno-fast-math -pthread -Wall -Wno-unused-function -Wno-unused-result -Wno-unused-value -Iinclude -I$(CSRC)
HOSTFLAGS := -std=c++17 -O2 -fPIC -ffp-contract=off -Wall -Wextra -Wno-unused-parameter -Iinclude -I$(HOST)

HOST_SRCS := $(HOST)/AABB.cpp $(HOST)/BVHTree.cpp $(HOST)/MeshData.cpp $(HOST)/GltfLoader.cpp $(HOST)/ImageIO.cpp $(HOST)/JpegDecode.cpp $(HOST)/PngDecode.cpp $(HOST)/TraceableObject.cpp $(HOST)/RayScene.cpp \
             $(HOST)/RaySceneBuilder.cpp $(HOST)/RayTracer.cpp $(HOST)/RayEngine.cpp $(HOST)/ProceduralMesh.cpp \
             $(HOST)/host_capi.cpp $(HOST)/gpu/HipStorage.cpp
HOST_HDRS := $(wildcard $(HOST)/*.h $(HOST)/gpu/*.h $(HOST)/glm/*.hpp $(HOST)/glm/gtc/*.hpp) include/rt_hip.h include/rt_host.h

all: $(LIBDIR)/librt_hip.so $(LIBDIR)/librt_host.so oracle/liboracle.so examples/rt_render

# Identity of the library's code: a hash over its sources and flags (rt_build_id()); profiles carry it.
BUILD_ID := $(shell cat $(CSRC)/*.h $(CSRC)/*.hip $(CSRC)/*.cpp include/rt_hip.h | sha1sum | cut -c1-12)

# The device assembly is kept (build/*.s) and audited before the library is installed: hand-written
# instructions next to a transcendental result they read (a hazard the compiler does not see inside asm
# blocks), and spill bytes of every kernel against the audited figures of tools/spill_budget.json.
BUILD := build
DEVICE_ASM := $(BUILD)/rt_hip-hip-amdgcn-amd-amdhsa-$(ARCH).s
# Host-only parts of the backend (stream validation, re-pack, tree builds; RCCL binding) are plain C++ built by
# g++: the sanitizer targets below build the same files without a GPU toolchain.
COMMIT_FLAGS := -std=c++17 -O2 -fPIC -ffp-contract=off -fno-fast-math -pthread -Wall -Wextra -Iinclude -I$(CSRC)
$(BUILD)/rt_commit.o: $(CSRC)/rt_commit.cpp $(CSRC)/rt_commit.h $(CSRC)/rt_device.h $(CSRC)/rt_math.h include/rt_hip.h
	@mkdir -p $(BUILD)
	$(CXX) $(COMMIT_FLAGS) -c $< -o $@
$(BUILD)/rt_comm.o: $(CSRC)/rt_comm.cpp $(CSRC)/rt_comm.h include/rt_hip.h
	@mkdir -p $(BUILD)
	$(CXX) $(COMMIT_FLAGS) -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -c $< -o $@

# The opt-in fast-arithmetic build of the shading kernels (RT_OPT_FAST_SHADE): same sources, the hardware's
# reciprocal / square root / transcendental instructions, contraction allowed.
FASTFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=fast -fno-hip-fp32-correctly-rounded-divide-sqrt -freciprocal-math \
             -pthread -Wall -Wno-unused-function -Wno-unused-result -Wno-unused-value -I$(CURDIR)/include -I$(CURDIR)/$(CSRC)
FAST_ASM := $(BUILD)/rt_shade_fast-hip-amdgcn-amd-amdhsa-$(ARCH).s
$(BUILD)/rt_shade_fast.o: $(CSRC)/rt_shade_fast.hip $(wildcard $(CSRC)/*.h)
	@mkdir -p $(BUILD)
	cd $(BUILD) && $(HIPCC) $(FASTFLAGS) -save-temps -c $(CURDIR)/$(CSRC)/rt_shade_fast.hip -o rt_shade_fast.o
	python3 tools/isa_audit.py $(FAST_ASM)

# The device-side tree build (hipcub radix sort): a translation unit of its own, the sort's templates take a while.
$(BUILD)/rt_tree_device.o: $(CSRC)/rt_tree_device.hip $(CSRC)/rt_tree_device.h $(CSRC)/rt_commit.h $(CSRC)/rt_device.h include/rt_hip.h
	@mkdir -p $(BUILD)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIBDIR)/librt_hip.so: $(BUILD)/rt_tree_device.o $(BUILD)/rt_shade_fast.o $(CSRC)/rt_hip.hip $(wildcard $(CSRC)/*.h) include/rt_hip.h $(BUILD)/rt_commit.o $(BUILD)/rt_comm.o tools/isa_audit.py tools/spill_budget.json
	@mkdir -p $(LIBDIR) $(BUILD)
	cd $(BUILD) && $(HIPCC) $(subst -Iinclude,-I$(CURDIR)/include,$(subst -I$(CSRC),-I$(CURDIR)/$(CSRC),$(HIPFLAGS))) $(EXTRA_HIPFLAGS) -DRT_BUILD_ID='"$(BUILD_ID)"' -save-temps -c $(CURDIR)/$(CSRC)/rt_hip.hip -o rt_hip.o
	$(HIPCC) --offload-arch=$(ARCH) -fPIC -pthread -shared $(BUILD)/rt_hip.o $(BUILD)/rt_shade_fast.o $(BUILD)/rt_tree_device.o $(BUILD)/rt_commit.o $(BUILD)/rt_comm.o -o $(BUILD)/librt_hip.so
	python3 tools/isa_audit.py $(DEVICE_ASM) --budget tools/spill_budget.json
	cp $(BUILD)/librt_hip.so $@

# Sanitizer builds of everything that runs on the CPU (g++ only, no GPU toolchain): the host layer, the image and
# glTF readers and the commit-time code of the backend, driven by tools/fuzz_host.cpp (which stands in for the
# backend's entry points).  `make asan` / `make tsan` build build/fuzz_host_asan / build/fuzz_host_tsan;
# `make sanitize` runs the self-test and a bounded fuzz pass under both.
FUZZ_SRCS := tools/fuzz_host.cpp $(CSRC)/rt_commit.cpp $(filter-out $(HOST)/gpu/HipStorage.cpp,$(HOST_SRCS)) $(HOST)/gpu/HipStorage.cpp
SAN_FLAGS := -std=c++17 -O1 -g -fno-omit-frame-pointer -ffp-contract=off -pthread -Iinclude -I$(CSRC) -I$(HOST)
asan: $(BUILD)/fuzz_host_asan
tsan: $(BUILD)/fuzz_host_tsan
$(BUILD)/fuzz_host_asan: $(FUZZ_SRCS) $(HOST_HDRS) $(CSRC)/rt_commit.h
	@mkdir -p $(BUILD)
	$(CXX) $(SAN_FLAGS) -fsanitize=address,undefined -fno-sanitize-recover=undefined $(FUZZ_SRCS) -o $@ -lz
$(BUILD)/fuzz_host_tsan: $(FUZZ_SRCS) $(HOST_HDRS) $(CSRC)/rt_commit.h
	@mkdir -p $(BUILD)
	$(CXX) $(SAN_FLAGS) -fsanitize=thread $(FUZZ_SRCS) -o $@ -lz
FUZZ_N ?= 20000
sanitize: asan tsan
	python3 tools/make_fuzz_seeds.py $(BUILD)/fuzz_seeds
	$(BUILD)/fuzz_host_asan selftest
	$(BUILD)/fuzz_host_asan streams $(FUZZ_N) 1
	$(BUILD)/fuzz_host_asan png $(FUZZ_N) 2 $(BUILD)/fuzz_seeds/*.png
	$(BUILD)/fuzz_host_asan jpeg $(FUZZ_N) 3 $(BUILD)/fuzz_seeds/*.jpg
	$(BUILD)/fuzz_host_asan glb 2000 4 $(BUILD)/fuzz_seeds/*.glb tests/golden/room.glb
	$(BUILD)/fuzz_host_tsan selftest

# Times validation, re-pack and the host tree builds on a scene written to files, without a device (tools/commit_host_time.cpp).
$(BUILD)/commit_host_time: tools/commit_host_time.cpp $(CSRC)/rt_commit.cpp $(CSRC)/rt_commit.h $(CSRC)/rt_device.h
	@mkdir -p $(BUILD)
	$(CXX) $(COMMIT_FLAGS) tools/commit_host_time.cpp $(CSRC)/rt_commit.cpp -o $@

# Microbenchmarks behind the roofline limits (tools/roofline_limits.sh runs the built binaries on the GPU box).
MICRO := tools/microbench/gather64 tools/microbench/chain tools/microbench/fetch_calib tools/microbench/mix_calib
microbench: $(MICRO)
tools/microbench/%: tools/microbench/%.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 $< -o $@

# Re-records the spill figures after a change that was looked at (tools/isa_audit.py --list shows them).
spill-budget:
	python3 tools/isa_audit.py $(DEVICE_ASM) --budget tools/spill_budget.json --write-budget

$(LIBDIR)/librt_host.so: $(HOST_SRCS) $(HOST_HDRS) $(LIBDIR)/librt_hip.so
	@mkdir -p $(LIBDIR)
	$(CXX) $(HOSTFLAGS) -shared $(HOST_SRCS) -o $@ -L$(LIBDIR) -lrt_hip -lz -Wl,-rpath,'$$ORIGIN'

# Headless counterpart of the reference's main program, on the C++ host API.
examples/rt_render: examples/render_main.cpp $(LIBDIR)/librt_host.so
	$(CXX) $(HOSTFLAGS) examples/render_main.cpp -o $@ -L$(LIBDIR) -lrt_host -lrt_hip -Wl,-rpath,'$$ORIGIN/../$(LIBDIR)'

oracle/liboracle.so: oracle/oracle.cpp oracle/oracle_spectral.inc $(CSRC)/rt_math.h
	$(MAKE) -C oracle liboracle.so

clean:
	rm -rf $(BUILD)
	rm -f $(LIBDIR)/*.so oracle/liboracle.so examples/rt_render

.PHONY: all clean spill-budget asan tsan sanitize microbench
