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

# The device assembly is kept (build/*.s) and audited before the library is installed: hand-written
# instructions next to a transcendental result they read (a hazard the compiler does not see inside asm
# blocks), and spill bytes of every kernel against the audited figures of tools/spill_budget.json.
BUILD := build
DEVICE_ASM := $(BUILD)/rt_hip-hip-amdgcn-amd-amdhsa-$(ARCH).s
$(LIBDIR)/librt_hip.so: $(CSRC)/rt_hip.hip $(wildcard $(CSRC)/*.h) include/rt_hip.h tools/isa_audit.py tools/spill_budget.json
	@mkdir -p $(LIBDIR) $(BUILD)
	cd $(BUILD) && $(HIPCC) $(subst -Iinclude,-I$(CURDIR)/include,$(subst -I$(CSRC),-I$(CURDIR)/$(CSRC),$(HIPFLAGS))) $(EXTRA_HIPFLAGS) -save-temps -shared $(CURDIR)/$(CSRC)/rt_hip.hip -o librt_hip.so
	python3 tools/isa_audit.py $(DEVICE_ASM) --budget tools/spill_budget.json
	cp $(BUILD)/librt_hip.so $@

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

.PHONY: all clean spill-budget
