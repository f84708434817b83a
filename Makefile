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

$(LIBDIR)/librt_hip.so: $(CSRC)/rt_hip.hip $(wildcard $(CSRC)/*.h) include/rt_hip.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -shared $(CSRC)/rt_hip.hip -o $@

$(LIBDIR)/librt_host.so: $(HOST_SRCS) $(HOST_HDRS) $(LIBDIR)/librt_hip.so
	@mkdir -p $(LIBDIR)
	$(CXX) $(HOSTFLAGS) -shared $(HOST_SRCS) -o $@ -L$(LIBDIR) -lrt_hip -lz -Wl,-rpath,'$$ORIGIN'

# Headless counterpart of the reference's main program, on the C++ host API.
examples/rt_render: examples/render_main.cpp $(LIBDIR)/librt_host.so
	$(CXX) $(HOSTFLAGS) examples/render_main.cpp -o $@ -L$(LIBDIR) -lrt_host -lrt_hip -Wl,-rpath,'$$ORIGIN/../$(LIBDIR)'

oracle/liboracle.so: oracle/oracle.cpp oracle/oracle_spectral.inc $(CSRC)/rt_math.h
	$(MAKE) -C oracle liboracle.so

clean:
	rm -f $(LIBDIR)/*.so oracle/liboracle.so examples/rt_render

.PHONY: all clean
