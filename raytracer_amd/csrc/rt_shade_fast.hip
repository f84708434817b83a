// rt_shade_fast.hip — the shading and connection kernels once more, compiled with the hardware's own
// arithmetic (RT_OPT_FAST_SHADE, opt-in): one-instruction reciprocals, square roots, exponentials, logarithms,
// sines and cosines and fused multiply-adds in place of the pinned, correctly rounded sequences that make
// the default kernels equal to the CPU checker bit for bit.  Same source (rt_kernels_shade.h, rt_bsdf.h,
// rt_device.h, rt_math.h), other compiler flags and RT_FAST_SHADE defined; the traversal kernels, the
// camera rays and the per-pixel sums stay in rt_hip.hip, so hit ids, distances and barycentrics are
// unchanged and only the radiance moves, within the stated RMSE.
//
// Everything here lives in namespaces of its own (rtm_fast, rtd_fast) so that the two builds of the
// shared inline functions never meet in the linker; the launchers below take the kernel arguments as
// plain bytes of the same layout.

#define RT_FAST_SHADE 1
#define rtm rtm_fast
#define rtd rtd_fast

#include <hip/hip_runtime.h>

#include <cstring>

#include "rt_wavefront.h"
#include "rt_kernels_shade.h"

#include "rt_shade_fast.h"

#undef rtm
#undef rtd

const size_t rtFastLayout[4] = {sizeof(rtd_fast::SceneView), sizeof(FrameParams), sizeof(Batch), sizeof(PathBuffers)};

void rtFastConnect(bool spectral, int grid, void *stream, const void *view, const void *batch, const void *paths, int stage, int countOn, int skipOccluded) {
    rtd_fast::SceneView sc;
    Batch bt;
    PathBuffers pb;
    std::memcpy(&sc, view, sizeof sc);
    std::memcpy(&bt, batch, sizeof bt);
    std::memcpy(&pb, paths, sizeof pb);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (spectral) hipLaunchKernelGGL(k_connect<true>, dim3(grid), dim3(kBlock), 0, s, sc, bt, pb, stage, countOn, skipOccluded);
    else hipLaunchKernelGGL(k_connect<false>, dim3(grid), dim3(kBlock), 0, s, sc, bt, pb, stage, countOn, skipOccluded);
}

void rtFastShade(bool spectral, int grid, void *stream, const void *view, const void *frame, const void *batch, const void *paths, int stage, int countOn) {
    rtd_fast::SceneView sc;
    FrameParams fp;
    Batch bt;
    PathBuffers pb;
    std::memcpy(&sc, view, sizeof sc);
    std::memcpy(&fp, frame, sizeof fp);
    std::memcpy(&bt, batch, sizeof bt);
    std::memcpy(&pb, paths, sizeof pb);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (spectral) hipLaunchKernelGGL(k_shade_spectral, dim3(grid), dim3(kBlock), 0, s, sc, fp, bt, pb, stage, countOn);
    else hipLaunchKernelGGL(k_shade_rgb, dim3(grid), dim3(kBlock), 0, s, sc, fp, bt, pb, stage, countOn);
}
