// rt_shade_fast.h — launchers of the fast-arithmetic shading kernels (rt_shade_fast.hip) as rt_hip.hip
// calls them.  The structures cross as bytes; their sizes are checked on both sides.
#pragma once

#include <cstddef>

void rtFastConnect(bool spectral, int grid, void *stream, const void *view, const void *batch, const void *paths, int stage, int countOn, int skipOccluded);
void rtFastShade(bool spectral, int grid, void *stream, const void *view, const void *frame, const void *batch, const void *paths, int stage, int countOn);
// sizeof SceneView, FrameParams, Batch, PathBuffers as this translation unit sees them.
extern const size_t rtFastLayout[4];
