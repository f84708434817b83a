// gpu::Storage — device-side home of one stream (reference src/gpu/Storage.h:5-9).
// This is the plug point: the reference's only implementation is the OpenGL
// texture-buffer one; here HipStorage forwards to the C ABI and HostStorage
// keeps a host copy for inspection.
#pragma once
#include "Buffer.h"

namespace gpu {

struct Storage {
    virtual ~Storage() = default;
    virtual void upload(const Buffer &buffer) = 0;
    virtual void bindToUnit(i32 unit) const = 0;
    // Status of the last transfer to the device (0 = ok).  The reference's GL storage traps on a GL
    // error instead; a backend behind a C ABI reports, and the scene hands the code on.
    virtual int status() const { return 0; }
};

}  // namespace gpu
