// gpu::Serializable — anything that can append itself to a gpu::Buffer
// (reference src/gpu/Serializable.h:5-8).
#pragma once
#include "Buffer.h"

namespace gpu {

struct Serializable {
    virtual ~Serializable() = default;
    virtual void serialize(Buffer &buffer) const = 0;
};

}  // namespace gpu
