// gpu::HostStorage — Storage that snapshots the uploaded words on the host.
// Used when a scene is built without a device (wire-format tests, the CPU
// checker) and as the mirror a HipStorage keeps so streams can be re-read.
#pragma once
#include "Storage.h"
#include <cstring>
#include <vector>

namespace gpu {

class HostStorage : public Storage {
    std::vector<f32> m_copy;
    mutable i32 m_unit = -1;

public:
    void upload(const Buffer &buffer) override {
        m_copy.resize(buffer.size() / sizeof(f32));
        if (!m_copy.empty()) std::memcpy(m_copy.data(), buffer.data(), m_copy.size() * sizeof(f32));
    }
    void bindToUnit(i32 unit) const override { m_unit = unit; }
    i32 unit() const { return m_unit; }
    const std::vector<f32> &words() const { return m_copy; }
};

}  // namespace gpu
