// gpu::HipStorage — Storage backed by the HIP backend's flat device buffers.
// Takes the place of gpu::TBOStorage (reference src/gpu/TBOStorage.cpp:9-15):
// upload() hands the words to rt_upload(); bindToUnit(n) names the slot
// (the unit numbers of src/RayScene.cpp:32-49) and performs the upload that
// was deferred until the slot was known.
#pragma once
#include "HostStorage.h"

struct rt_ctx;

namespace gpu {

class HipStorage : public Storage {
    rt_ctx *m_ctx;
    HostStorage m_mirror;
    mutable i32 m_slot;
    mutable bool m_dirty = false;
    mutable int m_status = 0;

public:
    explicit HipStorage(rt_ctx *ctx, i32 slot = -1) : m_ctx(ctx), m_slot(slot) {}
    void upload(const Buffer &buffer) override;
    void bindToUnit(i32 unit) const override;
    int status() const override { return m_status; }
    const std::vector<f32> &words() const { return m_mirror.words(); }
};

}  // namespace gpu
