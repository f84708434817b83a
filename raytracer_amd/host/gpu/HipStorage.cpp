#include "HipStorage.h"
#include "rt_hip.h"

namespace gpu {

void HipStorage::upload(const Buffer &buffer) {
    m_mirror.upload(buffer);
    m_dirty = true;
    if (m_slot > 0) bindToUnit(m_slot);
}

void HipStorage::bindToUnit(i32 unit) const {
    m_slot = unit;
    if (!m_dirty || m_ctx == nullptr) return;
    const auto &w = m_mirror.words();
    int rc = rt_upload(m_ctx, unit, w.data(), w.size() * sizeof(f32));
    ASSERT(rc == RT_OK);
    m_dirty = false;
}

}  // namespace gpu
