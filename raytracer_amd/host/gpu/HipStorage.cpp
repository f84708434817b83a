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
    m_status = rt_upload(m_ctx, unit, w.data(), w.size() * sizeof(f32));
    if (m_status == RT_OK) m_dirty = false;   // a failed upload is tried again at the next bind
}

}  // namespace gpu
