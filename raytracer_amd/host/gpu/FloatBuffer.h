// gpu::FloatBuffer — a Buffer whose element is one 32-bit word.  Integers are
// stored as their bit pattern, bools as 1.0f / 0.0f (wire rules of reference
// src/gpu/FloatBuffer.h:20-34, decoded by floatBitsToInt / bool(float) in
// src/RayScene.cpp:286-289,647).
#pragma once
#include "Buffer.h"
#include <cstring>
#include <vector>

namespace gpu {

class FloatBuffer : public Buffer {
    std::vector<f32> m_words;

    template <typename T> static f32 asWord(T v) {
        static_assert(sizeof(T) == sizeof(f32), "32-bit payloads only");
        f32 w;
        std::memcpy(&w, &v, sizeof w);
        return w;
    }

public:
    void push(const glm::vec3 &v) override { m_words.insert(m_words.end(), {v.x, v.y, v.z}); }
    void push(const glm::ivec3 &iv) override { m_words.insert(m_words.end(), {asWord(iv.x), asWord(iv.y), asWord(iv.z)}); }
    void push(const glm::vec2 &v) override { m_words.insert(m_words.end(), {v.x, v.y}); }
    void push(f32 val) override { m_words.push_back(val); }
    void push(i32 val) override { m_words.push_back(asWord(val)); }
    void push(u32 val) override { m_words.push_back(asWord(val)); }
    void push(bool val) override { m_words.push_back(val ? 1.0f : 0.0f); }
    void push(const char *bytes, size_t size) override {
        size_t n = size / sizeof(f32), at = m_words.size();
        m_words.resize(at + n);
        std::memcpy(m_words.data() + at, bytes, n * sizeof(f32));
    }

    const void *data() const override { return m_words.data(); }
    size_t size() const override { return m_words.size() * sizeof(f32); }
    void clear() override { m_words.clear(); }
};

}  // namespace gpu
