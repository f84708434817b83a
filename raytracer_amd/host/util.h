// Fixed-width aliases and the trap-on-failure ASSERT used throughout the host
// classes (same names as reference src/util.h:6-23).
#pragma once

#include <cstdint>
#include <cstddef>

using u8 = std::uint8_t;
using u16 = std::uint16_t;
using u32 = std::uint32_t;
using u64 = std::uint64_t;
using i8 = std::int8_t;
using i16 = std::int16_t;
using i32 = std::int32_t;
using i64 = std::int64_t;
using f32 = float;
using f64 = double;

#define ASSERT(cond)                   \
    do {                               \
        if (!(cond)) __builtin_trap(); \
    } while (0)
