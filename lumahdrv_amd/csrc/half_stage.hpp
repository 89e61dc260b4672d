// half_stage.hpp -- host-side float -> binary16 conversion with an exactness check (half_stage.cpp)
#pragma once
#include <cstddef>
#include <cstdint>

namespace lh {
bool f16c_available();
bool convert_f32_to_f16_checked(const float *src, uint16_t *dst, size_t n);   // call only when f16c_available()
}  // namespace lh
