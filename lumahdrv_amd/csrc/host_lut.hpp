// host_lut.hpp -- host-only helpers of host_lut.cpp that the HIP translation units call (plain C++: compiled by g++ with
// -ffp-contract=off against the host libm, like the table builder itself)
#pragma once
#include <cstddef>

namespace lh {
int ycbcr_luma_code_host(float t, const float *lut, int maxVal, float Lmax);   // t = 219 y + 16
void ycbcr_ytab_host(const float *lut, size_t n, float Lmax, float *out);
bool ycbcr_half_table_host(float sc, float Lmax, float *out);                 // 0x7C01 floats; false: not usable for this (sc, Lmax)
}  // namespace lh
