// floor(n / d) for the recoded coder's `range / total` (recode.cpp:826): n < 2^63 + 1, d < 256.  The GPU has no integer
// divide.  `__host__ __device__` so that tests/k1p_emul.cpp checks the very same function against the CPU's divide.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define AVR_DIV_HD __host__ __device__ inline
#else
#define AVR_DIV_HD inline
#endif

namespace avr {

// Through the FP64 pipe (full rate on CDNA, where 32-bit integer multiplies run at a quarter): two steps of 32 bits.
// With inv = fl(1/d), h = inv/2:  trunc(fma(a, inv, h)) == floor(a / d) for every integer a < 2^40 -- (a + 1/2)/d lies
// at least 1/(2d) >= 2^-9 away from an integer, and the two roundings move the result by at most a * 2^-52 <= 2^-12.
// Remainders and the 40-bit second dividend are integers below 2^53: exact.  d = 0 with inv = 0 gives 0.
AVR_DIV_HD uint64_t div_u64_small_f64(uint64_t n, double d, double inv) {
    const double hi = double(uint32_t(n >> 32)), lo = double(uint32_t(n)), h = 0.5 * inv;
    const double qh = __builtin_trunc(__builtin_fma(hi, inv, h));
    const double a = __builtin_fma(__builtin_fma(-qh, d, hi), 4294967296.0, lo);
    const double ql = __builtin_fma(a, inv, h);
    return (uint64_t(uint32_t(qh)) << 32) | uint32_t(ql);
}

}  // namespace avr
