// K2p: the recoded range coder (arithmetic_code<uint64_t, uint8_t>, recode.cpp:322-323, with
// p(1) = (range / total) * pos, recode.cpp:823-827) for batches of few, long slices, in three passes.
//
// What can and what cannot be done in parallel.  The coder's state is (low, range).  `range` obeys an exact 63-bit
// integer recurrence on its own previous value, range <- (range / total) * pos or range minus that
// (arithmetic_code.h:107-114), with a shift by 8 whenever it drops below 2^51 (:115-122): unlike CABAC's 9-bit range
// with its four-quarter table there is no small state to speculate on, so the recurrence is walked bin by bin, one
// lane per slice (pass 1) -- the wall of this path.  `low`, however, is a sum: every bin adds a term that depends on
// the range alone (:110), at a bit position given by the number of shifts so far.  With the range and the shift
// count noted at every chunk boundary,
//   pass 1  range recurrence per slice; per chunk of kChunk bins: range and bytes emitted at its start
//   pass 2  one lane per chunk: the real coder from (low = 0, the noted range), its bytes ADDED into 32-bit sums per
//           byte position of the slice (a byte it emits with the carry bit set simply adds 256: one into the byte
//           before), and what is left of its low at the chunk's end added over the eight positions that follow
//   pass 3  per slice: carries over the sums from the last byte, which leaves the reference's bytes and, in the eight
//           positions past the last emitted byte, the reference's final low; finish() (:128-144) as it is
// give the reference's bytes.  Everything here is `__host__ __device__`: tests/k2p_emul.cpp runs the same functions on
// the CPU against the oracle.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define AVR_K2P_HD __host__ __device__ inline
#else
#define AVR_K2P_HD inline
#endif

namespace avr {
namespace k2p {

constexpr uint32_t kChunk = 1024;                      // bins per chunk
constexpr uint64_t kOne = uint64_t(1) << 63;           // fixed_one (arithmetic_code.h:54-55)
constexpr uint64_t kMinRange = uint64_t(1) << 51;      // arithmetic_code.h:61-62
constexpr uint32_t kTail = 8;                          // byte positions a chunk's left-over low is spread over

// One bin (arithmetic_code.h:106-126 with recode.cpp:823-827).  `div(range, total)` = range / total, and 0 for total 0 (a
// padding record: it then changes nothing, without a branch of its own on the chain); `emit(byte9)` takes each byte
// shifted out, carry bit included (0 .. 511).  WITH_LOW = false: the range recurrence alone (pass 1).
// Returns false when the bin has probability zero (arithmetic_code.h:116-118): the slice is in error.
template <bool WITH_LOW, class Div, class Emit>
AVR_K2P_HD bool bin(uint64_t &low, uint64_t &range, uint32_t rec, Div &&div, Emit &&emit) {
    const uint32_t pos = (rec >> 1) & 0x7fu, total = pos + ((rec >> 8) & 0x7fu);                 // recode.cpp:825
    const uint32_t b = total ? rec & 1u : 0u;          // padding record (total 0): quotient 0, r1 = 0, range and low stay
    const uint64_t r1 = div(range, total) * pos;       // recode.cpp:826
    const uint64_t r0 = range - r1;                    // arithmetic_code.h:108
    if (WITH_LOW) low += b ? r0 : 0;
    range = b ? r1 : r0;
    if (range < kMinRange) {                           // :115
        if (range == 0) return false;                  // :116-118
        do {                                           // :120-122, renormalize_and_emit_digit<uint8_t> (:147-180)
            emit(WITH_LOW ? uint32_t(low >> 55) : 0u); // bit 8 of the byte: the carry (low >= fixed_one)
            if (WITH_LOW) low = (low & ((uint64_t(1) << 55) - 1)) << 8;
            range <<= 8;
        } while (range < (kMinRange << 4));
    }
    return true;
}

// The range recurrence alone, without a branch: the number of 8-bit shifts a renormalisation takes (arithmetic_code.h:115-122:
// below 2^51, shift until 2^55 is reached) follows from the position of the range's top bit, k = (62 - msb) / 8.  Pass 1 is
// one dependent chain of these per slice, and every branch on it costs more than the few operations that replace it.
// `bytes` counts the shifts; returns false for a bin of probability zero (the range is 0 then and stays 0).
template <class Div>
AVR_K2P_HD bool range_step(uint64_t &range, uint32_t &bytes, uint32_t rec, Div &&div) {
    const uint32_t pos = (rec >> 1) & 0x7fu, total = pos + ((rec >> 8) & 0x7fu);                 // recode.cpp:825
    const uint32_t b = total ? rec & 1u : 0u;
    const uint64_t r1 = div(range, total) * pos;       // recode.cpp:826
    const uint64_t r = b ? r1 : range - r1;            // arithmetic_code.h:108-114
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t msb = 63u - uint32_t(__clzll(r | 1));
#else
    const uint32_t msb = 63u - uint32_t(__builtin_clzll(r | 1));
#endif
    const uint32_t k = r < kMinRange ? (62u - msb) >> 3 : 0u;
    range = r << (8 * k);
    bytes += r ? k : 0u;
    return r != 0;
}

// ------------------------------------------------------------------ the range recurrence in double precision (pass 1, round 3)
//
// Pass 1 is one dependent chain per slice, and on this chip EVERY dependent instruction of a lone wave costs 8-11 cycles
// whatever it is (measured, tools/ubench/chain_latency.hip: v_fma_f64 9.0, v_trunc_f64 8.2, v_mad_u64_u32 10.7,
// v_lshlrev_b64 10.1, v_add_u32 8.9, s_mul_hi_u32 9.0, s_lshl_b64 8.4): what counts is the NUMBER of dependent
// instructions per bin.  The 64-bit integer form above needs about 24 (64-bit shift, two conversions to double, the
// two-step quotient, two conversions back, two 32 x 32 -> 64 multiplies, a subtract with borrow, a select, count
// leading zeros and its arithmetic, a compare); this form needs 12, all on the double-precision pipe:
//
//   range = H * 2^32 + L exactly, H and L integers held in doubles, H >= 0, |L| <= 2^47 (a REDUNDANT pair: L is signed
//   and need not be below 2^32).  With inv = fl(1 / total), h = inv / 2 (avr_div.h), ps = +pos for a 1, -pos for a 0,
//   nb = 0 for a 1, 1 for a 0:
//      qh = trunc(fma(H, inv, h))                 H / total                      (exact: H < 2^44)
//      rh = fma(-qh, total, H)                    its remainder, < total
//      a  = rh * 2^32 + L                         (|a| < 2^48; not formed: see the next line)
//      ql = floor(fma(rh, 2^32 inv, fma(L, inv, h)))   a / total, rounded down; exact for |a| < 2^51: (a + 1/2) / total is at least
//                                                 1 / (2 total) away from an integer, the three roundings move it by 1.5 |a / total| 2^-52.
//                                                 The inner fma needs L alone, which is there when the bin begins: off the chain
//   => range / total = qh * 2^32 + ql (recode.cpp:826), and the new range nb * range + (range / total) * ps
//      (arithmetic_code.h:107-114) limb by limb:   H1 = fma(qh, ps, nb * H),  L1 = fma(ql, ps, nb * L)   (|L1| < 2^48)
//      v  = fma(H1, 2^32, L1)                     the new range as ONE double: exact when it is below 2^53, and rounding is
//                                                 monotone, so v < 2^51 (and v < 2^47) decide exactly what the reference's
//                                                 comparisons decide (arithmetic_code.h:115-122): shift by 0, 8 or 16 bits
//      cL = (L1 + 1.5 * 2^84) - 1.5 * 2^84        L1 rounded to a multiple of 2^32 (the adder's own rounding)
//      Hn = fma(cL, 2^-32, H1),  Ln = L1 - cL     the same value with |Ln| <= 2^31
//      H = Hn * s,  L = Ln * s                    s = 1, 2^8 or 2^16: |L| <= 2^47
// The comparison and the carry run side by side: the chain is t1, qh, rh, t2, ql, L1, v | t, cmp | cL, select, select | Hn, H: 11 deep.
// A new range below 2^39 (three or more bytes at once -- only a record with pos or neg 0 does that -- or zero: a bin of
// probability zero) is outside what the two thresholds cover: vmin_hi notes it and the slice takes the integer form above.
struct RangeFP { double H, L; };
struct BinFP { double inv, h, d, ps, nb, inv32; };        // inv32 = 2^32 * inv (exact: a power of two)
constexpr double kTwo32 = 4294967296.0, kInvTwo32 = 1.0 / 4294967296.0;
constexpr double kSplit32 = 1.5 * 4294967296.0 * 4294967296.0 * 1048576.0;    // 1.5 * 2^84: ulp 2^32
constexpr double kTwo51 = 2251799813685248.0, kTwo47 = 140737488355328.0, kTwo39 = 549755813888.0;

AVR_K2P_HD RangeFP fp_from_u64(uint64_t r) { return RangeFP{double(uint32_t(r >> 32)), double(uint32_t(r))}; }
AVR_K2P_HD uint64_t fp_to_u64(const RangeFP &r) { return (uint64_t(r.H) << 32) + uint64_t(int64_t(r.L)); }
// the operands of a record (the kernel reads them from two small tables instead)
AVR_K2P_HD BinFP fp_operands(uint32_t rec) {
    const uint32_t pos = (rec >> 1) & 0x7fu, total = pos + ((rec >> 8) & 0x7fu), b = total ? rec & 1u : 0u;
    const double inv = total ? 1.0 / double(total) : 0.0;
    return BinFP{inv, 0.5 * inv, double(total), b ? double(pos) : -double(pos), b ? 0.0 : 1.0, kTwo32 * inv};
}
// The constants of a step, as VALUES: the kernel keeps them in scalar registers (a 32-bit literal forces the compiler into
// the two-operand form of the multiply-add, which overwrites an input that is still needed: a register copy per use).
struct FpConsts { double two32, inv_two32, split32, two51, two47; };
AVR_K2P_HD FpConsts fp_consts() { return FpConsts{kTwo32, kInvTwo32, kSplit32, kTwo51, kTwo47}; }
AVR_K2P_HD uint32_t fp_hi(double x) { uint64_t u; __builtin_memcpy(&u, &x, 8); return uint32_t(u >> 32); }
constexpr uint32_t kTwo39Hi = 0x42600000u;             // the high word of 2^39 as a double: fp_hi(v) < this iff 0 <= v < 2^39

// One bin; returns the number of BITS shifted out (0, 8, 16).  vmin_hi: the high word of the smallest new range seen (the
// new range is never negative, so the high words of the doubles order like the values).
AVR_K2P_HD uint32_t range_step_fp(RangeFP &r, uint32_t &vmin_hi, const BinFP &o, const FpConsts &K) {
    const double qh = __builtin_trunc(__builtin_fma(r.H, o.inv, o.h));
    const double rh = __builtin_fma(-qh, o.d, r.H);
    const double H1 = __builtin_fma(qh, o.ps, r.H * o.nb);
    // (rh * 2^32 + L) / total as rh * (2^32 / total) + (L / total + h): the second term does not wait for the first division, which
    // takes one multiply-add off the chain.  Three roundings of relative size 2^-53 instead of two: still exact for |a| < 2^51.
    const double ql = __builtin_floor(__builtin_fma(rh, o.inv32, __builtin_fma(r.L, o.inv, o.h)));
    const double L1 = __builtin_fma(ql, o.ps, r.L * o.nb);
    const double v = __builtin_fma(H1, K.two32, L1);
    const double t = L1 + K.split32;
    const double cL = t - K.split32;
    const double Hn = __builtin_fma(cL, K.inv_two32, H1), Ln = L1 - cL;
    const int sh = v < K.two51 ? (v < K.two47 ? 16 : 8) : 0;
    r.H = __builtin_ldexp(Hn, sh);
    r.L = __builtin_ldexp(Ln, sh);
    const uint32_t vh = fp_hi(v);
    vmin_hi = vh < vmin_hi ? vh : vmin_hi;
    return uint32_t(sh);
}

// What is left of a chunk's low, as kTail bytes behind the ones it emitted (byte j: bits 62 - 8j .. 55 - 8j of low, the
// first with the carry bit on top, the last with the seven bits that remain, shifted up by one).
template <class Emit>
AVR_K2P_HD void leftover(uint64_t low, Emit &&emit) {
    emit(uint32_t(low >> 55));
    for (uint32_t j = 1; j < 7; j++) emit(uint32_t(low >> (55 - 8 * j)) & 0xffu);
    emit(uint32_t(low & 0x7fu) << 1);
}

// The reference's final low from the kTail normalised bytes past the last emitted one.
AVR_K2P_HD uint64_t low_from_tail(const uint8_t t[kTail]) {
    uint64_t low = 0;
    for (uint32_t j = 0; j < 7; j++) low |= uint64_t(t[j]) << (55 - 8 * j);
    return low | (t[7] >> 1);
}

// finish() of arithmetic_code<uint64_t, uint8_t>::encoder (arithmetic_code.h:128-144) on the exact final (low, range):
// the bytes it appends (at most 9) and the carry it sends into the bytes before them.
AVR_K2P_HD uint32_t finish(uint64_t low, uint64_t range, uint8_t tail[9], uint32_t *carry) {
    for (uint64_t stop = kOne >> 1; stop > 0; stop >>= 1) {          // :131-137
        const uint64_t x = (low | stop) & ~(stop - 1);
        if (stop < range && low <= x && x < uint64_t(low + range)) { low = x; break; }
    }
    uint32_t n = 0, cy = 0;
    while (low != 0 && n < 9) {                                      // :139-142
        if (low >= kOne) { cy = 1; low -= kOne; }
        const uint32_t d = uint32_t(low >> 55);
        tail[n++] = uint8_t(d);
        low = (low - (uint64_t(d) << 55)) << 8;
    }
    *carry = cy;
    return n;
}

}  // namespace k2p
}  // namespace avr
