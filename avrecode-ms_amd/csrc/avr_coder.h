// Device-side binary range coder: the arithmetic of the reference's
//   arithmetic_code<FixedPoint, CompressedDigit, MinRange>::encoder
// (/root/reference/arithmetic_code.h:87-201) restated for one GPU lane per slice.
//
// Differences in form, none in the bytes produced:
//   * no callback: the caller evaluates probability_of_1(range) and passes r1
//     (the reference takes a std::function, arithmetic_code.h:106);
//   * no `overflow` vector (arithmetic_code.h:154-174,200).  A digit that a later carry
//     can still change is held as `pend` followed by a run of `nff` all-ones digits -- by
//     construction the only shapes the reference's deferred digits can take: the first
//     deferred digit is < all-ones, every digit deferred while the interval still
//     straddles fixed_one is all-ones, and a deferral that starts below fixed_one makes
//     everything before it final.  Output is therefore strictly append-only, which is
//     what lets a lane stream bytes to HBM without ever reading them back;
//   * FixedPoint arithmetic is done in exactly the reference's width (uint32_t for the
//     CABAC instantiation, cabac_code.h:18-24; uint64_t for recoded_code,
//     recode.cpp:322-323) so wrap-around in finish() matches (arithmetic_code.h:131-137).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace avr {

// Append-only byte writer into the lane's own output region.  Bytes are gathered into a
// 64-bit word and stored 8 at a time (aligned); writes past `cap` are dropped and show up
// as n > cap (AVR_SLICE_OVERFLOW).
struct ByteWriter {
    uint8_t *base;
    uint32_t n, cap;
    uint64_t acc;           // bytes [n & ~7, n) of the stream, little-endian in memory order

    __device__ void init(uint8_t *p, uint32_t capacity) { base = p; n = 0; cap = capacity; acc = 0; }

    __device__ __forceinline__ void put8(uint32_t b) {
        acc |= uint64_t(b & 0xff) << ((n & 7) * 8);
        n++;
        if ((n & 7) == 0) {
            if (n <= cap) *reinterpret_cast<uint64_t *>(base + n - 8) = acc;
            else spill(n - 8, 8);
            acc = 0;
        }
    }
    __device__ __forceinline__ void put16(uint32_t d) { put8(d >> 8); put8(d); }

    __device__ void spill(uint32_t from, uint32_t count) {      // byte-wise, bounds-checked
        for (uint32_t i = 0; i < count; i++)
            if (from + i < cap) base[from + i] = uint8_t(acc >> (8 * i));
    }
    __device__ void flush() {
        const uint32_t r = n & 7;
        if (r) spill(n - r, r);
    }
};

template <typename F, int FBITS, int DBITS>
struct RangeEncoder {
    static constexpr F kOne = F(1) << (FBITS - 1);              // fixed_one, arithmetic_code.h:54-55
    static constexpr uint32_t kDigitMask = (1u << DBITS) - 1;

    F low, range;
    int32_t pend;           // held-back digit (may still take a carry), -1 = none
    uint32_t nff;           // all-ones digits held back behind it
    ByteWriter w;

    __device__ void init(F initial_range, uint8_t *out, uint32_t cap) {
        low = 0; range = initial_range; pend = -1; nff = 0;     // arithmetic_code.h:98-99
        w.init(out, cap);
    }

    __device__ __forceinline__ void put_digit(uint32_t d) {
        if (DBITS == 16) w.put16(d); else w.put8(d);
    }

    // Release everything held back, with (carry = 1) or without the pending carry.
    __device__ __forceinline__ void release(uint32_t carry) {
        if (pend >= 0) {
            put_digit(uint32_t(pend) + carry);
            const uint32_t fill = carry ? 0u : kDigitMask;
            while (nff) { put_digit(fill); nff--; }
            pend = -1;
        }
    }

    // renormalize_and_emit_digit<CompressedDigit> (arithmetic_code.h:147-180)
    __device__ __forceinline__ void emit_digit() {
        constexpr int sh = FBITS - 1 - DBITS;                   // log2(most_significant_digit)
        if (low >= kOne) { release(1); low -= kOne; }           // :154-159
        const uint32_t digit = uint32_t(low >> sh);             // :164
        const uint32_t top = uint32_t(F(low + F(range - 1)) >> sh) & kDigitMask;   // :165
        if (digit == top) {                                     // :168-174
            release(0);
            put_digit(digit);
        } else if (digit == kDigitMask && pend >= 0) {          // still straddling fixed_one
            nff++;
        } else {                                                // :166-167, a fresh deferral
            release(0);
            pend = int32_t(digit);
        }
        low = F(F(low - (F(digit) << sh)) << DBITS);            // :177-178
        range = F(range << DBITS);                              // :179
    }

    // finish() (arithmetic_code.h:128-144): stop bit, then OutputDigit-sized (8-bit) digits.
    __device__ void finish() {
        for (F stop = kOne >> 1; stop > 0; stop >>= 1) {        // :131-137
            const F x = F((low | stop) & F(~F(stop - 1)));
            if (stop < range && low <= x && x < F(low + range)) { low = x; break; }
        }
        constexpr int sh8 = FBITS - 1 - 8;
        while (low != 0) {                                      // :139-142 (range = 1: never deferred)
            if (low >= kOne) { release(1); low -= kOne; }
            const uint32_t digit = uint32_t(low >> sh8);
            release(0);
            w.put8(digit);
            low = F(F(low - (F(digit) << sh8)) << 8);
        }
        range = 0;                                              // :143
        // digits still held back here are dropped, as the reference drops `overflow`
        pend = -1; nff = 0;
    }
};

}  // namespace avr
