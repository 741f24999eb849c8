// Device-side binary range coder: the arithmetic of the reference's
//   arithmetic_code<FixedPoint, CompressedDigit, MinRange>::encoder
// (/root/reference/arithmetic_code.h:87-201) restated for one GPU lane per slice.
//
// Differences in form, none in the bytes produced:
//   * no callback: the caller evaluates probability_of_1(range) and updates low/range
//     itself (the reference takes a std::function, arithmetic_code.h:106);
//   * no `overflow` vector (arithmetic_code.h:154-174,200).  A digit that a later carry
//     can still change is held as `pend` followed by a run of `nff` all-ones digits -- by
//     construction the only shapes the reference's deferred digits can take: the first
//     deferred digit is < all-ones, every digit deferred while the interval still
//     straddles fixed_one is all-ones, and a deferral that starts below fixed_one makes
//     everything before it final.  Output is therefore strictly append-only, which is
//     what lets a lane stream bytes to HBM without ever reading them back;
//   * the common case -- nothing held back, no carry, digit unambiguous -- is one compare
//     and an append; everything else goes through the (rare) general path;
//   * FixedPoint arithmetic is done in exactly the reference's width (uint32_t for the
//     CABAC instantiation, cabac_code.h:18-24; uint64_t for recoded_code,
//     recode.cpp:322-323) so wrap-around in finish() matches (arithmetic_code.h:131-137).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace avr {

// Append-only byte writer into the lane's own 8-byte-aligned output region.  Bytes are
// shifted into a 64-bit word in stream order and stored 8 at a time; writes past `cap` are
// dropped and show up as n > cap (AVR_SLICE_OVERFLOW).
struct ByteWriter {
    uint8_t *base;
    uint32_t n, cap;
    uint64_t acc;           // the last (n & 7) bytes of the stream, most recent in the low byte

    __device__ __forceinline__ void init(uint8_t *p, uint32_t capacity) { base = p; n = 0; cap = capacity; acc = 0; }

    __device__ __forceinline__ void store8() {
        if (n <= cap) {
            const uint32_t lo = uint32_t(acc), hi = uint32_t(acc >> 32);
            uint2 v;                                   // byte-swap: first stream byte at the lowest address
            v.x = __builtin_bswap32(hi);
            v.y = __builtin_bswap32(lo);
            *reinterpret_cast<uint2 *>(base + n - 8) = v;
        }
    }
    __device__ __forceinline__ void put8(uint32_t b) {
        acc = (acc << 8) | (b & 0xffu);
        n++;
        if ((n & 7) == 0) store8();
    }
    // n must be even (true whenever only 16-bit digits have been written so far)
    __device__ __forceinline__ void put16_even(uint32_t d) {
        acc = (acc << 16) | (d & 0xffffu);
        n += 2;
        if ((n & 7) == 0) store8();
    }
    __device__ void flush() {
        const uint32_t r = n & 7;
        for (uint32_t i = 0; i < r; i++)
            if (n - r + i < cap) base[n - r + i] = uint8_t(acc >> (8 * (r - 1 - i)));
    }
};

template <typename F, int FBITS, int DBITS>
struct RangeEncoder {
    static constexpr F kOne = F(1) << (FBITS - 1);              // fixed_one, arithmetic_code.h:54-55
    static constexpr uint32_t kDigitMask = (1u << DBITS) - 1;
    static constexpr int kShift = FBITS - 1 - DBITS;            // log2(most_significant_digit), :150

    F low, range;
    int32_t pend;           // held-back digit (may still take a carry), -1 = none
    uint32_t nff;           // all-ones digits held back behind it
    ByteWriter w;

    __device__ __forceinline__ void init(F initial_range, uint8_t *out, uint32_t cap) {
        low = 0; range = initial_range; pend = -1; nff = 0;     // arithmetic_code.h:98-99
        w.init(out, cap);
    }

    __device__ __forceinline__ void put_digit(uint32_t d) {
        if (DBITS == 16) w.put16_even(d); else w.put8(d);
    }

    // Release everything held back, with (carry = 1) or without the pending carry.
    __device__ void release(uint32_t carry) {
        put_digit(uint32_t(pend) + carry);
        const uint32_t fill = carry ? 0u : kDigitMask;
        while (nff) { put_digit(fill); nff--; }
        pend = -1;
    }

    // renormalize_and_emit_digit<CompressedDigit> (arithmetic_code.h:147-180)
    __device__ __forceinline__ void emit_digit() {
        const uint32_t carry = low >= kOne;                                         // :154
        const uint32_t digit = uint32_t(low >> kShift) & kDigitMask;                // :158,164
        const uint32_t top = uint32_t(F(low + F(range - 1)) >> kShift) & kDigitMask;   // :165
        const bool certain = digit == top;
        if (__builtin_expect(certain && pend < 0 && !carry, 1)) {
            put_digit(digit);                                                       // :173
        } else {
            // still straddling fixed_one: one more all-ones digit behind the held-back one
            const bool extend = !certain && !carry && pend >= 0 && digit == kDigitMask;
            if (pend >= 0 && !extend) release(carry);                               // :155-157, :169-172
            if (certain) put_digit(digit);
            else if (extend) nff++;
            else pend = int32_t(digit);                                             // :166-167
        }
        low = F(F(low & ((F(1) << kShift) - 1)) << DBITS);                          // :158,177-178
        range = F(range << DBITS);                                                  // :179
    }

    // finish() (arithmetic_code.h:128-144): stop bit, then OutputDigit-sized (8-bit) digits.
    __device__ void finish() {
        for (F stop = kOne >> 1; stop > 0; stop >>= 1) {        // :131-137
            const F x = F((low | stop) & F(~F(stop - 1)));
            if (stop < range && low <= x && x < F(low + range)) { low = x; break; }
        }
        constexpr int sh8 = FBITS - 1 - 8;
        while (low != 0) {                                      // :139-142 (range = 1: never deferred)
            const uint32_t carry = low >= kOne;
            if (carry) low -= kOne;
            if (pend >= 0) release(carry);
            const uint32_t digit = uint32_t(low >> sh8);
            w.put8(digit);
            low = F(F(low - (F(digit) << sh8)) << 8);
        }
        range = 0;                                              // :143
        pend = -1; nff = 0;      // digits still held back are dropped, as the reference drops `overflow`
    }
};

// The CABAC instantiation (arithmetic_code<uint32_t, uint16_t, 0x200>, cabac_code.h:18-24) in
// WRITE-THROUGH form, used by k_cabac_encode.  Every digit is appended as soon as it exists; when low
// crosses fixed_one the carry (arithmetic_code.h:154-157) is added into the bytes already produced,
// last byte first -- first the ones still in the 8-byte staging word, then, rarely, the lane's own
// bytes in HBM, which nobody else touches.  The bytes are the same as with deferral (that is what the
// reference's deferred digits plus the carry add up to); what changes is the cost of the common
// case: no certainty test and no held-back state, and since one of a wave's 64 lanes emits on most
// bins, the length of this path is paid per bin.  (Measured on config 5: held-back form 2.84 ms; this
// form with a 2-byte store per digit 3.63 ms -- 64 partial-line stores per instruction -- hence the
// staging word.)
struct CabacEncoder {
    static constexpr uint32_t kOne = 0x80000000u;       // fixed_one, arithmetic_code.h:54-55
    uint32_t low, range;
    ByteWriter w;

    __device__ __forceinline__ void init(uint32_t initial_range, uint8_t *out, uint32_t capacity) {
        low = 0; range = initial_range;
        w.init(out, capacity);
    }
    __device__ void carry_back() {
        const uint32_t r = w.n & 7;                     // bytes still in the staging word
        if (r) {
            const uint64_t mask = (uint64_t(1) << (8 * r)) - 1, v = (w.acc & mask) + 1;
            w.acc = (w.acc & ~mask) | (v & mask);
            if ((v >> (8 * r)) == 0) return;
        }
        uint32_t p = w.n - r;
        if (p > w.cap) return;                          // overflowed output: nothing stored to carry into
        while (p > 0) {
            p--;
            const uint32_t b = uint32_t(w.base[p]) + 1u;
            w.base[p] = uint8_t(b);
            if (b <= 0xffu) break;
        }
    }
    // renormalize_and_emit_digit<uint16_t> (arithmetic_code.h:147-180)
    __device__ __forceinline__ void emit_digit() {
        if (__builtin_expect(low >= kOne, 0)) { carry_back(); low -= kOne; }            // :154-159
        w.put16_even(low >> 15);                                                        // :158, :184-190
        low = (low & 0x7fffu) << 16;                                                    // :177-178
        range <<= 16;                                                                   // :179
    }
    // finish() (arithmetic_code.h:128-144): stop bit, then 8-bit digits until low is used up
    __device__ void finish() {
        for (uint32_t stop = kOne >> 1; stop > 0; stop >>= 1) {                         // :131-137
            const uint32_t x = (low | stop) & ~(stop - 1);
            if (stop < range && low <= x && x < uint32_t(low + range)) { low = x; break; }
        }
        while (low != 0) {                                                              // :139-142
            if (low >= kOne) { carry_back(); low -= kOne; }
            const uint32_t digit = low >> 23;
            w.put8(digit);
            low = (low - (digit << 23)) << 8;
        }
        range = 0;                                                                      // :143
    }
};

}  // namespace avr
