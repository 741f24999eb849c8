// Device-side binary range coder: the arithmetic of the reference's
//   arithmetic_code<FixedPoint, CompressedDigit, MinRange>::encoder
// (/root/reference/arithmetic_code.h:87-201) restated for one GPU lane per slice.
//
// Differences in form, none in the bytes produced:
//   * no callback: the caller evaluates probability_of_1(range) and updates low/range
//     itself (the reference takes a std::function, arithmetic_code.h:106);
//   * no `overflow` vector (arithmetic_code.h:154-174,200): digits are written through and a
//     carry is added into the bytes already produced (see CabacEncoder).  An earlier form held a
//     digit back together with a count of all-ones digits behind it -- the only shapes the
//     reference's deferred digits can take -- which kept the output append-only but cost a
//     certainty test per digit; measured slower on both kernels (config 5: K1 2.84 -> 2.70 ms,
//     K2 3.51 -> 3.00 ms);
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
    uint64_t prev;          // PAIR form only: the word before acc's, see put16_even_pair

    __device__ __forceinline__ void init(uint8_t *p, uint32_t capacity) { base = p; n = 0; cap = capacity; acc = 0; prev = 0; }

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
    // PAIR form (k_cabac_encode): the stream's words go out two at a time, 16 bytes a store -- the first of a pair waits in `prev`
    // (bytes [n & ~15, (n & ~15) + 8) of the stream while n & 8).  drain_pair() ends the form: put8 / store8 / flush take over.
    __device__ __forceinline__ void put16_even_pair(uint32_t d) {
        acc = (acc << 16) | (d & 0xffffu);
        n += 2;
        if ((n & 7) == 0) {
            if (n & 8) prev = acc;
            else if (n <= cap) {
                uint4 v;
                v.x = __builtin_bswap32(uint32_t(prev >> 32)); v.y = __builtin_bswap32(uint32_t(prev));
                v.z = __builtin_bswap32(uint32_t(acc >> 32));  v.w = __builtin_bswap32(uint32_t(acc));
                *reinterpret_cast<uint4 *>(base + n - 16) = v;
            } else if (n - 8 <= cap) {
                uint2 v;
                v.x = __builtin_bswap32(uint32_t(prev >> 32)); v.y = __builtin_bswap32(uint32_t(prev));
                *reinterpret_cast<uint2 *>(base + n - 16) = v;
            }
        }
    }
    __device__ __forceinline__ void drain_pair() {
        if ((n & 8) && (n & ~7u) <= cap) {
            uint2 v;
            v.x = __builtin_bswap32(uint32_t(prev >> 32)); v.y = __builtin_bswap32(uint32_t(prev));
            *reinterpret_cast<uint2 *>(base + (n & ~15u)) = v;
        }
    }
    __device__ void flush() {
        const uint32_t r = n & 7;
        for (uint32_t i = 0; i < r; i++)
            if (n - r + i < cap) base[n - r + i] = uint8_t(acc >> (8 * (r - 1 - i)));
    }
};

// The CABAC instantiation (arithmetic_code<uint32_t, uint16_t, 0x200>, cabac_code.h:18-24) in
// WRITE-THROUGH form, used by k_cabac_encode.  Every digit is appended as soon as it exists; when low
// crosses fixed_one the carry (arithmetic_code.h:154-157) is added into the bytes already produced,
// last byte first -- first the ones still in the 8-byte staging word, then, rarely, the lane's own
// bytes in HBM, which nobody else touches.  The bytes are the same as with deferral (that is what the
// reference's deferred digits plus the carry add up to); what changes is the cost of the common
// case: no certainty test and no held-back state, and since one of a wave's 64 lanes emits on most
// bins, the length of this path is paid per bin.  (With a 2-byte store per digit instead of the
// staging word: 3.63 ms on config 5 -- 64 partial-line stores per instruction.)
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
    __device__ void carry_back_pair() {                 // carry_back() while the writer is in its PAIR form
        const uint32_t r = w.n & 7;
        if (r) {
            const uint64_t mask = (uint64_t(1) << (8 * r)) - 1, v = (w.acc & mask) + 1;
            w.acc = (w.acc & ~mask) | (v & mask);
            if ((v >> (8 * r)) == 0) return;
        }
        if (w.n & 8) {                                  // the word that waits for its partner
            w.prev += 1;
            if (w.prev != 0) return;
        }
        uint32_t p = w.n & ~15u;
        if (p > w.cap) return;
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

// The recoded instantiation (arithmetic_code<uint64_t, uint8_t>, recode.cpp:322-323) in the same
// write-through form, used by k_range_encode: 8-bit digits, min_range 2^51 (arithmetic_code.h:61-62).
struct RangeEncoder64 {
    static constexpr uint64_t kOne = uint64_t(1) << 63; // fixed_one
    uint64_t low, range;
    ByteWriter w;

    __device__ __forceinline__ void init(uint64_t initial_range, uint8_t *out, uint32_t capacity) {
        low = 0; range = initial_range;
        w.init(out, capacity);
    }
    __device__ void carry_back() {                      // as CabacEncoder::carry_back
        const uint32_t r = w.n & 7;
        if (r) {
            const uint64_t mask = (uint64_t(1) << (8 * r)) - 1, v = (w.acc & mask) + 1;
            w.acc = (w.acc & ~mask) | (v & mask);
            if ((v >> (8 * r)) == 0) return;
        }
        uint32_t p = w.n - r;
        if (p > w.cap) return;
        while (p > 0) {
            p--;
            const uint32_t b = uint32_t(w.base[p]) + 1u;
            w.base[p] = uint8_t(b);
            if (b <= 0xffu) break;
        }
    }
    __device__ __forceinline__ void emit_digit() {      // arithmetic_code.h:147-180 with 8-bit digits
        if (__builtin_expect(low >= kOne, 0)) { carry_back(); low -= kOne; }
        w.put8(uint32_t(low >> 55));
        low = (low & ((uint64_t(1) << 55) - 1)) << 8;
        range <<= 8;
    }
    __device__ void finish() {                          // arithmetic_code.h:128-144
        for (uint64_t stop = kOne >> 1; stop > 0; stop >>= 1) {
            const uint64_t x = (low | stop) & ~(stop - 1);
            if (stop < range && low <= x && x < uint64_t(low + range)) { low = x; break; }
        }
        while (low != 0) {
            if (low >= kOne) { carry_back(); low -= kOne; }
            const uint32_t digit = uint32_t(low >> 55);
            w.put8(digit);
            low = (low - (uint64_t(digit) << 55)) << 8;
        }
        range = 0;
    }
};

}  // namespace avr
