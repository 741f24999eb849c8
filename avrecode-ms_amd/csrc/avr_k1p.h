// K1p: intra-slice parallel CABAC encode (the same bytes as K1, i.e. as cabac::encoder of
// /root/reference/cabac_code.h:26-82 on arithmetic_code.h, from many lanes per slice).
//
// Why a slice can be cut at all.  The reference codes a slice strictly serially
// (arithmetic_code.h:107-114: every bin narrows [low, low+range)).  But CABAC's arithmetic is
// exactly decomposable:
//   (1) context states evolve per context, independent of low/range (cabac_code.h:43-47);
//       phase A resolves them and rewrites every bin as a one-byte RESOLVED code
//           c = (s << 1) | bin   context bin coded in state s = 2*pStateIdx + valMPS, s <= 125
//           252, 253             bypass bin 0 / 1 (the slot of state 126, never a real context
//                                state: pStateIdx 63 is only reached by put_terminate)
//           254, 255             state 127 = (pStateIdx 63, valMPS 1) with bin 0 / 1: the table row
//                                of pStateIdx 63 is "LPS range 2" in every quarter, which is exactly
//                                put_terminate (cabac_code.h:57-67): terminate(b) is code 255 - b,
//                                and so is a context bin met at pStateIdx 63 with symbol b
//       so that for every non-bypass code  symbol = ((c >> 1) ^ c) & 1  and  row = rows[c >> 2];
//   (2) range, written R << norm with R in [256, 511] (what cabac_code.h:37-41 recomputes per
//       bin), is a finite-state chain over R that forgets its past at every coded LPS: after
//       an LPS R is rangeTabLPS[p][q] renormalised, so only the quarter q (2 bits) of the range
//       before that LPS matters.  A STRETCH runs from just after an LPS to the first LPS at or
//       past the next multiple of kChunk bins; phase B1 walks every stretch for the 4 possible
//       entry quarters (they merge within a few bins), phase B2 chains the 4->4 maps per slice
//       (serial, one table look-up per stretch) and so fixes every stretch's entry range and
//       its bit position T (total renormalisation shifts before it);
//   (3) low is a sum of per-bin terms at known bit positions (arithmetic_code.h:110), so phase
//       C codes each stretch with low = 0 from its true entry range, aligned to the global
//       16-bit digit grid, and ADDS its digits into a per-slice array of 32-bit digit sums;
//       phase D adds the carries up once, serially from the last digit, and applies the
//       reference's finish() (arithmetic_code.h:128-144) to the exact final (low, range).
//
// Everything below is `__host__ __device__`: the kernels in avr_k1p.hip are thin per-lane
// wrappers, and tests/k1p_emul.cpp runs the very same functions on the CPU (test build only).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define AVR_HD __host__ __device__ inline
#else
#define AVR_HD inline
#endif

namespace avr {
namespace k1p {

// Keep a value where it is computed (an empty asm statement the optimiser cannot look through): used to
// make a batch of independent LDS table reads issue together, ahead of the dependent arithmetic --
// hipcc otherwise sinks each read to its use, behind the previous bin's result, and every bin waits
// out a full LDS latency.
#if defined(__HIP_DEVICE_COMPILE__)
#define AVR_PIN2(a, b) asm volatile("" : "+v"(a), "+v"(b))
#else
#define AVR_PIN2(a, b) ((void)0)
#endif

constexpr uint32_t kChunk = 1024;              // bins per fixed chunk (one lane of B1 / C)
constexpr uint32_t kNone = 0xffffffffu;
constexpr uint32_t kMaxStretch = 16 * kChunk;  // a longer stretch sends the slice to the serial kernel
constexpr uint32_t kCodeBypass = 252;
constexpr uint32_t kCodePad = 252;             // padding of a resolved stream's last 16-byte group

struct alignas(16) U4 { uint32_t x, y, z, w; };

AVR_HD int clz32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __clz(x);
#else
    return x ? __builtin_clz(x) : 32;
#endif
}

// ---- resolved codes
AVR_HD uint32_t code_sym(uint32_t c) { return ((c >> 1) ^ c) & 1u; }
AVR_HD bool code_is_bypass(uint32_t c) { return (c >> 1) == 126u; }
// a coded "LPS" (symbol 1 on a non-bypass bin): the range after it depends on q only
AVR_HD bool code_is_boundary(uint32_t c) { return code_sym(c) && !code_is_bypass(c); }
// resolved code of put_terminate(b) and of a context bin met in state s (sym = bin ^ valMPS)
AVR_HD uint32_t code_terminate(uint32_t b) { return 255u - b; }
AVR_HD uint32_t code_context(uint32_t s, uint32_t bin) { return s >= 126 ? 255u - ((bin ^ s) & 1u) : (s << 1) | bin; }

// Range (9-bit, normalised) right after a boundary bin coded from quarter q, and its shift.
AVR_HD uint32_t post_lps_range(uint32_t row, uint32_t q, uint32_t *shift) {
    const uint32_t rl = (row >> (8 * q)) & 0xffu;
    const uint32_t sh = uint32_t(clz32(rl)) - 23;          // rl << sh in [256, 511]
    *shift = sh;
    return rl << sh;
}

// What phases B and C need to know about a resolved code, precomputed per code value (256 entries):
//   row   rangeTabLPS[p][0..3] of its state, one byte per range quarter (0 for a bypass bin)
//   meta  bit 0: the coded symbol (1 = the LPS side, cabac_code.h:34; 0 for bypass, whose halving of
//         the scale, cabac_code.h:52-54, is the extra shift instead); bit 1: the bin adds to low
//         (the symbol again, or for bypass the bin); bit 8: bypass
struct CodeEntry { uint32_t row, meta; };
AVR_HD CodeEntry code_entry(uint32_t c, const uint32_t *rows /* rows[p], p = pStateIdx */) {
    if (code_is_bypass(c)) return CodeEntry{0u, 0x100u | ((c & 1u) << 1)};
    return CodeEntry{rows[c >> 2], code_sym(c) * 3u};
}

// One bin on the normalised range, branch-free: returns the renormalisation shift it causes
// (= bits of output).  A bypass bin has row 0: rLPS = 0 leaves R alone, and meta adds its one shift.
// Either side's renormalisation is "shift the new range up to nine bits" (the MPS side's range - rLPS lies in [128, 511]:
// one shift iff below 256; the LPS side's is rLPS itself, cabac_code.h:40-41), so the side is chosen first and one count of
// leading zeros serves both.
AVR_HD int clz32_nz(uint32_t x) { return __builtin_clz(x); }   // x != 0
// rangeTabLPS[p][(range >> 6) & 3] out of the state's row (one byte per quarter), range in [256, 511]
AVR_HD uint32_t lps_range(uint32_t row, uint32_t R) {
#if defined(__HIP_DEVICE_COMPILE__)
    // one byte permute: range >> 6 is 4 .. 7, which as a selector means bytes 0 .. 3 of the first operand; the selector's
    // other three bytes are 0 = byte 0 of the second operand, the constant 0
    return __builtin_amdgcn_perm(row, 0u, R >> 6);
#else
    return (row >> ((R >> 3) & 24)) & 0xffu;
#endif
}
AVR_HD uint32_t mul24(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(a, b);
#else
    return a * b;
#endif
}
AVR_HD uint32_t step_range(const CodeEntry &e, uint32_t *R) {
    const uint32_t rl = lps_range(e.row, *R);
    const uint32_t rm = *R - rl;                           // MPS side: range - rLPS
    const uint32_t x = (e.meta & 1u) ? rl : rm;            // never 0: a coded LPS has rLPS >= 2, rm >= 128
    const uint32_t sh = uint32_t(clz32_nz(x)) - 23;
    *R = x << sh;
    return sh + (e.meta >> 8);
}

// The same entry laid out for the inner loops (16 bytes, one LDS read), every field an operand as it stands:
//   side   all ones for a coded symbol 1 (the LPS side), else 0: the new range is a bit-field insert of rLPS over range - rLPS
//   k      what the bin adds to low, in half units, as a factor of range - rLPS: 2 for a coded symbol 1 (cabac_code.h:37-39),
//          1 for a bypass 1 (:52-54: the range itself, rLPS being 0 there), 0 otherwise
//   adj    bypass - 23: leading zeros of the new range + adj = the bits the bin shifts out
struct alignas(16) CodeEntryC { uint32_t row, side, k, adj; };
AVR_HD CodeEntryC code_entry_c(const CodeEntry &e) {
    const uint32_t sym = e.meta & 1u, adds = (e.meta >> 1) & 1u, byp = e.meta >> 8;
    return CodeEntryC{e.row, sym ? ~0u : 0u, adds ? (byp ? 1u : 2u) : 0u, byp - 23u};
}
// One bin: the new range, what it adds to low (*v), returns the shift (the range's own: *sh0)
AVR_HD uint32_t step_range_c(const CodeEntryC &e, uint32_t *R, uint32_t *v) {
    const uint32_t rl = lps_range(e.row, *R);
    const uint32_t rm = *R - rl;
    const uint32_t x = (rl & e.side) | (rm & ~e.side);
    const uint32_t lz = uint32_t(clz32_nz(x));
    *v = mul24(rm, e.k);
    *R = x << (lz - 23);
    return lz + e.adj;
}

// ------------------------------------------------------------------ phase B1

struct Stretch {
    uint32_t first;          // index of the boundary bin that opens the stretch (kNone: chunk inactive;
                             // chunk 0 is always active and starts at bin 0 with R = 510)
    uint32_t end;            // one past the stretch's last bin
    uint32_t t_exit[4];      // shifts inside the stretch, per entry quarter
    uint16_t r_exit[4];      // normalised range after the last bin, per entry quarter
    uint8_t exit_q;          // quarter seen by the closing boundary bin, 2 bits per entry quarter
    uint8_t too_long;        // stretch exceeded kMaxStretch: the slice goes to the serial kernel
    uint8_t pad[2];
};

// Where a slice's resolved codes are read from: load16(i), i a multiple of 16, gives codes i .. i+15; byte(i) one code.
// LinearCodes: the slice's codes in a row (the layout of the public entry points: 16-byte aligned, readable up to the
// next multiple of 16).  The kernels have a second one for their own, wave-interleaved buffer (avr_k1p.hip).
struct LinearCodes {
    const uint8_t *p;
    AVR_HD U4 load16(uint32_t i) const { return *reinterpret_cast<const U4 *>(p + i); }
    AVR_HD uint32_t byte(uint32_t i) const { return p[i]; }
};

// Visit the resolved codes [from .. n) of `src` in order, 16 bytes per load.  f(i, code) returns true to stop.
template <class Src, class F>
AVR_HD void for_codes_in(const Src &src, uint32_t from, uint32_t n, F &&f) {
    for (uint32_t base = from & ~15u; base < n; base += 16) {
        const U4 v = src.load16(base);
        uint32_t w0 = v.x, w1 = v.y, w2 = v.z, w3 = v.w;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
        for (uint32_t k = 0; k < 4; k++) {
            const uint32_t d = w0;
            w0 = w1; w1 = w2; w2 = w3;
            const uint32_t i = base + 4 * k;
            // (i + b - from) < (n - from) is "from <= i + b < n" in one unsigned compare
            if (i + 0 - from < n - from && f(i + 0, d & 0xffu)) return;
            if (i + 1 - from < n - from && f(i + 1, (d >> 8) & 0xffu)) return;
            if (i + 2 - from < n - from && f(i + 2, (d >> 16) & 0xffu)) return;
            if (i + 3 - from < n - from && f(i + 3, d >> 24)) return;
        }
    }
}
template <class F>
AVR_HD void for_codes(const uint8_t *res, uint32_t from, uint32_t n, F &&f) { for_codes_in(LinearCodes{res}, from, n, f); }

// The same for codes that all get the same treatment: g(code) for every code of res[from .. to), no
// index, no early exit.  The bulk is taken 64 bytes per lane per trip (four 16-byte loads issued
// together): lanes stream from addresses a kilobyte apart, so a 16-byte load drags in a whole
// cache line per lane and nothing keeps it resident until the lane comes back for the rest.
template <class G, class G4>
AVR_HD void for_codes_all(const uint8_t *res, uint32_t from, uint32_t to, G &&g, G4 &&g4) {
    if (from >= to) return;
    auto group16 = [&](const U4 &v) {
        uint32_t w0 = v.x, w1 = v.y, w2 = v.z, w3 = v.w;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
        for (uint32_t k = 0; k < 4; k++) {
            const uint32_t d = w0;
            w0 = w1; w1 = w2; w2 = w3;
            g4(d);                                         // four codes, first in the low byte
        }
    };
    // code by code up to a 16-byte boundary, 16-byte groups up to a cache line, then lines, and back down
    const uint32_t head_end = ((from + 15) & ~15u) < to ? ((from + 15) & ~15u) : to;
    for_codes(res, from, head_end, [&](uint32_t, uint32_t c) { g(c); return false; });
    uint32_t base = head_end;
    for (; (base & 63) && base + 16 <= to; base += 16) group16(*reinterpret_cast<const U4 *>(res + base));
    if (base + 64 <= to) {                                 // the next line is in flight while this one is worked on
        const U4 *p = reinterpret_cast<const U4 *>(res + base);
        U4 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3];
        for (; base + 64 <= to; base += 64) {
            const U4 *q = reinterpret_cast<const U4 *>(res + (base + 128 <= to ? base + 64 : base));   // unconditional: see c_stretch_in
            const U4 n0 = q[0], n1 = q[1], n2 = q[2], n3 = q[3];
            group16(v0); group16(v1); group16(v2); group16(v3);
            v0 = n0; v1 = n1; v2 = n2; v3 = n3;
        }
    }
    for (; base + 16 <= to; base += 16) group16(*reinterpret_cast<const U4 *>(res + base));
    if (base < to) for_codes(res, base, to, [&](uint32_t, uint32_t c) { g(c); return false; });
}

// The table entries of the four codes in a dword, read together (see AVR_PIN2).
AVR_HD void code_entries4(const CodeEntry *codes, uint32_t d, CodeEntry e[4]) {
    e[0] = codes[d & 0xffu]; e[1] = codes[(d >> 8) & 0xffu]; e[2] = codes[(d >> 16) & 0xffu]; e[3] = codes[d >> 24];
    AVR_PIN2(e[0].row, e[0].meta); AVR_PIN2(e[1].row, e[1].meta); AVR_PIN2(e[2].row, e[2].meta); AVR_PIN2(e[3].row, e[3].meta);
}
AVR_HD void code_entries4(const CodeEntryC *codes, uint32_t d, CodeEntryC e[4]) {
    e[0] = codes[d & 0xffu]; e[1] = codes[(d >> 8) & 0xffu]; e[2] = codes[(d >> 16) & 0xffu]; e[3] = codes[d >> 24];
    AVR_PIN2(e[0].row, e[0].side); AVR_PIN2(e[1].row, e[1].side); AVR_PIN2(e[2].row, e[2].side); AVR_PIN2(e[3].row, e[3].side);
}

AVR_HD void b1_stretch(const uint8_t *res, uint32_t n, uint32_t chunk, const CodeEntry *codes, uint32_t max_stretch,
                       Stretch *o) {
    const uint32_t lo = chunk * kChunk, limit = lo + kChunk;
    uint32_t R[4], T[4] = {0, 0, 0, 0};
    uint32_t i;
    o->too_long = 0;
    o->exit_q = 0;
    o->pad[0] = o->pad[1] = 0;
    if (chunk == 0) {
        o->first = 0;                                      // opens at bin 0 with the initial range 510 (cabac_code.h:30)
        R[0] = R[1] = R[2] = R[3] = 510;
        i = 0;
    } else {
        uint32_t f = kNone;
        const uint32_t hi = limit < n ? limit : n;
        for_codes(res, lo, hi, [&](uint32_t idx, uint32_t c) { if (code_is_boundary(c)) { f = idx; return true; } return false; });
        if (f == kNone) {
            o->first = kNone; o->end = 0;
            for (int q = 0; q < 4; q++) { o->t_exit[q] = 0; o->r_exit[q] = 0; }
            return;
        }
        o->first = f;
        const uint32_t row = codes[res[f]].row;
        for (uint32_t q = 0; q < 4; q++) { uint32_t sh; R[q] = post_lps_range(row, q, &sh); }
        i = f + 1;
    }
    // Until the four candidates have merged.  That takes a while -- on config 2 a mean of 66 bins, and the
    // slowest of a wave's 64 lanes needs about 240 -- so the part of it that lies inside the chunk, where
    // no bin can close the stretch, is walked 16 codes per load like the merged part.
    bool closed = false;
    auto merged = [&]() { return R[0] == R[1] && R[1] == R[2] && R[2] == R[3]; };
    auto code_by_code = [&](uint32_t stop) {               // up to `stop`, or merged, or closed
        if (i >= stop) return;
        uint32_t i_next = stop;
        for_codes(res, i, stop, [&](uint32_t idx, uint32_t c) {
            if (merged()) { i_next = idx; return true; }
            const bool closing = idx >= limit && code_is_boundary(c);
            if (closing)
                for (uint32_t q = 0; q < 4; q++) o->exit_q |= uint8_t(((R[q] >> 6) & 3) << (2 * q));
            const CodeEntry e = codes[c];
            for (uint32_t q = 0; q < 4; q++) T[q] += step_range(e, &R[q]);
            i_next = idx + 1;
            if (closing) { closed = true; return true; }
            return false;
        });
        i = i_next;
    };
    {
        const uint32_t interior_end = limit < n ? limit : n;
        const uint32_t a16 = (i + 15) & ~15u;
        code_by_code(a16 < interior_end ? a16 : interior_end);
        while (!merged() && i + 16 <= interior_end) {      // i is a multiple of 16 here
            const U4 v = *reinterpret_cast<const U4 *>(res + i);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
            for (uint32_t k = 0; k < 4; k++) {
                CodeEntry e[4];
                code_entries4(codes, w[k], e);
                for (uint32_t b = 0; b < 4; b++)
                    for (uint32_t q = 0; q < 4; q++) T[q] += step_range(e[b], &R[q]);
            }
            i += 16;
        }
        code_by_code(n);
    }
    // merged: one range, 16 codes per load.  Below `limit` no bin can close the stretch.
    uint32_t Rm = R[0], Tm = 0, end = i;
    if (!closed && i < n) {
        const uint32_t interior_end = limit < n ? limit : n;
        if (i < interior_end) {
            for_codes_all(res, i, interior_end, [&](uint32_t c) { Tm += step_range(codes[c], &Rm); },
                          [&](uint32_t d) {
                              CodeEntry e[4];
                              code_entries4(codes, d, e);
                              Tm += step_range(e[0], &Rm); Tm += step_range(e[1], &Rm);
                              Tm += step_range(e[2], &Rm); Tm += step_range(e[3], &Rm);
                          });
            i = interior_end;
        }
        end = n;
        for_codes(res, i, n, [&](uint32_t idx, uint32_t c) {
            const bool closing = idx >= limit && code_is_boundary(c);
            if (closing) o->exit_q = uint8_t(((Rm >> 6) & 3) * 0x55u);
            Tm += step_range(codes[c], &Rm);
            if (closing) { end = idx + 1; return true; }
            if (idx - lo > max_stretch) { o->too_long = 1; end = idx + 1; return true; }
            return false;
        });
        R[0] = R[1] = R[2] = R[3] = Rm;
    }
    o->end = end;
    for (uint32_t q = 0; q < 4; q++) { o->t_exit[q] = T[q] + Tm; o->r_exit[q] = uint16_t(R[q]); }
}

// ------------------------------------------------------------------ phase B2 (one lane per slice)

struct Entry {               // what phase C needs per active stretch
    uint32_t t_start;        // shifts before the stretch = bit position of its first output bit
    uint32_t q;              // entry quarter
};

struct SliceTotals {
    uint32_t t_total;        // shifts over the whole slice
    uint32_t r_final;        // normalised range after the last bin
    uint32_t bad;            // 1: a stretch was too long
    uint32_t pad;
};

AVR_HD void b2_chain(const Stretch *st, uint32_t n_chunks, Entry *en, SliceTotals *tot) {
    uint32_t T = 0, q = 0, r = 510, bad = 0;
    for (uint32_t c = 0; c < n_chunks; c++) {
        if (st[c].first == kNone) continue;
        en[c].t_start = T;
        en[c].q = q;
        T += st[c].t_exit[q];
        r = st[c].r_exit[q];
        bad |= st[c].too_long;
        q = (st[c].exit_q >> (2 * q)) & 3;
    }
    tot->t_total = T;
    tot->r_final = r;
    tot->bad = bad;
    tot->pad = 0;
}

// Number of 16-bit digits the reference has produced (emitted or deferred) after t shifts:
// its range is R << (22 - t + 16 nd) and a digit is produced whenever that exponent would
// drop below 1 (min_range 0x200, cabac_code.h:22, arithmetic_code.h:115-122).
AVR_HD uint32_t ref_digits(uint32_t t) { return t <= 21 ? 0 : (t - 21 + 15) / 16; }

// ------------------------------------------------------------------ phase C (one lane per active stretch)

// Digit sums: 32-bit per 16-bit digit of the slice's code string.  `Adder` supplies
//   void store(uint32_t digit_index, uint32_t v)   exclusive position, plain store; called with consecutive indices
//   void add(uint32_t digit_index, uint32_t v)     shared position, atomic add
//   void flush()                                   no store() follows: whatever the adder holds back goes out
// A stretch shares its first two digits with the windows of earlier stretches and its final
// window (two digits) with later ones; everything between is its own (argument in DESIGN.md).
//
// The coder is kept in normalised form: R in [256, 511] as in B1, and low as a 64-bit integer L2 in
// units of half the reference's scale, i.e. with the reference's range R << e (arithmetic_code.h:107-114,
// cabac_code.h:37-41) L2 = 2 * low >> e, which is exact: every term of low is a multiple of the
// scale it was added at, and a bypass bin's R/2 (cabac_code.h:52-54) is what the factor 2 is for.
// A 16-bit digit of the code string is due whenever e drops to 0 or below (arithmetic_code.h:115-122);
// in this form that is just a bit position `sp` = 15 - e of L2/2 reaching 15.  Nothing forces the
// digit out at that very bin: L2 has room for 4 more bins, so the bulk loop looks after every 4th
// bin, at the same place for every lane of a wave -- where the reference's form has one lane or
// another emitting at almost every bin.  A digit taken late has the carries of the bins in between
// already in it (it can reach 2^17); the sums are integers and phase D carries them on.
template <class Src, class Adder>
AVR_HD void c_stretch_in(const Src &src, const Stretch &st, const Entry &en, uint32_t chunk,
                         const CodeEntryC *codes, Adder &S) {
    uint32_t R, from;
    if (chunk == 0) { R = 510; from = 0; }
    else { uint32_t sh; R = post_lps_range(codes[src.byte(st.first)].row, en.q, &sh); from = st.first + 1; }
    const uint32_t phase = en.t_start & 15, g0 = en.t_start >> 4;
    uint64_t L2 = 0;
    int sp = int(phase) - 7;                               // e = 22 - phase at the start (cabac_code.h:30 shifted onto the digit grid)
    uint32_t j = 0;                                        // digits produced
    auto bin_e = [&](const CodeEntryC &e) {
        uint32_t v;
        const uint32_t sh = step_range_c(e, &R, &v);
        L2 = (L2 + v) << sh;
        sp += int(sh);
    };
    auto bin = [&](uint32_t c) { bin_e(codes[c]); };
    auto digits = [&]() {                                  // every digit that is due, oldest (topmost) first
        while (sp >= 15) {
            const uint32_t d = uint32_t(L2 >> (sp + 1));
            L2 &= (uint64_t(2) << sp) - 1;
            if (j < 2) S.add(g0 + j, d); else S.store(g0 + j, d);
            j++;
            sp -= 16;
        }
    };
    const uint32_t to = st.end;
    if (from < to) {
        auto four = [&](uint32_t d) {                      // 4 bins shift by at most 28: sp <= 14 + 28, L2 < 2^61
            CodeEntryC e[4];
            code_entries4(codes, d, e);
            bin_e(e[0]); bin_e(e[1]); bin_e(e[2]); bin_e(e[3]);
            digits();
        };
        auto group16 = [&](const U4 &v) { four(v.x); four(v.y); four(v.z); four(v.w); };
        // code by code up to a 16-byte boundary, 16-byte groups up to a cache line, then lines, and back down
        const uint32_t head_end = ((from + 15) & ~15u) < to ? ((from + 15) & ~15u) : to;
        for_codes_in(src, from, head_end, [&](uint32_t, uint32_t c) { bin(c); digits(); return false; });
        uint32_t base = head_end;
        for (; (base & 63) && base + 16 <= to; base += 16) group16(src.load16(base));
        U4 v0{0, 0, 0, 0}, v1 = v0, v2 = v0, v3 = v0;
        if (base + 64 <= to) { v0 = src.load16(base); v1 = src.load16(base + 16); v2 = src.load16(base + 32); v3 = src.load16(base + 48); }
        for (; base + 64 <= to; base += 64) {              // a whole cache line per lane per trip, the next one in flight
            const uint32_t w[16] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
            // the next line unconditionally (past the last whole line: this one again, a hit) -- a load inside a branch is waited for at the branch's end
            const uint32_t nb = base + 128 <= to ? base + 64 : base;
            v0 = src.load16(nb); v1 = src.load16(nb + 16); v2 = src.load16(nb + 32); v3 = src.load16(nb + 48);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 4
#endif
            for (uint32_t k = 0; k < 16; k++) four(w[k]);
        }
        for (; base + 16 <= to; base += 16) group16(src.load16(base));
        if (base < to) for_codes_in(src, base, to, [&](uint32_t, uint32_t c) { bin(c); digits(); return false; });
    }
    S.flush();
    // what is left is the coder's window: the top of digit g0 + j (with any carry) and 15 - e bits of the next
    if (sp >= 0) {
        S.add(g0 + j, uint32_t(L2 >> (sp + 1)));
        S.add(g0 + j + 1, uint32_t(L2 & ((uint64_t(2) << sp) - 1)) << (15 - sp));
    } else {
        S.add(g0 + j, uint32_t(L2 << (-sp - 1)) );
    }
}

template <class Adder>
AVR_HD void c_stretch(const uint8_t *res, const Stretch &st, const Entry &en, uint32_t chunk,
                      const CodeEntryC *codes, Adder &S) {
    c_stretch_in(LinearCodes{res}, st, en, chunk, codes, S);
}

// ------------------------------------------------------------------ phase D (one lane per slice)

// finish() of arithmetic_code<uint32_t,uint16_t,0x200>::encoder (arithmetic_code.h:128-144) on the
// exact final window.  Returns the carry it sends into the digits and the bytes it appends.
AVR_HD uint32_t d_finish(uint32_t low, uint32_t range, uint8_t tail[5], uint32_t *carry) {
    for (uint32_t stop = 1u << 30; stop > 0; stop >>= 1) {         // :131-137
        const uint32_t x = (low | stop) & ~(stop - 1);
        if (stop < range && low <= x && x < uint32_t(low + range)) { low = x; break; }
    }
    uint32_t n = 0, cy = 0;
    while (low != 0 && n < 5) {                                    // :139-142
        if (low >= 0x80000000u) { cy = 1; low -= 0x80000000u; }
        const uint32_t d = low >> 23;
        tail[n++] = uint8_t(d);
        low = (low - (d << 23)) << 8;
    }
    *carry = cy;
    return n;
}

// S: the slice's digit sums (ref_digits(t_total) + 2 entries are read).  Writes the final byte
// string to out (capacity cap) and returns its length.
AVR_HD uint32_t d_slice(const uint32_t *S, const SliceTotals &tot, uint8_t *out, uint32_t cap) {
    const uint32_t nd = ref_digits(tot.t_total);
    // the reference's final low: the two window digits, with whatever carry bit they hold
    const uint32_t low = uint32_t((uint64_t(S[nd]) << 15) + (S[nd + 1] >> 1));
    const uint32_t range = tot.r_final << (22 - tot.t_total + 16 * nd);
    uint8_t tail[5];
    uint32_t carry;
    const uint32_t n_tail = d_finish(low, range, tail, &carry);
    for (uint32_t i = nd; i-- > 0;) {                              // add the carries up, last digit first
        const uint32_t v = S[i] + carry;
        carry = v >> 16;
        if (2 * i + 1 < cap) { out[2 * i] = uint8_t(v >> 8); out[2 * i + 1] = uint8_t(v); }
    }
    for (uint32_t k = 0; k < n_tail; k++)
        if (2 * nd + k < cap) out[2 * nd + k] = tail[k];
    return 2 * nd + n_tail;
}

}  // namespace k1p
}  // namespace avr
