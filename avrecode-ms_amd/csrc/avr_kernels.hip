// HIP kernels of the arithmetic re-encode path for gfx950 (MI355X).
//
//   K1  k_cabac_encode   cabac::encoder over recorded bins        (cabac_code.h:26-82
//                        on arithmetic_code<uint32_t,uint16_t,0x200>, decompress direction,
//                        recode.cpp:1442-1481)
//   K2  k_range_encode   recoded_code::encoder over recorded bins (arithmetic_code<uint64_t,
//                        uint8_t>, compress direction, recode.cpp:1075-1103, 823-827)
//   k_pack_tiles         slice-major records -> wave-interleaved tiles
//   k_synth_*            seeded synthetic bin streams (avr_synth.h)
//
// Mapping: one lane per slice (each slice has its own coder object in the reference,
// recode.cpp:1270, 1525, so slices are independent and a slice is strictly serial).
// One wave (64 slices) per workgroup.  Integer work only; no MFMA.
//
// LDS per workgroup (K1):
//   [0, 1 KiB)          packed CABAC table, 128 x 8 B (avr_tables.h)
//   [1 KiB, ...)        context states, dword (k, lane) at 4*(k*64 + lane) holds the four
//                       state bytes 4k..4k+3 of the lane's slice: the bank is lane % 32 for
//                       every context, so the data-dependent state read and write of a bin
//                       never conflict across lanes.
#include <hip/hip_runtime.h>
#include <string.h>

#include <mutex>
#include <type_traits>
#include <vector>
#include <stdint.h>
#include <stdlib.h>

#include "avr_coder.h"
#include "avr_div.h"
#include "avr_internal.h"
#include "avr_k1p.h"
#include "avr_synth.h"
#include "avr_tables.h"

namespace avr {

static __device__ const CabacTables d_tables = make_cabac_tables();

// ------------------------------------------------------------------ record fetch

// Where lane `lane` of processing slot `g` finds chunk c (8 records = 16 bytes).
template <bool TILED>
struct ChunkSource {
    const uint4 *p;
    uint32_t stride;       // in uint4
    __device__ ChunkSource(const void *recs, const uint64_t *off, uint32_t g, uint32_t slice) {
        if (TILED) {       // off = tile_off (16-byte units), one entry per 64 slots
            p = reinterpret_cast<const uint4 *>(recs) + off[g >> 6] + (g & 63);
            stride = 64;
        } else {           // off = rec_off (records), indexed by slice
            p = reinterpret_cast<const uint4 *>(reinterpret_cast<const uint16_t *>(recs) + off[slice]);
            stride = 1;
        }
    }
    __device__ __forceinline__ uint4 load(uint32_t c) const { return p[size_t(c) * stride]; }
    // the same read marked non-temporal: records are read once, and what they would push out of L2 are the output lines being filled
    __device__ __forceinline__ uint4 load_nt(uint32_t c) const {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p + size_t(c) * stride));
        return make_uint4(v.x, v.y, v.z, v.w);
    }
};

// ------------------------------------------------------------------ K1

// One CABAC bin (cabac_code.h:33-67 on arithmetic_code.h:106-126), written branch-free up to
// the renormalisation: every lane does the same LDS read / table read / LDS write whatever
// kind of bin it holds, so a wave never splits on context-vs-bypass.
//   * where a bin's state byte lives comes from one look-up, sel_off[selector]: contexts have their
//     byte in the lane's column of the state rows; bypass, terminate and everything else (padding,
//     selectors that are no context of the batch) go to pseudo contexts whose pseudo states never move;
//   * the pseudo states are table rows of their own: 128 bypass (flag in .y: the coded range is
//     range / 2), 130 terminate (LPS range 2 in every range quarter, valMPS 0), 132 no-op (LPS range 0,
//     and a flag in .y that forces symbol 0, so such a record changes nothing whatever its bin);
//   * 134 / 135 are the no-op again, but 134's successor is 135 and 135 is its own: the pseudo context that
//     starts in 134 reads 135 once any bin has gone through it.  That is where a context goes that the slice
//     declares but the (sampled) census of the batch did not see -- the lane finds out at the end, at no cost
//     per bin, and hands its slice back (AVR_SLICE_RETRY_SERIAL: coded by a second launch without renumbering).
//   * 131 / 133 remember put_terminate(1): the terminate context goes 130 -> 131 at the first one and 131 -> 133 at any terminate bin
//     after it (both rows are no-ops); the caller reads the byte once, at the end of the slice (k_cabac_encode).
struct CabacLane {
    CabacEncoder e;

    __device__ __forceinline__ void bin(uint32_t rec, uint32_t off, const uint2 *tab, uint8_t *st_lane) {
        uint8_t *sp = st_lane + off;                             // state byte of (context, lane): dword (k >> 2, lane), byte k & 3
        uint32_t s = *sp;
        // The empty asm statements pin the two LDS reads where they are written: without them
        // hipcc sinks each read into a branch on the bin kind (it is only "needed" on one side of
        // a select), which splits the wave and exposes the full LDS latency behind every branch.
        asm volatile("" : "+v"(s));
        uint2 ent = tab[s];
        asm volatile("" : "+v"(ent.x), "+v"(ent.y));
        // normalize = floor(log2(range / 0x100)) (cabac_code.h:37,59,70-79); range != 0 here
        const int norm = 23 - __builtin_clz(e.range);
        const uint32_t q = (e.range >> (norm + 6)) & 3;          // (range_approx & 0x180) >> 7, :39-40
        const uint32_t r_tab = ((ent.x >> (q * 8)) & 0xffu) << norm;              // :40-41, :60
        // bypass (:53): its table row is 0 and the top bit of ent.y is set, so this is an OR, not a branch
        const uint32_t r1 = r_tab | ((e.range >> 1) & uint32_t(int32_t(ent.y) >> 31));
        const uint32_t sym = (rec ^ s) & 1 & ~(ent.y >> 30);     // :34 (pseudo-states have valMPS 0; the no-op row forces 0)
        const uint32_t r0 = e.range - r1;                        // arithmetic_code.h:107-114
        e.low += sym ? r0 : 0u;
        e.range = sym ? r1 : r0;
        *sp = uint8_t(ent.y >> (8 * sym));                       // cabac_code.h:43-47
        if (e.range < 0x200u) e.emit_digit();                    // arithmetic_code.h:115-122 (one digit)
    }
};

// The same bin in NORMALISED form (the form of K1p's phase C, avr_k1p.h: c_stretch): the range as 9 bits R in [256, 511]
// with the reference's range = R << e (cabac_code.h:37-41), low as the 64-bit integer L2 = 2 low >> e, and sp = 15 - e the
// bit of L2 / 2 where the next 16-bit digit of the code string begins.  A digit is due when e drops to 0 (sp >= 15,
// arithmetic_code.h:115-122) -- but nothing forces it out at that very bin: L2 has room for four more bins (28 shifts at
// most), so the caller takes the due digits after every fourth bin, at the same instruction for all lanes of the wave,
// where the reference's form has one lane or another in its emit branch at almost every bin.  A digit taken late has
// the carries of the bins in between in it (bit 16 set): carry_back(), as for low >= fixed_one in the reference's form.
struct CabacLaneN {
    uint32_t R;
    uint64_t L2;
    int32_t sp;
    CabacEncoder e;                                              // its writer, and finish() in the reference's form

    __device__ __forceinline__ void init(uint8_t *out, uint32_t capacity) {
        R = 510; L2 = 0; sp = -7;                                // range 510 << 22 = 0x7F800000 (cabac_code.h:30)
        e.init(0x7F800000u, out, capacity);
    }
    // tabn[(state << 1) | bin]: the entry of avr_k1p.h's CodeEntryC -- { the state's LPS ranges, the side as a mask, what the bin
    // adds to low as a factor of range - rLPS, bypass - 23 } -- with the successor state in the factor's top byte, which the 24-bit
    // multiply does not see (norm_entry() below)
    __device__ __forceinline__ void bin(uint32_t rec, uint32_t off, const uint4 *tabn, uint8_t *st_lane) {
        uint8_t *spb = st_lane + off;
        uint32_t s = *spb;
        asm volatile("" : "+v"(s));                              // see CabacLane::bin
        uint4 ent = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(tabn) + ((s << 5) | ((rec & 1u) << 4)));
        asm volatile("" : "+v"(ent.x), "+v"(ent.y));
        uint32_t v;
        const uint32_t sh = k1p::step_range_c(k1p::CodeEntryC{ent.x, ent.y, ent.z, ent.w}, &R, &v);
        L2 = (L2 + v) << sh;
        sp += int32_t(sh);
        *spb = uint8_t(ent.z >> 24);                                             // cabac_code.h:43-47
    }
    // Four bins with their four state bytes read TOGETHER, before the first is coded: a bin's state is then one LDS round trip (its
    // table entry) behind the previous bin's instead of two.  A bin whose context one of the group's earlier bins has just moved
    // takes that bin's successor state instead of the byte read (the latest wins; where the states live does not depend on them,
    // so the comparisons are off the chain); the four writes follow in order.  Measured variant: k_cabac_encode FORM 4.
    __device__ __forceinline__ void bin4(const uint32_t (&rec)[4], const uint32_t (&off)[4], const uint4 *tabn, uint8_t *st_lane) {
        uint32_t s[4], next[4];
#pragma unroll
        for (int j = 0; j < 4; j++) s[j] = st_lane[off[j]];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint32_t sj = s[j];
#pragma unroll
            for (int i = 0; i < j; i++) sj = off[j] == off[i] ? next[i] : sj;
            asm volatile("" : "+v"(sj));
            uint4 ent = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(tabn) + ((sj << 5) | ((rec[j] & 1u) << 4)));
            asm volatile("" : "+v"(ent.x), "+v"(ent.y));
            uint32_t v;
            const uint32_t sh = k1p::step_range_c(k1p::CodeEntryC{ent.x, ent.y, ent.z, ent.w}, &R, &v);
            L2 = (L2 + v) << sh;
            sp += int32_t(sh);
            next[j] = ent.z >> 24;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) st_lane[off[j]] = uint8_t(next[j]);
    }
    // the table entry of (state or pseudo state s, bin): see k_cabac_encode for the pseudo states
    static __device__ uint4 norm_entry(uint32_t s, uint32_t bin) {
        uint32_t row, sym, next, byp = 0;
        if (s < 128) {
            row = d_tables.packed[s][0];
            sym = (bin ^ s) & 1u;                                                // the bin is not valMPS (cabac_code.h:34)
            next = (d_tables.packed[s][1] >> (8 * sym)) & 0xffu;
        } else {
            const uint32_t t = s - 128;
            row = t == 2 ? 0x02020202u : 0u;                                     // terminate: LPS range 2, valMPS 0 (:59-61)
            sym = t == 2 ? bin : 0u;
            byp = t == 0;                                                        // bypass: the bin itself goes to low (:52-54)
            next = t == 6 ? 135u : (t == 3 || t == 5) ? 133u : (t == 2 && bin) ? 131u : s;   // 131: put_terminate(1) has been; 133: and a terminate bin after it
        }
        const uint32_t k = byp ? bin : 2u * sym;
        return make_uint4(row, sym ? ~0u : 0u, k | next << 24, byp - 23u);
    }
    template <bool PAIR = false>                                 // PAIR: the output in 16-byte stores (ByteWriter::put16_even_pair)
    __device__ __forceinline__ void digits() {                   // every digit that is due, oldest first
        while (sp >= 15) {
            uint32_t d = uint32_t(L2 >> (sp + 1));
            L2 &= (uint64_t(2) << sp) - 1;
            if (__builtin_expect(d >> 16, 0)) { if constexpr (PAIR) e.carry_back_pair(); else e.carry_back(); d &= 0xffffu; }
            if constexpr (PAIR) e.w.put16_even_pair(d); else e.w.put16_even(d);
            sp -= 16;
        }
    }
    template <bool PAIR = false>
    __device__ void finish() {                                   // back to the reference's (low, range), then its finish()
        digits<PAIR>();
        if constexpr (PAIR) e.w.drain_pair();
        const uint32_t ex = uint32_t(15 - sp);                   // 1 .. 22
        uint64_t low = L2 << (ex - 1);
        if (low >= CabacEncoder::kOne) { e.carry_back(); low -= CabacEncoder::kOne; }
        e.low = uint32_t(low);
        e.range = R << ex;
        e.finish();
    }
};

// The same bin with the emitter BASELINE.json's north star names -- "wavefront ballots used for output-bit packing": a digit is not
// shifted into a register word but dropped into the lane's column of a staging area in LDS (16 slots of 64 lanes; slot-major, so
// the data-dependent slot index never makes a bank conflict), and after every eight bins ONE ballot tells whether any lane has a
// 16-byte row of eight digits ready; those lanes gather their row and store it whole.  Measured against the 8-byte register word
// of CabacEncoder on config 5: kept as a test-build variant (hook k1_emit_lds), not shipped.
struct alignas(8) Row16 { uint32_t x, y, z, w; };
struct CabacLaneS {
    CabacEncoder e;                                              // low / range; its byte writer takes over in finish()
    uint32_t *stage;                                             // this lane's column: slot j at stage[64 j]
    uint32_t cnt, flushed;                                       // digits produced / stored (flushed: a multiple of 8)

    __device__ __forceinline__ void init(uint8_t *out, uint32_t capacity, uint32_t *column) {
        e.init(0x7F800000u, out, capacity);
        stage = column; cnt = flushed = 0;
    }
    __device__ void carry() {                                    // arithmetic_code.h:154-157: into the staged digits, then the stored bytes
        for (uint32_t i = cnt; i > flushed;) {
            i--;
            const uint32_t d = (stage[64 * (i & 15u)] + 1u) & 0xffffu;
            stage[64 * (i & 15u)] = d;
            if (d) return;
        }
        uint32_t p = flushed * 2;
        if (p > e.w.cap) return;
        while (p > 0) {
            p--;
            const uint32_t b = uint32_t(e.w.base[p]) + 1u;
            e.w.base[p] = uint8_t(b);
            if (b <= 0xffu) break;
        }
    }
    __device__ __forceinline__ void bin(uint32_t rec, uint32_t off, const uint2 *tab, uint8_t *st_lane) {
        uint8_t *sp = st_lane + off;
        uint32_t s = *sp;
        asm volatile("" : "+v"(s));
        uint2 ent = tab[s];
        asm volatile("" : "+v"(ent.x), "+v"(ent.y));
        const int norm = 23 - __builtin_clz(e.range);
        const uint32_t q = (e.range >> (norm + 6)) & 3;
        const uint32_t r_tab = ((ent.x >> (q * 8)) & 0xffu) << norm;
        const uint32_t r1 = r_tab | ((e.range >> 1) & uint32_t(int32_t(ent.y) >> 31));
        const uint32_t sym = (rec ^ s) & 1 & ~(ent.y >> 30);
        const uint32_t r0 = e.range - r1;
        e.low += sym ? r0 : 0u;
        e.range = sym ? r1 : r0;
        *sp = uint8_t(ent.y >> (8 * sym));
        if (e.range < 0x200u) {                                  // arithmetic_code.h:115-122, the digit into the staging column
            if (__builtin_expect(e.low >= CabacEncoder::kOne, 0)) { carry(); e.low -= CabacEncoder::kOne; }
            stage[64 * (cnt & 15u)] = e.low >> 15;
            cnt++;
            e.low = (e.low & 0x7fffu) << 16;
            e.range <<= 16;
        }
    }
    __device__ __forceinline__ void rows() {                     // after every eight bins: at most 15 digits are staged at any time
        const bool ready = cnt - flushed >= 8u;
        if (__ballot(ready)) {
            if (ready) {
                uint32_t d[8];
#pragma unroll
                for (uint32_t j = 0; j < 8; j++) d[j] = stage[64 * ((flushed + j) & 15u)];
                Row16 v;                                         // digits most significant byte first (arithmetic_code.h:184-190)
                v.x = __builtin_bswap32(d[0] << 16 | d[1]); v.y = __builtin_bswap32(d[2] << 16 | d[3]);
                v.z = __builtin_bswap32(d[4] << 16 | d[5]); v.w = __builtin_bswap32(d[6] << 16 | d[7]);
                if (flushed * 2 + 16 <= e.w.cap) *reinterpret_cast<Row16 *>(e.w.base + flushed * 2) = v;
                flushed += 8;
            }
        }
    }
    __device__ void finish() {                                   // what is still staged goes through the byte writer, then finish() as it stands
        e.w.n = flushed * 2;
        e.w.acc = 0;
        for (uint32_t i = flushed; i < cnt; i++) e.w.put16_even(stage[64 * (i & 15u)]);
        e.finish();
    }
};

// The normalised form with CabacLaneS's emitter: the due digits go into the lane's staging column, whole 16-byte rows out
// by one ballot per eight bins.  The normalised form waits where the other one computes (62 % of its wave-cycles on config 5);
// if it waited for its own output -- 8-byte stores of a lane's own, ten times the bytes in 64-byte requests -- this would be
// the faster kernel.  It is not: 2.88 against 2.49 ms on config 5.  A measured variant (test hook k1_emit_lds=2).
struct CabacLaneNS : CabacLaneN {
    uint32_t *stage;                                             // this lane's column: slot j at stage[64 j]
    uint32_t cnt, flushed;                                       // digits produced / stored (flushed: a multiple of 8)

    __device__ __forceinline__ void init(uint8_t *out, uint32_t capacity, uint32_t *column) {
        CabacLaneN::init(out, capacity);
        stage = column; cnt = flushed = 0;
    }
    __device__ void carry() {                                    // arithmetic_code.h:154-157: into the staged digits, then the stored bytes
        for (uint32_t i = cnt; i > flushed;) {
            i--;
            const uint32_t d = (stage[64 * (i & 15u)] + 1u) & 0xffffu;
            stage[64 * (i & 15u)] = d;
            if (d) return;
        }
        uint32_t p = flushed * 2;
        if (p > e.w.cap) return;
        while (p > 0) {
            p--;
            const uint32_t b = uint32_t(e.w.base[p]) + 1u;
            e.w.base[p] = uint8_t(b);
            if (b <= 0xffu) break;
        }
    }
    __device__ __forceinline__ void digits() {                   // every digit that is due, oldest first (at most 5 per eight bins: 16 slots do)
        while (sp >= 15) {
            uint32_t d = uint32_t(L2 >> (sp + 1));
            L2 &= (uint64_t(2) << sp) - 1;
            if (__builtin_expect(d >> 16, 0)) { carry(); d &= 0xffffu; }
            stage[64 * (cnt & 15u)] = d;
            cnt++;
            sp -= 16;
        }
    }
    __device__ __forceinline__ void rows() {                     // after every eight bins
        const bool ready = cnt - flushed >= 8u;
        if (__ballot(ready)) {
            if (ready) {
                uint32_t d[8];
#pragma unroll
                for (uint32_t j = 0; j < 8; j++) d[j] = stage[64 * ((flushed + j) & 15u)];
                Row16 v;                                         // digits most significant byte first (arithmetic_code.h:184-190)
                v.x = __builtin_bswap32(d[0] << 16 | d[1]); v.y = __builtin_bswap32(d[2] << 16 | d[3]);
                v.z = __builtin_bswap32(d[4] << 16 | d[5]); v.w = __builtin_bswap32(d[6] << 16 | d[7]);
                if (flushed * 2 + 16 <= e.w.cap) *reinterpret_cast<Row16 *>(e.w.base + flushed * 2) = v;
                flushed += 8;
            }
        }
    }
    __device__ void finish() {                                   // back to the reference's (low, range); what is staged through the byte writer; its finish()
        digits();
        const uint32_t ex = uint32_t(15 - sp);                   // 1 .. 22
        uint64_t low = L2 << (ex - 1);
        if (low >= CabacEncoder::kOne) { carry(); low -= CabacEncoder::kOne; }
        e.low = uint32_t(low);
        e.range = R << ex;
        e.w.n = flushed * 2;
        e.w.acc = 0;
        for (uint32_t i = flushed; i < cnt; i++) e.w.put16_even(stage[64 * (i & 15u)]);
        e.finish();
    }
};

constexpr uint32_t kCensusStride = 16;                           // the one-lane-per-slice kernel renumbers from a 1-in-16 sample
constexpr uint32_t kK1Waves = 4;                                 // waves per workgroup (fewer when the state rows are large): they share the two tables
constexpr uint32_t kK1MaxWaves = 16;

// table / index: the dense renumbering of the batch's contexts (k_k1p_densemap), or null: contexts as the caller
// numbers them.  n_rows: contexts the kernel keeps states for (dense count, or n_states); init_states / final_states
// rows are n_states wide, in the caller's numbering.
// FORM: 1 = normalised form (CabacLaneN: shipped; its output in 16-byte stores), 5 = that with 8-byte stores, 4 = with four state bytes read ahead,
// 0 = the coder as cabac_code.h writes it (CabacLane), 2 = that with its digits
// staged in LDS (CabacLaneS), 3 = the normalised form with them (CabacLaneNS); 0, 2 and 3 are measured variants of the test build.
template <bool TILED, int FORM>
__global__ __launch_bounds__(64 * kK1MaxWaves) void k_cabac_encode(
    const void *recs, const uint64_t *off, const uint32_t *n_bins, const uint32_t *order,
    uint32_t n_slices, const uint8_t *init_states, uint32_t n_states, const uint16_t *table, const uint16_t *index, uint32_t n_rows,
    uint8_t *out, const uint64_t *out_off, uint32_t *out_len, int32_t *status,
    uint8_t *final_states, int32_t want_status) {
    extern __shared__ uint32_t lds[];                            // per wave: state dwords [(n_rows + 4 + 3) / 4][64]; FORM 2: then 16 x 64 staging slots per wave
    constexpr bool NORM = FORM == 1 || FORM == 3 || FORM == 4 || FORM == 5;           // 5: FORM 1 with 8-byte output stores (rounds 1-3)
    __shared__ uint2 tab[136];                                   // 128 states + pseudo-states 128..135
    __shared__ uint4 tabn[NORM ? 272 : 1];                       // the normalised form's: by (state, bin)
    __shared__ uint32_t sel_off[2048];                           // selector -> byte offset of its state in the lane's column (up to 256 rows of 256 bytes, + 3)

    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t rows4 = (n_rows + 4 + 3) >> 2;                // contexts, then four pseudo contexts
    uint32_t *st32 = lds + wv * rows4 * 64;
    uint8_t *st_lane = reinterpret_cast<uint8_t *>(st32) + lane * 4;
    const bool in_range = g < n_slices;
    const uint32_t slice = in_range ? (order ? order[g] : g) : 0;
    // Only slices whose status is `want_status` are coded: AVR_SLICE_OK in the normal case (a slice
    // flagged by the packer is skipped), AVR_SLICE_RETRY_SERIAL when slices are handed back (by K1p, or by
    // the renumbered launch of this kernel).  A workgroup without any leaves before it fills a table.
    int32_t st = in_range ? status[slice] : AVR_SLICE_OK;
    const bool active = in_range && st == want_status;
    if (!__syncthreads_or(active)) {
        if (in_range && want_status == AVR_SLICE_OK) out_len[slice] = 0;
        return;
    }
    for (uint32_t i = threadIdx.x; i < 128; i += blockDim.x)
        tab[i] = make_uint2(d_tables.packed[i][0], d_tables.packed[i][1]);
    if (threadIdx.x < 8) {                                       // 128: bypass, 130: terminate (LPS range 2), 131 / 133: after put_terminate(1), 132 / 134 / 135: no-op
        const uint32_t t = threadIdx.x, ps = 128 + t, next = t == 6 ? 135u : (t == 3 || t == 5) ? 133u : ps, lps = t == 2 ? 131u : next;
        tab[ps] = make_uint2(t == 2 ? 0x02020202u : 0u, (t == 0 ? 0x80000000u : t >= 3 ? 0x40000000u : 0u) | next | lps << 8);
    }
    if constexpr (NORM)
        for (uint32_t i = threadIdx.x; i < 272; i += blockDim.x) tabn[i] = CabacLaneN::norm_entry(i >> 1, i & 1u);
    for (uint32_t base = threadIdx.x; base < 2048; base += 8 * blockDim.x) {     // (eight entries a trip: their loads in flight together)
        uint32_t dense[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) {
            const uint32_t sel = base + u * blockDim.x;
            dense[u] = table ? uint32_t(table[sel & 1023u]) : (sel < n_states ? sel : 0xffffu);
        }
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) {
            const uint32_t sel = base + u * blockDim.x;
            if (sel >= 2048) break;
            uint32_t k;
            if (sel < 1024) k = dense[u] < n_rows ? dense[u] : sel < n_states ? n_rows + 3 : n_rows + 2;   // n_rows + 3: a context of the slice that has no row: the census missed it
            else k = sel == AVR_SEL_BYPASS ? n_rows : sel == AVR_SEL_TERMINATE ? n_rows + 1 : n_rows + 2;
            sel_off[sel] = ((k >> 2) << 8) + (k & 3);
        }
    }

    if (active) st = AVR_SLICE_OK;
    const uint32_t nb = active ? n_bins[slice] : 0;

    // states: global (n_states bytes per slice, caller's numbering) -> the LDS columns of the wave's lanes.  The wave takes its 64
    // slices one after the other, lane k fetching the slice's state of row k (r4: 64 independent loads of one slice's row each; the
    // lanes used to gather their own columns, row by row -- one load per row with 64 lines in it, each behind a load of index[k]).
    {
        for (uint32_t k4 = 0; k4 < rows4; k4++) {                // the pseudo contexts (and padding); the contexts' own bytes follow
            uint32_t v = 0;
            for (uint32_t b = 0; b < 4; b++) {
                const uint32_t k = 4 * k4 + b;
                const uint32_t sv = k < n_rows ? 0u : k == n_rows ? 128u : k == n_rows + 1 ? 130u : k == n_rows + 3 ? 134u : 132u;
                v |= sv << (8 * b);
            }
            st32[k4 * 64 + lane] = v;
        }
        uint8_t *st_wave = reinterpret_cast<uint8_t *>(st32);
        const uint64_t act = __ballot(active);
        for (uint32_t k0 = 0; k0 < n_rows; k0 += 64) {
            const uint32_t k = k0 + lane;
            const uint32_t col = k < n_rows ? (index ? uint32_t(index[k]) : k) : 0xffffu;   // 0xffff: a row beyond the batch's contexts (launch sized by a guess)
            uint8_t *dst = st_wave + ((k >> 2) << 8) + (k & 3);
            const uint32_t colc = col < n_states ? col : 0u;     // (every load is made, sixteen in flight: a lane without a slice has slice 0)
            for (uint32_t j0 = 0; j0 < 64; j0 += 16) {
                uint8_t v[16];
#pragma unroll
                for (uint32_t u = 0; u < 16; u++) {              // (the slice's row is a scalar address, the lane's column the offset)
                    const uint8_t *row = init_states + size_t(uint32_t(__builtin_amdgcn_readlane(int(slice), int(j0 + u)))) * n_states;
                    v[u] = row[colc];
                }
#pragma unroll
                for (uint32_t u = 0; u < 16; u++)
                    if (((act >> (j0 + u)) & 1u) && col < n_states) dst[(j0 + u) * 4] = v[u];
            }
        }
    }
    __syncthreads();

    typename std::conditional<FORM == 3, CabacLaneNS, typename std::conditional<FORM == 1 || FORM == 4 || FORM == 5, CabacLaneN,
        typename std::conditional<FORM == 2, CabacLaneS, CabacLane>::type>::type>::type L;
    const uint64_t o0 = in_range ? out_off[slice] : 0;
    const uint32_t cap = in_range ? uint32_t(out_off[slice + 1] - o0) : 0;
    if constexpr (FORM == 1 || FORM == 4 || FORM == 5) L.init(out + o0, cap);
    else if constexpr (FORM == 2 || FORM == 3) L.init(out + o0, cap, lds + (blockDim.x >> 6) * rows4 * 64 + wv * 1024 + lane);
    else L.e.init(0x7F800000u, out + o0, cap);                   // cabac_code.h:30

    const ChunkSource<TILED> src(recs, off, in_range ? g : 0, slice);
    const uint32_t n_chunks = (nb + 7) >> 3;
    const uint4 nop4 = make_uint4(AVR_NOP_CABAC2, AVR_NOP_CABAC2, AVR_NOP_CABAC2, AVR_NOP_CABAC2);
    // every load is unconditional (the index clamped to the slice's last chunk; what a clamped load returns is never coded): a load
    // inside a branch is waited for before the branch ends -- the "two chunks ahead" of rounds 1-3 waited out a memory latency per
    // eight bins (the same in k_range_encode since round 2; tools/ubench/read_patterns)
    const uint32_t last_chunk = n_chunks ? n_chunks - 1 : 0;
    // put_terminate(1) (cabac_code.h:63-65) ends the slice: in a well-formed stream it is the last record, and what follows it in its
    // 16-byte chunk is padding that changes nothing.  No bin is tested for it (r4; rounds 1-3 compared every record with it, three
    // instructions a bin): the terminate pseudo context's state remembers -- 130 -> 131 at the first put_terminate(1), 131 -> 133 at
    // any terminate bin after it -- and the lane looks at that byte and at its last record once, at the end.
    auto code8 = [&](const uint4 &v) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t offs[8];                                        // where the eight bins' states live: independent of the states, read ahead
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) offs[k] = sel_off[((w[k >> 1] >> ((k & 1) * 16)) >> 1) & 0x7ffu];
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) {
            const uint32_t rec = (w[k >> 1] >> ((k & 1) * 16)) & 0xffffu;
            if constexpr (FORM == 4) {
                if ((k & 3) == 0) {
                    const uint32_t r4[4] = {w[k >> 1] & 0xffffu, w[k >> 1] >> 16, w[(k >> 1) + 1] & 0xffffu, w[(k >> 1) + 1] >> 16};
                    const uint32_t o4[4] = {offs[k], offs[k + 1], offs[k + 2], offs[k + 3]};
                    L.bin4(r4, o4, tabn, st_lane);
                }
            } else
            if constexpr (NORM) L.bin(rec, offs[k], tabn, st_lane);
            else L.bin(rec, offs[k], tab, st_lane);
            if constexpr (FORM == 1) { if ((k & 3) == 3) L.template digits<true>(); }
            else if constexpr (NORM) { if ((k & 3) == 3) L.digits(); }
            if constexpr (FORM == 2 || FORM == 3) { if (k == 7) L.rows(); }
        }
    };
    // The records come two chunks ahead, marked non-temporal (read once: nothing of them need stay in L2).
    uint4 ca = nop4, cb = nop4;
    if (n_chunks) { ca = src.load_nt(0); cb = src.load_nt(min(1u, last_chunk)); }
    for (uint32_t c = 0; c < n_chunks; c++) {
        const uint4 nx2 = src.load_nt(min(c + 2, last_chunk));
        code8(ca);
        ca = cb;
        cb = nx2;
    }
    constexpr uint32_t kTerm1 = (AVR_SEL_TERMINATE << 1) | 1;
    bool after_finish = false;                                   // a bin after the put_terminate(1), which was finish()
    if (active && nb) {
        const uint32_t tstate = st_lane[(((n_rows + 1) >> 2) << 8) + ((n_rows + 1) & 3)];
        if (tstate != 130u) {
            const uint4 v = src.load(last_chunk);
            const uint32_t i = (nb - 1) & 7u, d = i < 2 ? v.x : i < 4 ? v.y : i < 6 ? v.z : v.w;
            after_finish = tstate != 131u || ((d >> ((i & 1u) * 16)) & 0xffffu) != kTerm1;
        }
    }
    if (in_range) {
        if (active) {
            const bool missed = st_lane[(((n_rows + 3) >> 2) << 8) + ((n_rows + 3) & 3)] == 135u;
            if (after_finish) st = AVR_SLICE_BAD_RECORD;
            else if constexpr (FORM == 1) L.template finish<true>();
            else if constexpr (FORM != 0) L.finish();
            else L.e.finish();                                   // cabac_code.h:63-65 / ~encoder(), arithmetic_code.h:100
            L.e.w.flush();
            if (st == AVR_SLICE_OK && L.e.w.n > cap) st = AVR_SLICE_OVERFLOW;
            if (missed) { st = AVR_SLICE_RETRY_SERIAL; L.e.w.n = 0; }   // whatever else: the bytes are not the slice's
        }
        if (active) { out_len[slice] = L.e.w.n; status[slice] = st; }
        else if (want_status == AVR_SLICE_OK) out_len[slice] = 0;
    }
    if (final_states) {                                          // (final_states starts out as a copy of init_states when a renumbering is in use)
        const uint64_t done = __ballot(in_range && active && st != AVR_SLICE_RETRY_SERIAL);
        const uint8_t *st_wave = reinterpret_cast<const uint8_t *>(st32);
        for (uint32_t k0 = 0; k0 < n_rows; k0 += 64) {           // slice by slice, lane k the state of row k: see the way in
            const uint32_t k = k0 + lane;
            const uint32_t col = k < n_rows ? (index ? uint32_t(index[k]) : k) : 0xffffu;
            const uint8_t *from = st_wave + ((k >> 2) << 8) + (k & 3);
#pragma unroll 8
            for (uint32_t j = 0; j < 64; j++) {
                uint8_t *row = final_states + size_t(uint32_t(__builtin_amdgcn_readlane(int(slice), int(j)))) * n_states;
                if (((done >> j) & 1u) && col < n_states) row[col] = from[j * 4];
            }
        }
    }
}

// Which contexts the records of a tile / a slice use (one bit per selector < 1024): the census behind the dense
// renumbering, for the one-lane-per-slice kernel.  One workgroup per kCensusTiles x 64 slices; plain LDS flag stores, then the
// global words are only touched while a bit is still missing (see k_k1p_census).
//
// `stride` > 1: a sample -- every stride-th 16-byte chunk of each slice (a tile's chunk rows r with r % stride ==
// tile % stride).  The renumbering only has to hold the contexts that matter for the LDS footprint; a slice with a
// bin in a context the sample missed is handed back by k_cabac_encode and coded by the launch without renumbering.
constexpr uint32_t kCensusTiles = 8;                              // tiles (of 64 slices) to a workgroup
template <bool TILED>
__global__ __launch_bounds__(256) void k_k1_census(const void *recs, const uint64_t *off, const uint32_t *n_bins, const uint32_t *order,
                                                   uint32_t n_slices, uint32_t *used, uint32_t stride) {
    __shared__ uint8_t flag[1024];
    __shared__ uint32_t bm[32];
    reinterpret_cast<uint32_t *>(flag)[threadIdx.x] = 0;
    if (threadIdx.x < 32) bm[threadIdx.x] = 0;
    __syncthreads();
    auto take = [&](const uint4 &v) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) {
            const uint32_t sel = ((w[j >> 1] >> ((j & 1) * 16)) >> 1) & 0x7ffu;
            if (sel < 1024) flag[sel] = 1;
        }
    };
    const uint32_t n_tiles = (n_slices + 63) / 64;
    for (uint32_t tile = blockIdx.x * kCensusTiles; tile < n_tiles && tile < (blockIdx.x + 1) * kCensusTiles; tile++) {
        if (TILED) {                                             // the tile is one contiguous run of 16-byte chunks
            const uint4 *p = reinterpret_cast<const uint4 *>(recs) + off[tile];
            const uint64_t rows = (off[tile + 1] - off[tile]) >> 6;              // 64 chunks (one per lane) to a row
            for (uint64_t r = tile % stride + uint64_t(stride) * (threadIdx.x >> 6); r < rows; r += 4 * stride) take(p[r * 64 + (threadIdx.x & 63)]);
        } else {
            for (uint32_t l = 0; l < 64; l++) {
                const uint32_t g = tile * 64 + l;
                if (g >= n_slices) break;
                const uint32_t slice = order ? order[g] : g;
                const uint4 *p = reinterpret_cast<const uint4 *>(reinterpret_cast<const uint16_t *>(recs) + off[slice]);
                const uint32_t n = (n_bins[slice] + 7) >> 3;
                for (uint32_t i = (tile + l) % stride + stride * threadIdx.x; i < n; i += 256 * stride) take(p[i]);
            }
        }
    }
    __syncthreads();
    const uint32_t f = reinterpret_cast<const uint32_t *>(flag)[threadIdx.x];
    const uint32_t nib = (f & 1u) | ((f >> 7) & 2u) | ((f >> 14) & 4u) | ((f >> 21) & 8u);
    if (nib) atomicOr(&bm[threadIdx.x >> 3], nib << ((threadIdx.x & 7) * 4));
    __syncthreads();
    if (threadIdx.x < 32) {
        const uint32_t mine = bm[threadIdx.x];
        if (mine & ~__hip_atomic_load(&used[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(&used[threadIdx.x], mine);
    }
}

// ------------------------------------------------------------------ K2

// A deeper read-ahead than two chunks was measured and buys nothing even at one wave per SIMD: the bin-to-bin
// dependency chain, not HBM latency, is the time.  (Round 2's measurement variants -- the integer long division above,
// the range recurrence alone -- are gone from the library: their numbers are in DESIGN.md section 4.)
template <bool TILED>
__global__ __launch_bounds__(64) void k_range_encode(
    const void *recs, const uint64_t *off, const uint32_t *n_bins, const uint32_t *order,
    uint32_t n_slices, uint8_t *out, const uint64_t *out_off, uint32_t *out_len, int32_t *status) {
    __shared__ double inv_d[256];                                // fl(1 / d)
    const uint32_t lane = threadIdx.x;
    for (uint32_t d = lane; d < 256; d += 64) inv_d[d] = d ? 1.0 / double(d) : 0.0;
    __syncthreads();
    const uint32_t g = blockIdx.x * 64 + lane;
    if (g >= n_slices) return;
    const uint32_t slice = order ? order[g] : g;
    int32_t st = status[slice];
    const bool active = st == AVR_SLICE_OK;
    const uint32_t nb = active ? n_bins[slice] : 0;

    RangeEncoder64 e;
    const uint64_t o0 = out_off[slice];
    const uint32_t cap = uint32_t(out_off[slice + 1] - o0);
    e.init(uint64_t(1) << 63, out + o0, cap);                    // arithmetic_code.h:96-97

    const ChunkSource<TILED> src(recs, off, g, slice);
    const uint32_t n_chunks = (nb + 7) >> 3;
    bool dead = false;
    // The operands of a chunk's eight bins (bin, pos, 1/total from LDS) depend on the records alone: they are all
    // fetched before the first bin is coded, so no LDS latency sits on the range -> range chain.
    auto code_chunk = [&](const uint4 cur) {
        const uint32_t w[4] = {cur.x, cur.y, cur.z, cur.w};
        uint32_t bin[8], pos[8], tot[8];
        double inv[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t rec = (w[k >> 1] >> (16 * (k & 1))) & 0xffffu;
            pos[k] = (rec >> 1) & 0x7f;
            tot[k] = pos[k] + ((rec >> 8) & 0x7f);               // recode.cpp:825
            bin[k] = tot[k] ? rec & 1 : 0;                       // total 0: a no-op (padding) record, whatever its bin
            inv[k] = inv_d[tot[k]];                              // 0 for total 0: quotient 0, range unchanged
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint64_t quot = div_u64_small_f64(e.range, double(tot[k]), inv[k]);
            const uint64_t r1 = quot * pos[k];                   // recode.cpp:826
            const uint64_t r0 = e.range - r1;                    // arithmetic_code.h:108
            e.low += bin[k] ? r0 : 0;
            e.range = bin[k] ? r1 : r0;
            if (e.range < (uint64_t(1) << 51)) {                 // min_range, arithmetic_code.h:61-62,115
                if (e.range == 0) { st = AVR_SLICE_ZERO_PROB; dead = true; }           // :116-118 (range stays 0 from here on: 0 / total * pos,
                                                                                     // nothing more is emitted; the chunk loop below ends the walk)
                else do e.emit_digit(); while (e.range < (uint64_t(1) << 55));         // :120-122
            }
        }
    };
    // every load is unconditional (the index is clamped to the slice's last chunk; what a clamped load returns is never
    // coded), so the compiler keeps counted waits instead of draining the queue at a branch
    const uint32_t last = n_chunks ? n_chunks - 1 : 0;
    uint4 cur = make_uint4(0, 0, 0, 0), nx1 = cur;
    if (n_chunks) { cur = src.load(0); nx1 = src.load(min(1u, last)); }
    for (uint32_t c = 0; c < n_chunks && !dead; c++) {
        const uint4 nx2 = src.load(min(c + 2, last));
        code_chunk(cur);
        cur = nx1;
        nx1 = nx2;
    }
    if (active) {
        if (st == AVR_SLICE_OK) e.finish();                      // recode.cpp:1100
        e.w.flush();
        if (st == AVR_SLICE_OK && e.w.n > cap) st = AVR_SLICE_OVERFLOW;
    }
    out_len[slice] = active ? e.w.n : 0;
    status[slice] = st;
}

// ------------------------------------------------------------------ pack: slice-major -> tiles

// One workgroup (64 lanes) per tile.  Lane l copies the chunks of its slice; a wave-wide
// store instruction writes 1 KiB contiguous.  Records past a slice's end become no-op records.
// Every record passes through here exactly once, so this is also where selectors are
// validated: a slice with a record the coders cannot take gets status AVR_SLICE_BAD_RECORD and
// is skipped by the encode kernel (status must be zeroed by the caller beforehand).
__global__ __launch_bounds__(64) void k_pack_tiles(
    int kind, uint32_t n_states, const uint16_t *recs, const uint64_t *rec_off, const uint32_t *n_bins,
    const uint32_t *order, uint32_t n_slices, const uint64_t *tile_off, uint4 *tiles, int32_t *status) {
    const uint32_t lane = threadIdx.x, t = blockIdx.x;
    const uint32_t g = t * 64 + lane;
    const bool active = g < n_slices;
    const uint32_t slice = active ? (order ? order[g] : g) : 0;
    const uint32_t nb = active ? n_bins[slice] : 0;
    const uint32_t my_chunks = (nb + 7) >> 3;
    const uint32_t tile_chunks = uint32_t((tile_off[t + 1] - tile_off[t]) >> 6);
    const uint4 *src = reinterpret_cast<const uint4 *>(recs + (active ? rec_off[slice] : 0));
    uint4 *dst = tiles + tile_off[t] + lane;
    const uint32_t nop = kind == AVR_KIND_CABAC ? AVR_NOP_CABAC : AVR_NOP_RANGE;
    const uint32_t nop2 = nop | (nop << 16);
    bool bad = false;
    for (uint32_t c = 0; c < tile_chunks; c++) {
        uint4 v = make_uint4(nop2, nop2, nop2, nop2);
        if (c < my_chunks) {
            v = src[c];
            const uint32_t valid = nb - c * 8;                   // records valid in this chunk (>= 1)
            uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) {
                uint32_t r = (w[k >> 1] >> ((k & 1) * 16)) & 0xffffu;
                if (k >= valid) {
                    r = nop;
                } else if (kind == AVR_KIND_CABAC) {
                    const uint32_t sel = r >> 1;                 // bit 12..15 must be clear too
                    bad |= !(sel < n_states || sel == AVR_SEL_BYPASS || sel == AVR_SEL_TERMINATE);
                } else {
                    bad |= (r & 0x8000u) || ((r >> 1) & 0x7f) + ((r >> 8) & 0x7f) == 0;
                }
                w[k >> 1] = (k & 1) ? ((w[k >> 1] & 0xffffu) | (r << 16)) : ((w[k >> 1] & 0xffff0000u) | r);
            }
            v = make_uint4(w[0], w[1], w[2], w[3]);
        }
        dst[size_t(c) * 64] = v;
    }
    if (active && bad) status[slice] = AVR_SLICE_BAD_RECORD;
}

// ------------------------------------------------------------------ dense context ids

// A slice's contexts are identified by their offset in libavcodec's cabac_state[1024]
// (recode.cpp:325 keys on the address), but a stream touches far fewer: state bytes cost LDS
// (64 lanes x n_states per wave), so a batch is renumbered onto the contexts it uses.
// census: which selectors < 1024 occur in a flat array of CABAC records (tiles or slice-major).
__global__ __launch_bounds__(256) void k_context_census(const uint16_t *recs, uint64_t n, uint32_t *bitmap) {
    __shared__ uint32_t bm[32];
    if (threadIdx.x < 32) bm[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t stride = uint64_t(gridDim.x) * 256 * 8;
    for (uint64_t i = (uint64_t(blockIdx.x) * 256 + threadIdx.x) * 8; i < n; i += stride) {
        const uint4 v = *reinterpret_cast<const uint4 *>(recs + i);       // n is a multiple of 8 (whole chunks)
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t sel = ((w[j >> 1] >> ((j & 1) * 16)) >> 1) & 0x7ffu;
            if (sel < 1024) atomicOr(&bm[sel >> 5], 1u << (sel & 31));
        }
    }
    __syncthreads();
    if (threadIdx.x < 32 && bm[threadIdx.x]) atomicOr(&bitmap[threadIdx.x], bm[threadIdx.x]);
}

// remap: selector s < 1024 -> table[s] (dense id), other selectors unchanged; in place.
__global__ __launch_bounds__(256) void k_context_remap(uint16_t *recs, uint64_t n, const uint16_t *table) {
    __shared__ uint16_t t[1024];
    for (uint32_t i = threadIdx.x; i < 1024; i += 256) t[i] = table[i];
    __syncthreads();
    const uint64_t stride = uint64_t(gridDim.x) * 256 * 8;
    for (uint64_t i = (uint64_t(blockIdx.x) * 256 + threadIdx.x) * 8; i < n; i += stride) {
        uint4 v = *reinterpret_cast<uint4 *>(recs + i);
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t sh = (j & 1) * 16;
            const uint32_t rec = (w[j >> 1] >> sh) & 0xffffu, sel = (rec >> 1) & 0x7ffu;
            const uint32_t out = sel < 1024 ? ((rec & 1) | (uint32_t(t[sel]) << 1)) : rec;
            w[j >> 1] = (w[j >> 1] & ~(0xffffu << sh)) | (out << sh);
        }
        *reinterpret_cast<uint4 *>(recs + i) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// dst[slice][j] = src[slice][index[j]] (gather, n_dst per slice) or dst[slice][index[j]] = src[slice][j] (scatter)
__global__ __launch_bounds__(256) void k_states_permute(const uint8_t *src, uint32_t n_src, uint8_t *dst, uint32_t n_dst,
                                                        const uint16_t *index, uint32_t n_index, uint64_t n_slices, int scatter) {
    const uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n_slices * n_index) return;
    const uint64_t s = i / n_index;
    const uint32_t j = uint32_t(i % n_index);
    if (scatter) dst[s * n_dst + index[j]] = src[s * n_src + j];
    else dst[s * n_dst + j] = src[s * n_src + index[j]];
}

// ------------------------------------------------------------------ compact: per-slice regions -> dense

// One workgroup per slice; the copy is contiguous on both sides.  dense_off[i+1]-dense_off[i]
// is min(out_len[i], capacity) as computed by the host.
__global__ __launch_bounds__(64) void k_compact(const uint8_t *out, const uint64_t *out_off, const uint64_t *dense_off,
                                                uint32_t n_slices, uint8_t *dense) {
    const uint32_t i = blockIdx.x;
    if (i >= n_slices) return;
    const uint8_t *src = out + out_off[i];
    uint8_t *dst = dense + dense_off[i];
    const uint32_t len = uint32_t(dense_off[i + 1] - dense_off[i]);
    for (uint32_t k = threadIdx.x; k < len; k += 64) dst[k] = src[k];
}

// ------------------------------------------------------------------ synthetic streams

struct TileRecordSink {               // gathers 8 records, stores one 16-byte chunk
    uint4 *dst;                       // chunk 0 of this lane
    uint32_t stride;                  // chunks between two of its chunks: 64 in a tile, 1 slice-major
    uint32_t n;
    uint32_t nop2;                    // two no-op records (padding)
    uint32_t w[4];
    __device__ TileRecordSink(uint4 *d, uint32_t nop, uint32_t stride_ = 64) : dst(d), stride(stride_), n(0), nop2(nop | (nop << 16)) {
        w[0] = w[1] = w[2] = w[3] = nop2;
    }
    __device__ void put_record(uint16_t rec) {
        const uint32_t j = n & 7, sh = (j & 1) * 16;
        w[j >> 1] = (w[j >> 1] & ~(0xffffu << sh)) | (uint32_t(rec) << sh);
        n++;
        if ((n & 7) == 0) {
            dst[size_t((n >> 3) - 1) * stride] = make_uint4(w[0], w[1], w[2], w[3]);
            w[0] = w[1] = w[2] = w[3] = nop2;
        }
    }
    __device__ void flush(uint32_t tile_chunks) {
        uint32_t c = n >> 3;
        if (n & 7) { dst[size_t(c) * stride] = make_uint4(w[0], w[1], w[2], w[3]); c++; }
        for (; c < tile_chunks; c++) dst[size_t(c) * stride] = make_uint4(nop2, nop2, nop2, nop2);
    }
};

__global__ __launch_bounds__(64) void k_synth_count(
    int workload, uint32_t scale, uint64_t seed, uint64_t first_slice, int kind,
    uint32_t n_slices, uint32_t *n_bins) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n_slices) return;
    CountSink cs;
    CabacSink<CountSink> sink(cs);       // K2 streams have the same length as K1 streams
    synth_slice(workload, scale, seed, first_slice + i, sink);
    (void)kind;
    n_bins[i] = cs.n;
}

__global__ __launch_bounds__(64) void k_synth_tiles(
    int workload, uint32_t scale, uint64_t seed, uint64_t first_slice, int kind,
    uint32_t n_slices, const uint32_t *order, const uint64_t *tile_off, uint4 *tiles,
    uint8_t *init_states, uint32_t n_states) {
    const uint32_t lane = threadIdx.x, t = blockIdx.x;
    const uint32_t g = t * 64 + lane;
    const uint32_t tile_chunks = uint32_t((tile_off[t + 1] - tile_off[t]) >> 6);
    TileRecordSink rs(tiles + tile_off[t] + lane, kind == AVR_KIND_CABAC ? AVR_NOP_CABAC : AVR_NOP_RANGE);
    if (g < n_slices) {
        const uint32_t slice = order ? order[g] : g;
        if (kind == AVR_KIND_CABAC) {
            CabacSink<TileRecordSink> sink(rs);
            synth_slice(workload, scale, seed, first_slice + slice, sink);
            if (init_states)
                for (uint32_t c = 0; c < n_states; c++)
                    init_states[size_t(slice) * n_states + c] = synth_init_state(c, seed, first_slice + slice);
        } else {
            ModelSink<TileRecordSink> sink(rs);
            synth_slice(workload, scale, seed, first_slice + slice, sink);
        }
    }
    rs.flush(tile_chunks);
}

__global__ __launch_bounds__(64) void k_synth_slices(
    int workload, uint32_t scale, uint64_t seed, uint64_t first_slice, int kind,
    uint32_t n_slices, const uint64_t *rec_off, uint16_t *recs, uint8_t *init_states, uint32_t n_states) {
    const uint32_t slice = blockIdx.x * 64 + threadIdx.x;
    if (slice >= n_slices) return;
    const uint32_t chunks = uint32_t((rec_off[slice + 1] - rec_off[slice]) >> 3);
    TileRecordSink rs(reinterpret_cast<uint4 *>(recs + rec_off[slice]), kind == AVR_KIND_CABAC ? AVR_NOP_CABAC : AVR_NOP_RANGE, 1);
    if (kind == AVR_KIND_CABAC) {
        CabacSink<TileRecordSink> sink(rs);
        synth_slice(workload, scale, seed, first_slice + slice, sink);
        if (init_states)
            for (uint32_t c = 0; c < n_states; c++)
                init_states[size_t(slice) * n_states + c] = synth_init_state(c, seed, first_slice + slice);
    } else {
        ModelSink<TileRecordSink> sink(rs);
        synth_slice(workload, scale, seed, first_slice + slice, sink);
    }
    rs.flush(chunks);
}

// One-byte K1 records (AVR_KIND_CABAC8: bin | dense selector << 1; include/avrecode_ms_amd.h) widened into the two-byte records every K1
// kernel reads -- what the batch API does with a batch that came over PCIe in half the bytes.  One thread per 8 records: 8 bytes in,
// 16 out, both coalesced; a slice's records past its n_bins (the staging buffer pads every slice to 8) become no-ops.  Slice i's bytes
// are at byte rec_off[i] of `in`, its records at record rec_off[i] of `out`: the same offsets, a multiple of 8 each.
__global__ __launch_bounds__(256) void k_expand_records8(const uint8_t *in, const uint64_t *rec_off, const uint32_t *n_bins, uint32_t n_slices,
                                                         uint32_t n_states, uint64_t total, uint16_t *out) {
    const uint64_t g = (uint64_t(blockIdx.x) * 256 + threadIdx.x) * 8;           // first of this thread's eight records
    if (g >= total) return;
    uint32_t lo = 0, hi = n_slices;                              // the slice g lies in: the last i with rec_off[i] <= g
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (rec_off[mid] <= g) lo = mid; else hi = mid; }
    const uint64_t first = g - rec_off[lo];
    const uint32_t n = n_bins[lo];
    const uint2 v = *reinterpret_cast<const uint2 *>(in + g);
    const uint32_t w[2] = {v.x, v.y};
    uint32_t o[4];
#pragma unroll
    for (uint32_t j = 0; j < 8; j++) {
        const uint32_t b = (w[j >> 2] >> (8 * (j & 3))) & 0xffu, sel8 = b >> 1;
        uint32_t sel = sel8 == AVR_SEL8_BYPASS ? AVR_SEL_BYPASS : sel8 == AVR_SEL8_TERMINATE ? AVR_SEL_TERMINATE : sel8 < n_states ? sel8 : 1027u;
        uint32_t rec = (sel << 1) | (b & 1u);
        if (first + j >= n) rec = AVR_NOP_CABAC;
        if (j & 1) o[j >> 1] |= rec << 16; else o[j >> 1] = rec;
    }
    *reinterpret_cast<uint4 *>(out + g) = make_uint4(o[0], o[1], o[2], o[3]);
}

// ------------------------------------------------------------------ launchers

// The renumbering's scratch (4.25 KiB: used[32] + n_dense | table[1024] | index[1024]): one per (device, stream), made on
// first use and kept -- work on one stream is ordered, so consecutive calls may share it, and no call allocates.
// (One thread per stream: the census of a call and the kernel that reads its table are separate launches with a 4-byte wait between
// them, so a second host thread enqueueing on the SAME stream in that window would overwrite the table -- as include/avrecode_ms_amd.h
// says of every device-resident entry point, a stream is used by one thread at a time.)
namespace {
struct ScratchSlot { int dev; hipStream_t s; uint8_t *p; };
std::vector<ScratchSlot> g_scratch;
std::mutex g_scratch_mu;
}  // namespace
static hipError_t stream_scratch(hipStream_t s, uint8_t **out) {
    using Slot = ScratchSlot;
    std::vector<Slot> &slots = g_scratch;
    std::mutex &mu = g_scratch_mu;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    for (const Slot &x : slots)
        if (x.dev == dev && x.s == s) { *out = x.p; return hipSuccess; }
    uint8_t *p = nullptr;
    if ((e = hipMalloc(reinterpret_cast<void **>(&p), 256 + 4096)) != hipSuccess) return e;
    slots.push_back(Slot{dev, s, p});
    *out = p;
    return hipSuccess;
}

void forget_side_stream(hipStream_t s);                          // avr_k2p.hip
// A stream is about to be destroyed (avr_batch_destroy): the resources kept per (device, stream) go with it, so that a process that
// makes and destroys batches (one per file, say) does not collect them.
void forget_stream(hipStream_t s) {
    {
        std::lock_guard<std::mutex> lock(g_scratch_mu);
        for (size_t i = 0; i < g_scratch.size();)
            if (g_scratch[i].s == s) { (void)hipFree(g_scratch[i].p); g_scratch.erase(g_scratch.begin() + long(i)); } else i++;
    }
    forget_side_stream(s);
}

// The one-lane-per-slice kernel keeps 64 x (contexts) state bytes in LDS per wave, so it renumbers the batch onto the
// contexts its records use (census -> k_k1p_densemap -> one 4-byte read-back to size the launch), applied to the
// records as they are loaded (sel_off): nothing is rewritten, init_states / final_states stay in the caller's numbering.
// dense = false (and every hand-over from K1p, want_status != OK): the caller's numbering as it is.
hipError_t launch_cabac_encode(bool tiled, hipStream_t s, const void *recs, const uint64_t *off,
                               const uint32_t *n_bins, const uint32_t *order, uint32_t n_slices,
                               const uint8_t *init_states, uint32_t n_states, uint8_t *out,
                               const uint64_t *out_off, uint32_t *out_len, int32_t *status,
                               uint8_t *final_states, int32_t want_status, bool dense, const DenseHint *hint) {
    if (n_slices == 0) return hipSuccess;
    hipError_t err;
    uint32_t n_rows = n_states;
    uint8_t *scratch = nullptr;                                  // used[32] + n_dense | table[1024] | index[1024]
    const uint16_t *table = nullptr, *index = nullptr;
    bool retry = false;                                          // the census was a sample: a second launch takes what it missed
    if (dense && want_status == AVR_SLICE_OK && n_states > 8 && !no_dense()) {
        if ((err = stream_scratch(s, &scratch)) != hipSuccess) return err;
        uint32_t *used = reinterpret_cast<uint32_t *>(scratch);
        uint16_t *t = reinterpret_cast<uint16_t *>(scratch + 256);
        if ((err = hipMemsetAsync(used, 0, 256, s)) != hipSuccess) return err;
        const dim3 cgrid(((n_slices + 63) / 64 + kCensusTiles - 1) / kCensusTiles);
        const uint32_t stride = test_hooks().census_stride ? test_hooks().census_stride : kCensusStride;
        retry = stride > 1;
        if (tiled) hipLaunchKernelGGL(k_k1_census<true>, cgrid, dim3(256), 0, s, recs, off, n_bins, order, n_slices, used, stride);
        else hipLaunchKernelGGL(k_k1_census<false>, cgrid, dim3(256), 0, s, recs, off, n_bins, order, n_slices, used, stride);
        if ((err = launch_densemap(s, used, t, t + 1024, used + 32)) != hipSuccess) return err;
        uint32_t nd = 0;
        if (hint && hint->rows) {                                // sized by the caller's guess: nothing waits (see DenseHint)
            nd = hint->rows;
            retry = true;
            if (hint->host_count && (err = hipMemcpyAsync(hint->host_count, used + 32, 4, hipMemcpyDeviceToHost, s)) != hipSuccess) return err;
        } else {
            if ((err = hipMemcpyAsync(&nd, used + 32, 4, hipMemcpyDeviceToHost, s)) != hipSuccess) return err;
            if ((err = hipStreamSynchronize(s)) != hipSuccess) return err;
            if (hint && hint->host_count) *hint->host_count = nd;
        }
        // a selector >= n_states is not a context of the slice even if it occurs: the table only serves selectors < n_states
        // (k_pack_tiles has flagged such records; the kernel sends them to the no-op row either way, see sel_off)
        n_rows = nd < n_states ? nd : n_states;
        table = t;
        index = t + 1024;
        if (final_states && (err = hipMemcpyAsync(final_states, init_states, size_t(n_slices) * n_states, hipMemcpyDeviceToDevice, s)) != hipSuccess)
            return err;                                          // contexts without bins keep their state
    }
    auto launch = [&](uint32_t rows, const uint16_t *tb, const uint16_t *ix, int32_t want) -> hipError_t {
        const uint32_t per_wave = ((rows + 4 + 3) / 4) * 256 + (test_hooks().k1_emit_lds ? 4096u : 0u);     // (+ the staging slots of FORMs 2, 3)
        uint32_t want_waves = test_hooks().k1_waves ? test_hooks().k1_waves : kK1Waves;
        if (want_waves > kK1MaxWaves) want_waves = kK1MaxWaves;
        const uint32_t waves = per_wave * want_waves <= 60 * 1024 ? want_waves : per_wave * 4 <= 60 * 1024 ? 4 : per_wave * 2 <= 60 * 1024 ? 2 : 1;
        const uint32_t lds = waves * per_wave;
        const dim3 grid((n_slices + 64 * waves - 1) / (64 * waves)), block(64 * waves);
        // Shipped: the coder in normalised form with the digits taken every fourth bin, in step across the wave (CabacLaneN) --
        // since round 3's table entries (avr_k1p.h, CodeEntryC) 31 VALU instructions a bin against the 48 of the form that reads
        // like cabac_code.h, and 2.44 against 2.55 ms per step on config 5.  Test hooks: k1_form_ref = that form (CabacLane),
        // k1_emit_lds = 1: it with the digits staged in LDS (CabacLaneS), 2: the shipped form with them (CabacLaneNS); same bytes all.
        const int form = test_hooks().k1_emit_lds ? (test_hooks().k1_emit_lds == 2 ? 3 : 2) : test_hooks().k1_form_ref ? 0 : test_hooks().k1_fwd ? 4 : test_hooks().k1_words8 ? 5 : 1;
        auto kern = tiled ? (form == 5 ? k_cabac_encode<true, 5> : form == 4 ? k_cabac_encode<true, 4> : form == 1 ? k_cabac_encode<true, 1> : form == 2 ? k_cabac_encode<true, 2> : form == 3 ? k_cabac_encode<true, 3> : k_cabac_encode<true, 0>)
                          : (form == 5 ? k_cabac_encode<false, 5> : form == 4 ? k_cabac_encode<false, 4> : form == 1 ? k_cabac_encode<false, 1> : form == 2 ? k_cabac_encode<false, 2> : form == 3 ? k_cabac_encode<false, 3> : k_cabac_encode<false, 0>);
        if (lds > 48 * 1024) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, grid, block, lds, s, recs, off, n_bins, order, n_slices, init_states, n_states, tb, ix, rows, out,
                           out_off, out_len, status, final_states, want);
        return hipGetLastError();
    };
    err = launch(n_rows, table, index, want_status);
    if (err == hipSuccess && retry) err = launch(n_states, nullptr, nullptr, AVR_SLICE_RETRY_SERIAL);
    return err;
}

hipError_t launch_range_encode(bool tiled, hipStream_t s, const void *recs, const uint64_t *off,
                               const uint32_t *n_bins, const uint32_t *order, uint32_t n_slices,
                               uint8_t *out, const uint64_t *out_off, uint32_t *out_len,
                               int32_t *status) {
    if (n_slices == 0) return hipSuccess;
    const dim3 grid((n_slices + 63) / 64), block(64);
    if (tiled) hipLaunchKernelGGL(k_range_encode<true>, grid, block, 0, s, recs, off, n_bins, order, n_slices, out, out_off, out_len, status);
    else hipLaunchKernelGGL(k_range_encode<false>, grid, block, 0, s, recs, off, n_bins, order, n_slices, out, out_off, out_len, status);
    return hipGetLastError();
}

hipError_t launch_pack_tiles(hipStream_t s, int kind, uint32_t n_states, const uint16_t *recs,
                             const uint64_t *rec_off, const uint32_t *n_bins, const uint32_t *order,
                             uint32_t n_slices, const uint64_t *tile_off, void *tiles, int32_t *status) {
    if (n_slices == 0) return hipSuccess;
    const dim3 grid((n_slices + 63) / 64), block(64);
    hipLaunchKernelGGL(k_pack_tiles, grid, block, 0, s, kind, n_states, recs, rec_off, n_bins, order, n_slices,
                       tile_off, reinterpret_cast<uint4 *>(tiles), status);
    return hipGetLastError();
}

hipError_t launch_context_census(hipStream_t s, const uint16_t *recs, uint64_t n, uint32_t *bitmap) {
    if (n == 0) return hipSuccess;
    const uint64_t want = (n / 8 + 255) / 256;
    hipLaunchKernelGGL(k_context_census, dim3(uint32_t(want < 4096 ? want : 4096)), dim3(256), 0, s, recs, n, bitmap);
    return hipGetLastError();
}

hipError_t launch_context_remap(hipStream_t s, uint16_t *recs, uint64_t n, const uint16_t *table) {
    if (n == 0) return hipSuccess;
    const uint64_t want = (n / 8 + 255) / 256;
    hipLaunchKernelGGL(k_context_remap, dim3(uint32_t(want < 8192 ? want : 8192)), dim3(256), 0, s, recs, n, table);
    return hipGetLastError();
}

hipError_t launch_states_permute(hipStream_t s, const uint8_t *src, uint32_t n_src, uint8_t *dst, uint32_t n_dst,
                                 const uint16_t *index, uint32_t n_index, uint64_t n_slices, int scatter) {
    const uint64_t total = n_slices * n_index;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(k_states_permute, dim3(uint32_t((total + 255) / 256)), dim3(256), 0, s, src, n_src, dst, n_dst, index,
                       n_index, n_slices, scatter);
    return hipGetLastError();
}

hipError_t launch_compact(hipStream_t s, const uint8_t *out, const uint64_t *out_off, const uint32_t *out_len,
                          const uint64_t *dense_off, uint32_t n_slices, uint8_t *dense) {
    (void)out_len;
    if (n_slices == 0) return hipSuccess;
    hipLaunchKernelGGL(k_compact, dim3(n_slices), dim3(64), 0, s, out, out_off, dense_off, n_slices, dense);
    return hipGetLastError();
}

hipError_t launch_synth_count(hipStream_t s, int workload, uint32_t scale, uint64_t seed,
                              uint64_t first_slice, int kind, uint32_t n_slices, uint32_t *n_bins) {
    if (n_slices == 0) return hipSuccess;
    hipLaunchKernelGGL(k_synth_count, dim3((n_slices + 63) / 64), dim3(64), 0, s, workload, scale, seed,
                       first_slice, kind, n_slices, n_bins);
    return hipGetLastError();
}

hipError_t launch_synth_slices(hipStream_t s, int workload, uint32_t scale, uint64_t seed, uint64_t first_slice,
                               int kind, uint32_t n_slices, const uint64_t *rec_off, uint16_t *recs,
                               uint8_t *init_states, uint32_t n_states) {
    if (n_slices == 0) return hipSuccess;
    hipLaunchKernelGGL(k_synth_slices, dim3((n_slices + 63) / 64), dim3(64), 0, s, workload, scale, seed, first_slice,
                       kind, n_slices, rec_off, recs, init_states, n_states);
    return hipGetLastError();
}

hipError_t launch_synth_tiles(hipStream_t s, int workload, uint32_t scale, uint64_t seed,
                              uint64_t first_slice, int kind, uint32_t n_slices, const uint32_t *order,
                              const uint64_t *tile_off, void *tiles, uint8_t *init_states,
                              uint32_t n_states) {
    if (n_slices == 0) return hipSuccess;
    hipLaunchKernelGGL(k_synth_tiles, dim3((n_slices + 63) / 64), dim3(64), 0, s, workload, scale, seed,
                       first_slice, kind, n_slices, order, tile_off, reinterpret_cast<uint4 *>(tiles),
                       init_states, n_states);
    return hipGetLastError();
}

hipError_t launch_expand_records8(hipStream_t s, const uint8_t *in, const uint64_t *rec_off, const uint32_t *n_bins, uint32_t n_slices,
                                  uint32_t n_states, uint64_t total, uint16_t *out) {
    if (n_slices == 0 || total == 0) return hipSuccess;
    const uint64_t threads = total / 8;
    hipLaunchKernelGGL(k_expand_records8, dim3(uint32_t((threads + 255) / 256)), dim3(256), 0, s, in, rec_off, n_bins, n_slices, n_states, total, out);
    return hipGetLastError();
}

}  // namespace avr
