// K2p kernels: the recoded range coder (compress direction) for batches of few, long slices.  The decomposition and the
// per-lane functions are in avr_k2p.h; this file maps them to lanes:
//
//   k_k2p_ranges_fp pass 1  lane per slice       the range recurrence (double-precision form); range and bytes emitted at every chunk start
//   k_k2p_ranges_wave       wave per slice       the same for batches of up to 1 024 slices: the lanes unpack the operands side by side
//   k_k2p_ranges   pass 1   lane per slice       the same in 64-bit integers: the slices the first form hands over (rare)
//   k_k2p_zero     pass 2a  lane per chunk       zero the positions of the byte sums that are ADDED into
//   k_k2p_code     pass 2b  lane per chunk       the coder from (low = 0, noted range); bytes added into 32-bit sums
//   k_k2p_finish   pass 3   workgroup per slice  carries from the last byte, finish(), bytes out
//
// Input is the slice-major record layout (a slice's records consecutive, padded with no-op records to a multiple of 8).
// Pass 1 is the wall: one lane per slice, one dependent chain of a 63-bit division, a multiply and a compare per bin
// (DESIGN.md section 4); passes 2 and 3 together take a few percent of it.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <vector>

#include "avr_div.h"
#include "avr_internal.h"
#include "avr_k2p.h"

namespace avr {

using namespace k2p;
static_assert(k2p::kChunk == AVR_CHUNK_BINS, "the chunk plan of the C ABI is made for this chunk size");

namespace {

struct U4 { uint32_t x, y, z, w; };

// A slice's status is written from two streams at once while pass 2 of one segment runs beside pass 1 of the next: pass 1 finds bad
// records and hands slices over (RETRY_SERIAL), pass 2 finds regions too small (OVERFLOW).  So that what a slice reports does not
// depend on which came first: a bad record overwrites anything; OVERFLOW and RETRY_SERIAL only replace "no error yet" (a slice handed
// over is walked again from its start and finds its overflow again; one in error already is left alone).
__device__ __forceinline__ void note_status(int32_t *status, int32_t code) {
    if (code == AVR_SLICE_BAD_RECORD) atomicExch(status, code);
    else atomicCAS(status, AVR_SLICE_OK, code);
}

struct K2Plan {
    const uint16_t *recs;
    const uint64_t *rec_off;
    const uint32_t *n_bins;
    const uint32_t *chunk_base, *chunk_slice;
    const uint64_t *out_off;        // the slice's bytes in `out`, and its sums in S, start here (capacity n_bins + 16 at least)
};

// fl(1 / d) for avr_div.h, per workgroup
__device__ __forceinline__ void fill_inv(double *inv) {
    for (uint32_t d = threadIdx.x; d < 256; d += blockDim.x) inv[d] = d ? 1.0 / double(d) : 0.0;
}

// Eight records: operands first (they depend on the records alone: no LDS latency on the range chain), then the bins.
template <bool WITH_LOW, class Emit>
__device__ __forceinline__ bool eight(const U4 &v, const double *inv, uint64_t &low, uint64_t &range, Emit &&emit) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t rec[8];
    double iv[8];
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) {
        rec[k] = (w[k >> 1] >> (16 * (k & 1))) & 0xffffu;
        iv[k] = inv[((rec[k] >> 1) & 0x7fu) + ((rec[k] >> 8) & 0x7fu)];
    }
    bool ok = true;
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) {
        const double i1 = iv[k];
        const auto div = [i1](uint64_t r, uint32_t t) { return div_u64_small_f64(r, double(t), i1); };
        // pass 1 leaves the walk at a bin of probability zero (measured faster than walking on: 105 against 118 ms on
        // config 2); pass 2 only sees slices that have none, and a test per bin is what it can do without
        if (WITH_LOW) bin<true>(low, range, rec[k], div, emit);
        else ok = ok && bin<false>(low, range, rec[k], div, emit);
    }
    return ok;
}

// Pass 1.  One lane per slice; ck_range / ck_pos: the range and the number of bytes emitted when chunk c of the slice
// begins; fin_range / fin_pos: at the slice's end.  A zero-probability bin (arithmetic_code.h:116-118) ends the slice
// with AVR_SLICE_ZERO_PROB.
__global__ __launch_bounds__(64) void k_k2p_ranges(K2Plan p, uint32_t n_slices, uint64_t *ck_range, uint32_t *ck_pos,
                                                  uint64_t *fin_range, uint32_t *fin_pos, int32_t *status, int32_t want_status) {
    __shared__ double inv[256];
    const uint32_t s = blockIdx.x * 64 + threadIdx.x;
    const bool mine = s < n_slices && status[s] == want_status;
    if (!__syncthreads_or(mine)) return;                         // the usual case when this is the hand-over launch: nothing to do
    fill_inv(inv);
    __syncthreads();
    if (!mine) return;
    const uint32_t n = p.n_bins[s], c0 = p.chunk_base[s];
    const U4 *r = reinterpret_cast<const U4 *>(p.recs + p.rec_off[s]);
    const uint32_t n_groups = (n + 7) >> 3, last = n_groups ? n_groups - 1 : 0;
    uint64_t range = kOne;                                       // arithmetic_code.h:96-97
    uint32_t pos = 0;
    bool ok = true;                                              // sticky: a bin of probability zero puts the slice in error
    // eight records: operands first (they depend on the records alone), then the chain -- range_step, no branch in it
    auto group = [&](const U4 &v) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t rec[8];
        double iv[8];
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) {
            rec[k] = (w[k >> 1] >> (16 * (k & 1))) & 0xffffu;
            iv[k] = inv[((rec[k] >> 1) & 0x7fu) + ((rec[k] >> 8) & 0x7fu)];
        }
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) {
            const double i1 = iv[k];
            ok &= range_step(range, pos, rec[k], [i1](uint64_t r, uint32_t t) { return div_u64_small_f64(r, double(t), i1); });
        }
    };
    // A cache line of records (four groups) per trip, two lines in flight ahead of the chain; every load is unconditional
    // (the index clamped to the slice's last group: what a clamped load returns is never coded).
    auto line = [&](uint32_t g, U4 v[4]) {
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) v[k] = r[g + k < last ? g + k : last];
    };
    U4 cur[4], nx1[4];
    if (n_groups) { line(0, cur); line(4, nx1); }
    uint32_t g = 0;
    for (; g + 4 <= n_groups; g += 4) {
        U4 nx2[4];
        line(g + 8, nx2);
        if ((g & (kChunk / 8 - 1)) == 0) { ck_range[c0 + (g >> 7)] = range; ck_pos[c0 + (g >> 7)] = pos; }
        group(cur[0]); group(cur[1]); group(cur[2]); group(cur[3]);
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) { cur[k] = nx1[k]; nx1[k] = nx2[k]; }
    }
    if (g < n_groups) {                                          // the slice's last, partial line (chunks start on whole lines)
        if ((g & (kChunk / 8 - 1)) == 0) { ck_range[c0 + (g >> 7)] = range; ck_pos[c0 + (g >> 7)] = pos; }
        for (uint32_t k = 0; g + k < n_groups; k++) group(cur[k]);
    }
    if (n_groups == 0) { ck_range[c0] = range; ck_pos[c0] = 0; }  // an empty slice still has its one chunk
    fin_range[s] = range;
    fin_pos[s] = pos;
    status[s] = ok ? (want_status == AVR_SLICE_OK ? AVR_SLICE_OK : AVR_SLICE_DONE) : AVR_SLICE_ZERO_PROB;   // DONE: parked for the hand-over's own passes 2, 3
}

// Pass 1, double-precision form (range_step_fp, avr_k2p.h): the same chunk notes from a chain of 12 dependent
// instructions per bin instead of 24.  A record's operands come from small LDS tables -- by total: { 1 / total,
// 1 / (2 total) } and { total, 2^32 / total }, by (pos, bin): { +-pos, bin ? 0 : 1 }.  A slice with a new range below 2^39 anywhere (a record with pos or neg 0, or a bin of probability zero) is handed
// to k_k2p_ranges (status AVR_SLICE_RETRY_SERIAL), which covers everything.  Also validates: bit 15 of a record must be clear.
struct TotA { double inv, h; };                           // by 2 total + bin: 1 / total, 1 / (2 total)
struct TotB { double d, inv32; };                         // by 2 total + bin: total, 2^32 / total
struct PosEntry { double ps, nb; };                       // by the record's low byte (2 pos + bin): +-pos, bin ? 0 : 1
//
// A lone wave issues one instruction every four to five cycles whatever it is, and the walk is bound by that count (about 30 a
// bin), not by the chain alone (tools/ubench/chain_latency: 105 cycles a bin with the operands in registers): so the tables are
// laid out for the fewest address instructions -- every entry 16 bytes and both tables by total indexed by 2 total + bin, which
// is the record's low byte plus twice its high byte: (low byte << 4) addresses PosEntry, that plus (high byte << 5) the other
// two -- and the operands of a group of eight records are fetched while the group BEFORE it is walked, so that no wait for LDS
// ever stalls the chain.
//
// The walk is launched in SEGMENTS of chunks [seg_begin, seg_end) of every slice (a lane picks up its range and byte count from
// the note of chunk seg_begin, which the segment before it left): pass 2 of a segment then runs on a second stream while pass 1
// walks the next one -- pass 1 keeps a handful of the chip's 1 024 SIMDs busy, pass 2 takes the rest (launch_k2p).
// (block: which 64 slices; long_chunks: slices of that many chunks or more are not this kernel's -- k_k2p_ranges_hybrid)
__device__ __forceinline__ void ranges_fp_body(uint32_t block, uint32_t long_chunks, K2Plan p, uint32_t n_slices, uint64_t *ck_range, uint32_t *ck_pos,
                                               uint64_t *fin_range, uint32_t *fin_pos, int32_t *status, uint32_t seg_begin, uint32_t seg_end) {
    __shared__ TotA tot_a[512];
    __shared__ TotB tot_b[512];
    __shared__ PosEntry pos_tab[256];
    for (uint32_t i = threadIdx.x; i < 512; i += 64) {
        const uint32_t d = i >> 1;
        const double inv = d ? 1.0 / double(d) : 0.0;
        tot_a[i] = TotA{inv, 0.5 * inv};
        tot_b[i] = TotB{double(d), 4294967296.0 * inv};
        if (i < 256) pos_tab[i] = PosEntry{(i & 1u) ? double(d) : -double(d), (i & 1u) ? 0.0 : 1.0};
    }
    __syncthreads();
    const uint32_t s = block * 64 + threadIdx.x;
    if (s >= n_slices || status[s] != AVR_SLICE_OK) return;
    const uint32_t n = p.n_bins[s], c0 = p.chunk_base[s];
    if (p.chunk_base[s + 1] - c0 >= long_chunks) return;
    const U4 *r = reinterpret_cast<const U4 *>(p.recs + p.rec_off[s]);
    const uint32_t n_groups = (n + 7) >> 3, last = n_groups ? n_groups - 1 : 0;
    const uint32_t g_begin = seg_begin * (kChunk / 8);
    if (seg_begin && g_begin >= n_groups) return;                // the slice ended in an earlier segment
    const uint32_t g_end = uint64_t(seg_end) * (kChunk / 8) < n_groups ? seg_end * (kChunk / 8) : n_groups;
    RangeFP rg = fp_from_u64(seg_begin ? ck_range[c0 + seg_begin] : kOne);          // arithmetic_code.h:96-97
    FpConsts K = fp_consts();
    asm volatile("" : "+s"(K.two32), "+s"(K.inv_two32), "+s"(K.split32), "+s"(K.two51), "+s"(K.two47));   // in scalar registers, see FpConsts
    uint32_t vmin_hi = 0xffffffffu;
    uint32_t pos8 = seg_begin ? ck_pos[c0 + seg_begin] * 8u : 0u, high = 0;          // pos8: BITS shifted out so far
    // a padding record (0) reads { -0, 1 }: a 0 of probability one.  Anything else with total 0, or pos 0 and bin 1, makes a
    // range of zero, which vmin_hi notes.
    uint32_t four = 4;
    asm volatile("" : "+v"(four));
    auto fetch = [&](uint32_t w0, uint32_t w1, BinFP o[4]) {     // half a group: the operands of four records
        high |= w0 | w1;
        uint32_t pos16[4], hi7[4], tot16[4];
        // (low byte of the record) << 4 is one SDWA shift; + (high byte << 5): a bit-field extract and a shift-add
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(pos16[0]) : "v"(four), "v"(w0));
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(pos16[1]) : "v"(four), "v"(w0));
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(pos16[2]) : "v"(four), "v"(w1));
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(pos16[3]) : "v"(four), "v"(w1));
        hi7[0] = __builtin_amdgcn_ubfe(w0, 8, 7); hi7[1] = __builtin_amdgcn_ubfe(w0, 24, 7);
        hi7[2] = __builtin_amdgcn_ubfe(w1, 8, 7); hi7[3] = __builtin_amdgcn_ubfe(w1, 24, 7);
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) asm("v_lshl_add_u32 %0, %1, 5, %2" : "=v"(tot16[k]) : "v"(hi7[k]), "v"(pos16[k]));
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
            const TotA ta = *reinterpret_cast<const TotA *>(reinterpret_cast<const uint8_t *>(tot_a) + tot16[k]);
            const TotB tb = *reinterpret_cast<const TotB *>(reinterpret_cast<const uint8_t *>(tot_b) + tot16[k]);
            const PosEntry pe = *reinterpret_cast<const PosEntry *>(reinterpret_cast<const uint8_t *>(pos_tab) + pos16[k]);
            o[k] = BinFP{ta.inv, ta.h, tb.d, pe.ps, pe.nb, tb.inv32};
        }
        __builtin_amdgcn_sched_barrier(0);                       // the reads stay here, ahead of the walk below them (a lone wave is
                                                                 // bound by its instruction count, not by where the fills sit)
    };
    auto walk = [&](const BinFP o[4]) {
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) pos8 += range_step_fp(rg, vmin_hi, o[k], K);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto note = [&](uint32_t g) { ck_range[c0 + (g >> 7)] = fp_to_u64(rg); ck_pos[c0 + (g >> 7)] = pos8 >> 3; };
    auto line = [&](uint32_t g, U4 v[4]) {
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) v[k] = r[g + k < last ? g + k : last];
    };
    U4 cur[4], nx1[4];
    BinFP oa[4], ob[4];
    if (n_groups) { line(g_begin, cur); line(g_begin + 4, nx1); fetch(cur[0].x, cur[0].y, oa); }
    uint32_t g = g_begin;
    for (; g + 4 <= g_end; g += 4) {                             // (g_end is a whole number of lines unless it is the slice's end)
        U4 nx2[4];
        line(g + 8, nx2);
        if ((g & (kChunk / 8 - 1)) == 0) note(g);
        __builtin_amdgcn_s_waitcnt(0xc07f);                      // lgkmcnt(0), here where it costs nothing: all that is in flight is the
                                                                 // trip before's last fetch, a walk old (else the compiler waits behind the next)
        fetch(cur[0].z, cur[0].w, ob); walk(oa);
        fetch(cur[1].x, cur[1].y, oa); walk(ob);
        fetch(cur[1].z, cur[1].w, ob); walk(oa);
        fetch(cur[2].x, cur[2].y, oa); walk(ob);
        fetch(cur[2].z, cur[2].w, ob); walk(oa);
        fetch(cur[3].x, cur[3].y, oa); walk(ob);
        fetch(cur[3].z, cur[3].w, ob); walk(oa);
        fetch(nx1[0].x, nx1[0].y, oa); walk(ob);                 // (behind the slice's end: a copy of its last group, never walked)
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) { cur[k] = nx1[k]; nx1[k] = nx2[k]; }
    }
    if (g < g_end) {                                             // the slice's last, partial line (chunks start on whole lines)
        if ((g & (kChunk / 8 - 1)) == 0) note(g);
        fetch(cur[0].z, cur[0].w, ob); walk(oa); walk(ob);
        if (g + 1 < g_end) { fetch(cur[1].x, cur[1].y, oa); fetch(cur[1].z, cur[1].w, ob); walk(oa); walk(ob); }
        if (g + 2 < g_end) { fetch(cur[2].x, cur[2].y, oa); fetch(cur[2].z, cur[2].w, ob); walk(oa); walk(ob); }
    }
    if (g_end < n_groups) note(g_end);                           // where the next segment picks up
    else {
        if (n_groups == 0) { ck_range[c0] = kOne; ck_pos[c0] = 0; }
        fin_range[s] = fp_to_u64(rg);
        fin_pos[s] = pos8 >> 3;
    }
    // (pass 2 of an earlier segment writes the same word from the second stream: a bad record wins over everything, and a slice that is
    // in error already is not asked to be walked again -- see note_status)
    if (high & 0x80008000u) note_status(&status[s], AVR_SLICE_BAD_RECORD);
    else if (vmin_hi < kTwo39Hi) note_status(&status[s], AVR_SLICE_RETRY_SERIAL);   // the integer form walks it again, from the start
}

// Pass 1 for batches with no more slices than the chip has SIMDs: a WAVE per slice.  The walk is bound by its instruction
// count (above), and with a lane per slice a third of it is getting at the operands: addresses out of the record, three
// data-dependent LDS reads, their waits.  Here the 64 lanes of the slice's wave do that side by side for 64 records at a time
// -- a lane per record: its table look-ups, then the six doubles into slot `lane` of a ring in LDS -- and the walk reads
// slot after slot at IMMEDIATE offsets from one base register (every lane the same address: a broadcast): three reads and
// no address arithmetic per record, about 25 instructions where the lane-per-slice kernel has 31.5.  All lanes walk the same
// range; lane 0 writes the notes.  Same notes, same statuses, same segments as k_k2p_ranges_fp.
constexpr uint32_t kRingBins = 64, kRingBytes = kRingBins * 48;
__device__ __forceinline__ void ranges_wave_body(uint32_t s, uint32_t long_chunks, K2Plan p, uint32_t n_slices, uint64_t *ck_range, uint32_t *ck_pos,
                                                 uint64_t *fin_range, uint32_t *fin_pos, int32_t *status, uint32_t seg_begin, uint32_t seg_end) {
    __shared__ TotA tot_a[512];
    __shared__ TotB tot_b[512];
    __shared__ PosEntry pos_tab[256];
    __shared__ __attribute__((aligned(16))) uint8_t ring[2 * kRingBytes];
    const uint32_t lane = threadIdx.x;
    if (s >= n_slices || status[s] != AVR_SLICE_OK) return;     // (the whole wave alike)
    const uint32_t n = p.n_bins[s], c0 = p.chunk_base[s];
    if (p.chunk_base[s + 1] - c0 < long_chunks) return;          // (k_k2p_ranges_hybrid: a lane's)
    const uint32_t n_batches = (n + kRingBins - 1) / kRingBins;
    const uint32_t b_begin = seg_begin * (kChunk / kRingBins);
    if (seg_begin && b_begin >= n_batches) return;               // the slice ended in an earlier segment
    for (uint32_t i = lane; i < 512; i += 64) {
        const uint32_t d = i >> 1;
        const double inv = d ? 1.0 / double(d) : 0.0;
        tot_a[i] = TotA{inv, 0.5 * inv};
        tot_b[i] = TotB{double(d), 4294967296.0 * inv};
        if (i < 256) pos_tab[i] = PosEntry{(i & 1u) ? double(d) : -double(d), (i & 1u) ? 0.0 : 1.0};
    }
    __syncthreads();
    const uint16_t *r = p.recs + p.rec_off[s];
    const uint32_t b_end = uint64_t(seg_end) * (kChunk / kRingBins) < n_batches ? seg_end * (kChunk / kRingBins) : n_batches;
    RangeFP rg = fp_from_u64(seg_begin ? ck_range[c0 + seg_begin] : kOne);           // arithmetic_code.h:96-97
    FpConsts K = fp_consts();
    asm volatile("" : "+s"(K.two32), "+s"(K.inv_two32), "+s"(K.split32), "+s"(K.two51), "+s"(K.two47));
    uint32_t vmin_hi = 0xffffffffu;
    uint32_t pos8 = seg_begin ? ck_pos[c0 + seg_begin] * 8u : 0u, high = 0;
    // a record behind the slice's last is read as 0: { -0, 1 }, a 0 of probability one (k_k2p_ranges_fp)
    auto load = [&](uint32_t b) -> uint32_t { const uint32_t i = b * kRingBins + lane; return i < n ? uint32_t(r[i]) : 0u; };
    auto unpack = [&](uint32_t rec, uint32_t half) {             // this lane's record into slot `lane` of ring half `half`
        high |= rec;
        const uint32_t pos16 = (rec & 0xffu) << 4, tot16 = pos16 + (((rec >> 8) & 0x7fu) << 5);
        const TotA ta = *reinterpret_cast<const TotA *>(reinterpret_cast<const uint8_t *>(tot_a) + tot16);
        const TotB tb = *reinterpret_cast<const TotB *>(reinterpret_cast<const uint8_t *>(tot_b) + tot16);
        const PosEntry pe = *reinterpret_cast<const PosEntry *>(reinterpret_cast<const uint8_t *>(pos_tab) + pos16);
        double *slot = reinterpret_cast<double *>(ring + half * kRingBytes + lane * 48);
        *reinterpret_cast<double2 *>(slot) = make_double2(ta.inv, ta.h);
        *reinterpret_cast<double2 *>(slot + 2) = make_double2(tb.d, tb.inv32);
        *reinterpret_cast<double2 *>(slot + 4) = make_double2(pe.ps, pe.nb);
    };
    auto note = [&](uint32_t b) {
        if (lane == 0) { ck_range[c0 + b / (kChunk / kRingBins)] = fp_to_u64(rg); ck_pos[c0 + b / (kChunk / kRingBins)] = pos8 >> 3; }
    };
    if (b_begin < b_end) {
        uint32_t rec1 = load(b_begin + 1);
        unpack(load(b_begin), b_begin & 1u);
        for (uint32_t b = b_begin; b < b_end; b++) {
            const uint32_t rec2 = load(b + 2);                   // two batches ahead of the walk (masked behind the slice's end)
            if ((b & (kChunk / kRingBins - 1)) == 0) note(b);
            unpack(rec1, (b + 1) & 1u);                          // the next batch's operands: LDS works in order, the walk below finds its own complete
            uint32_t base = (b & 1u) * kRingBytes;
            asm volatile("" : "+v"(base));                       // one base register, the 192 reads at immediate offsets
            BinFP oa[4], ob[4];
            auto fetch = [&](uint32_t k0, BinFP o[4]) {
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) {
                    const double2 a = *reinterpret_cast<const double2 *>(ring + base + (k0 + k) * 48);
                    const double2 c = *reinterpret_cast<const double2 *>(ring + base + (k0 + k) * 48 + 16);
                    const double2 e = *reinterpret_cast<const double2 *>(ring + base + (k0 + k) * 48 + 32);
                    o[k] = BinFP{a.x, a.y, c.x, e.x, e.y, c.y};
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            auto walk = [&](const BinFP o[4]) {
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) pos8 += range_step_fp(rg, vmin_hi, o[k], K);
                __builtin_amdgcn_sched_barrier(0);
            };
            fetch(0, oa);
#pragma unroll
            for (uint32_t k0 = 0; k0 < kRingBins; k0 += 8) {
                fetch(k0 + 4, ob); walk(oa);
                if (k0 + 8 < kRingBins) fetch(k0 + 8, oa);
                walk(ob);
            }
            rec1 = rec2;
        }
    }
    if (b_end < n_batches) note(b_end);                          // where the next segment picks up
    else if (lane == 0) {
        if (n_batches == 0) { ck_range[c0] = kOne; ck_pos[c0] = 0; }
        fin_range[s] = fp_to_u64(rg);
        fin_pos[s] = pos8 >> 3;
    }
    const bool bad = __any((high & 0x8000u) != 0);
    if (lane == 0) {
        if (bad) note_status(&status[s], AVR_SLICE_BAD_RECORD);
        else if (vmin_hi < kTwo39Hi) note_status(&status[s], AVR_SLICE_RETRY_SERIAL);   // the integer form walks it again, from the start
    }
}

__global__ __launch_bounds__(64) void k_k2p_ranges_fp(K2Plan p, uint32_t n_slices, uint64_t *ck_range, uint32_t *ck_pos,
                                                     uint64_t *fin_range, uint32_t *fin_pos, int32_t *status, uint32_t seg_begin, uint32_t seg_end) {
    ranges_fp_body(blockIdx.x, 0xffffffffu, p, n_slices, ck_range, ck_pos, fin_range, fin_pos, status, seg_begin, seg_end);
}
__global__ __launch_bounds__(64) void k_k2p_ranges_wave(K2Plan p, uint32_t n_slices, uint64_t *ck_range, uint32_t *ck_pos,
                                                       uint64_t *fin_range, uint32_t *fin_pos, int32_t *status, uint32_t seg_begin, uint32_t seg_end) {
    ranges_wave_body(blockIdx.x, 0u, p, n_slices, ck_range, ck_pos, fin_range, fin_pos, status, seg_begin, seg_end);
}
// Both in one launch, for a ragged batch of more slices than SIMDs: a lane per slice packs 64 slices into a wave and leaves
// most of the chip idle, and the step takes as long as the longest slice -- so the longest slices, as many as there are SIMDs
// left over, get a wave each (blocks lane_blocks .. of the grid), which walks 13 % faster.  *long_chunks (k_k2p_threshold) is
// the chunk count from which a slice counts as long.
__global__ __launch_bounds__(64) void k_k2p_ranges_hybrid(uint32_t lane_blocks, const uint32_t *long_chunks, K2Plan p, uint32_t n_slices,
                                                         uint64_t *ck_range, uint32_t *ck_pos, uint64_t *fin_range, uint32_t *fin_pos,
                                                         int32_t *status, uint32_t seg_begin, uint32_t seg_end) {
    const uint32_t t = long_chunks[0];                           // long_chunks[1]: how many long slices, [2 ..]: which
    if (blockIdx.x < lane_blocks) ranges_fp_body(blockIdx.x, t, p, n_slices, ck_range, ck_pos, fin_range, fin_pos, status, seg_begin, seg_end);
    else if (blockIdx.x - lane_blocks < long_chunks[1])
        ranges_wave_body(long_chunks[2 + blockIdx.x - lane_blocks], t, p, n_slices, ck_range, ck_pos, fin_range, fin_pos, status, seg_begin, seg_end);
}
// long_chunks[0] = the smallest power of two 2^k such that at most max_long slices have 2^k chunks or more (0xffffffff: there is
// none), [1] = how many slices that is, [2 ..] = those slices.
__global__ __launch_bounds__(1024) void k_k2p_threshold(const uint32_t *chunk_base, uint32_t n_slices, uint32_t max_long, uint32_t *long_chunks) {
    __shared__ uint32_t hist[33];                                // hist[b]: slices whose chunk count has b significant bits
    __shared__ uint32_t thr, count;
    if (threadIdx.x < 33) hist[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t s = threadIdx.x; s < n_slices; s += 1024) atomicAdd(&hist[32 - __clz(chunk_base[s + 1] - chunk_base[s])], 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0xffffffffu, above = 0;
        for (int b = 32; b >= 1; b--) {                          // slices of 2^(b-1) chunks or more: those with b or more significant bits
            above += hist[b];
            if (above > max_long) break;
            t = 1u << (b - 1);
        }
        long_chunks[0] = thr = t;
        count = 0;
    }
    __syncthreads();
    for (uint32_t s = threadIdx.x; s < n_slices; s += 1024)
        if (chunk_base[s + 1] - chunk_base[s] >= thr) long_chunks[2 + atomicAdd(&count, 1u)] = s;
    __syncthreads();
    if (threadIdx.x == 0) long_chunks[1] = count;
}

// Which chunk a lane of passes 2a / 2b takes.  seg_len > 0: lane i takes chunk seg_begin + i % seg_len of slice i / seg_len (a
// segment of every slice, lanes packed); seg_len == 0: lane i takes global chunk i if it lies at or behind seg_begin in its
// slice (the last, open-ended segment -- and the whole slice when there is only one).  Returns false when there is nothing to do:
// no such chunk, a slice whose status is not `want`, or a chunk that would write past the slice's region, in which case the slice
// is flagged AVR_SLICE_OVERFLOW here (positions grow along the slice: the chunks before it fitted, the ones behind it do not).
struct ChunkPick { uint32_t s, gc, c, n, last; };
__device__ __forceinline__ bool pick_chunk(const K2Plan &p, uint32_t i, uint32_t n_slices, uint32_t total_chunks, uint32_t seg_begin, uint32_t seg_len,
                                           const uint32_t *ck_pos, const uint32_t *fin_pos, int32_t *status, int32_t want, ChunkPick *o) {
    uint32_t s, gc;
    if (seg_len) {
        s = i / seg_len;
        if (s >= n_slices) return false;
        const uint32_t c = seg_begin + (i - s * seg_len);
        gc = p.chunk_base[s] + c;
        if (gc >= p.chunk_base[s + 1]) return false;
    } else {
        gc = i;
        if (gc >= total_chunks) return false;
        s = p.chunk_slice[gc];
        if (gc - p.chunk_base[s] < seg_begin) return false;
    }
    if (status[s] != want) return false;
    o->s = s; o->gc = gc; o->c = gc - p.chunk_base[s]; o->n = p.n_bins[s];
    o->last = gc + 1 == p.chunk_base[s + 1];
    const uint32_t end_pos = o->last ? fin_pos[s] : ck_pos[gc + 1];
    if (uint64_t(end_pos) + kTail > p.out_off[s + 1] - p.out_off[s]) { atomicCAS(&status[s], want, AVR_SLICE_OVERFLOW); return false; }
    return true;
}

// Pass 2a: the positions that will be ADDED into are zeroed; every other position gets one plain store.  Added into are the kTail
// positions from every chunk's end on (its own left-over low, the first bytes of the chunk behind it, and, at the slice's end, what
// finish() reads), and the first kTail of the slice.  A chunk's lane zeroes those behind ITS end that the chunk before it has not
// already zeroed -- [max(end, start + kTail), end + kTail) -- so that no position is zeroed twice: with the passes running segment
// by segment, a second zeroing would come after the adds of the segment before.
__global__ __launch_bounds__(256) void k_k2p_zero(K2Plan p, uint32_t n_slices, uint32_t total_chunks, const uint32_t *ck_pos, const uint32_t *fin_pos,
                                                 int32_t *status, uint32_t *S, int32_t want, uint32_t seg_begin, uint32_t seg_len) {
    ChunkPick k;
    if (!pick_chunk(p, blockIdx.x * 256 + threadIdx.x, n_slices, total_chunks, seg_begin, seg_len, ck_pos, fin_pos, status, want, &k)) return;
    uint32_t *base = S + p.out_off[k.s];
    const uint32_t start = ck_pos[k.gc], end = k.last ? fin_pos[k.s] : ck_pos[k.gc + 1];
    if (k.c == 0)
        for (uint32_t j = 0; j < kTail; j++) base[j] = 0;
    for (uint32_t j = end > start + kTail ? end : start + kTail; j < end + kTail; j++) base[j] = 0;
}

// Pass 2.  One lane per chunk: the coder itself over the chunk's bins, from low = 0 and the range noted by pass 1; every
// byte it shifts out (carry bit included) is added to the sum of its position in the slice, and what is left of low at
// the end to the kTail positions behind.  Lanes of neighbouring chunks add into the same positions where they meet.
__global__ __launch_bounds__(256) void k_k2p_code(K2Plan p, uint32_t n_slices, uint32_t total_chunks, const uint64_t *ck_range, const uint32_t *ck_pos,
                                                 const uint32_t *fin_pos, int32_t *status, uint32_t *S, int32_t want, uint32_t seg_begin, uint32_t seg_len) {
    __shared__ double inv[256];
    fill_inv(inv);
    __syncthreads();
    ChunkPick k;
    if (!pick_chunk(p, blockIdx.x * 256 + threadIdx.x, n_slices, total_chunks, seg_begin, seg_len, ck_pos, fin_pos, status, want, &k)) return;
    const uint32_t s = k.s, gc = k.gc, n = k.n, i0 = k.c * kChunk;
    const uint32_t i1 = i0 + kChunk < n ? i0 + kChunk : n;
    const uint16_t *r = p.recs + p.rec_off[s];
    uint32_t *at = S + p.out_off[s] + ck_pos[gc];
    uint32_t *const shared_end = at + kTail;                     // up to here the chunks before this one may have left bytes of their low
    uint64_t low = 0, range = ck_range[gc];
    auto add = [&](uint32_t v) {
        if (at < shared_end) atomicAdd(at, v); else *at = v;     // (behind it the position is this chunk's alone, until its own left-over)
        at++;
    };
    auto add_shared = [&](uint32_t v) { atomicAdd(at, v); at++; };
    // a cache line of records per trip, the next one in flight (a slice's records are padded to whole groups of 8)
    uint32_t i = i0;
    U4 v0{0, 0, 0, 0}, v1 = v0, v2 = v0, v3 = v0;
    if (i + 32 <= i1) { const U4 *q = reinterpret_cast<const U4 *>(r + i); v0 = q[0]; v1 = q[1]; v2 = q[2]; v3 = q[3]; }
    for (; i + 32 <= i1; i += 32) {
        // the next line unconditionally (at the chunk's last whole line: that line again, a hit): a load inside a branch is waited for
        // at the branch's end, which made every trip wait out a memory latency (tools/ubench/read_patterns)
        const U4 *qn = reinterpret_cast<const U4 *>(r + (i + 64 <= i1 ? i + 32 : i));
        const U4 n0 = qn[0], n1 = qn[1], n2 = qn[2], n3 = qn[3];
        eight<true>(v0, inv, low, range, add); eight<true>(v1, inv, low, range, add);
        eight<true>(v2, inv, low, range, add); eight<true>(v3, inv, low, range, add);
        v0 = n0; v1 = n1; v2 = n2; v3 = n3;
    }
    for (; i < i1; i += 8) eight<true>(*reinterpret_cast<const U4 *>(r + i), inv, low, range, add);
    leftover(low, add_shared);
}

// Pass 3.  One 64-lane workgroup per slice.  Positions [0, P + kTail) of the slice's sums, P = bytes emitted: carries from
// the last position (tiles of kFinTile through LDS: loads and byte stores by all lanes, the carry chain by lane 0);
// the kTail positions past P then hold the reference's final low, to which finish() (arithmetic_code.h:128-144) is
// applied as it stands.
constexpr uint32_t kFinTile = 4096, kFinSeg = kFinTile / 64, kFinRow = kFinSeg + 1;   // a lane's segment, padded by one word: no bank conflicts
__global__ __launch_bounds__(64) void k_k2p_finish(K2Plan p, const uint64_t *fin_range, const uint32_t *fin_pos, const uint32_t *S,
                                                  uint8_t *out, uint32_t *out_len, int32_t *status, int32_t want) {
    __shared__ uint32_t dig[64 * kFinRow];
    __shared__ uint32_t cout[65];
    __shared__ uint8_t tail_bytes[kTail];
    const uint32_t s = blockIdx.x, t = threadIdx.x;
    if (status[s] != want) {                                     // (a slice parked as DONE is the hand-over launch's to finish)
        if (t == 0 && want == AVR_SLICE_OK && status[s] != AVR_SLICE_DONE) out_len[s] = 0;
        return;
    }
    const uint32_t P = fin_pos[s];
    const uint32_t *Ss = S + p.out_off[s];
    uint8_t *o = out + p.out_off[s];
    const uint32_t cap = uint32_t(p.out_off[s + 1] - p.out_off[s]);
    uint32_t tile_carry = 0;                                     // into the last position of the tile being done (same in every lane)
    // Tiles from the last position.  A tile is aligned to its END: entry e of the tile (position lo + e - pad) sits in segment
    // e / kFinSeg, and segment 63 ends at the tile's last position; a short first tile leaves its low segments empty (zeros).
    for (uint32_t hi = P + kTail; hi > 0;) {
        const uint32_t lo = hi > kFinTile ? hi - kFinTile : 0, cnt = hi - lo, pad = kFinTile - cnt;
        for (uint32_t e = t; e < kFinTile; e += 64) dig[(e / kFinSeg) * kFinRow + e % kFinSeg] = e >= pad ? Ss[lo + e - pad] : 0u;
        __syncthreads();
        // every lane its own segment from the segment's last entry, then the carries between segments until none is left:
        // integer addition in another order.  After the first sweep every entry is a byte, so a carry that arrives only
        // travels through 0xff entries: the loop ends after a round or two.
        uint32_t *seg = dig + t * kFinRow;
        uint32_t c = t == 63 ? tile_carry : 0u;
        for (uint32_t i = kFinSeg; i-- > 0;) { const uint32_t v = seg[i] + c; seg[i] = v & 0xffu; c = v >> 8; }
        uint32_t out_carry = 0;                                  // what leaves segment 0: into the next tile
        for (;;) {
            cout[t] = c;
            __syncthreads();
            const uint32_t into = t < 63 ? cout[t + 1] : 0u;     // from the segment behind (higher positions)
            out_carry += cout[0];
            const bool more = __any(into != 0);
            __syncthreads();
            if (!more) break;
            c = into;
            for (uint32_t i = kFinSeg; c && i-- > 0;) { const uint32_t v = seg[i] + c; seg[i] = v & 0xffu; c = v >> 8; }
        }
        tile_carry = out_carry;
        __syncthreads();
        for (uint32_t e = t + (pad & ~63u); e < kFinTile; e += 64) {
            if (e < pad) continue;
            const uint32_t at = lo + e - pad, v = dig[(e / kFinSeg) * kFinRow + e % kFinSeg];
            if (at >= P) tail_bytes[at - P] = uint8_t(v);
            else if (at < cap) o[at] = uint8_t(v);
        }
        hi = lo;
        __syncthreads();
    }
    if (t == 0) {
        uint8_t tail[9];
        uint32_t cy;
        const uint32_t n_tail = finish(low_from_tail(tail_bytes), fin_range[s], tail, &cy);
        for (uint32_t k = 0; k < n_tail; k++)
            if (P + k < cap) o[P + k] = tail[k];
        for (uint32_t i = P < cap ? P : cap; cy && i-- > 0;) { const uint32_t b = uint32_t(o[i]) + 1u; o[i] = uint8_t(b); cy = b >> 8; }
        out_len[s] = P + n_tail;
        status[s] = P + n_tail > cap ? AVR_SLICE_OVERFLOW : AVR_SLICE_OK;
    }
}

inline uint64_t up256(uint64_t x) { return (x + 255) & ~uint64_t(255); }

}  // namespace

// workspace: ck_range, ck_pos per chunk; fin_range, fin_pos per slice; 32-bit sums per output byte position
size_t k2p_workspace_bytes(size_t n_slices, uint32_t total_chunks, uint64_t out_total) {
    return size_t(up256(uint64_t(total_chunks) * 8) + up256(uint64_t(total_chunks) * 4) + up256(n_slices * 8) + up256(n_slices * 4) + 4096 +
                  up256(out_total * 4 + 64));
}

// The second stream pass 2 runs on while pass 1 walks on, with its events: one set per (device, caller's stream), made on
// first use and kept (work on the caller's stream is ordered, so consecutive calls may share it).
namespace {
constexpr uint32_t kMaxSegments = 8;
struct Side { int dev; hipStream_t main, side; hipEvent_t seg[kMaxSegments], join; };
std::vector<Side *> g_side_pool;
std::mutex g_side_mu;
hipError_t side_stream(hipStream_t s, Side **out) {
    std::vector<Side *> &pool = g_side_pool;
    std::mutex &mu = g_side_mu;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    for (Side *x : pool)
        if (x->dev == dev && x->main == s) { *out = x; return hipSuccess; }
    Side *x = new Side{dev, s, nullptr, {}, nullptr};
    if ((e = hipStreamCreateWithFlags(&x->side, hipStreamNonBlocking)) != hipSuccess) { delete x; return e; }
    for (uint32_t k = 0; k < kMaxSegments && e == hipSuccess; k++) e = hipEventCreateWithFlags(&x->seg[k], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&x->join, hipEventDisableTiming);
    if (e != hipSuccess) { delete x; return e; }
    pool.push_back(x);
    *out = x;
    return hipSuccess;
}
}  // namespace

// The caller's stream is going away (avr_batch_destroy): what was kept for it -- the second stream and its events -- goes with it.
void forget_side_stream(hipStream_t s) {
    std::lock_guard<std::mutex> lock(g_side_mu);
    for (size_t i = 0; i < g_side_pool.size();) {
        Side *x = g_side_pool[i];
        if (x->main != s) { i++; continue; }
        (void)hipStreamSynchronize(x->side);
        for (uint32_t k = 0; k < kMaxSegments; k++) (void)hipEventDestroy(x->seg[k]);
        (void)hipEventDestroy(x->join);
        (void)hipStreamDestroy(x->side);
        delete x;
        g_side_pool.erase(g_side_pool.begin() + long(i));
    }
}

hipError_t launch_k2p(hipStream_t s, const uint16_t *recs, const uint64_t *rec_off, const uint32_t *n_bins, uint32_t n_slices,
                      const uint32_t *chunk_base, const uint32_t *chunk_slice, uint32_t total_chunks, uint64_t out_total,
                      void *workspace, uint8_t *out, const uint64_t *out_off, uint32_t *out_len, int32_t *status) {
    if (n_slices == 0) return hipSuccess;
    (void)out_total;
    uint8_t *w = static_cast<uint8_t *>(workspace);
    uint64_t *ck_range = reinterpret_cast<uint64_t *>(w);    w += up256(uint64_t(total_chunks) * 8);
    uint32_t *ck_pos = reinterpret_cast<uint32_t *>(w);      w += up256(uint64_t(total_chunks) * 4);
    uint64_t *fin_range = reinterpret_cast<uint64_t *>(w);   w += up256(uint64_t(n_slices) * 8);
    uint32_t *fin_pos = reinterpret_cast<uint32_t *>(w);     w += up256(uint64_t(n_slices) * 4);
    uint32_t *long_chunks = reinterpret_cast<uint32_t *>(w); w += 4096;     // the hybrid pass 1's threshold, count and list of long slices (at most 1 022)
    uint32_t *S = reinterpret_cast<uint32_t *>(w);
    const K2Plan p{recs, rec_off, n_bins, chunk_base, chunk_slice, out_off};
    const dim3 slice_grid((n_slices + 63) / 64), chunk_grid((total_chunks + 255) / 256);
    // Segments: an eighth of the average slice each, the last one open-ended (it takes whatever the longer slices have left).
    const uint32_t avg = total_chunks / n_slices;
    uint32_t seg_len = avg >= 32 ? (avg + kMaxSegments - 1) / kMaxSegments : 0;
    if (test_hooks().k2p_seg_len) seg_len = test_hooks().k2p_seg_len;       // test build only: short slices through many segments
    const uint32_t n_seg = seg_len ? kMaxSegments : 1;
    Side *side = nullptr;
    hipError_t e;
    if (n_seg > 1 && (e = side_stream(s, &side)) != hipSuccess) return e;
    hipStream_t s2 = side ? side->side : s;
    // a wave per slice while every slice's wave has a SIMD of its own (1 024 of them); beyond, a lane per slice and -- while the
    // lanes' waves leave SIMDs over -- a wave each for the longest slices (test hook k2p_wave: 1 / 2 = the one / the other for all)
    constexpr uint32_t kSimds = 1024;
    const uint32_t form = test_hooks().k2p_wave ? test_hooks().k2p_wave : n_slices <= kSimds ? 1u : slice_grid.x + 64 <= kSimds ? 3u : 2u;
    if (form == 3) hipLaunchKernelGGL(k_k2p_threshold, dim3(1), dim3(1024), 0, s, chunk_base, n_slices, kSimds - slice_grid.x, long_chunks);
    for (uint32_t k = 0; k < n_seg; k++) {
        const bool open = k + 1 == n_seg;
        const uint32_t begin = k * seg_len, end = open ? 0xffffffffu / kChunk : begin + seg_len;
        if (form == 1)
            hipLaunchKernelGGL(k_k2p_ranges_wave, dim3(n_slices), dim3(64), 0, s, p, n_slices, ck_range, ck_pos, fin_range, fin_pos, status, begin, end);
        else if (form == 3)
            hipLaunchKernelGGL(k_k2p_ranges_hybrid, dim3(kSimds), dim3(64), 0, s, slice_grid.x, long_chunks, p, n_slices, ck_range, ck_pos,
                               fin_range, fin_pos, status, begin, end);
        else
            hipLaunchKernelGGL(k_k2p_ranges_fp, slice_grid, dim3(64), 0, s, p, n_slices, ck_range, ck_pos, fin_range, fin_pos, status, begin, end);
        if (side) {
            if ((e = hipEventRecord(side->seg[k], s)) != hipSuccess || (e = hipStreamWaitEvent(s2, side->seg[k], 0)) != hipSuccess) {
                (void)hipStreamSynchronize(s2);                  // the caller only knows its own stream: nothing of this call is left behind on the other
                return e;
            }
        }
        const dim3 grid = open ? chunk_grid : dim3(uint32_t((uint64_t(n_slices) * seg_len + 255) / 256));
        const uint32_t len = open ? 0 : seg_len;
        hipLaunchKernelGGL(k_k2p_zero, grid, dim3(256), 0, s2, p, n_slices, total_chunks, ck_pos, fin_pos, status, S, AVR_SLICE_OK, begin, len);
        hipLaunchKernelGGL(k_k2p_code, grid, dim3(256), 0, s2, p, n_slices, total_chunks, ck_range, ck_pos, fin_pos, status, S, AVR_SLICE_OK, begin, len);
    }
    if (side) {
        if ((e = hipEventRecord(side->join, s2)) != hipSuccess || (e = hipStreamWaitEvent(s, side->join, 0)) != hipSuccess) {
            (void)hipStreamSynchronize(s2);
            return e;
        }
    }
    hipLaunchKernelGGL(k_k2p_finish, dim3(n_slices), dim3(64), 0, s, p, fin_range, fin_pos, S, out, out_len, status, AVR_SLICE_OK);
    // The slices the double-precision walk handed over (AVR_SLICE_RETRY_SERIAL: a range below 2^39 somewhere), start to end in the
    // integer form; a workgroup without any leaves at once, so these four launches cost microseconds when there is none.
    hipLaunchKernelGGL(k_k2p_ranges, slice_grid, dim3(64), 0, s, p, n_slices, ck_range, ck_pos, fin_range, fin_pos, status, AVR_SLICE_RETRY_SERIAL);
    hipLaunchKernelGGL(k_k2p_zero, chunk_grid, dim3(256), 0, s, p, n_slices, total_chunks, ck_pos, fin_pos, status, S, AVR_SLICE_DONE, 0u, 0u);
    hipLaunchKernelGGL(k_k2p_code, chunk_grid, dim3(256), 0, s, p, n_slices, total_chunks, ck_range, ck_pos, fin_pos, status, S, AVR_SLICE_DONE, 0u, 0u);
    hipLaunchKernelGGL(k_k2p_finish, dim3(n_slices), dim3(64), 0, s, p, fin_range, fin_pos, S, out, out_len, status, AVR_SLICE_DONE);
    return hipGetLastError();
}

}  // namespace avr
