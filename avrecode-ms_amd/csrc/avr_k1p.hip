// K1p kernels: intra-slice parallel CABAC encode for batches of few, long slices.
// The algorithm and the per-lane functions of phases B-D are in avr_k1p.h; this file maps
// them to lanes and adds phase A (context-state resolution), which is a per-slice stable
// counting sort by context followed by one state chain per (slice, context):
//
//   k_k1p_hist      A1  workgroup per sort block    per-block count of every context
//   k_k1p_scan      A2  workgroup per slice         run start of every context, block offsets
//   k_k1p_scatter   A3  wave per sort block         stable rank (ballot multisplit) -> sorted order
//   k_k1p_spec      A4a lane per sorted segment     walk the entered run from the two extreme states
//   k_k1p_link      A4b lane per sorted segment     true entry state of every segment
//   k_k1p_chain     A4c lane per sorted segment     state before each bin (cabac_code.h:43-47) -> resolved codes
//   k_k1p_entry     A5a thread per 4 contexts       state of every context at the start of every chunk
//   k_k1p_replay    A5b lane per chunk              resolved code of every bin, in stream order
//   k_k1p_b1        B1  lane per chunk              stretch summaries for the 4 entry quarters
//   k_k1p_b2        B2  lane per slice              chain the summaries: entry range + bit position
//   k_k1p_zero          workgroup per slice         zero the digit sums that will be used
//   k_k1p_c         C   lane per chunk              code each stretch, add its digits
//   k_k1p_d         D   workgroup per slice         finish(), carries (segmented), bytes
//
// Input is the slice-major record layout (a slice's bins must be consecutive for the sort and
// for the chunk lanes).  Results are byte-identical to k_cabac_encode (tests/test_gpu_k1p.py);
// a slice the scheme declines (no coded LPS for 16 chunks) is coded by k_cabac_encode itself.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "avr_coder.h"
#include "avr_internal.h"
#include "avr_k1p.h"
#include "avr_tables.h"

namespace avr {

static __device__ const CabacTables d_tables = make_cabac_tables();

using namespace k1p;

constexpr uint32_t kSortBlock = AVR_SORT_BLOCK_BINS;

struct Plan {                       // device pointers of the caller's plan (avr_chunk_plan) + record layout
    const uint16_t *recs;
    const uint64_t *rec_off;
    const uint32_t *n_bins;
    const uint64_t *res_off;
    const uint32_t *chunk_base, *chunk_slice;
    const uint32_t *blk_base, *blk_slice;
    const uint64_t *dig_off;
    uint32_t n_states;              // sort keys: the contexts the batch uses, numbered densely 0 .. n_states-1 (k_k1p_densemap)
    uint32_t ns_full;               // contexts per slice as the caller numbers them (init_states / final_states rows)
    const uint16_t *table;          // [1024] caller's context number -> dense id (kNotUsed: occurs nowhere in the batch)
    const uint16_t *index;          // [n_states] dense id -> caller's context number
};
constexpr uint32_t kNotUsed = 0xffffu;

// ------------------------------------------------------------------ phase A

// Also the one place every record of this path is examined: a selector that is no context of the
// slice, bypass or terminate, or a bin after put_terminate(1), flags the slice AVR_SLICE_BAD_RECORD.
//
// Contexts are counted under the caller's numbering (the offset of the state byte in cabac_state[],
// recode.cpp:325: up to 1024, of which a stream touches few).  The census of the contexts the batch
// uses falls out of the same pass -- the non-zero columns -- as a 1024-bit map; k_k1p_densemap turns it
// into the dense numbering every later kernel sorts and indexes by, applied to the records as they are
// loaded (one LDS look-up), so no pass over the records exists for the renumbering.
__global__ __launch_bounds__(256) void k_k1p_hist(Plan p, int32_t *status, uint16_t *hist16, uint32_t *used) {
    __shared__ uint32_t cnt[AVR_MAX_STATES];
    __shared__ uint32_t bm[32];
    const uint32_t b = blockIdx.x, s = p.blk_slice[b], nk = p.ns_full;
    for (uint32_t k = threadIdx.x; k < nk; k += 256) cnt[k] = 0;
    if (threadIdx.x < 32) bm[threadIdx.x] = 0;
    __syncthreads();
    if (status[s] == AVR_SLICE_OK) {
        const uint32_t n = p.n_bins[s], i0 = (b - p.blk_base[s]) * kSortBlock;
        const uint32_t i1 = i0 + kSortBlock < n ? i0 + kSortBlock : n;
        const uint16_t *r = p.recs + p.rec_off[s];
        // 8 records (one 16-byte chunk) per thread per trip; a slice's padding records are no-ops,
        // and i0 is a multiple of 8, so whole chunks can be read up to the padded end
        // Records that are no context bin: bypass (selector 1024) and terminate (1025) are the values
        // 2048..2051; anything else is bad, and so is put_terminate(1) = 2051 anywhere but last.
        uint32_t worst = 0;                                      // max over non-context records of (record - 2048), wrapping
        for (uint32_t i = i0 + threadIdx.x * 8; i < i1; i += 256 * 8) {
            const uint4 v = *reinterpret_cast<const uint4 *>(r + i);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) {
                const uint32_t rec = (w[j >> 1] >> ((j & 1) * 16)) & 0xffffu, sel = rec >> 1;   // sel keeps bits 12..15: a set one is bad too
                if (i + j < i1) {
                    if (sel < nk) atomicAdd(&cnt[sel], 1u);
                    else {
                        const uint32_t t = rec - 2048u + (i + j + 1 == n ? 0u : (rec == 2051u ? 4u : 0u));
                        worst = worst > t ? worst : t;
                    }
                }
            }
        }
        const bool bad = worst > 3u;
        if (bad) status[s] = AVR_SLICE_BAD_RECORD;
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < nk; k += 256) {
        const uint32_t c = cnt[k];
        hist16[size_t(b) * nk + k] = uint16_t(c);                // <= kSortBlock
        if (c) atomicOr(&bm[k >> 5], 1u << (k & 31));
    }
    __syncthreads();
    // bits only ever get set, so a (possibly stale) plain read tells which are still missing: after the
    // first few blocks nothing is, and no block touches the shared words any more
    if (threadIdx.x < 32) {
        const uint32_t mine = bm[threadIdx.x];
        if (mine & ~__hip_atomic_load(&used[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(&used[threadIdx.x], mine);
    }
}

// used (1024 bits) -> table[caller's number] = dense id, index[dense id] = caller's number, *n_dense.
__global__ __launch_bounds__(1024) void k_k1p_densemap(const uint32_t *used, uint16_t *table, uint16_t *index, uint32_t *n_dense) {
    __shared__ uint32_t sc[1024];
    const uint32_t k = threadIdx.x;
    const uint32_t bit = (used[k >> 5] >> (k & 31)) & 1u;
    sc[k] = bit;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        const uint32_t v = k >= d ? sc[k - d] : 0;
        __syncthreads();
        sc[k] += v;
        __syncthreads();
    }
    const uint32_t id = sc[k] - bit;
    table[k] = bit ? uint16_t(id) : uint16_t(kNotUsed);
    if (bit) index[id] = uint16_t(k);
    if (k == 1023) *n_dense = sc[1023];
}

// hist16[b][caller's number] -> boff[b][k] = position (within the slice's sorted order) of the first bin of
// context k (dense) in block b; run_start[s][k] = first position of context k, run_start[s][nk] = number of
// context bins.
__global__ __launch_bounds__(1024) void k_k1p_scan(Plan p, const uint16_t *hist16, uint32_t *boff, uint32_t *run_start) {
    __shared__ uint32_t sc[1024];
    const uint32_t s = blockIdx.x, k = threadIdx.x, nk = p.n_states, ns = p.ns_full;
    const uint32_t b0 = p.blk_base[s], b1 = p.blk_base[s + 1];
    const uint32_t col = k < nk ? p.index[k] : 0;
    uint32_t total = 0;
    if (k < nk) {
        uint32_t b = b0;
        for (; b + 8 <= b1; b += 8) {                            // eight loads in flight: the loop is their latency
            uint32_t c[8];
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) c[j] = hist16[size_t(b + j) * ns + col];
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) total += c[j];
        }
        for (; b < b1; b++) total += hist16[size_t(b) * ns + col];
    }
    sc[k] = total;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {                    // inclusive scan over contexts
        const uint32_t v = k >= d ? sc[k - d] : 0;
        __syncthreads();
        sc[k] += v;
        __syncthreads();
    }
    uint32_t run = sc[k] - total;                                // exclusive
    if (k < nk) {
        run_start[size_t(s) * (nk + 1) + k] = run;
        uint32_t b = b0;
        for (; b + 8 <= b1; b += 8) {
            uint32_t c[8];
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) c[j] = hist16[size_t(b + j) * ns + col];
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) { boff[size_t(b + j) * nk + k] = run; run += c[j]; }
        }
        for (; b < b1; b++) {
            const uint32_t c = hist16[size_t(b) * ns + col];
            boff[size_t(b) * nk + k] = run;
            run += c;
        }
    }
    if (k == 1023) run_start[size_t(s) * (nk + 1) + nk] = sc[1023];
}

// One workgroup of 4 waves per sort block; wave w takes the w-th quarter of the block.  Bins are
// taken 64 at a time in stream order; the lanes holding the same context find each other with one
// ballot per key bit, which gives every bin its rank among them (stable), and the first of them
// advances the context's running position.  A wave is a serial chain through those running
// positions, so the block is quartered to shorten it: the quarters are counted first (LDS atomics),
// each wave starts from the block's local start of the context plus what the quarters before it hold.
//
// The block is sorted into LDS first and copied out afterwards: its bins of one context go to
// consecutive global positions, so the copy-out stores are runs of consecutive bytes, where a
// direct scatter would be 64 different cache lines per store instruction (measured: the direct
// form spent 75 % of its wave cycles stalled on issuing those stores).
constexpr uint32_t kQuarter = kSortBlock / 4, kQuarterBatches = kQuarter / 64;

template <uint32_t KEY_BITS>
__global__ __launch_bounds__(256, 8) void k_k1p_scatter(Plan p, const int32_t *status, const uint32_t *boff,
                                                     const uint32_t *run_start, uint8_t *sorted, uint32_t *qoff) {
    // LDS (dynamic, sized by the number of contexts so that more blocks fit a CU):
    extern __shared__ uint32_t scatter_lds[];
    const uint32_t nk = p.n_states, nk_pad = (nk + 63) & ~63u;
    uint32_t *cnt = scatter_lds;                                 // [4][nk_pad]: next local position, per wave
    uint32_t *delta = cnt + 4 * nk_pad;                          // global position - local position
    uint16_t *kbuf = reinterpret_cast<uint16_t *>(delta + nk_pad);   // context of every locally sorted bin
    uint8_t *lbuf = reinterpret_cast<uint8_t *>(kbuf + kSortBlock);  // the bins, locally sorted
    __shared__ uint16_t tab[1024];                               // caller's context number -> dense id
    __shared__ uint32_t sh_n_local;
    const uint32_t b = blockIdx.x, s = p.blk_slice[b], t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (status[s] != AVR_SLICE_OK) return;
    const uint32_t n = p.n_bins[s], i0 = (b - p.blk_base[s]) * kSortBlock;
    const uint32_t i1 = i0 + kSortBlock < n ? i0 + kSortBlock : n;
    const uint16_t *r = p.recs + p.rec_off[s];
    // this wave's records: batch j of the quarter is r[q0 + 64 j + lane]
    const uint32_t q0 = i0 + w * kQuarter;
    uint32_t recs[kQuarterBatches];
    const uint16_t *rq = r + q0 + lane;
    if (q0 + kQuarter <= i1) {                                   // a whole quarter (all but a slice's last block): no bounds to check
#pragma unroll
        for (uint32_t j = 0; j < kQuarterBatches; j++) recs[j] = rq[64 * j];
    } else {
#pragma unroll
        for (uint32_t j = 0; j < kQuarterBatches; j++) recs[j] = q0 + 64 * j + lane < i1 ? rq[64 * j] : uint32_t(AVR_NOP_CABAC);
    }
    for (uint32_t k = t; k < 4 * nk_pad; k += 256) cnt[k] = 0;
    for (uint32_t k = t; k < 1024; k += 256) tab[k] = p.table[k];
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < kQuarterBatches; j++) {             // renumber (dense ids), count the quarter
        const uint32_t raw = (recs[j] >> 1) & 0x7ffu;
        const uint32_t sel = raw < 1024u ? uint32_t(tab[raw]) : raw | 0x8000u;      // not a context: anything >= nk
        recs[j] = (recs[j] & 1u) | (sel << 1);
        if (sel < nk) atomicAdd(&cnt[w * nk_pad + sel], 1u);
    }
    __syncthreads();
    if (w == 0) {
        // local exclusive scan of the block's per-context counts: 16 consecutive contexts per lane
        const bool last_block = b + 1 == p.blk_base[s + 1];
        const uint32_t *g0 = boff + size_t(b) * nk;
        const uint32_t *g1 = last_block ? run_start + size_t(s) * (nk + 1) + 1 : g0 + nk;   // where context k stops
        uint32_t mine[16], sum = 0;
#pragma unroll
        for (uint32_t j = 0; j < 16; j++) {
            const uint32_t k = lane * 16 + j;
            mine[j] = k < nk ? g1[k] - g0[k] : 0;
            sum += mine[j];
        }
        uint32_t incl = sum;
        for (uint32_t d = 1; d < 64; d <<= 1) {
            const uint32_t v = __shfl_up(incl, d);
            if (lane >= d) incl += v;
        }
        uint32_t run = incl - sum;
        if (lane == 63) sh_n_local = incl;                       // context bins in this block
#pragma unroll
        for (uint32_t j = 0; j < 16; j++) {
            const uint32_t k = lane * 16 + j;
            if (k < nk) {
                delta[k] = g0[k] - run;
                uint32_t at = run;                               // quarter w starts after the quarters before it
                for (uint32_t q = 0; q < 4; q++) { const uint32_t c = cnt[q * nk_pad + k]; cnt[q * nk_pad + k] = at; at += c; }
            }
            run += mine[j];
        }
    }
    __syncthreads();
    // where every quarter's bins of every context start in the slice's sorted order (for k_k1p_replay)
    for (uint32_t k = t; k < nk; k += 256) {
        const uint32_t d = delta[k];
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) qoff[(size_t(b) * 4 + q) * nk + k] = cnt[q * nk_pad + k] + d;
    }
    __syncthreads();
    uint32_t *my_cnt = cnt + w * nk_pad;
    const uint64_t lt = (uint64_t(1) << lane) - 1;
#pragma unroll
    for (uint32_t j = 0; j < kQuarterBatches; j++) {
        uint32_t rec = recs[j];
        asm volatile("" : "+v"(rec));                            // keeps this batch's ballots here: hoisted to the top, the 16 x KEY_BITS
                                                                 // lane masks outlive the SGPR file and are spilled (measured 1.45x slower)
        const uint32_t sel = rec >> 1;
        const bool is_ctx = sel < nk;
        // lanes holding the same context: for every key bit keep the lanes whose bit equals mine,
        // mask &= ~(ballot(bit) ^ (my bit ? ~0 : 0)), one three-input bit operation per half
        const uint64_t m0 = __ballot(is_ctx);
        uint32_t mask_lo = uint32_t(m0), mask_hi = uint32_t(m0 >> 32);
#pragma unroll
        for (uint32_t bit = 0; bit < KEY_BITS; bit++) {
            const uint32_t mine = uint32_t(__builtin_amdgcn_sbfe(int32_t(rec), bit + 1, 1));   // bit of the selector, as 0 / ~0
            const uint64_t m = __ballot(mine != 0);
            mask_lo &= ~(uint32_t(m) ^ mine);
            mask_hi &= ~(uint32_t(m >> 32) ^ mine);
        }
        const uint64_t mask = uint64_t(mask_lo) | uint64_t(mask_hi) << 32;
        if (is_ctx) {
            const uint32_t rank = __popcll(mask & lt);
            const uint32_t start = my_cnt[sel];
            const uint32_t at = start + rank;
            lbuf[at] = uint8_t(rec & 1);
            kbuf[at] = uint16_t(sel);
            if (rank == 0) my_cnt[sel] = start + __popcll(mask);
        }
    }
    __syncthreads();
    uint8_t *so = sorted + p.res_off[s];
    const uint32_t n_local = sh_n_local;
    for (uint32_t j = t; j < n_local; j += 256) so[j + delta[kbuf[j]]] = lbuf[j];
}

// ---- A4: state chains over the sorted order.
//
// The bins of one context are a contiguous run of `sorted`, in stream order; the state before each
// is a chain from the slice's initial state (cabac_code.h:43-47).  A hot context's run is tens of
// thousands of bins long, so the sorted order is cut into SEGMENTS of kChunk positions, one lane
// each.  A run that starts inside a segment starts from its known initial state.  For the run a
// segment is entered in the middle of, the state is not known yet -- but the CABAC transition
// functions are MONOTONE in the order
//     (62,MPS 0) < (61,0) < ... < (0,0) < (0,1) < ... < (62,MPS 1)          [pStateIdx, valMPS]
// (tests/test_k1p_emul.py checks this on the tables), so if the two extreme states 124 and 125
// have reached the same state after some bins, every possible state has.  k_k1p_spec walks the
// entered run from both extremes (and the run it is left in, exactly, if that run started inside);
// k_k1p_link gives every segment its true entry state (replaying predecessors only where the
// extremes did not meet, which real streams essentially never do over 1024 bins of one context);
// k_k1p_chain walks every segment once more and writes the resolved codes.

struct Seg {
    uint32_t k_first;        // context whose run contains the segment's first position
    uint16_t lo, hi;         // state at the segment's end of the run it is left in, from the extremes
                             // 124 / 125 (equal: exact); only meaningful if that run goes on
    uint8_t enters_mid;      // the first position is not the start of its run
    uint8_t leaves_mid;      // the last position is not the end of its run
    uint8_t empty;           // no positions
    uint8_t pad;
};

__device__ __forceinline__ uint32_t chain_next(const uint32_t *next, uint32_t st, uint32_t bin) {
    const uint32_t nx = next[st];
    return ((bin ^ st) & 1) ? (nx >> 8) : (nx & 0xffu);
}

// Walk sorted positions [from, to) of one run.  WRITE: replace each bin by its resolved code.
// Two states are carried (for the speculative walk); pass the same value twice for an exact walk.
template <bool WRITE>
__device__ __forceinline__ void chain_walk(const uint32_t *next, uint8_t *so, uint32_t from, uint32_t to,
                                           uint32_t &a, uint32_t &b) {
    uint32_t at = from;
    auto one = [&](uint32_t bin) {
        const uint32_t code = code_context(a, bin);
        a = chain_next(next, a, bin);
        if (!WRITE) b = chain_next(next, b, bin);
        return code;
    };
    auto group16 = [&](U4 v, uint32_t where) {
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint32_t sh = (j & 3) * 8;
            const uint32_t code = one((w[j >> 2] >> sh) & 1u);
            if (WRITE) w[j >> 2] = (w[j >> 2] & ~(0xffu << sh)) | (code << sh);
        }
        if (WRITE) *reinterpret_cast<U4 *>(so + where) = U4{w[0], w[1], w[2], w[3]};
    };
    for (; at < to && (at & 15); at++) { const uint32_t c = one(so[at]); if (WRITE) so[at] = uint8_t(c); }
    for (; at + 16 <= to && (at & 63); at += 16) group16(*reinterpret_cast<const U4 *>(so + at), at);
    for (; at + 64 <= to; at += 64) {                    // a whole cache line per lane per trip (see for_codes_all)
        const U4 *p = reinterpret_cast<const U4 *>(so + at);
        const U4 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3];
        group16(v0, at); group16(v1, at + 16); group16(v2, at + 32); group16(v3, at + 48);
    }
    for (; at + 16 <= to; at += 16) group16(*reinterpret_cast<const U4 *>(so + at), at);
    for (; at < to; at++) { const uint32_t c = one(so[at]); if (WRITE) so[at] = uint8_t(c); }
}

// context whose run [rs[k], rs[k+1]) contains sorted position `at` (at < rs[nk])
__device__ __forceinline__ uint32_t run_of(const uint32_t *rs, uint32_t nk, uint32_t at) {
    uint32_t lo = 0, hi = nk;                                    // invariant: rs[lo] <= at < rs[hi]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (rs[mid] <= at) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void k_k1p_spec(Plan p, uint32_t total_chunks, const int32_t *status,
                                                  const uint32_t *run_start, const uint8_t *init_states,
                                                  uint8_t *sorted, Seg *seg) {
    __shared__ uint32_t next[128];
    if (threadIdx.x < 128) next[threadIdx.x] = d_tables.packed[threadIdx.x][1];
    __syncthreads();
    const uint32_t gc = blockIdx.x * 256 + threadIdx.x;
    if (gc >= total_chunks) return;
    const uint32_t s = p.chunk_slice[gc], nk = p.n_states;
    Seg o{0, 0, 0, 0, 0, 1, 0};
    const uint32_t *rs = run_start + size_t(s) * (nk + 1);
    const uint32_t from = (gc - p.chunk_base[s]) * kChunk;
    if (status[s] != AVR_SLICE_OK || from >= rs[nk]) { seg[gc] = o; return; }
    const uint32_t to = from + kChunk < rs[nk] ? from + kChunk : rs[nk];
    uint8_t *so = sorted + p.res_off[s];
    const uint8_t *init = init_states + size_t(s) * p.ns_full;   // rows in the caller's numbering
    const uint32_t k0 = run_of(rs, nk, from);
    o.empty = 0;
    o.k_first = k0;
    o.enters_mid = rs[k0] < from;
    const uint32_t k1 = run_of(rs, nk, to - 1);                 // the run the segment is left in
    o.leaves_mid = rs[k1 + 1] > to;
    if (o.leaves_mid) {
        const uint32_t i0 = init[p.index[k1]] & 127u;
        uint32_t a, b;
        if (k1 == k0 && o.enters_mid && i0 < 126) { a = 124; b = 125; }        // entered and left in the same run
        else a = b = i0;                                          // the run starts here, or its state never moves
        if (i0 < 126) chain_walk<false>(next, so, k1 == k0 ? from : rs[k1], to, a, b);
        o.lo = uint16_t(a);
        o.hi = uint16_t(b);
    }
    seg[gc] = o;
}

// entry[gc] = state of the run the segment is entered in the middle of, at its first position.
__global__ __launch_bounds__(256) void k_k1p_link(Plan p, uint32_t total_chunks, const int32_t *status,
                                                  const uint32_t *run_start, const uint8_t *init_states,
                                                  uint8_t *sorted, const Seg *seg, uint8_t *entry) {
    __shared__ uint32_t next[128];
    if (threadIdx.x < 128) next[threadIdx.x] = d_tables.packed[threadIdx.x][1];
    __syncthreads();
    const uint32_t gc = blockIdx.x * 256 + threadIdx.x;
    if (gc >= total_chunks) return;
    const Seg me = seg[gc];
    if (me.empty || !me.enters_mid) { entry[gc] = 0; return; }
    const uint32_t s = p.chunk_slice[gc];
    // back to the nearest predecessor whose exit is known without its own entry
    uint32_t j = gc - 1;                                         // c >= 1 because the segment is entered mid-run
    while (seg[j].lo != seg[j].hi) j--;                          // ends at the latest where the run starts (exact walk)
    uint32_t st = seg[j].lo;
    if (j + 1 < gc) {                                            // replay the predecessors whose extremes did not meet
        uint8_t *so = sorted + p.res_off[s];
        for (uint32_t r = j + 1; r < gc; r++) {
            const uint32_t from = (r - p.chunk_base[s]) * kChunk;
            uint32_t a = st, b = st;
            chain_walk<false>(next, so, from, from + kChunk, a, b);
            st = a;
        }
    }
    (void)run_start; (void)init_states; (void)status;
    entry[gc] = uint8_t(st);
}

__global__ __launch_bounds__(256) void k_k1p_chain(Plan p, uint32_t total_chunks, const int32_t *status,
                                                   const uint32_t *run_start, const uint8_t *init_states,
                                                   uint8_t *sorted, const Seg *seg, const uint8_t *entry,
                                                   uint8_t *final_states) {
    __shared__ uint32_t next[128];
    // two bins per look-up: pair[st] = { state after a first bin 0 | 1,  state after bins 00 | 01 | 10 | 11
    // (first bin in the higher index bit) }: the walk is a chain of dependent LDS reads, this halves it
    __shared__ uint2 pair[128];
    if (threadIdx.x < 128) {
        const uint32_t st = threadIdx.x, nx = d_tables.packed[st][1];
        next[st] = nx;
        uint32_t mid[2], end = 0;
        for (uint32_t b0 = 0; b0 < 2; b0++) {
            mid[b0] = ((b0 ^ st) & 1) ? (nx >> 8) : (nx & 0xffu);
            const uint32_t n2 = d_tables.packed[mid[b0]][1];
            for (uint32_t b1 = 0; b1 < 2; b1++) end |= (((b1 ^ mid[b0]) & 1) ? (n2 >> 8) : (n2 & 0xffu)) << (8 * (2 * b0 + b1));
        }
        pair[st] = make_uint2(mid[0] | mid[1] << 8, end);
    }
    __syncthreads();
    const uint32_t gc = blockIdx.x * 256 + threadIdx.x;
    if (gc >= total_chunks) return;
    const uint32_t s = p.chunk_slice[gc], nk = p.n_states;
    if (status[s] != AVR_SLICE_OK) return;
    const uint32_t *rs = run_start + size_t(s) * (nk + 1);
    // rows of init_states / final_states are in the caller's numbering; final_states starts out as a copy of
    // init_states (launch_resolve), so only contexts that have bins in the slice are written here
    const uint8_t *init = init_states + size_t(s) * p.ns_full;
    uint8_t *fin = final_states ? final_states + size_t(s) * p.ns_full : nullptr;
    const uint16_t *idx = p.index;
    const uint32_t c = gc - p.chunk_base[s], from = c * kChunk;
    const Seg me = seg[gc];
    if (me.empty) return;
    const uint32_t to = from + kChunk < rs[nk] ? from + kChunk : rs[nk];
    uint8_t *so = sorted + p.res_off[s];
    uint32_t k = me.k_first;
    uint32_t st = me.enters_mid ? uint32_t(entry[gc]) : (init[idx[k]] & 127u);
    uint32_t run_end = rs[k + 1];                                // > from: the segment's first position lies in run k
    // run k ends at `run_end`: its final state, then on to the next run that has bins
    auto next_run = [&]() {
        if (fin) fin[idx[k]] = uint8_t(st);
        k++;
        while (k < nk && rs[k + 1] == rs[k]) k++;
        if (k < nk) { st = init[idx[k]] & 127u; run_end = rs[k + 1]; } else run_end = 0xffffffffu;
    };
    // The segment is walked in aligned 16-byte groups whatever runs it holds (a cold context's run is
    // a few bins: walking run by run would mean byte loads and stores for most of such a segment).
    auto group16 = [&](U4 &v, uint32_t at) {
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
        if (at + 16 <= run_end && at + 16 <= to) {               // no run ends inside
#pragma unroll
            for (int j = 0; j < 16; j += 2) {
                const uint32_t sh = (j & 3) * 8, d = w[j >> 2] >> sh, b0 = d & 1u, b1 = (d >> 8) & 1u;
                const uint2 e = pair[st];
                const uint32_t mid = (e.x >> (8 * b0)) & 0xffu;
                const uint32_t c0 = code_context(st, b0), c1 = code_context(mid, b1);
                st = (e.y >> (8 * (2 * b0 + b1))) & 0xffu;
                w[j >> 2] = (w[j >> 2] & ~(0xffffu << sh)) | ((c0 | c1 << 8) << sh);
            }
        } else {
            for (uint32_t j = 0; j < 16 && at + j < to; j++) {
                if (at + j == run_end) next_run();
                const uint32_t sh = (j & 3) * 8, bin = (w[j >> 2] >> sh) & 1u;
                const uint32_t code = code_context(st, bin);
                st = chain_next(next, st, bin);
                w[j >> 2] = (w[j >> 2] & ~(0xffu << sh)) | (code << sh);
            }
        }
        v = U4{w[0], w[1], w[2], w[3]};
    };
    uint32_t at = from;                                          // a multiple of kChunk; `sorted` is padded past rs[nk]
    if (at + 64 <= to) {                                         // a whole cache line per lane per trip, the next one in flight
        U4 *q = reinterpret_cast<U4 *>(so + at);
        U4 v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3];
        for (; at + 64 <= to; at += 64) {
            q = reinterpret_cast<U4 *>(so + at);
            U4 n0 = v0, n1 = v1, n2 = v2, n3 = v3;
            if (at + 128 <= to) { n0 = q[4]; n1 = q[5]; n2 = q[6]; n3 = q[7]; }
            group16(v0, at); group16(v1, at + 16); group16(v2, at + 32); group16(v3, at + 48);
            q[0] = v0; q[1] = v1; q[2] = v2; q[3] = v3;
            v0 = n0; v1 = n1; v2 = n2; v3 = n3;
        }
    }
    for (; at < to; at += 16) {
        U4 *q = reinterpret_cast<U4 *>(so + at);
        U4 v = q[0];
        group16(v, at);
        q[0] = v;
    }
    if (to == run_end) next_run();                               // the run (and any empty ones after it) ends with the segment
}

// A5: one lane per chunk (a quarter of a sort block), its records in stream order.  After the chains
// the resolved code at sorted position qoff[chunk][k] holds the state of context k when the chunk is
// entered, so a lane loads those states (one byte per context, in LDS, laid out (k, lane) like
// k_cabac_encode's) and simply plays the chunk's bins: state before the bin -> resolved code,
// cabac_code.h:43-47 -> next state.  Everything it touches in HBM is read or written once, in order:
// records in, codes out.  (The form this replaced fetched every bin's code back from the sorted
// order through a 32-bit position the scatter kernel had stored per bin: 8 more bytes of traffic per
// bin and a gather of 64 cache lines per load; 0.97 ms on config 2.)
//
// The step is branch-free.  Bypass, terminate and padding records go through pseudo contexts nk+1..
// whose pseudo states 128.. never move, and one LDS table gives, per state, both successors and
// both resolved codes:  T[st] = next if MPS | next if LPS << 8 | code(bin 0) << 16 | code(bin 1) << 24.
constexpr uint32_t kStBypass = 128, kStTerminate = 129, kStPad = 130, kStNone = 131, kReplayTable = 192;

// The entry states of every chunk, four contexts per thread: est[chunk][k] = state of context k at the
// chunk's first bin, read off the resolved code at the sorted position where the chunk's bins of k
// start (a context without a bin from there on gets whatever lies at the end of its run: it is never
// looked at).  Done apart from the replay so that a replay lane starts from independent loads.
__global__ __launch_bounds__(256) void k_k1p_entry(Plan p, uint32_t total_chunks, const int32_t *status, const uint32_t *qoff,
                                                  const uint8_t *sorted, const uint8_t *init_states, uint32_t *est,
                                                  uint32_t chunks_per_block, uint32_t recip) {
    // Consecutive threads take consecutive context groups of one chunk: the chunk's offsets are read and
    // its states written as one contiguous row.  (Measured: chunk-minor threads with est[kw][chunk], which
    // would make the sorted-order reads neighbours instead, is 3x slower: the offset reads become strided.)
    // A workgroup takes chunks_per_block = 256 / nkw whole rows; thread -> (row, group) by a multiply with
    // recip = ceil(2^16 / nkw), exact for thread ids below 256.
    const uint32_t nk = p.n_states, nkw = (nk + 3) >> 2;
    const uint32_t row = (threadIdx.x * recip) >> 16, kw = threadIdx.x - row * nkw, k0 = kw * 4;
    const uint32_t gc = blockIdx.x * chunks_per_block + row;
    if (row >= chunks_per_block || gc >= total_chunks) return;
    const uint32_t s = p.chunk_slice[gc];
    if (status[s] != AVR_SLICE_OK) return;
    const uint32_t c = gc - p.chunk_base[s];
    if (c * kChunk >= p.n_bins[s]) return;
    const uint8_t *so = sorted + p.res_off[s];
    const uint32_t *bo = qoff + (size_t(p.blk_base[s]) * 4 + c) * nk;    // chunk c is quarter c & 3 of sort block c >> 2
    const uint8_t *init = init_states + size_t(s) * p.ns_full;
    uint32_t word = 0;
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) {
        const uint32_t k = k0 + j < nk ? k0 + j : nk - 1;
        const uint32_t cd = so[bo[k]];
        word |= (cd < 252 ? cd >> 1 : init[p.index[k]] & 127u) << (8 * j);  // pStateIdx 63 never moves
    }
    est[size_t(gc) * nkw + kw] = word;
}

__global__ __launch_bounds__(256) void k_k1p_replay(Plan p, uint32_t total_chunks, const uint32_t *est, uint8_t *res,
                                                    const int32_t *status) {
    extern __shared__ uint32_t replay_lds[];                     // T[kReplayTable], then per wave: state dwords [(nk+8)/4][64]
    __shared__ uint16_t tab[1024];                               // caller's context number -> dense id
    uint32_t *T = replay_lds;
    const uint32_t lane = threadIdx.x & 63, nk = p.n_states;
    uint8_t *stb = reinterpret_cast<uint8_t *>(replay_lds + kReplayTable) + (threadIdx.x >> 6) * (((nk + 8) >> 2) << 8) + lane * 4;
    for (uint32_t k = threadIdx.x; k < 1024; k += blockDim.x) tab[k] = p.table[k];
    for (uint32_t st = threadIdx.x; st < kReplayTable; st += blockDim.x) {
        uint32_t e;
        if (st < 128) e = d_tables.packed[st][1] | code_context(st, 0) << 16 | code_context(st, 1) << 24;
        else {
            const uint32_t c0 = st == kStBypass ? kCodeBypass : st == kStTerminate ? code_terminate(0) : kCodePad;
            const uint32_t c1 = st == kStBypass ? kCodeBypass | 1 : st == kStTerminate ? code_terminate(1) : kCodePad;
            e = st | st << 8 | c0 << 16 | c1 << 24;
        }
        T[st] = e;
    }
    __syncthreads();
    const uint32_t gc = blockIdx.x * blockDim.x + threadIdx.x;
    if (gc >= total_chunks) return;
    const uint32_t s = p.chunk_slice[gc];
    if (status[s] != AVR_SLICE_OK) return;
    const uint32_t c = gc - p.chunk_base[s];
    const uint32_t n = p.n_bins[s], i0 = c * kChunk;
    const uint32_t i1 = i0 + kChunk < n ? i0 + kChunk : n;
    if (i0 >= n) return;
    {
        const uint32_t nkw = (nk + 3) >> 2;
        const uint32_t *e = est + size_t(gc) * nkw;
        for (uint32_t kw = 0; kw < nkw; kw++) *reinterpret_cast<uint32_t *>(stb + kw * 256) = e[kw];
        const uint32_t pseudo[5] = {kStNone, kStBypass, kStTerminate, kStPad, kStNone};
#pragma unroll
        for (uint32_t j = 0; j < 5; j++) stb[((nk + j) >> 2) * 256 + ((nk + j) & 3)] = uint8_t(pseudo[j]);
    }
    const uint16_t *r = p.recs + p.rec_off[s];
    uint8_t *ro = res + p.res_off[s];
    // 8 records (16 bytes) -> 8 codes
    auto eight = [&](const U4 &v, uint32_t &c0, uint32_t &c1) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t cc[2] = {0, 0};
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) {
            const uint32_t rec = w[j >> 1] >> ((j & 1) * 16);
            const uint32_t sel = (rec >> 1) & 0x7ffu, bin = rec & 1;
            // contexts get their dense id; 1024 (bypass), 1025 (terminate), 1026 (no-op) -> nk+1, nk+2, nk+3
            const uint32_t over = (sel < 1023u ? 1023u : sel > 1027u ? 1027u : sel) - 1023u;
            const uint32_t dense = tab[sel & 1023u];
            const uint32_t kk = (sel < 1024u && dense < nk ? dense : nk) + over;
            uint8_t *sp = stb + ((kk & ~3u) << 6) + (kk & 3);
            const uint32_t st = *sp;
            const uint32_t e = T[st];
            *sp = uint8_t(e >> (8 * ((bin ^ st) & 1)));
            cc[j >> 2] |= ((e >> (16 + 8 * bin)) & 0xffu) << (8 * (j & 3));
        }
        c0 = cc[0]; c1 = cc[1];
    };
    // a slice's records are padded with no-ops to a multiple of 8, its codes to a multiple of 16
    uint32_t i = i0;
    U4 v0, v1, v2, v3;
    if (i + 32 <= i1) {
        const U4 *q = reinterpret_cast<const U4 *>(r + i);
        v0 = q[0]; v1 = q[1]; v2 = q[2]; v3 = q[3];
    }
    for (; i + 32 <= i1; i += 32) {                              // a cache line of records per trip, the next one in flight
        U4 n0 = v0, n1 = v1, n2 = v2, n3 = v3;
        if (i + 64 <= i1) {
            const U4 *q = reinterpret_cast<const U4 *>(r + i + 32);
            n0 = q[0]; n1 = q[1]; n2 = q[2]; n3 = q[3];
        }
        U4 a, b;
        eight(v0, a.x, a.y); eight(v1, a.z, a.w); eight(v2, b.x, b.y); eight(v3, b.z, b.w);
        *reinterpret_cast<U4 *>(ro + i) = a;
        *reinterpret_cast<U4 *>(ro + i + 16) = b;
        v0 = n0; v1 = n1; v2 = n2; v3 = n3;
    }
    for (; i < i1; i += 16) {
        const U4 *q = reinterpret_cast<const U4 *>(r + i);
        const U4 nop{AVR_NOP_CABAC2, AVR_NOP_CABAC2, AVR_NOP_CABAC2, AVR_NOP_CABAC2};
        const U4 t0 = q[0], t1 = i + 8 < i1 ? q[1] : nop;
        U4 a;
        eight(t0, a.x, a.y); eight(t1, a.z, a.w);
        *reinterpret_cast<U4 *>(ro + i) = a;
    }
}

// ------------------------------------------------------------------ phases B1, B2, C

// code_entry() from the device copy of the tables (d_tables.packed[2 p][0] = rangeTabLPS[p][0..3])
__device__ __forceinline__ CodeEntry device_code_entry(uint32_t c) {
    if (code_is_bypass(c)) return CodeEntry{0u, 0x100u | ((c & 1u) << 1)};
    return CodeEntry{d_tables.packed[2 * (c >> 2)][0], code_sym(c) * 3u};
}

__global__ __launch_bounds__(256) void k_k1p_b1(Plan p, uint32_t total_chunks, const uint8_t *res,
                                                const int32_t *status, Stretch *st, uint32_t max_stretch) {
    __shared__ CodeEntry codes[256];
    codes[threadIdx.x] = device_code_entry(threadIdx.x);
    __syncthreads();
    const uint32_t gc = blockIdx.x * 256 + threadIdx.x;
    if (gc >= total_chunks) return;
    const uint32_t slice = p.chunk_slice[gc];
    if (status[slice] != AVR_SLICE_OK) { st[gc].first = kNone; st[gc].too_long = 0; return; }
    Stretch o;
    b1_stretch(res + p.res_off[slice], p.n_bins[slice], gc - p.chunk_base[slice], codes, max_stretch, &o);
    st[gc] = o;
}

// One workgroup per slice: the stretch summaries are staged through LDS a tile at a time (coalesced),
// thread 0 runs the serial 4->4 chain on them (b2_chain's loop, fed from LDS instead of ~600
// dependent trips to HBM), and the entries go back out coalesced.
constexpr uint32_t kB2Tile = 1024;

__global__ __launch_bounds__(256) void k_k1p_b2(Plan p, const int32_t *status, const Stretch *st, Entry *en,
                                                SliceTotals *tot) {
    __shared__ Stretch tile[kB2Tile];
    __shared__ Entry ent[kB2Tile];
    __shared__ uint8_t xq[kB2Tile], qin[kB2Tile];               // exit quarter per entry quarter (2 bits each); entry quarter
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t carry[4];                                // T, q, r, bad across tiles
    const uint32_t s = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (status[s] != AVR_SLICE_OK) {
        if (t == 0) { tot[s].t_total = 0; tot[s].r_final = 510; tot[s].bad = 0; tot[s].pad = 0; }
        return;
    }
    const uint32_t c0 = p.chunk_base[s], nc = p.chunk_base[s + 1] - c0;
    if (t == 0) { carry[0] = 0; carry[1] = 0; carry[2] = 510; carry[3] = 0; }
    for (uint32_t base = 0; base < nc; base += kB2Tile) {
        const uint32_t cnt = nc - base < kB2Tile ? nc - base : kB2Tile;
        const uint32_t *src = reinterpret_cast<const uint32_t *>(st + c0 + base);
        uint32_t *dst = reinterpret_cast<uint32_t *>(tile);
        for (uint32_t i = t; i < cnt * (sizeof(Stretch) / 4); i += 256) dst[i] = src[i];
        __syncthreads();
        // the only serial part is the quarter: one byte look-up per stretch (an inactive chunk maps q to q)
        for (uint32_t c = t; c < kB2Tile; c += 256) xq[c] = c < cnt && tile[c].first != kNone ? tile[c].exit_q : uint8_t(0xE4);
        __syncthreads();
        if (t == 0) {
            uint32_t q = carry[1];
            for (uint32_t c = 0; c < cnt; c++) { qin[c] = uint8_t(q); q = (xq[c] >> (2 * q)) & 3; }
            carry[1] = q;
        }
        __syncthreads();
        // everything else follows from the entry quarters: shifts (prefix sum), range, flags.  Thread t
        // takes stretches 4t .. 4t+3 of the tile.
        uint32_t tv[4], sum = 0, bad = 0, last = kNone;
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
            const uint32_t c = 4 * t + j;
            const bool on = c < cnt && tile[c].first != kNone;
            tv[j] = on ? tile[c].t_exit[qin[c]] : 0;
            sum += tv[j];
            if (on) { bad |= tile[c].too_long; last = c; }
        }
        uint32_t incl = sum;
        for (uint32_t d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(incl, d); if (lane >= d) incl += v; }
        if (lane == 63) wsum[w] = incl;
        // the last active stretch of the tile gives the range; any too_long flags the slice
        const uint64_t has = __ballot(last != kNone);
        __syncthreads();
        uint32_t run = carry[0] + incl - sum;
        for (uint32_t v = 0; v < w; v++) run += wsum[v];
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
            const uint32_t c = 4 * t + j;
            if (c < cnt) { ent[c].t_start = run; ent[c].q = qin[c]; }
            run += tv[j];
        }
        if (bad) atomicOr(&carry[3], 1u);
        __syncthreads();
        if (t == 255) carry[0] = run;                            // total shifts so far
        // range after the tile: the highest active stretch (waves in order, the last one that has any wins)
        for (uint32_t v = 0; v < 4; v++) {
            if (w == v && has && lane == 63 - uint32_t(__builtin_clzll(has))) carry[2] = tile[last].r_exit[qin[last]];
            __syncthreads();
        }
        uint32_t *eo = reinterpret_cast<uint32_t *>(en + c0 + base);
        const uint32_t *ei = reinterpret_cast<const uint32_t *>(ent);
        for (uint32_t i = t; i < cnt * (sizeof(Entry) / 4); i += 256) eo[i] = ei[i];
        __syncthreads();
    }
    if (t == 0) { tot[s].t_total = carry[0]; tot[s].r_final = carry[2]; tot[s].bad = carry[3]; tot[s].pad = 0; }
}

__global__ __launch_bounds__(256) void k_k1p_zero(Plan p, const SliceTotals *tot, uint32_t *S) {
    const uint32_t s = blockIdx.x;
    const uint32_t n = ref_digits(tot[s].t_total) + 2;
    uint32_t *d = S + p.dig_off[s];
    for (uint32_t i = threadIdx.x; i < n; i += 256) d[i] = 0;
}

struct DeviceAdder {
    uint32_t *S;
    __device__ void store(uint32_t i, uint32_t v) { S[i] = v; }
    __device__ void add(uint32_t i, uint32_t v) { atomicAdd(&S[i], v); }
};

__global__ __launch_bounds__(256) void k_k1p_c(Plan p, uint32_t total_chunks, const uint8_t *res,
                                               const Stretch *st, const Entry *en, const SliceTotals *tot,
                                               uint32_t *S) {
    __shared__ CodeEntry codes[256];
    codes[threadIdx.x] = device_code_entry(threadIdx.x);
    __syncthreads();
    const uint32_t gc = blockIdx.x * 256 + threadIdx.x;
    if (gc >= total_chunks) return;
    const Stretch o = st[gc];
    if (o.first == kNone) return;
    const uint32_t slice = p.chunk_slice[gc];
    if (tot[slice].bad) return;
    DeviceAdder add{S + p.dig_off[slice]};
    c_stretch(res + p.res_off[slice], o, en[gc], gc - p.chunk_base[slice], codes, add);
}

// ------------------------------------------------------------------ phase D

// One workgroup per slice.  Thread 0 applies finish() to the exact final window; then the digit
// sums are normalised tile by tile from the last digit: each thread adds up kSeg digits with
// carry-in 0 and reports (carry-out, "all ones"), one thread chains the 256 segments, each
// thread fixes its segment up, and the bytes go out coalesced.
constexpr uint32_t kSeg = 33;                  // odd: thread t's digits start at LDS word 33 t (conflict-free)
constexpr uint32_t kTile = 256 * kSeg;

__global__ __launch_bounds__(256) void k_k1p_d(Plan p, const SliceTotals *tot, const uint32_t *S,
                                               uint8_t *out, const uint64_t *out_off, uint32_t *out_len,
                                               int32_t *status, uint32_t force_retry_every) {
    __shared__ uint32_t dig[kTile];
    __shared__ uint32_t seg_g[256], seg_cin[256];
    __shared__ uint64_t seg_gm[4], seg_pm[4];
    __shared__ uint32_t sh_carry;
    const uint32_t s = blockIdx.x, t = threadIdx.x;
    if (status[s] != AVR_SLICE_OK) { if (t == 0) out_len[s] = 0; return; }
    const SliceTotals T = tot[s];
    // force_retry_every (test switch AVR_K1P_FORCE_RETRY=n, 0 = off): every n-th slice is handed to the serial
    // kernel as if phase D had met the carry pattern it does not resolve -- the hand-over is then proven on
    // every run of the tests, not only when that pattern occurs
    if (T.bad || (force_retry_every && s % force_retry_every == 0)) { if (t == 0) status[s] = AVR_SLICE_RETRY_SERIAL; return; }
    const uint32_t *Ss = S + p.dig_off[s];
    uint8_t *o = out + out_off[s];
    const uint32_t cap = uint32_t(out_off[s + 1] - out_off[s]);
    const uint32_t nd = ref_digits(T.t_total);
    if (t == 0) {
        const uint32_t low = uint32_t((uint64_t(Ss[nd]) << 15) + (Ss[nd + 1] >> 1));
        const uint32_t range = T.r_final << (22 - T.t_total + 16 * nd);
        uint8_t tail[5];
        uint32_t carry;
        const uint32_t n_tail = d_finish(low, range, tail, &carry);
        for (uint32_t k = 0; k < n_tail; k++) {
            if (2 * nd + k < cap) o[2 * nd + k] = tail[k];
        }
        sh_carry = carry;
        out_len[s] = 2 * nd + n_tail;
        if (2 * nd + n_tail > cap) status[s] = AVR_SLICE_OVERFLOW;
    }
    __syncthreads();
    uint32_t carry_in = sh_carry;                        // into the last digit of the current tile
    for (uint32_t hi = nd; hi > 0;) {                    // tiles from the low-order end: digits [lo, hi)
        const uint32_t lo = hi > kTile ? hi - kTile : 0, cnt = hi - lo;
        for (uint32_t i = t; i < cnt; i += 256) dig[i] = Ss[lo + i];
        __syncthreads();
        // segment t covers tile digits [a, b); the LAST segment (highest t) is the low-order end
        const uint32_t a = t * kSeg < cnt ? t * kSeg : cnt, b = (t + 1) * kSeg < cnt ? (t + 1) * kSeg : cnt;
        uint32_t c = 0, all_ones = 1;
        for (uint32_t i = b; i-- > a;) {
            const uint32_t v = dig[i] + c;
            dig[i] = v & 0xffffu;
            c = v >> 16;
            all_ones &= (v & 0xffffu) == 0xffffu;
        }
        // Carries run from segment 255 (the low-order end) down to segment 0.  Digit sums overlap where
        // stretches meet, so they are small integers, not bits: segment k sends on what it made itself
        // (c), plus one if it is all ones and receives anything.  WHETHER a segment receives anything
        // is a carry chain over (generate = c > 0, propagate = all ones) -- solved for 64 segments at
        // a time by one 64-bit addition of the two lane masks, bit-reversed so that it runs upward.
        const bool gen = c > 0, prop = all_ones && !gen;
        const uint64_t gm = __builtin_bitreverse64(__ballot(gen)), pm = __builtin_bitreverse64(__ballot(prop));
        if ((t & 63) == 0) { seg_gm[t >> 6] = gm; seg_pm[t >> 6] = pm; }
        __syncthreads();
        uint32_t into = carry_in > 0;                    // into segment 255, then into each wave's top segment
        bool recv = false;
        for (uint32_t v = 4; v-- > 0;) {
            const uint64_t a = seg_gm[v] | seg_pm[v], b2 = seg_gm[v], sum = a + b2 + into;
            if (v == (t >> 6)) recv = (__builtin_bitreverse64(sum ^ a ^ b2) >> (t & 63)) & 1;   // carry into my segment
            into = uint32_t(((a & b2) | ((a | b2) & ~sum)) >> 63);
        }
        seg_g[t] = c + ((all_ones && recv) ? 1u : 0u);   // what segment t sends to segment t - 1
        __syncthreads();
        // the tile's own carry-in goes to its last segment that has digits (a partial tile -- the
        // highest-order one -- leaves the segments behind it empty; they only pass the chain on)
        const uint32_t n_seg = (cnt + kSeg - 1) / kSeg;
        seg_cin[t] = t + 1 >= n_seg ? carry_in : seg_g[t + 1];
        if (t == 0) sh_carry = seg_g[0];                 // into the next (higher-order) tile
        __syncthreads();
        if (t < n_seg && seg_cin[t]) {
            uint32_t c2 = seg_cin[t];
            for (uint32_t i = b; i-- > a && c2;) {
                const uint32_t v = dig[i] + c2;
                dig[i] = v & 0xffffu;
                c2 = v >> 16;
            }
            // What was sent on assumed that only an all-ones segment overflows when it receives.  A
            // carry of 2 or more into a segment of the shape ffff ... ffff fffe would too (it takes two
            // overlapping windows and 33 particular digits); then the slice is handed to the serial kernel.
            if (c2 != ((all_ones && recv) ? 1u : 0u)) status[s] = AVR_SLICE_RETRY_SERIAL;
        }
        __syncthreads();
        for (uint32_t i = t; i < cnt; i += 256) {
            const uint32_t v = dig[i], at = 2 * (lo + i);
            if (at + 1 < cap) *reinterpret_cast<uint16_t *>(o + at) = uint16_t((v >> 8) | (v << 8));
        }
        carry_in = sh_carry;
        hi = lo;
        __syncthreads();
    }
}

// ------------------------------------------------------------------ serial coder from resolved codes
//
// One lane per slice over one-byte resolved codes: each code IS the (symbol, *state) pair cabac::encoder::put
// takes (cabac_code.h:33), so this is cabac_code.h:33-67 on arithmetic_code.h:106-126 with no state table at
// all -- the per-code entry (LPS ranges of the state's four range quarters, coded symbol) comes from a
// 256-entry LDS table.  Two uses: the slices phase D hands back (want_status = AVR_SLICE_RETRY_SERIAL: the
// carry pattern of k_k1p_d, or the test switch), and batches of many short slices, where one lane per
// slice fills the chip and the per-chunk machinery of K1p would be all overhead.
// Slice i's codes are at codes + res_off[i] (16-byte aligned, readable up to the next multiple of 16).
__global__ __launch_bounds__(64) void k_cabac_encode_codes(const uint8_t *codes_in, const uint64_t *res_off, const uint32_t *n_bins,
                                                           const uint32_t *order, uint32_t n_slices, uint8_t *out,
                                                           const uint64_t *out_off, uint32_t *out_len, int32_t *status,
                                                           int32_t want_status) {
    __shared__ CodeEntry codes[256];
    for (uint32_t c = threadIdx.x; c < 256; c += 64) codes[c] = device_code_entry(c);
    __syncthreads();
    const uint32_t g = blockIdx.x * 64 + threadIdx.x;
    if (g >= n_slices) return;
    const uint32_t slice = order ? order[g] : g;
    if (status[slice] != want_status) {
        if (want_status == AVR_SLICE_OK) out_len[slice] = 0;
        return;
    }
    const uint32_t n = n_bins[slice];
    const uint8_t *res = codes_in + res_off[slice];
    const uint64_t o0 = out_off[slice];
    const uint32_t cap = uint32_t(out_off[slice + 1] - o0);
    CabacEncoder e;
    e.init(0x7F800000u, out + o0, cap);                          // cabac_code.h:30
    auto bin = [&](uint32_t c) {
        const CodeEntry ce = codes[c];
        const int norm = 23 - __builtin_clz(e.range);            // cabac_code.h:37, 70-79
        const uint32_t q = (e.range >> (norm + 6)) & 3;          // :39-40
        const uint32_t r_tab = ((ce.row >> (8 * q)) & 0xffu) << norm;            // :40-41; put_terminate :59-60 is the row of pStateIdx 63
        const uint32_t r1 = (ce.meta >> 8) ? e.range >> 1 : r_tab;               // put_bypass :52-54
        const uint32_t sym = (ce.meta >> 1) & 1u;                // the coded symbol (for a bypass bin: the bin)
        const uint32_t r0 = e.range - r1;                        // arithmetic_code.h:107-114
        e.low += sym ? r0 : 0u;
        e.range = sym ? r1 : r0;
        if (e.range < 0x200u) e.emit_digit();                    // :115-122
    };
    uint32_t i = 0;
    if (n >= 16) {
        U4 v = *reinterpret_cast<const U4 *>(res);
        for (; i + 16 <= n; i += 16) {                           // the next group is in flight while this one is coded
            const U4 nx = i + 32 <= n ? *reinterpret_cast<const U4 *>(res + i + 16) : v;
            uint32_t w0 = v.x, w1 = v.y, w2 = v.z, w3 = v.w;
#pragma unroll 1
            for (uint32_t k = 0; k < 4; k++) {
                const uint32_t d = w0;
                w0 = w1; w1 = w2; w2 = w3;
                bin(d & 0xffu); bin((d >> 8) & 0xffu); bin((d >> 16) & 0xffu); bin(d >> 24);
            }
            v = nx;
        }
    }
    for (; i < n; i++) bin(res[i]);
    e.finish();                                                  // cabac_code.h:63-65 / ~encoder(), arithmetic_code.h:100
    e.w.flush();
    out_len[slice] = e.w.n;
    status[slice] = e.w.n > cap ? AVR_SLICE_OVERFLOW : AVR_SLICE_OK;
}

// ------------------------------------------------------------------ launcher

namespace {
inline uint64_t up256(uint64_t x) { return (x + 255) & ~uint64_t(255); }
}

// Phase A: records + initial states -> resolved codes `res` (slice i at res + res_off[i]).
// `w` is workspace for the sort (sorted bins, histograms, run starts, segments), laid out for the
// caller's context count; the kernels index it by the dense count, which is known after the histogram
// pass (the one host round trip of the path: four bytes, to size the later launches).
struct ResolveLayout {
    uint64_t sorted, hist16, boff, run_start, seg, entry, qoff, est, meta, total;
};
static inline ResolveLayout resolve_layout(size_t n_slices, uint32_t ns, const avr_chunk_plan *pl) {
    ResolveLayout L;
    uint64_t at = 0;
    auto take = [&](uint64_t bytes) { const uint64_t o = at; at += up256(bytes); return o; };
    L.sorted = take(pl->res_total + 32);
    L.hist16 = take(uint64_t(pl->total_blocks) * ns * 2 + 16);
    L.boff = take(uint64_t(pl->total_blocks) * ns * 4 + 16);
    L.run_start = take(n_slices * uint64_t(ns + 1) * 4 + 16);
    L.seg = take(uint64_t(pl->total_chunks) * sizeof(Seg));
    L.entry = take(uint64_t(pl->total_chunks) + 16);
    L.qoff = take(uint64_t(pl->total_blocks) * ns * 16 + 16);
    L.est = take(uint64_t(pl->total_chunks) * ((ns + 3) / 4) * 4 + 16);
    L.meta = take(256 + 2048 + 2048);                            // used[32] + n_dense, table[1024], index[1024]
    L.total = at;
    return L;
}

static hipError_t launch_resolve(hipStream_t s, Plan p, uint32_t n_slices, const uint8_t *init_states,
                                 const avr_chunk_plan *pl, uint8_t *w, uint8_t *res, int32_t *status, uint8_t *final_states) {
    const uint32_t ns = p.ns_full;
    const ResolveLayout L = resolve_layout(n_slices, ns, pl);
    uint8_t *sorted = w + L.sorted;
    uint16_t *hist16 = reinterpret_cast<uint16_t *>(w + L.hist16);
    uint32_t *boff = reinterpret_cast<uint32_t *>(w + L.boff);
    uint32_t *run_start = reinterpret_cast<uint32_t *>(w + L.run_start);
    Seg *seg = reinterpret_cast<Seg *>(w + L.seg);
    uint8_t *entry = w + L.entry;
    uint32_t *qoff = reinterpret_cast<uint32_t *>(w + L.qoff);
    uint32_t *est = reinterpret_cast<uint32_t *>(w + L.est);
    uint32_t *used = reinterpret_cast<uint32_t *>(w + L.meta);   // [32], then n_dense
    uint32_t *n_dense = used + 32;
    uint16_t *table = reinterpret_cast<uint16_t *>(w + L.meta + 256), *index = table + 1024;
    p.table = table;
    p.index = index;
    hipError_t e;
    if ((e = hipMemsetAsync(used, 0, 256, s)) != hipSuccess) return e;
    if (final_states && ns && (e = hipMemcpyAsync(final_states, init_states, size_t(n_slices) * ns, hipMemcpyDeviceToDevice, s)) != hipSuccess)
        return e;                                                // contexts without bins keep their state
    hipLaunchKernelGGL(k_k1p_hist, dim3(pl->total_blocks), dim3(256), 0, s, p, status, hist16, used);
    hipLaunchKernelGGL(k_k1p_densemap, dim3(1), dim3(1024), 0, s, used, table, index, n_dense);
    uint32_t n_states = 0;
    if ((e = hipMemcpyAsync(&n_states, n_dense, 4, hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
    p.n_states = n_states;
    uint32_t key_bits = 0;
    while ((1u << key_bits) < n_states) key_bits++;
    const uint32_t chunk_blocks = (pl->total_chunks + 255) / 256;
    if (n_states > 0) {
        hipLaunchKernelGGL(k_k1p_scan, dim3(n_slices), dim3(1024), 0, s, p, hist16, boff, run_start);
        const uint32_t scatter_lds = 20 * ((n_states + 63) & ~63u) + 3 * kSortBlock;
        auto scatter = key_bits <= 1 ? k_k1p_scatter<1> : key_bits == 2 ? k_k1p_scatter<2> : key_bits == 3 ? k_k1p_scatter<3> :
                       key_bits == 4 ? k_k1p_scatter<4> : key_bits == 5 ? k_k1p_scatter<5> : key_bits == 6 ? k_k1p_scatter<6> :
                       key_bits == 7 ? k_k1p_scatter<7> : key_bits == 8 ? k_k1p_scatter<8> : key_bits == 9 ? k_k1p_scatter<9> :
                                                                                               k_k1p_scatter<10>;
        hipLaunchKernelGGL(scatter, dim3(pl->total_blocks), dim3(256), scatter_lds, s, p, status, boff, run_start, sorted, qoff);
        hipLaunchKernelGGL(k_k1p_spec, dim3(chunk_blocks), dim3(256), 0, s, p, pl->total_chunks, status, run_start, init_states,
                           sorted, seg);
        hipLaunchKernelGGL(k_k1p_link, dim3(chunk_blocks), dim3(256), 0, s, p, pl->total_chunks, status, run_start, init_states,
                           sorted, seg, entry);
        hipLaunchKernelGGL(k_k1p_chain, dim3(chunk_blocks), dim3(256), 0, s, p, pl->total_chunks, status, run_start, init_states,
                           sorted, seg, entry, final_states);
        const uint32_t nkw = (n_states + 3) / 4, cpb = 256 / nkw;           // nkw <= 256
        hipLaunchKernelGGL(k_k1p_entry, dim3((pl->total_chunks + cpb - 1) / cpb), dim3(256), 0, s, p, pl->total_chunks, status, qoff,
                           sorted, init_states, est, cpb, (65536 + nkw - 1) / nkw);
    }
    // the waves of a workgroup share the two tables; each has its own state rows: as many waves as 60 KiB hold (1 .. 4)
    const uint32_t per_wave = ((n_states + 8) / 4) * 256;
    const uint32_t replay_waves = per_wave * 4 <= 60 * 1024 ? 4 : per_wave * 2 <= 60 * 1024 ? 2 : 1;
    const uint32_t replay_lds = kReplayTable * 4 + replay_waves * per_wave;
    if (replay_lds > 60 * 1024) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_k1p_replay), hipFuncAttributeMaxDynamicSharedMemorySize, int(replay_lds));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_k1p_replay, dim3((pl->total_chunks + 64 * replay_waves - 1) / (64 * replay_waves)), dim3(64 * replay_waves),
                       replay_lds, s, p, pl->total_chunks, est, res, status);
    return hipGetLastError();
}

// Phases B-D: resolved codes -> bytes.  `w` is workspace for stretches, entries, totals, digit sums.
static hipError_t launch_code(hipStream_t s, const Plan &p, uint32_t n_slices, const avr_chunk_plan *pl, uint8_t *w,
                              const uint8_t *res, uint32_t max_stretch, uint8_t *out, const uint64_t *out_off,
                              uint32_t *out_len, int32_t *status) {
    Stretch *st = reinterpret_cast<Stretch *>(w);            w += up256(uint64_t(pl->total_chunks) * sizeof(Stretch));
    Entry *en = reinterpret_cast<Entry *>(w);                w += up256(uint64_t(pl->total_chunks) * sizeof(Entry));
    SliceTotals *tot = reinterpret_cast<SliceTotals *>(w);   w += up256(n_slices * sizeof(SliceTotals));
    uint32_t *S = reinterpret_cast<uint32_t *>(w);
    const uint32_t chunk_blocks = (pl->total_chunks + 255) / 256;
    hipLaunchKernelGGL(k_k1p_b1, dim3(chunk_blocks), dim3(256), 0, s, p, pl->total_chunks, res, status, st, max_stretch);
    hipLaunchKernelGGL(k_k1p_b2, dim3(n_slices), dim3(256), 0, s, p, status, st, en, tot);
    hipLaunchKernelGGL(k_k1p_zero, dim3(n_slices), dim3(256), 0, s, p, tot, S);
    hipLaunchKernelGGL(k_k1p_c, dim3(chunk_blocks), dim3(256), 0, s, p, pl->total_chunks, res, st, en, tot, S);
    uint32_t force_retry_every = 0;                              // test switch, see k_k1p_d
    if (const char *f = getenv("AVR_K1P_FORCE_RETRY")) force_retry_every = uint32_t(strtoul(f, nullptr, 10));
    hipLaunchKernelGGL(k_k1p_d, dim3(n_slices), dim3(256), 0, s, p, tot, S, out, out_off, out_len, status, force_retry_every);
    return hipGetLastError();
}

hipError_t launch_cabac_encode_codes(hipStream_t s, const uint8_t *codes, const uint64_t *res_off, const uint32_t *n_bins,
                                     const uint32_t *order, uint32_t n_slices, uint8_t *out, const uint64_t *out_off,
                                     uint32_t *out_len, int32_t *status, int32_t want_status) {
    if (n_slices == 0) return hipSuccess;
    hipLaunchKernelGGL(k_cabac_encode_codes, dim3((n_slices + 63) / 64), dim3(64), 0, s, codes, res_off, n_bins, order, n_slices,
                       out, out_off, out_len, status, want_status);
    return hipGetLastError();
}

static inline uint64_t resolve_ws_bytes(size_t n_slices, uint32_t n_states, const avr_chunk_plan *pl) {
    return resolve_layout(n_slices, n_states, pl).total;
}

size_t k1p_code_workspace_bytes(size_t n_slices, const avr_chunk_plan *pl);
size_t k1p_workspace_bytes(size_t n_slices, uint32_t n_states, const avr_chunk_plan *pl) {
    return size_t(up256(pl->res_total + 32) + resolve_ws_bytes(n_slices, n_states, pl)) + k1p_code_workspace_bytes(n_slices, pl);
}

hipError_t launch_k1p(hipStream_t s, const uint16_t *recs, const uint64_t *rec_off, const uint32_t *n_bins,
                      uint32_t n_slices, const uint8_t *init_states, uint32_t n_states, const avr_chunk_plan *pl,
                      void *workspace, uint8_t *out, const uint64_t *out_off, uint32_t *out_len, int32_t *status,
                      uint8_t *final_states) {
    if (n_slices == 0) return hipSuccess;
    uint8_t *w = static_cast<uint8_t *>(workspace);
    uint8_t *res = w;                                        w += up256(pl->res_total + 32);
    const Plan p{recs, rec_off, n_bins, pl->res_off, pl->chunk_base, pl->chunk_slice, pl->blk_base, pl->blk_slice,
                 pl->dig_off, 0, n_states, nullptr, nullptr};
    hipError_t e = launch_resolve(s, p, n_slices, init_states, pl, w, res, status, final_states);
    if (e != hipSuccess) return e;
    w += resolve_ws_bytes(n_slices, n_states, pl);
    e = launch_code(s, p, n_slices, pl, w, res, kMaxStretch, out, out_off, out_len, status);
    if (e != hipSuccess) return e;
    // slices the scheme declined (status AVR_SLICE_RETRY_SERIAL) are coded by the serial kernel
    return launch_cabac_encode(false, s, recs, rec_off, n_bins, nullptr, n_slices, init_states, n_states, out, out_off,
                               out_len, status, nullptr, AVR_SLICE_RETRY_SERIAL);
}

// The two stages on their own: phase A into a caller-owned code buffer ...
size_t k1p_resolve_workspace_bytes(size_t n_slices, uint32_t n_states, const avr_chunk_plan *pl) {
    return size_t(resolve_ws_bytes(n_slices, n_states, pl));
}
hipError_t launch_k1p_resolve(hipStream_t s, const uint16_t *recs, const uint64_t *rec_off, const uint32_t *n_bins,
                              uint32_t n_slices, const uint8_t *init_states, uint32_t n_states, const avr_chunk_plan *pl,
                              void *workspace, uint8_t *codes, int32_t *status, uint8_t *final_states) {
    if (n_slices == 0) return hipSuccess;
    const Plan p{recs, rec_off, n_bins, pl->res_off, pl->chunk_base, pl->chunk_slice, pl->blk_base, pl->blk_slice,
                 pl->dig_off, 0, n_states, nullptr, nullptr};
    return launch_resolve(s, p, n_slices, init_states, pl, static_cast<uint8_t *>(workspace), codes, status, final_states);
}
// ... and phases B-D from resolved codes (no stretch is declined for its length here: a stretch without an
// LPS is simply walked to its end by one lane); a slice phase D hands back is coded by k_cabac_encode_codes
size_t k1p_code_workspace_bytes(size_t n_slices, const avr_chunk_plan *pl) {
    return size_t(up256(uint64_t(pl->total_chunks) * sizeof(Stretch)) + up256(uint64_t(pl->total_chunks) * sizeof(Entry)) +
                  up256(n_slices * sizeof(SliceTotals)) + up256(pl->dig_total * 4 + 16));
}
hipError_t launch_k1p_code(hipStream_t s, const uint8_t *codes, const uint32_t *n_bins, uint32_t n_slices,
                           const avr_chunk_plan *pl, void *workspace, uint8_t *out, const uint64_t *out_off,
                           uint32_t *out_len, int32_t *status) {
    if (n_slices == 0) return hipSuccess;
    const Plan p{nullptr, nullptr, n_bins, pl->res_off, pl->chunk_base, pl->chunk_slice, pl->blk_base, pl->blk_slice,
                 pl->dig_off, 0, 0, nullptr, nullptr};
    hipError_t e = launch_code(s, p, n_slices, pl, static_cast<uint8_t *>(workspace), codes, 0xffffffffu, out, out_off, out_len, status);
    if (e != hipSuccess) return e;
    return launch_cabac_encode_codes(s, codes, pl->res_off, n_bins, nullptr, n_slices, out, out_off, out_len, status,
                                     AVR_SLICE_RETRY_SERIAL);
}

}  // namespace avr
