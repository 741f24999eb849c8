// K1p kernels: intra-slice parallel CABAC encode for batches of few, long slices.
// The algorithm and the per-lane functions of phases B-D are in avr_k1p.h; this file maps
// them to lanes and adds phase A (context-state resolution; see "phase A" below):
//
//   k_k1p_census    A   32 blocks of 4096 bins / WG   which contexts the batch uses, from a 1-in-16 sample of the cache lines
//   k_k1p_densemap  A   one workgroup                dense numbering of those contexts
//   k_k1p_tn        A   thread per table entry       state after n = 0..8 bins, per state and bin pattern
//   k_k1p_local     A   lane per chunk               counting sort of the chunk's bins by context (lane-serial); validates every record
//   k_k1p_ctxchain  A   lane per (slice, context)    state chain through the slice -> state of every context per chunk
//   k_k1p_replay    A+B1 lane per chunk              resolved code of every bin, in stream order, and the
//                                                    chunk's stretch summary for the 4 entry quarters
//   k_k1p_b1        B1  lane per chunk               the same summaries from finished codes (resolved-code entry)
//   k_k1p_b2        B2  workgroup per slice          chain the summaries: entry range + bit position; zero the digit sums
//   k_k1p_c         C   lane per chunk               code each stretch, add its digits
//   k_k1p_d         D   workgroup per slice          finish(), carries (segmented), bytes
//   k_cabac_encode_codes  lane per slice             serial coder from resolved codes (hand-over of phase D; short slices)
//
// Input is the slice-major record layout (a slice's bins must be consecutive for the chunk lanes).
// Results are byte-identical to k_cabac_encode (tests/test_gpu_k1p.py); a slice the scheme declines
// (no coded LPS for 16 chunks) is coded by k_cabac_encode itself.
#include <hip/hip_runtime.h>

#include <mutex>
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

// The kernels' own buffer of resolved codes, wave-interleaved: 16-byte group g (of 64) of global chunk gc is at
//   base + (((gc / 64) * 64 + g) * 64 + gc % 64) * 16,
// so the 64 lanes of a wave -- 64 consecutive chunks -- store (k_k1p_replay) and load (k_k1p_c) group g as one
// contiguous kilobyte, where slice-major codes cost a cache line per lane per 16-byte access.  chunk0 = the slice's
// first global chunk; i = index of a code in the slice.  (The public two-stage entry points keep slice-major codes.)
struct TileCodes {
    const uint8_t *base;
    uint32_t chunk0;
    __device__ __forceinline__ size_t at(uint32_t i) const {
        const uint32_t gc = chunk0 + (i >> 10), g = (i >> 4) & 63u;
        return ((size_t(gc >> 6) * 64 + g) * 64 + (gc & 63u)) * 16;
    }
    __device__ __forceinline__ U4 load16(uint32_t i) const { return *reinterpret_cast<const U4 *>(base + at(i)); }
    __device__ __forceinline__ uint32_t byte(uint32_t i) const { return base[at(i) + (i & 15u)]; }
};
__host__ __device__ inline uint64_t tile_codes_bytes(uint32_t total_chunks) { return (uint64_t(total_chunks) + 63) / 64 * 64 * 1024; }

// ------------------------------------------------------------------ phase A
//
// Context states evolve per context, independent of low / range (cabac_code.h:43-47): the state a bin is
// coded in is a function of the bins the same context coded before it.  A slice is cut into chunks of
// kChunk bins; what phase A has to find is the state of every context at the start of every chunk
// (`est`), from which one lane per chunk replays its bins in stream order (k_k1p_replay).
//
//   k_k1p_census    32 blocks of 4096 bins    which contexts the batch uses (from a sample)
//   k_k1p_densemap  one workgroup             dense numbering of those contexts
//   k_k1p_local     lane per chunk            counting sort of the chunk's bins by context, lane-serial: per
//                                             context the number of bins, and the bins themselves as a bit string
//   k_k1p_ctxchain  lane per (slice, context) the context's state chain through the slice, chunk by chunk over
//                                             those bit strings, eight bins per table look-up -> est
//   k_k1p_replay    lane per chunk            resolved code of every bin, in stream order
//
// Why this shape.  A 1080p stream spreads its bins over ~90 contexts, the hottest of which gets 5 % of them:
// about 50 bins of a 1024-bin chunk, far too few for any shortcut through the state machine, so each
// context's bins have to be walked in order -- the chain is serial per (slice, context), 512 x 86 of them in
// BASELINE.json's configs[1], and short per chunk.  What it needs is every context's bins side by side:
// a sort by context.  Round 1 sorted whole slices (a wave ranks 64 bins with one ballot per key bit: 1.1
// vector instructions per bin, three passes over the records, four more over the sorted copy).  Sorting
// inside the chunk instead lets ONE LANE do it with private counters in LDS -- a count pass and a
// placement pass of a dozen instructions per bin at full SIMD width, no cross-lane ranking at all -- and
// the result is tiny: 1024 bits and one 16-bit end position per context per chunk.

// Which contexts the batch uses: one bit per context that occurs (`used`), under the caller's numbering (the offset
// of the state byte in cabac_state[], recode.cpp:325: up to 1024, of which a stream touches few).  `stride` > 1: from
// a sample -- of each block of kSortBlock bins (64 cache lines of 64 records) the lines l with l % stride ==
// block % stride, whole lines so that the sample moves 1/stride of the bytes.  The
// renumbering exists for the LDS footprint of the kernels behind it, so it has to hold the contexts that matter, not
// all of them: k_k1p_local, which looks every record up anyway, finds the slices with a bin in a context the sample
// missed (and the records that are no record at all), and those slices take a second pass with a full count.
// kCensusBlocks blocks to a workgroup, eight threads (one cache line per trip) to a block.
constexpr uint32_t kCensusBlocks = 32;
__global__ __launch_bounds__(256) void k_k1p_census(Plan p, uint32_t total_blocks, const int32_t *status, uint32_t *used, uint32_t stride) {
    __shared__ uint8_t flag[AVR_MAX_STATES];
    __shared__ uint32_t bm[32];
    for (uint32_t k = threadIdx.x; k < AVR_MAX_STATES / 4; k += 256) reinterpret_cast<uint32_t *>(flag)[k] = 0;
    if (threadIdx.x < 32) bm[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t b = blockIdx.x * kCensusBlocks + (threadIdx.x >> 3), t = threadIdx.x & 7, nk = p.ns_full;
    if (b < total_blocks) {
        const uint32_t s = p.blk_slice[b];
        if (status[s] == AVR_SLICE_OK) {
            const uint32_t n = p.n_bins[s], i0 = (b - p.blk_base[s]) * kSortBlock;
            const uint32_t i1 = i0 + kSortBlock < n ? i0 + kSortBlock : n;
            const uint16_t *r = p.recs + p.rec_off[s];
            // a slice's padding records are no-ops and i0 is a multiple of 8: whole groups can be read up to the padded end
            for (uint32_t l = b % stride; l < kSortBlock / 64; l += stride) {
                const uint32_t i = i0 + l * 64 + t * 8;
                if (i >= i1) continue;
                const uint4 v = *reinterpret_cast<const uint4 *>(r + i);
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (uint32_t j = 0; j < 8; j++) {
                    const uint32_t sel = ((w[j >> 1] >> ((j & 1) * 16)) & 0xffffu) >> 1;       // keeps bits 12..15: no context then
                    if (i + j < i1 && sel < nk) flag[sel] = 1;   // plain store: every writer writes the same value
                }
            }
        }
    }
    __syncthreads();
    {
        const uint32_t f = reinterpret_cast<const uint32_t *>(flag)[threadIdx.x];          // contexts 4t .. 4t+3
        const uint32_t nib = (f & 1u) | ((f >> 7) & 2u) | ((f >> 14) & 4u) | ((f >> 21) & 8u);
        if (nib) atomicOr(&bm[threadIdx.x >> 3], nib << ((threadIdx.x & 7) * 4));
    }
    __syncthreads();
    // bits only ever get set, so a (possibly stale) plain read tells which are still missing: after the
    // first few blocks nothing is, and no block touches the shared words any more
    if (threadIdx.x < 32) {
        const uint32_t mine = bm[threadIdx.x];
        if (mine & ~__hip_atomic_load(&used[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(&used[threadIdx.x], mine);
    }
}

// OK <-> DONE around the second pass: a -> b and c -> d in one sweep
__global__ __launch_bounds__(256) void k_k1p_swap(int32_t *status, uint32_t n, int32_t a, int32_t b, int32_t c, int32_t d) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int32_t v = status[i];
    if (v == a) status[i] = b;
    else if (v == c) status[i] = d;
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
    if (k >= sc[1023]) index[k] = uint16_t(kNotUsed);            // rows a launch sized by a guess has beyond the count: no context
    if (k == 1023) *n_dense = sc[1023];
}

// State after n = 0 .. 8 bins, per state and bin pattern (first bin in the low bit; cabac_code.h:43-47):
//   tn[(128 << n) - 128 + (st << n | bits)],  bits < 2^n                           (kTnBytes in all)
constexpr uint32_t kTnBytes = 128 * 511;
__global__ __launch_bounds__(256) void k_k1p_tn(uint8_t *tn) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= kTnBytes) return;
    const uint32_t n = 31 - __clz((i >> 7) + 1), j = i - ((128u << n) - 128u);
    uint32_t st = j >> n;
    for (uint32_t b = 0; b < n; b++) {
        const uint32_t nx = d_tables.packed[st][1], bin = (j >> b) & 1u;
        st = ((bin ^ st) & 1u) ? (nx >> 8) & 0xffu : nx & 0xffu;
    }
    tn[i] = uint8_t(st);
}

// A lane's walk over the 16-byte record groups [i0, i1) of its chunk (8 records a group; a slice's records are padded with no-ops to
// a whole group): a cache line of 64 bytes a trip, the next trip's line in flight meanwhile.  The loads of the next line are
// UNCONDITIONAL -- past the chunk's last whole line the address is clamped to that line (a hit) instead of the load being skipped: hipcc
// puts a load inside a branch behind an exec mask and waits for it (s_waitcnt vmcnt) before the branch's end, which turned "the next
// line in flight" into "every trip waits out a memory latency" in rounds 1-3 (tools/ubench/read_patterns: 1.4 against 4.3 TB/s for
// this very pattern at ten waves a CU; more than one line ahead is slower again: 2.4 TB/s with two, 1.0 with four).
template <class F>
__device__ __forceinline__ void for_record_groups(const uint16_t *r, uint32_t i0, uint32_t i1, F &&f) {
    uint32_t i = i0;
    if (i + 32 <= i1) {
        const uint32_t last = i0 + ((i1 - i0) / 32u - 1u) * 32u;                 // where the chunk's last whole line starts
        const U4 *q = reinterpret_cast<const U4 *>(r + i);
        U4 v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3];
        for (; i + 32 <= i1; i += 32) {
            const U4 *qn = reinterpret_cast<const U4 *>(r + (i + 32 <= last ? i + 32 : last));
            const U4 n0 = qn[0], n1 = qn[1], n2 = qn[2], n3 = qn[3];
            f(v0); f(v1); f(v2); f(v3);
            v0 = n0; v1 = n1; v2 = n2; v3 = n3;
        }
    }
    for (; i < i1; i += 8) f(*reinterpret_cast<const U4 *>(r + i));
}

// One lane per chunk: counting sort of the chunk's context bins by (dense) context, in LDS, serially.
//   pass 1  count the bins of every context                    cnt[k][lane]++           (16-bit, lane-private)
//   scan    exclusive prefix over the contexts                 cnt[k][lane] = first position of context k
//   pass 2  place: position = cnt[k][lane]++, the bin goes to bit `position` of the lane's 1024-bit string
// after which cnt[k][lane] is where context k's bins END (context k's bins are bits [end[k-1], end[k])).
// Out: lbits[gc][32 dwords], lend[gc][nk] (16-bit), both written through a transposed read of the LDS
// arrays so that a wave's stores are contiguous.
//
// The counters of contexts 2j and 2j+1 share a dword (a chunk has at most 1024 bins: no carry from the low
// half), and every update is an LDS atomic on the lane's own dword: ds_add_u32 without return in pass 1,
// ds_add_rtn_u32 in pass 2, ds_or_b32 for the bit.  Nothing else touches those dwords, so "atomic" only means
// that the LDS does the read-modify-write itself, in issue order: the eight bins of a 16-byte group are
// eight independent instructions in flight, where a load / add / store per bin is a chain of LDS round
// trips (the compiler must keep them in order: two bins of a group may hit the same counter).
// Row j of `cnt` and row j of `bits` are 64 lanes wide: the data-dependent accesses never conflict across
// lanes.  What a record's 11-bit selector means is one look-up: sel_tab[selector] = byte offset of the
// counter row | 16 for the high half.  Rows past the contexts: nk = terminate bins (sorted like a context: their
// values end up side by side behind the last context's), nk + 1 = bypass and padding, nk + 2 = a context of the
// slice that has no dense id (the sampled census missed it), nk + 3 = no selector of the slice at all.  The last
// three count from position 1024, i.e. land in a spare row of `bits`: no branch on the bin kind anywhere.
//
// This is also where every record of the path is examined (the census only samples).  A lane flags its slice
//   AVR_SLICE_BAD_RECORD    if row nk + 3 is not empty, a record has a bit above its selector set, or a
//                           put_terminate(1) is anywhere but last (the terminate row holds a 1 that is not the
//                           slice's last bin: a handful of bits to look at per chunk);
//   AVR_SLICE_RETRY_CENSUS  (internal) if row nk + 2 is not empty, counting such slices in *n_retry.
// SelT: uint16_t while the offsets fit (up to 500 contexts) -- with 86 contexts eight waves' rows and a table of 4 KiB are the
// CU's 160 KiB to the byte -- uint32_t beyond.
template <class SelT>
__global__ __launch_bounds__(512) void k_k1p_local(Plan p, uint32_t total_chunks, int32_t *status, uint32_t *lbits,
                                                   uint16_t *lend, uint32_t *n_retry) {
    extern __shared__ uint32_t local_lds[];                      // per wave: bits[33][64], then cnt[(nk + 5) / 2][64] (two 16-bit counters each)
    __shared__ SelT sel_tab[2048];
    const uint32_t nk = p.n_states, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t cnt_rows = (nk + 5) / 2;                      // contexts 0 .. nk-1 and the rows nk .. nk+3
    const uint32_t wave_dwords = (33 + cnt_rows) * 64;
    uint32_t *bits = local_lds + wv * wave_dwords;
    uint32_t *cnt = bits + 33 * 64;
    for (uint32_t sel = threadIdx.x; sel < 2048; sel += blockDim.x) {
        uint32_t k;
        if (sel < 1024u) {
            const uint32_t d = p.table[sel];
            k = d < nk ? d : sel < p.ns_full ? nk + 2 : nk + 3;
        } else k = sel == AVR_SEL_TERMINATE ? nk : (sel == AVR_SEL_BYPASS || sel == (AVR_NOP_CABAC >> 1)) ? nk + 1 : nk + 3;
        sel_tab[sel] = SelT((k >> 1) * 256u | (k & 1u) * 16u);
    }
    for (uint32_t i = lane; i < wave_dwords; i += 64) bits[i] = 0;
    __syncthreads();
    const uint32_t gc0 = (blockIdx.x * (blockDim.x >> 6) + wv) * 64, gc = gc0 + lane;
    uint32_t i0 = 0, i1 = 0, n = 0, s = 0;
    bool mine = false;                                           // the lane has a chunk of a live slice
    const uint16_t *r = p.recs;
    if (gc < total_chunks) {
        s = p.chunk_slice[gc];
        if (status[s] == AVR_SLICE_OK) {
            mine = true;
            n = p.n_bins[s];
            i0 = (gc - p.chunk_base[s]) * kChunk;
            i1 = i0 + kChunk < n ? i0 + kChunk : n;
            if (i0 > i1) i0 = i1;
            r = p.recs + p.rec_off[s];
        }
    }
    uint32_t *my_cnt = cnt + lane;                               // counters of contexts 2j, 2j+1 at my_cnt[64 j]
    uint32_t *my_bits = bits + lane;                             // dword j at my_bits[64 j]
    uint8_t *cnt_b = reinterpret_cast<uint8_t *>(my_cnt), *bits_b = reinterpret_cast<uint8_t *>(my_bits);
    // Visit the chunk's 16-byte groups (8 records; a slice's records are padded with no-ops to a whole group), a
    // cache line of records per trip with the next one in flight (see for_codes_all).
    auto for_groups = [&](auto &&f) { for_record_groups(r, i0, i1, f); };
    uint32_t high = 0;                                           // OR of all records: bits 12..15 must stay clear
    for_groups([&](const U4 &v) {                                // pass 1
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        high |= (w[0] | w[1]) | (w[2] | w[3]);
        uint32_t e[8];
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) e[j] = sel_tab[((w[j >> 1] >> ((j & 1) * 16)) >> 1) & 0x7ffu];
#pragma unroll
        for (uint32_t j = 0; j < 8; j++)                         // 1 << e: the shift takes the low five bits, 0 or 16
            __hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(cnt_b + (e[j] & ~255u)), 1u << (e[j] & 31u), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
    });
    auto count_of = [&](uint32_t k) { return (my_cnt[64 * (k >> 1)] >> (16 * (k & 1u))) & 0xffffu; };
    const uint32_t n_term = count_of(nk), n_missed = count_of(nk + 2), n_invalid = count_of(nk + 3) | (high & 0xf000f000u);
    uint32_t term_at = 0;                                        // where the terminate row starts
    {
        uint32_t run = 0;                                        // exclusive prefix over the contexts and the terminate row; the rest starts at 1024
        for (uint32_t j = 0; j < cnt_rows; j++) {
            const uint32_t c = my_cnt[64 * j], c0 = c & 0xffffu, c1 = c >> 16;
            const uint32_t s0 = 2 * j <= nk ? run : 1024u, s1 = 2 * j + 1 <= nk ? run + c0 : 1024u;
            if (2 * j == nk) term_at = s0;
            if (2 * j + 1 == nk) term_at = s1;
            my_cnt[64 * j] = s0 | s1 << 16;
            run += (2 * j <= nk ? c0 : 0u) + (2 * j + 1 <= nk ? c1 : 0u);
        }
    }
    for_groups([&](const U4 &v) {                                // pass 2
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t e[8], pos[8];
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) e[j] = sel_tab[((w[j >> 1] >> ((j & 1) * 16)) >> 1) & 0x7ffu];
#pragma unroll
        for (uint32_t j = 0; j < 8; j++)
            pos[j] = __hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(cnt_b + (e[j] & ~255u)), 1u << (e[j] & 31u), __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) {
            const uint32_t at = (pos[j] >> (e[j] & 31u)) & 2047u;            // < 1024: a context bin's place; the spare row counts from 1024
            const uint32_t row = at >> 5 < 32u ? at >> 5 : 32u;
            const uint32_t bin = (w[j >> 1] >> ((j & 1) * 16)) & 1u;
            __hip_atomic_fetch_or(reinterpret_cast<uint32_t *>(bits_b + row * 256u), bin << (at & 31u), __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    });
    if (mine) {
        uint32_t ones = 0;                                       // put_terminate(1) among the chunk's terminate bins
        for (uint32_t pos = term_at; pos < term_at + n_term;) {
            const uint32_t lo = pos & 31u, take = 32u - lo < term_at + n_term - pos ? 32u - lo : term_at + n_term - pos;
            const uint32_t mask = (take == 32u ? 0xffffffffu : (1u << take) - 1u) << lo;
            ones += __popc(my_bits[64 * (pos >> 5)] & mask);
            pos += take;
        }
        constexpr uint32_t kTerm1 = (AVR_SEL_TERMINATE << 1) | 1;
        const bool bad_term = ones > 1u || (ones == 1u && !(i1 == n && r[n - 1] == kTerm1));   // one 1: it is the last bin iff the last bin is one
        if (n_invalid || bad_term) status[s] = AVR_SLICE_BAD_RECORD;
        else if (n_missed && atomicCAS(&status[s], AVR_SLICE_OK, AVR_SLICE_RETRY_CENSUS) == AVR_SLICE_OK) atomicAdd(n_retry, 1u);
    }
    // out, transposed: flat element f of the wave's 64 rows <-> (chunk f / row, column f % row)
    {
        uint32_t *dst = lbits + size_t(gc0) * 32;
        const uint32_t lim = gc0 < total_chunks ? (total_chunks - gc0 < 64 ? total_chunks - gc0 : 64) * 32 : 0;
        for (uint32_t f = lane; f < lim; f += 64) dst[f] = bits[64 * (f & 31u) + (f >> 5)];
    }
    if (nk) {
        uint16_t *dst = lend + size_t(gc0) * nk;
        const uint32_t lim = gc0 < total_chunks ? (total_chunks - gc0 < 64 ? total_chunks - gc0 : 64) * nk : 0;
        uint32_t ch = lane / nk, k = lane - ch * nk;
        const uint32_t dch = 64 / nk, dk = 64 - dch * nk;
        for (uint32_t f = lane; f < lim; f += 64) {
            dst[f] = uint16_t(cnt[64 * (k >> 1) + ch] >> (16 * (k & 1u)));
            ch += dch; k += dk;
            if (k >= nk) { k -= nk; ch++; }
        }
    }
}

// One lane per (slice, context), a wave = kChainLanes consecutive contexts of one slice: the state chain of the
// context through the slice (cabac_code.h:43-47), chunk by chunk.  At chunk c the lane notes the state
// (est[gc][k]: what k_k1p_replay starts chunk c from), then walks the context's bins of the chunk -- bits
// [end[k-1], end[k]) of the chunk's bit string -- up to eight per look-up (k_k1p_tn's table, in LDS).  The loads
// are one row of `lend` and pieces of one 128-byte bit string per chunk; they do not depend on the state and
// run ahead of the chain: end positions four chunks ahead, the 64-bit window a run starts in two ahead.
// They are unconditional (the arrays are padded past the last chunk, and in front for k = 0): a load inside a
// branch is waited for at the end of the branch, which would put its whole latency into every step.
//
// The chain is a sequence of dependent look-ups and loads, chunk after chunk: what hides its latency is other
// waves, and a batch has only slices x contexts lanes to give (44 000 for 512 slices of a 1080p clip).  So a
// wave takes only kChainLanes of them -- the other lanes stay idle -- which also bounds a step by the longest
// of kChainLanes runs instead of the longest of 64.  (Measured and dropped: the batch in two halves on two streams,
// one kernel apart, so that one half's chains run under the other half's sort and replay.  Those kernels, one lane
// per chunk at under five waves per SIMD, are short of waves themselves: every kernel of a half got slower by more
// than the overlap gave back, 2.05 against 1.95 ms per step.  Also without effect: twice the read-ahead (end positions
// eight chunks ahead, windows four), and whole-byte look-ups without the arithmetic on n between them -- a step is the
// hottest lane's seven or so dependent LDS round trips, and neither its loads nor its VALU work.  The step as it stands --
// four look-ups without a branch, four more if any lane needs them, one shift-add between two look-ups -- is 3 % faster
// than the loops it replaced; taking out, for the measurement, the stores, or the window loads, changes nothing.)
constexpr uint32_t kChainLanes = 22, kChainWaves = 8;         // kChainLanes: the fewest lanes a wave takes
__global__ __launch_bounds__(64 * kChainWaves) void k_k1p_ctxchain(Plan p, uint32_t n_slices, uint32_t groups, uint32_t chain_lanes, const int32_t *status,
                                                      const uint8_t *tng, const uint32_t *lbits, const uint16_t *lend,
                                                      const uint8_t *init_states, uint8_t *est, uint8_t *final_states) {
    __shared__ uint8_t tn[kTnBytes];
    for (uint32_t i = threadIdx.x; i < kTnBytes / 16; i += 64 * kChainWaves) reinterpret_cast<uint4 *>(tn)[i] = reinterpret_cast<const uint4 *>(tng)[i];
    __syncthreads();
    const uint32_t wave = blockIdx.x * kChainWaves + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    // group g of a slice takes contexts g, g + groups, g + 2 groups ...: neighbouring contexts (the hot ones of a
    // stream sit next to each other: the coefficient contexts of one block category) go to different waves
    // (groups == 0, r4: a batch with pairs enough for full waves -- 64 (slice, context) pairs to a wave across slice boundaries, no idle lanes:
    // a stream of 86 contexts used to give every slice two waves of 43)
    const uint32_t nk = p.n_states, pair = wave * 64 + lane;
    const uint32_t s = groups ? wave / groups : pair / nk, k = groups ? (wave - s * groups) + lane * groups : pair - s * nk;
    if ((groups && lane >= chain_lanes) || s >= n_slices || status[s] != AVR_SLICE_OK || k >= nk) return;
    const uint32_t col = p.index[k], row4 = ((nk + 3) >> 2) << 2;
    if (col >= p.ns_full) return;                                // a row beyond the batch's contexts (launch sized by a guess)
    uint32_t st = init_states[size_t(s) * p.ns_full + col] & 127u;
    const uint32_t c0 = p.chunk_base[s], nc = (p.n_bins[s] + kChunk - 1) / kChunk;
    const uint16_t *le = lend + size_t(c0) * nk + k;             // end[k] of the chunk being requested (four ahead)
    const uint32_t *bw_ahead = lbits + size_t(c0) * 32;          // bit string of the chunk whose window is being requested (two ahead)
    const uint32_t *bw_cur = bw_ahead;                           // bit string of the chunk being walked
    uint8_t *eo = est + size_t(c0) * row4 + k;
    auto ends = [&]() {                                          // (end[k-1], end[k]), then on to the next chunk
        const uint32_t e1 = le[0], em = le[-1];
        le += nk;
        return make_uint2(k ? em : 0u, e1);
    };
    auto window = [&](const uint32_t *bw, uint32_t pos) {        // bits 32 (pos / 32) .. + 127 of a chunk's string
        const uint32_t wi = (pos >> 5) & 31u;                    // (past dword 31: the next chunk's, never used -- a run ends by bit 1024)
        return make_uint4(bw[wi], bw[wi + 1], bw[wi + 2], bw[wi + 3]);
    };
    uint2 e0 = ends(), e1 = ends(), e2 = ends(), e3 = ends();
    uint4 w0 = window(bw_ahead, e0.x), w1 = window(bw_ahead + 32, e1.x);
    bw_ahead += 64;
    for (uint32_t c = 0; c < nc; c++) {
        const uint2 e4 = ends();
        const uint4 w2 = window(bw_ahead, e2.x);
        bw_ahead += 32;
        *eo = uint8_t(st);
        eo += row4;
        uint32_t pos = e0.x;
        const uint32_t end = e0.y;
        // The step is a chain of dependent look-ups, and every branch on it costs about as much as a look-up: the first 32
        // bins are four look-ups without one (a look-up of n = 0 bins is the identity: rows 0 .. 127 of the table), the next
        // 32 four more if any lane of the wave has them; what is left after 64 -- rare -- goes the general way.
        uint32_t left = (end > pos && st < 126) ? end - pos : 0u;    // pStateIdx 63 never moves
        const uint32_t sh = pos & 31u;
        auto four = [&](uint32_t avail) {
#pragma unroll
            for (uint32_t i = 0; i < 4; i++) {
                if (i && !__any(left != 0)) break;
                const uint32_t n = left < 8u ? left : 8u;
                uint32_t off = (128u << n) - 128u + (avail & ((1u << n) - 1u));         // all of the index that does not wait for the state
                asm volatile("" : "+v"(off));                                          // (kept whole: the compiler would split it up again)
                st = tn[(st << n) + off];                                              // ... and the one shift-add that does
                avail >>= 8;
                left -= n;
            }
        };
        four(__builtin_amdgcn_alignbit(w0.y, w0.x, sh));
        if (__any(left != 0)) {
            four(__builtin_amdgcn_alignbit(w0.z, w0.y, sh));
            if (__any(left != 0)) {
                pos += 64;
                uint32_t avail = __builtin_amdgcn_alignbit(w0.w, w0.z, sh);      // bins 64 .. 95 are in the window as well
                uint32_t have = 32;
                while (left) {
                    if (have == 0) {                             // a run of more than 96 bins: fetch as it goes
                        const uint4 win = window(bw_cur, pos);
                        avail = __builtin_amdgcn_alignbit(win.y, win.x, pos & 31u);
                        have = 32;
                    }
                    const uint32_t n = left < 8u ? left : 8u;
                    uint32_t off = (128u << n) - 128u + (avail & ((1u << n) - 1u));
                    asm volatile("" : "+v"(off));
                    st = tn[(st << n) + off];
                    avail >>= 8;
                    left -= n; pos += n; have -= 8;
                }
            }
        }
        bw_cur += 32;
        e0 = e1; e1 = e2; e2 = e3; e3 = e4;
        w0 = w1; w1 = w2;
    }
    if (final_states) final_states[size_t(s) * p.ns_full + col] = uint8_t(st);
}

// ------------------------------------------------------------------ the chains in segments
//
// k_k1p_ctxchain's time is its longest lane: the hottest context of the longest slice, 600 chunks in a row with about seven
// dependent look-ups each (0.29 of config 2's 1.7 ms).  What cuts a chain is that the state machine is MONOTONE: order the 126
// states by the probability they give a 1 -- (valMPS 0, pStateIdx 62) first, (valMPS 1, pStateIdx 62) last -- and the state
// after a bin never overtakes: s <= s' implies T(s, b) <= T(s', b) for either bin value (an MPS moves every state one step
// towards its own end, an LPS moves it back by transIdxLPS, which is non-decreasing in pStateIdx: cabac_code.h:43-47 with
// ITU-T H.264 Table 9-45; tests/test_k1p_emul.py walks all pairs).  So when the walks from the two extreme states have met,
// every state in between has met them too, and from there on the context's state does not depend on where the segment was
// entered.  Each (slice, context) chain is cut into n_segs segments of equal chunk counts (how many: kChainWaveSlots below):
//   k_k1p_chain_seg   lane per (slice, context, segment): both extreme states through the segment's chunks, side by side (two
//                     independent look-up chains in flight); from the chunk where they have met, the state of every chunk is
//                     noted in `est` as final.  Summary: exit state, the chunk they met at -- and, for the segments they do NOT
//                     meet in, which are the ones with few bins, those bins themselves (up to kSegBits of them, as a bit string:
//                     a run of one value needs about 80 bins for the walks to meet, 18 LPS steps down and 62 MPS steps up).
//   k_k1p_chain_fix   lane per (slice, context, segment): the segment's true entry state -- the exit of the nearest earlier
//                     segment whose walks met (or the slice's initial state), taken through the bit strings of the segments
//                     between -- and from it the chunks before the meeting point once more, this time for `est`.  A segment in
//                     between with more than kSegBits bins whose walks did not meet (nothing forbids it) is simply walked again,
//                     chunk by chunk, by the lanes that need its exit state.
// Batches of short slices (a segment would be a chunk or two) keep k_k1p_ctxchain, the start-to-end walk.
constexpr uint32_t kMaxChainSegs = 16, kSegWaves = 16;           // 16 waves share the 64 KiB look-up table: two such workgroups per CU
// How many segments (r4): as many as keep every wave of the launch resident at once -- the chip holds 8 192 of these waves (two workgroups
// of 16 a CU), a launch takes as long as its longest lane's walk, and a second round of workgroups would double that.  Lanes are dealt
// 64 (slice, context) pairs to a wave across slice boundaries (round 3 gave a slice's 86 contexts two waves of 43 lanes): 688 full waves a
// segment for config 2 instead of 1 024, which is what lets its chains be cut in 11 instead of 8.
constexpr uint32_t kChainWaveSlots = 8192;
constexpr uint32_t kSegBits = 128;                               // bins a segment's bit string holds
struct alignas(16) SegSummary {
    uint64_t bits[2];         // the segment's bins of this context, first bin in bit 0 of bits[0] (valid when n_bins <= kSegBits)
    uint32_t n_bins;
    uint8_t exit_state;       // state after the segment when the walks met
    uint8_t met;              // 1: the walks met (exit_state valid, chunks from met_chunk on are noted)
    uint16_t met_chunk;       // first chunk of the segment (relative to its start) whose entry state is noted; segment length if none
};

// One chunk's step of a context's chain for NS states at once: the context's bins of the chunk are bits [pos, end) of the chunk's
// bit string (window w: 128 bits from dword pos / 32), taken eight per look-up.  Same structure as k_k1p_ctxchain's step.
template <int NS>
__device__ __forceinline__ void chain_step(const uint8_t *tn, const uint32_t *bw_cur, uint32_t pos, uint32_t end, const uint4 &w0, uint32_t (&st)[NS]) {
    uint32_t left = end > pos ? end - pos : 0u;
    const uint32_t sh = pos & 31u;
    auto four = [&](uint32_t avail) {
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) {
            if (i && !__any(left != 0)) break;
            const uint32_t n = left < 8u ? left : 8u;
            uint32_t off = (128u << n) - 128u + (avail & ((1u << n) - 1u));
            asm volatile("" : "+v"(off));
#pragma unroll
            for (int q = 0; q < NS; q++) st[q] = tn[(st[q] << n) + off];
            avail >>= 8;
            left -= n;
        }
    };
    four(__builtin_amdgcn_alignbit(w0.y, w0.x, sh));
    if (__any(left != 0)) {
        four(__builtin_amdgcn_alignbit(w0.z, w0.y, sh));
        if (__any(left != 0)) {
            pos += 64;
            uint32_t avail = __builtin_amdgcn_alignbit(w0.w, w0.z, sh);
            uint32_t have = 32;
            while (left) {
                if (have == 0) {
                    const uint32_t wi = (pos >> 5) & 31u;
                    avail = __builtin_amdgcn_alignbit(bw_cur[wi + 1], bw_cur[wi], pos & 31u);
                    have = 32;
                }
                const uint32_t n = left < 8u ? left : 8u;
                uint32_t off = (128u << n) - 128u + (avail & ((1u << n) - 1u));
                asm volatile("" : "+v"(off));
#pragma unroll
                for (int q = 0; q < NS; q++) st[q] = tn[(st[q] << n) + off];
                avail >>= 8;
                left -= n; pos += n; have -= 8;
            }
        }
    }
}

// Which (slice, context, segment) a lane of the two kernels below has; false: none.
struct ChainLane { uint32_t s, k, seg, c_begin, c_end, nc, col; };
__device__ __forceinline__ bool chain_lane(const Plan &p, uint32_t n_slices, uint32_t n_segs, const int32_t *status, ChainLane *o) {
    const uint32_t wave = blockIdx.x * kSegWaves + (threadIdx.x >> 6), lane = threadIdx.x & 63, nk = p.n_states;
    const uint32_t seg = wave % n_segs, pair = (wave / n_segs) * 64 + lane;          // pair = slice * nk + context
    const uint32_t s = pair / nk, k = pair - s * nk;
    if (s >= n_slices || status[s] != AVR_SLICE_OK) return false;
    const uint32_t col = p.index[k];
    if (col >= p.ns_full) return false;
    const uint32_t nc = (p.n_bins[s] + kChunk - 1) / kChunk, seg_len = (nc + n_segs - 1) / n_segs;
    const uint32_t c_begin = seg * seg_len < nc ? seg * seg_len : nc, c_end = c_begin + seg_len < nc ? c_begin + seg_len : nc;
    *o = ChainLane{s, k, seg, c_begin, c_end, nc, col};
    return true;
}

__global__ __launch_bounds__(64 * kSegWaves) void k_k1p_chain_seg(Plan p, uint32_t n_slices, uint32_t n_segs, const int32_t *status,
                                                       const uint8_t *tng, const uint32_t *lbits, const uint16_t *lend, uint8_t *est, SegSummary *summ) {
    __shared__ uint8_t tn[kTnBytes];
    for (uint32_t i = threadIdx.x; i < kTnBytes / 16; i += 64 * kSegWaves) reinterpret_cast<uint4 *>(tn)[i] = reinterpret_cast<const uint4 *>(tng)[i];
    __syncthreads();
    ChainLane L;
    if (!chain_lane(p, n_slices, n_segs, status, &L)) return;
    const uint32_t nk = p.n_states, row4 = ((nk + 3) >> 2) << 2, k = L.k;
    const uint32_t c0 = p.chunk_base[L.s] + L.c_begin, n_ch = L.c_end - L.c_begin;
    const uint16_t *le = lend + size_t(c0) * nk + k;
    const uint32_t *bw_ahead = lbits + size_t(c0) * 32, *bw_cur = bw_ahead;
    uint8_t *eo = est + size_t(c0) * row4 + k;
    auto ends = [&]() { const uint32_t e1 = le[0], em = le[-1]; le += nk; return make_uint2(k ? em : 0u, e1); };
    auto window = [&](const uint32_t *bw, uint32_t pos) { const uint32_t wi = (pos >> 5) & 31u; return make_uint4(bw[wi], bw[wi + 1], bw[wi + 2], bw[wi + 3]); };
    uint2 e0 = ends(), e1 = ends(), e2 = ends(), e3 = ends();
    uint4 w0 = window(bw_ahead, e0.x), w1 = window(bw_ahead + 32, e1.x);
    bw_ahead += 64;
    uint32_t st[2] = {124u, 125u};                               // the two ends of the order: (valMPS 0, pStateIdx 62), (valMPS 1, pStateIdx 62)
    uint32_t met_chunk = n_ch, n_bins = 0;
    uint64_t bits0 = 0, bits1 = 0;
    for (uint32_t c = 0; c < n_ch; c++) {
        const uint2 e4 = ends();
        const uint4 w2 = window(bw_ahead, e2.x);
        bw_ahead += 32;
        if (st[0] == st[1]) {                                    // met: this chunk's entry state is what it is whatever the segment was entered in
            *eo = uint8_t(st[0]);
            met_chunk = met_chunk < c ? met_chunk : c;
        }
        eo += row4;
        const uint32_t pos = e0.x, end = e0.y, cnt = end > pos ? end - pos : 0u;
        if (n_bins < kSegBits && cnt) {                          // the bins themselves, while they fit: what a segment without a meeting point is carried on by
            const uint32_t sh = pos & 31u;                       // (up to 64 of a chunk's: with more than that the string is full soon anyway, and then unused)
            const uint64_t lo = uint64_t(__builtin_amdgcn_alignbit(w0.y, w0.x, sh)) | uint64_t(__builtin_amdgcn_alignbit(w0.z, w0.y, sh)) << 32;
            const uint64_t got = cnt < 64u ? lo & ((uint64_t(1) << cnt) - 1) : lo;
            if (n_bins < 64u) { bits0 |= got << n_bins; bits1 |= n_bins ? got >> (64u - n_bins) : 0u; }
            else bits1 |= got << (n_bins - 64u);
        }
        n_bins += cnt > 64u ? kSegBits + 1u : cnt;              // a chunk with more than 64 of them: the string is not kept
        chain_step<2>(tn, bw_cur, pos, end, w0, st);
        bw_cur += 32;
        e0 = e1; e1 = e2; e2 = e3; e3 = e4;
        w0 = w1; w1 = w2;
    }
    SegSummary o;
    // (met_chunk has 16 bits: a meeting point beyond chunk 65 534 of the segment -- a slice of half a gigabin -- is reported as "did not meet",
    // which k_k1p_chain_fix answers by walking the whole segment again: slower for that pair, never wrong)
    o.bits[0] = bits0; o.bits[1] = bits1; o.n_bins = n_bins; o.exit_state = uint8_t(st[0]); o.met = st[0] == st[1] && met_chunk < 0xffffu;
    o.met_chunk = uint16_t(met_chunk < 0xffffu ? met_chunk : 0xffffu);
    summ[(size_t(L.s) * nk + k) * n_segs + L.seg] = o;
}

// A context's chain through chunks [c_first, c_first + n_ch) of its slice from state `st`, chunk by chunk (k_k1p_ctxchain's loop); NOTE: the
// state at the start of every chunk goes to `est`.  Returns the state after the last chunk.
template <bool NOTE>
__device__ __forceinline__ uint32_t chain_chunks(const uint8_t *tn, const uint32_t *lbits, const uint16_t *lend, uint8_t *est, uint32_t nk, uint32_t row4,
                                                 uint32_t k, uint32_t c_first, uint32_t n_ch, uint32_t st0) {
    const uint16_t *le = lend + size_t(c_first) * nk + k;
    const uint32_t *bw_ahead = lbits + size_t(c_first) * 32, *bw_cur = bw_ahead;
    uint8_t *eo = est + size_t(c_first) * row4 + k;
    auto ends = [&]() { const uint32_t e1 = le[0], em = le[-1]; le += nk; return make_uint2(k ? em : 0u, e1); };
    auto window = [&](const uint32_t *bw, uint32_t pos) { const uint32_t wi = (pos >> 5) & 31u; return make_uint4(bw[wi], bw[wi + 1], bw[wi + 2], bw[wi + 3]); };
    uint2 e0 = ends(), e1 = ends(), e2 = ends(), e3 = ends();
    uint4 w0 = window(bw_ahead, e0.x), w1 = window(bw_ahead + 32, e1.x);
    bw_ahead += 64;
    uint32_t st[1] = {st0};
    for (uint32_t c = 0; c < n_ch; c++) {
        const uint2 e4 = ends();
        const uint4 w2 = window(bw_ahead, e2.x);
        bw_ahead += 32;
        if (NOTE) { *eo = uint8_t(st[0]); eo += row4; }
        if (st[0] < 126u) chain_step<1>(tn, bw_cur, e0.x, e0.y, w0, st);      // pStateIdx 63 never moves
        bw_cur += 32;
        e0 = e1; e1 = e2; e2 = e3; e3 = e4;
        w0 = w1; w1 = w2;
    }
    return st[0];
}

__global__ __launch_bounds__(64 * kSegWaves) void k_k1p_chain_fix(Plan p, uint32_t n_slices, uint32_t n_segs, const int32_t *status,
                                                       const uint8_t *tng, const uint32_t *lbits, const uint16_t *lend, const uint8_t *init_states,
                                                       uint8_t *est, const SegSummary *summ, uint8_t *final_states, uint32_t force_walk_every) {
    __shared__ uint8_t tn[kTnBytes];
    for (uint32_t i = threadIdx.x; i < kTnBytes / 16; i += 64 * kSegWaves) reinterpret_cast<uint4 *>(tn)[i] = reinterpret_cast<const uint4 *>(tng)[i];
    __syncthreads();
    ChainLane L;
    if (!chain_lane(p, n_slices, n_segs, status, &L)) return;
    const uint32_t nk = p.n_states, row4 = ((nk + 3) >> 2) << 2, k = L.k;
    const SegSummary *mine = summ + (size_t(L.s) * nk + k) * n_segs;
    const uint32_t init = init_states[size_t(L.s) * p.ns_full + L.col] & 127u;
    const uint32_t c_slice = p.chunk_base[L.s], seg_len = (L.nc + n_segs - 1) / n_segs;
    // test hook (always 0 in the product): for every n-th pair the summaries count for nothing -- every earlier segment is walked again
    const bool distrust = force_walk_every && (L.s * nk + k) % force_walk_every == 0;
    // The true entry state: back to the nearest segment whose walks met (or the slice's start), then forward through the ones between --
    // their bins as a bit string where they fit, chunk by chunk where not.  A context parked at pStateIdx 63 stays there.
    uint32_t from = L.seg, entry = init;
    if (init < 126u) {
        while (from > 0 && (distrust || !mine[from - 1].met)) from--;
        if (from > 0) entry = mine[from - 1].exit_state;
        for (uint32_t g = from; g < L.seg; g++) {
            const SegSummary sg = mine[g];
            if (sg.n_bins <= kSegBits && !distrust) {
                uint64_t b = sg.bits[0], b_hi = sg.bits[1];
                for (uint32_t left = sg.n_bins; left;) {
                    const uint32_t n = left < 8u ? left : 8u;
                    entry = tn[(128u << n) - 128u + (entry << n) + (uint32_t(b) & ((1u << n) - 1u))];
                    b = (b >> 8) | (b_hi << 56); b_hi >>= 8; left -= n;
                }
            } else {
                const uint32_t g_begin = g * seg_len < L.nc ? g * seg_len : L.nc, g_end = g_begin + seg_len < L.nc ? g_begin + seg_len : L.nc;
                entry = chain_chunks<false>(tn, lbits, lend, est, nk, row4, k, c_slice + g_begin, g_end - g_begin, entry);
            }
        }
    }
    const SegSummary me = mine[L.seg];
    const uint32_t n_ch = L.c_end - L.c_begin;
    const bool met = me.met && init < 126u && !distrust;
    const uint32_t stop = met && me.met_chunk < n_ch ? me.met_chunk : n_ch;          // chunks [0, stop): before the meeting point
    const uint32_t exit_state = chain_chunks<true>(tn, lbits, lend, est, nk, row4, k, c_slice + L.c_begin, stop, entry);
    // the slice's last segment that has chunks leaves the final state (an empty slice: its initial one)
    if (final_states && (L.c_end == L.nc) && (L.c_begin < L.nc || L.seg == 0))
        final_states[size_t(L.s) * p.ns_full + L.col] = uint8_t(stop < n_ch ? me.exit_state : exit_state);
}

__device__ __forceinline__ CodeEntry device_code_entry(uint32_t c);
// per code: { low byte of rLPS << shift per range quarter, the shift per quarter } (see step_pair)
__device__ __forceinline__ uint2 device_codes2(uint32_t c);

// Replay + phase B1: one lane per chunk, its records in stream order.  est[gc][k] is the state of context k
// when the chunk is entered, so a lane loads those states (one byte per context, in LDS, laid out (k, lane)
// like k_cabac_encode's) and simply plays the chunk's bins: state before the bin -> resolved code,
// cabac_code.h:43-47 -> next state.  Records in, codes out, each read or written once, in order.
//
// The same walk yields the chunk's stretch summary (b1_stretch of avr_k1p.h, which k_k1p_b1 runs over
// finished codes when a caller brings its own): with the state in hand, the bin's LPS ranges are one more
// field of the same table entry, so phase B1 costs its arithmetic and no pass of its own.  A stretch runs to
// the first coded LPS at or past the chunk's end: the lane keeps playing the next chunk's records for that (its
// private state table stays exact), without writing their codes -- those belong to the next chunk's lane.
//
// The step is branch-free up to B1's modes.  Two look-ups per bin: sel_off[selector] = where the bin's state
// byte lives in the lane's column (bypass, terminate, padding and anything that is no context of the batch go
// to pseudo contexts whose pseudo states 128.. never move), and info[state << 1 | bin] = { LPS ranges of the
// state's four range quarters, next state | resolved code << 8 | B1's meta << 16 }.
constexpr uint32_t kStBypass = 128, kStTerminate = 129, kStPad = 130, kStNone = 131, kReplayStates = 132;

// B1's four candidate ranges, two to a register (16 bits each), stepped with the packed 16-bit instructions: per pair
// one byte permute picks each candidate's LPS range out of the state's row by the candidate's own range quarter, and the
// LPS side (range renormalised, its shift) comes out of two more rows of the same shape instead of a count-leading-zeros
// per candidate (codes2[code] = { rown: low byte of rLPS << shift per quarter, shrow: the shift per quarter }).
typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, a) - __builtin_bit_cast(u16x2, b)); }
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, a) + __builtin_bit_cast(u16x2, b)); }
__device__ __forceinline__ uint32_t pk_shl(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, a) << __builtin_bit_cast(u16x2, b)); }
__device__ __forceinline__ uint32_t pk_shr8(uint32_t a) { return (a >> 8) & 0x00ff00ffu; }
// one bin on a pair of candidates (step_range of avr_k1p.h, twice): returns the two shifts, packed
__device__ __forceinline__ uint32_t step_pair(uint32_t &Rp, uint32_t row, uint32_t rown, uint32_t shrow, bool sym, uint32_t extra) {
    const uint32_t sel = ((Rp >> 6) & 0x00030003u) | 0x0c000c00u;   // bytes 0 and 2: the candidate's quarter; 1 and 3: zero
    const uint32_t rl = __builtin_amdgcn_perm(row, row, sel);
    const uint32_t rm = pk_sub(Rp, rl);                              // MPS side: range - rLPS, in [128, 511]
    const uint32_t shm = pk_shr8(rm) ^ 0x00010001u;                  // one shift iff below 256
    const uint32_t rn = __builtin_amdgcn_perm(rown, rown, sel) | 0x01000100u;   // LPS side, renormalised: in [256, 511]
    const uint32_t shl = __builtin_amdgcn_perm(shrow, shrow, sel);
    Rp = sym ? rn : pk_shl(rm, shm);
    return (sym ? shl : shm) + extra;
}

// Phase B1 as the kernels walk it (the logic of b1_stretch, avr_k1p.h, which stays the CPU's statement of it): a chunk's
// stretch summary from its bins in order, eight at a time.  A bin comes as the table entry { x: LPS ranges of its state's
// four range quarters, y: (code << 8) | (meta << 16), z, w: CodeEntryC's side and adj } (k_k1p_replay has it from its state
// look-up, k_k1p_b1 from the code).
// mode: 0 = looking for the LPS that opens the stretch, 1 = four candidate ranges, 2 = they have met, 3 = closed.
struct B1Walk {
    uint32_t i0, i1, limit, n, max_stretch;
    const uint2 *codes2;
    uint32_t mode, Rp0, Rp1, Tp0, Tp1, T[4], Rm, Tm, end;  // candidates 0, 1 in Rp0 (low, high half), 2, 3 in Rp1; Tp: their shifts since the last flush
    bool merged;
    Stretch o;

    __device__ __forceinline__ void init(uint32_t chunk, uint32_t i0_, uint32_t i1_, uint32_t n_, uint32_t max_stretch_, const uint2 *codes2_) {
        i0 = i0_; i1 = i1_; limit = i0_ + kChunk; n = n_; max_stretch = max_stretch_; codes2 = codes2_;
        mode = 0; Rp0 = Rp1 = 0x01fe01feu; Tp0 = Tp1 = 0; T[0] = T[1] = T[2] = T[3] = 0; Rm = 510; Tm = 0; end = 0;
        merged = false;
        o.first = kNone; o.end = 0; o.exit_q = 0; o.too_long = 0; o.pad[0] = o.pad[1] = 0;
        for (int q = 0; q < 4; q++) { o.t_exit[q] = 0; o.r_exit[q] = 0; }
        if (chunk == 0) { o.first = 0; mode = 2; merged = true; }    // opens at bin 0 with the initial range 510 (cabac_code.h:30)
    }
    __device__ __forceinline__ void flush_t() {                  // at least every 8 bins: 8 x 9 shifts fit 16 bits with room
        T[0] += Tp0 & 0xffffu; T[1] += Tp0 >> 16; T[2] += Tp1 & 0xffffu; T[3] += Tp1 >> 16;
        Tp0 = Tp1 = 0;
    }
    __device__ __forceinline__ bool met() const { return Rp0 == Rp1 && (Rp0 >> 16) == (Rp0 & 0xffffu); }
    __device__ __forceinline__ void one(uint32_t idx, const uint4 &e) {       // bin idx (< n), any mode
        const CodeEntry ce{e.x, e.y >> 16};
        const bool boundary = ce.meta & 1u;                      // a coded LPS (code_is_boundary)
        if (mode == 0) {
            if (idx < i1 && boundary) {
                o.first = idx;
                const uint32_t rown = codes2[(e.y >> 8) & 0xffu].x;              // post_lps_range for the four quarters
                Rp0 = __builtin_amdgcn_perm(rown, rown, 0x0c010c00u) | 0x01000100u;
                Rp1 = __builtin_amdgcn_perm(rown, rown, 0x0c030c02u) | 0x01000100u;
                mode = 1;
            }
        } else if (mode == 1) {
            const bool closing = idx >= limit && boundary;
            if (closing)
                o.exit_q |= uint8_t(((Rp0 >> 6) & 3u) | ((Rp0 >> 22) & 3u) << 2 | ((Rp1 >> 6) & 3u) << 4 | ((Rp1 >> 22) & 3u) << 6);
            const uint2 e2 = codes2[(e.y >> 8) & 0xffu];
            const uint32_t extra = (ce.meta >> 8) * 0x00010001u;                 // a bypass bin's one shift
            Tp0 = pk_add(Tp0, step_pair(Rp0, ce.row, e2.x, e2.y, boundary, extra));
            Tp1 = pk_add(Tp1, step_pair(Rp1, ce.row, e2.x, e2.y, boundary, extra));
            if (closing) { end = idx + 1; mode = 3; }
            else if (met()) { Rm = Rp0 & 0xffffu; mode = 2; merged = true; }
        } else if (mode == 2) {
            const bool closing = idx >= limit && boundary;
            if (closing) o.exit_q = uint8_t(((Rm >> 6) & 3) * 0x55u);
            Tm += step_merged(e);
            if (closing) { end = idx + 1; mode = 3; }
            else if (idx >= limit && idx - i0 > max_stretch) { o.too_long = 1; end = idx + 1; mode = 3; }
        }
    }
    __device__ __forceinline__ uint32_t step_merged(const uint4 &e) {         // step_range on the one range left
        uint32_t v;
        return step_range_c(CodeEntryC{e.x, e.z, 0u, e.w}, &Rm, &v);
    }
    __device__ __forceinline__ void group(uint32_t base, const uint4 e[8]) {  // bins base .. base+7
        if (mode == 2 && base + 8 <= i1) {                       // merged and inside the chunk: nothing can close the stretch
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) Tm += step_merged(e[j]);
        } else if (mode == 1 && base + 8 <= i1) {                // four candidates, inside the chunk: the same, and no branch per bin
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) {
                const uint2 e2 = codes2[(e[j].y >> 8) & 0xffu];
                const bool sym = (e[j].y >> 16) & 1u;
                const uint32_t extra = (e[j].y >> 24) * 0x00010001u;
                Tp0 = pk_add(Tp0, step_pair(Rp0, e[j].x, e2.x, e2.y, sym, extra));
                Tp1 = pk_add(Tp1, step_pair(Rp1, e[j].x, e2.x, e2.y, sym, extra));
            }
            flush_t();
            // candidates that have met stay together: looking once per group is enough, and the group's shifts are in T[] either way
            if (met()) { Rm = Rp0 & 0xffffu; mode = 2; merged = true; }
        } else if (mode != 3) {
            for (uint32_t j = 0; j < 8; j++) if (base + j < n) one(base + j, e[j]);
            flush_t();
        }
    }
    __device__ __forceinline__ void chunk_done() { if (mode == 0) mode = 3; }  // no LPS in the chunk: no stretch opens here (first stays kNone)
    __device__ __forceinline__ void tail(uint32_t base, const uint4 e[8]) {      // past the chunk, until the stretch closes
        for (uint32_t j = 0; j < 8; j++) if (base + j < n && mode != 3) one(base + j, e[j]);
        flush_t();
    }
    __device__ __forceinline__ const Stretch &finish() {
        if (o.first != kNone) {
            o.end = mode != 3 ? n : end;                         // not closed: ran to the end of the slice
            for (uint32_t q = 0; q < 4; q++) {                   // once merged, Rm / Tm carried on for all four candidates
                const uint32_t rq = ((q & 2u) ? Rp1 : Rp0) >> (16 * (q & 1u)) & 0xffffu;
                o.t_exit[q] = T[q] + (merged ? Tm : 0u);
                o.r_exit[q] = uint16_t(merged ? Rm : rq);
            }
        }
        return o;
    }
};

template <bool TILE_CODES>
__global__ __launch_bounds__(256) void k_k1p_replay(Plan p, uint32_t total_chunks, const uint32_t *est, uint8_t *res,
                                                    const int32_t *status, Stretch *stretch, uint32_t max_stretch) {
    extern __shared__ uint32_t replay_lds[];                     // per wave: state dwords [(nk+8)/4][64]
    __shared__ uint4 info[2 * kReplayStates];
    __shared__ uint2 codes2[256];
    __shared__ uint32_t sel_off[2048];
    const uint32_t lane = threadIdx.x & 63, nk = p.n_states;
    for (uint32_t c = threadIdx.x; c < 256; c += blockDim.x) codes2[c] = device_codes2(c);
    uint8_t *stb = reinterpret_cast<uint8_t *>(replay_lds) + (threadIdx.x >> 6) * (((nk + 8) >> 2) << 8) + lane * 4;
    for (uint32_t sel = threadIdx.x; sel < 2048; sel += blockDim.x) {
        // contexts get their dense id; 1024 (bypass), 1025 (terminate), 1026 (no-op) -> nk+1, nk+2, nk+3; the rest nk / nk+4
        const uint32_t over = (sel < 1023u ? 1023u : sel > 1027u ? 1027u : sel) - 1023u;
        const uint32_t dense = sel < 1024u ? uint32_t(p.table[sel]) : kNotUsed;
        const uint32_t kk = (dense < nk ? dense : nk) + over;
        sel_off[sel] = ((kk & ~3u) << 6) + (kk & 3u);
    }
    for (uint32_t i = threadIdx.x; i < 2 * kReplayStates; i += blockDim.x) {
        const uint32_t st = i >> 1, bin = i & 1u;
        uint32_t next, code;
        if (st < 128) {
            const uint32_t nx = d_tables.packed[st][1];
            next = ((bin ^ st) & 1u) ? (nx >> 8) & 0xffu : nx & 0xffu;
            code = code_context(st, bin);
        } else {
            next = st;
            code = st == kStBypass ? kCodeBypass | bin : st == kStTerminate ? code_terminate(bin) : kCodePad;
        }
        const CodeEntry ce = device_code_entry(code);
        const CodeEntryC cc = code_entry_c(ce);
        info[i] = make_uint4(ce.row, next | code << 8 | ce.meta << 16, cc.side, cc.adj);
    }
    __syncthreads();
    const uint32_t gc = blockIdx.x * blockDim.x + threadIdx.x;
    if (gc >= total_chunks) return;
    const uint32_t s = p.chunk_slice[gc];
    Stretch o;
    o.first = kNone; o.end = 0; o.exit_q = 0; o.too_long = 0; o.pad[0] = o.pad[1] = 0;
    for (int q = 0; q < 4; q++) { o.t_exit[q] = 0; o.r_exit[q] = 0; }
    if (status[s] != AVR_SLICE_OK) { stretch[gc] = o; return; }
    const uint32_t c = gc - p.chunk_base[s];
    const uint32_t n = p.n_bins[s], i0 = c * kChunk;
    const uint32_t i1 = i0 + kChunk < n ? i0 + kChunk : n;      // end of the chunk's own bins = where a closing LPS may come from
    {
        const uint32_t nkw = (nk + 3) >> 2;
        const uint32_t *e = est + size_t(gc) * nkw;
        for (uint32_t kw = 0; kw < nkw; kw++) *reinterpret_cast<uint32_t *>(stb + kw * 256) = e[kw];
        const uint32_t pseudo[5] = {kStNone, kStBypass, kStTerminate, kStPad, kStNone};
#pragma unroll
        for (uint32_t j = 0; j < 5; j++) stb[((nk + j) >> 2) * 256 + ((nk + j) & 3)] = uint8_t(pseudo[j]);
    }
    const uint16_t *r = p.recs + p.rec_off[s];
    // where the chunk's codes go: bin i of the slice at ro + (i - ro_i0) * ro_scale / 16 ... in whole 16-byte groups only
    uint8_t *ro = TILE_CODES ? res + ((size_t(gc >> 6) * 64) * 64 + (gc & 63u)) * 16 : res + p.res_off[s];
    auto put16 = [&](uint32_t i, const U4 &v) {                  // codes i .. i+15 of the slice (i a multiple of 16, in this chunk)
        if (TILE_CODES) *reinterpret_cast<U4 *>(ro + size_t((i - i0) >> 4) * 1024) = v;
        else *reinterpret_cast<U4 *>(ro + i) = v;
    };
    B1Walk b1;
    b1.init(c, i0, i1, n, max_stretch, codes2);
    // 8 records (16 bytes) -> 8 codes; e[] = their table entries
    auto eight = [&](const U4 &v, uint32_t &c0, uint32_t &c1, uint4 e[8]) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t off[8];
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) off[j] = sel_off[((w[j >> 1] >> ((j & 1) * 16)) >> 1) & 0x7ffu];
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) {
            const uint32_t bin = (w[j >> 1] >> ((j & 1) * 16)) & 1u;
            uint8_t *sp = stb + off[j];
            const uint32_t st = *sp;
            e[j] = info[(st << 1) | bin];
            *sp = uint8_t(e[j].y);
        }
        // the eight codes (byte 1 of .y) side by side: two byte permutes and an OR per four
        c0 = __builtin_amdgcn_perm(e[1].y, e[0].y, 0x0c0c0501u) | __builtin_amdgcn_perm(e[3].y, e[2].y, 0x05010c0cu);
        c1 = __builtin_amdgcn_perm(e[5].y, e[4].y, 0x0c0c0501u) | __builtin_amdgcn_perm(e[7].y, e[6].y, 0x05010c0cu);
    };
    // a slice's records are padded with no-ops to a multiple of 8, its codes to a multiple of 16
    uint32_t i = i0;
    if (i0 < n) {
        U4 v0{0, 0, 0, 0}, v1 = v0, v2 = v0, v3 = v0;
        const uint32_t last = i + 32 <= i1 ? i0 + ((i1 - i0) / 32u - 1u) * 32u : i0;      // where the chunk's last whole line starts
        if (i + 32 <= i1) {
            const U4 *q = reinterpret_cast<const U4 *>(r + i);
            v0 = q[0]; v1 = q[1]; v2 = q[2]; v3 = q[3];
        }
        for (; i + 32 <= i1; i += 32) {                          // a cache line of records per trip, the next one in flight
            // (loaded unconditionally, the address clamped to the chunk's last whole line: see for_record_groups)
            const U4 *qn = reinterpret_cast<const U4 *>(r + (i + 32 <= last ? i + 32 : last));
            const U4 n0 = qn[0], n1 = qn[1], n2 = qn[2], n3 = qn[3];
            U4 a, b;
            uint4 e[8];
            eight(v0, a.x, a.y, e); b1.group(i, e);
            eight(v1, a.z, a.w, e); b1.group(i + 8, e);
            eight(v2, b.x, b.y, e); b1.group(i + 16, e);
            eight(v3, b.z, b.w, e); b1.group(i + 24, e);
            put16(i, a);
            put16(i + 16, b);
            v0 = n0; v1 = n1; v2 = n2; v3 = n3;
        }
        for (; i < i1; i += 16) {
            const U4 *q = reinterpret_cast<const U4 *>(r + i);
            const U4 nop{AVR_NOP_CABAC2, AVR_NOP_CABAC2, AVR_NOP_CABAC2, AVR_NOP_CABAC2};
            const U4 t0 = q[0], t1 = i + 8 < i1 ? q[1] : nop;
            U4 a;
            uint4 e[8];
            eight(t0, a.x, a.y, e); b1.group(i, e);
            eight(t1, a.z, a.w, e); b1.group(i + 8, e);
            put16(i, a);
        }
        b1.chunk_done();
        // past the chunk: on through the next chunk's bins until the stretch closes (codes not written: not this lane's)
        for (i = i0 + kChunk; b1.mode != 3 && i < n; i += 8) {
            uint32_t c0, c1;
            uint4 e[8];
            eight(*reinterpret_cast<const U4 *>(r + i), c0, c1, e);
            b1.tail(i, e);
        }
    }
    stretch[gc] = b1.finish();
}

// ------------------------------------------------------------------ phases B1, B2, C

// code_entry() from the device copy of the tables (d_tables.packed[2 p][0] = rangeTabLPS[p][0..3])
__device__ __forceinline__ CodeEntry device_code_entry(uint32_t c) {
    if (code_is_bypass(c)) return CodeEntry{0u, 0x100u | ((c & 1u) << 1)};
    return CodeEntry{d_tables.packed[2 * (c >> 2)][0], code_sym(c) * 3u};
}

__device__ __forceinline__ uint2 device_codes2(uint32_t c) {
    const uint32_t row = device_code_entry(c).row;
    uint32_t rown = 0, shrow = 0;
    for (uint32_t q = 0; q < 4; q++) {
        const uint32_t rl = (row >> (8 * q)) & 0xffu;
        if (rl) { uint32_t sh; const uint32_t rn = post_lps_range(row, q, &sh); rown |= (rn & 0xffu) << (8 * q); shrow |= sh << (8 * q); }
    }
    return make_uint2(rown, shrow);
}

// Phase B1 from finished codes (the resolved-code entry points; k_k1p_replay does the same walk on the fly): one lane per
// chunk, a cache line of codes per trip with the next one in flight, eight codes to a group of B1Walk.
// `tile` != null: the chunk's codes are also written there wave-interleaved (TileCodes), for phase C to read.
__global__ __launch_bounds__(256) void k_k1p_b1(Plan p, uint32_t total_chunks, const uint8_t *res,
                                                const int32_t *status, Stretch *st, uint32_t max_stretch, uint8_t *tile) {
    __shared__ uint4 cinfo[256];                                 // per code: { LPS ranges of its state, (code << 8) | (meta << 16), side, adj }
    __shared__ uint2 codes2[256];
    {
        const CodeEntry ce = device_code_entry(threadIdx.x);
        const CodeEntryC cc = code_entry_c(ce);
        cinfo[threadIdx.x] = make_uint4(ce.row, threadIdx.x << 8 | ce.meta << 16, cc.side, cc.adj);
        codes2[threadIdx.x] = device_codes2(threadIdx.x);
    }
    __syncthreads();
    const uint32_t gc = blockIdx.x * 256 + threadIdx.x;
    if (gc >= total_chunks) return;
    const uint32_t slice = p.chunk_slice[gc];
    if (status[slice] != AVR_SLICE_OK) { st[gc].first = kNone; st[gc].too_long = 0; return; }
    const uint32_t c = gc - p.chunk_base[slice], n = p.n_bins[slice], i0 = c * kChunk;
    const uint32_t i1 = i0 + kChunk < n ? i0 + kChunk : n;
    const uint8_t *r = res + p.res_off[slice];
    B1Walk b1;
    b1.init(c, i0, i1, n, max_stretch, codes2);
    auto eight = [&](uint32_t lo, uint32_t hi, uint4 e[8]) {     // eight codes, first in lo's low byte
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) e[j] = cinfo[((j < 4 ? lo : hi) >> (8 * (j & 3))) & 0xffu];
    };
    uint8_t *to = tile ? tile + ((size_t(gc >> 6) * 64) * 64 + (gc & 63u)) * 16 : nullptr;
    auto sixteen = [&](uint32_t base, const U4 &v) {
        if (to) *reinterpret_cast<U4 *>(to + size_t((base - i0) >> 4) * 1024) = v;
        uint4 e[8];
        eight(v.x, v.y, e); b1.group(base, e);
        eight(v.z, v.w, e); b1.group(base + 8, e);
    };
    // a slice's codes are padded to a multiple of 16 (with a group to spare): whole 16-byte loads up to the padded end
    uint32_t i = i0;
    if (i0 < n) {
        U4 v0{0, 0, 0, 0}, v1 = v0, v2 = v0, v3 = v0;
        if (i + 64 <= i1) { const U4 *q = reinterpret_cast<const U4 *>(r + i); v0 = q[0]; v1 = q[1]; v2 = q[2]; v3 = q[3]; }
        for (; i + 64 <= i1; i += 64) {
            const U4 *qn = reinterpret_cast<const U4 *>(r + (i + 128 <= i1 ? i + 64 : i));     // unconditional: see for_record_groups
            const U4 n0 = qn[0], n1 = qn[1], n2 = qn[2], n3 = qn[3];
            sixteen(i, v0); sixteen(i + 16, v1); sixteen(i + 32, v2); sixteen(i + 48, v3);
            v0 = n0; v1 = n1; v2 = n2; v3 = n3;
        }
        for (; i < i1; i += 16) sixteen(i, *reinterpret_cast<const U4 *>(r + i));
        b1.chunk_done();
        for (i = i0 + kChunk; b1.mode != 3 && i < n; i += 16) {  // past the chunk, until the stretch closes
            const U4 v = *reinterpret_cast<const U4 *>(r + i);
            uint4 e[8];
            eight(v.x, v.y, e); b1.tail(i, e);
            if (b1.mode != 3) { eight(v.z, v.w, e); b1.tail(i + 8, e); }
        }
    }
    st[gc] = b1.finish();
}

// One workgroup per slice: the stretch summaries are staged through LDS a tile at a time (coalesced),
// thread 0 runs the serial 4->4 chain on them (b2_chain's loop, fed from LDS instead of ~600
// dependent trips to HBM), and the entries go back out coalesced.
constexpr uint32_t kB2Tile = 1024, kB2Seg = 16;                  // kB2Tile / kB2Seg segments x 4 quarters = the workgroup's 256 threads

__global__ __launch_bounds__(256) void k_k1p_b2(Plan p, const int32_t *status, const Stretch *st, Entry *en,
                                                SliceTotals *tot, uint32_t *S) {
    __shared__ Stretch tile[kB2Tile];
    __shared__ Entry ent[kB2Tile];
    __shared__ uint8_t xq[kB2Tile], qin[kB2Tile];               // exit quarter per entry quarter (2 bits each); entry quarter
    __shared__ uint8_t segmap[256], segin[kB2Tile / kB2Seg];     // per segment of kB2Seg stretches: exit quarter per entry quarter; entry quarter
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
        // ... and that chain in three short legs instead of one long one: segments of 16 stretches are walked for each of
        // the four entry quarters at once (thread = segment x quarter), one thread strings the 64 segment maps together,
        // then every segment is walked again from its now known entry quarter: 16 + 64 + 16 dependent look-ups, not 1024.
        {
            const uint32_t seg = t >> 2, c_lo = seg * kB2Seg;
            uint32_t q = t & 3u;
            for (uint32_t c = c_lo; c < c_lo + kB2Seg; c++) q = (xq[c] >> (2 * q)) & 3;     // (past cnt: identity maps)
            segmap[t] = uint8_t(q);
        }
        __syncthreads();
        if (t == 0) {
            uint32_t q = carry[1];
            for (uint32_t sg = 0; sg < kB2Tile / kB2Seg; sg++) { segin[sg] = uint8_t(q); q = segmap[4 * sg + q]; }
            carry[1] = q;                                        // identity past cnt: the quarter after the tile's last stretch
        }
        __syncthreads();
        if (t < kB2Tile / kB2Seg) {
            uint32_t q = segin[t];
            for (uint32_t c = t * kB2Seg; c < (t + 1) * kB2Seg; c++) { qin[c] = uint8_t(q); q = (xq[c] >> (2 * q)) & 3; }
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
    // the digit sums phase C will add into: zeroed here, where the slice's digit count has just become known
    __syncthreads();
    const uint32_t n = ref_digits(carry[0]) + 2;
    uint32_t *d = S + p.dig_off[s];
    for (uint32_t i = t; i < n; i += 256) d[i] = 0;
}

// A stretch's own digit sums go out in aligned 16-byte blocks of four: a lane's stretch is somewhere of its own in the slice's
// sums, so a 4-byte store per digit is a partial cache line per digit per lane (round 2: 487 MB written for 62 MB of sums).
struct DeviceAdder {
    uint32_t *S;
    uint32_t b0 = 0, b1 = 0, b2 = 0, b3 = 0, have = 0, blk = 0;  // the block being filled: its digits, which of them are in (bits), its first index
    __device__ __forceinline__ void store(uint32_t i, uint32_t v) {       // consecutive i
        const uint32_t a = (uint32_t(reinterpret_cast<uintptr_t>(S) >> 2) + i) & 3u;     // where the digit sits in its aligned 16 bytes
        if (have == 0) blk = i - a;
        b0 = a == 0 ? v : b0; b1 = a == 1 ? v : b1; b2 = a == 2 ? v : b2; b3 = a == 3 ? v : b3;
        have |= 1u << a;
        if (a == 3) {
            if (have == 15u) *reinterpret_cast<uint4 *>(S + blk) = make_uint4(b0, b1, b2, b3);
            else flush();
            have = 0;
        }
    }
    __device__ void flush() {
        if (have & 1u) S[blk] = b0;
        if (have & 2u) S[blk + 1] = b1;
        if (have & 4u) S[blk + 2] = b2;
        if (have & 8u) S[blk + 3] = b3;
        have = 0;
    }
    __device__ void add(uint32_t i, uint32_t v) { atomicAdd(&S[i], v); }
};

template <bool TILE_CODES>
__global__ __launch_bounds__(256) void k_k1p_c(Plan p, uint32_t total_chunks, const uint8_t *res,
                                               const Stretch *st, const Entry *en, const SliceTotals *tot,
                                               uint32_t *S) {
    __shared__ CodeEntryC codes[256];
    codes[threadIdx.x] = code_entry_c(device_code_entry(threadIdx.x));
    __syncthreads();
    const uint32_t gc = blockIdx.x * 256 + threadIdx.x;
    if (gc >= total_chunks) return;
    const Stretch o = st[gc];
    if (o.first == kNone) return;
    const uint32_t slice = p.chunk_slice[gc];
    if (tot[slice].bad) return;
    DeviceAdder add;
    add.S = S + p.dig_off[slice];
    if (TILE_CODES) c_stretch_in(TileCodes{res, p.chunk_base[slice]}, o, en[gc], gc - p.chunk_base[slice], codes, add);
    else c_stretch(res + p.res_off[slice], o, en[gc], gc - p.chunk_base[slice], codes, add);
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
    if (status[s] != AVR_SLICE_OK) { if (t == 0 && status[s] != AVR_SLICE_DONE) out_len[s] = 0; return; }   // DONE: coded by the pass before
    const SliceTotals T = tot[s];
    // force_retry_every (test hook k1p_force_retry_every = n, 0 = off; always 0 in the product library): every n-th slice is handed to the serial
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
// `w` is workspace (per-chunk bit strings, end positions and entry states, tables), laid out for the
// caller's context count; the kernels index it by the dense count, which is known after the census
// (the one host round trip of the path: four bytes, to size the later launches).
struct ResolveLayout {
    uint64_t lbits, lend, est, stretch, meta, summ, total;
};
static inline ResolveLayout resolve_layout(size_t n_slices, uint32_t ns, const avr_chunk_plan *pl) {
    ResolveLayout L;
    uint64_t at = 0;
    auto take = [&](uint64_t bytes) { const uint64_t o = at; at += up256(bytes); return o; };
    L.lbits = take(uint64_t(pl->total_chunks + 64) * 128);       // + 64 chunks: the chains read a few chunks ahead, unconditionally
    L.lend = take(uint64_t(pl->total_chunks + 64) * ns * 2 + 256) + 128;     // a pad in front: k_k1p_ctxchain reads lend[-1]
    L.est = take(uint64_t(pl->total_chunks) * ((ns + 3) / 4) * 4 + 16);
    L.stretch = take(uint64_t(pl->total_chunks) * sizeof(Stretch));
    L.meta = take(256 + 2048 + 2048 + kTnBytes);                 // used[32] + n_dense, table[1024], index[1024], tn
    L.summ = take(uint64_t(n_slices) * ns * kMaxChainSegs * sizeof(SegSummary));    // the segmented chains' summaries
    L.total = at;
    return L;
}

// k_k1p_tn's table is a constant: made once per device (on the stream of the first call that needs it, which every
// later call is ordered behind only by its own use of the device: the event makes that explicit) and kept.
static hipError_t tn_table(hipStream_t s, const uint8_t **out) {
    struct Slot { uint8_t *p = nullptr; hipEvent_t ready = nullptr; };
    static Slot slots[16];
    static std::mutex mu;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    Slot &x = slots[dev & 15];
    if (!x.p) {
        uint8_t *buf = nullptr;
        if ((e = hipMalloc(reinterpret_cast<void **>(&buf), kTnBytes + 256)) != hipSuccess) return e;
        if ((e = hipEventCreateWithFlags(&x.ready, hipEventDisableTiming)) != hipSuccess) return e;
        hipLaunchKernelGGL(k_k1p_tn, dim3((kTnBytes + 255) / 256), dim3(256), 0, s, buf);
        if ((e = hipEventRecord(x.ready, s)) != hipSuccess) return e;
        x.p = buf;
    }
    *out = x.p;
    return hipStreamWaitEvent(s, x.ready, 0);
}

static hipError_t launch_resolve(hipStream_t s, Plan p, uint32_t n_slices, const uint8_t *init_states,
                                 const avr_chunk_plan *pl, uint8_t *w, uint8_t *res, int32_t *status, uint8_t *final_states,
                                 uint32_t max_stretch, const Stretch **stretch_out, const DenseHint *hint = nullptr,
                                 uint32_t stride = 1, bool second_pass = false, uint32_t *retry_count = nullptr,
                                 bool tile_codes = false) {
    const uint32_t ns = p.ns_full;
    const ResolveLayout L = resolve_layout(n_slices, ns, pl);
    uint32_t *lbits = reinterpret_cast<uint32_t *>(w + L.lbits);
    uint16_t *lend = reinterpret_cast<uint16_t *>(w + L.lend);
    uint8_t *est = w + L.est;
    Stretch *stretch = reinterpret_cast<Stretch *>(w + L.stretch);
    if (stretch_out) *stretch_out = stretch;
    uint32_t *used = reinterpret_cast<uint32_t *>(w + L.meta);   // [32], then n_dense, then the number of slices for the second pass
    uint32_t *n_dense = used + 32, *n_retry = used + 33;
    uint16_t *table = reinterpret_cast<uint16_t *>(w + L.meta + 256), *index = table + 1024;
    const uint8_t *tn = nullptr;
    p.table = table;
    p.index = index;
    hipError_t e;
    if ((e = hipMemsetAsync(used, 0, 256, s)) != hipSuccess) return e;
    if (final_states && ns && !second_pass &&
        (e = hipMemcpyAsync(final_states, init_states, size_t(n_slices) * ns, hipMemcpyDeviceToDevice, s)) != hipSuccess)
        return e;                                                // contexts without bins keep their state
    hipLaunchKernelGGL(k_k1p_census, dim3((pl->total_blocks + kCensusBlocks - 1) / kCensusBlocks), dim3(256), 0, s, p, pl->total_blocks, status, used, stride);
    hipLaunchKernelGGL(k_k1p_densemap, dim3(1), dim3(1024), 0, s, used, table, index, n_dense);
    if ((e = tn_table(s, &tn)) != hipSuccess) return e;
    uint32_t n_states = 0;
    if (hint && hint->rows) {                                    // sized by the caller's guess, checked by the caller afterwards (DenseHint)
        n_states = hint->rows < ns ? hint->rows : ns;
        if (hint->host_count && (e = hipMemcpyAsync(hint->host_count, n_dense, 4, hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
    } else {
        if ((e = hipMemcpyAsync(&n_states, n_dense, 4, hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
        if (hint && hint->host_count) *hint->host_count = n_states;
    }
    p.n_states = n_states;
    {
        // the waves of a workgroup share the renumbering table; each has its own counters and bit strings
        // as many waves to a CU as its LDS takes (the kernel waits on LDS round trips: two waves a SIMD against one and a half is
        // what there is to win), as workgroups of w waves: the w with the most waves resident, the smaller workgroup on a tie
        const uint32_t per_wave = (33 + (n_states + 5) / 2) * 64 * 4;
        const bool narrow = n_states <= 500;                     // (rows up to (n_states + 3) / 2 = 251: offsets below 2^16)
        const uint32_t kLdsPerCu = 160 * 1024, kStatic = narrow ? 2048 * 2 : 2048 * 4;
        auto local = narrow ? k_k1p_local<uint16_t> : k_k1p_local<uint32_t>;
        uint32_t waves = 1, best = 0;
        for (uint32_t w = 1; w <= 8; w++) {
            const uint32_t need = w * per_wave + kStatic;
            if (need > kLdsPerCu) break;
            const uint32_t resident = kLdsPerCu / need * w;
            if (resident > best) { best = resident; waves = w; }
        }
        if (const uint32_t v = test_hooks().local_waves) waves = v;
        uint32_t lds = waves * per_wave;
        if (lds > 60 * 1024) {
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(local), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
            if (e != hipSuccess && waves > 1) {                  // a runtime that grants a workgroup less than the CU has: the small workgroups
                (void)hipGetLastError();
                waves = per_wave * 2 <= 60 * 1024 ? 2 : 1;
                lds = waves * per_wave;
                e = lds > 60 * 1024 ? hipFuncSetAttribute(reinterpret_cast<const void *>(local), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds))
                                    : hipSuccess;
            }
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(local, dim3((pl->total_chunks + 64 * waves - 1) / (64 * waves)), dim3(64 * waves), lds, s, p,
                           pl->total_chunks, status, lbits, lend, n_retry);
        // How many slices k_k1p_local set aside for the second pass.  Read here, not at the end of the pass: the kernels that
        // follow are launched while the device is still busy with this one's successors only for a moment, where a wait
        // after the last kernel would leave the device idle until the caller's next launch.
        if (stride > 1 && !second_pass) {
            if (hint && hint->rows && hint->host_retry) {        // the caller looks when it waits (DenseHint)
                if ((e = hipMemcpyAsync(hint->host_retry, n_retry, 4, hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
            } else if (retry_count) {
                if ((e = hipMemcpyAsync(retry_count, n_retry, 4, hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
                if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
            }
        }
    }
    if (n_states > 0) {
        // lanes per wave: enough waves to hide the chain's latency (about two per SIMD), but not so few lanes per wave that
        // issuing the waves becomes the limit (measured on 512 slices x 86 contexts: 8 lanes 0.77 ms, 16 0.44, 22 .. 64 0.29)
        uint32_t chain_lanes = uint32_t((uint64_t(n_slices) * n_states + 2047) / 2048);
        chain_lanes = chain_lanes < kChainLanes ? kChainLanes : chain_lanes > 64 ? 64 : chain_lanes;
        if (const uint32_t v = test_hooks().chain_lanes) chain_lanes = v <= 64 ? v : 64;     // tuning switch (test build)
        const bool packed = chain_lanes == 64;                   // lanes to spare: full waves of 64 pairs (see k_k1p_ctxchain)
        const uint32_t groups = packed ? 0u : (n_states + chain_lanes - 1) / chain_lanes;
        const uint32_t whole_waves = packed ? uint32_t((uint64_t(n_slices) * n_states + 63) / 64) : n_slices * groups;
        const dim3 whole((whole_waves + kChainWaves - 1) / kChainWaves), block(64 * kChainWaves);
        // Long slices: the chains in n_segs segments (walks from both extreme states, see k_k1p_chain_seg), then whatever pair
        // they could not settle start to end; short ones (a segment would be a chunk or two): start to end at once.
        // (what the segments buy is latency: eight times the lanes for an eighth of the dependent length.  Once a batch has lanes
        // enough to keep the LDS busy with table look-ups -- config 4: 16 384 slices x 86 contexts -- they only add look-ups: measured
        // 17.5 against 15.5 ms per step there, 0.277 against 0.289 ms for the chains of config 2's 512 slices, and the smaller the
        // batch the larger the gain: the chains do not get shorter with fewer slices, everything else does.)
        const bool few_lanes = uint64_t(n_slices) * n_states <= 49152;
        // segments: as many as keep the launch in one round of workgroups, at least four chunks each on average, at most kMaxChainSegs
        const uint32_t pair_waves = uint32_t((uint64_t(n_slices) * n_states + 63) / 64);
        uint32_t n_segs = kChainWaveSlots / (pair_waves * (hint && hint->sharing ? hint->sharing : 1u));   // (measured with two parts of config 2: 8 / 11 / 12 / 16 segments 1.35 / 1.33 / 1.36 / 1.36 ms)
        n_segs = n_segs > kMaxChainSegs ? kMaxChainSegs : n_segs;
        const uint64_t by_length = uint64_t(pl->total_chunks) / (uint64_t(n_slices) * 4);
        if (n_segs > by_length) n_segs = uint32_t(by_length);
        if (const uint32_t v = test_hooks().chain_nsegs) n_segs = v < kMaxChainSegs ? v : kMaxChainSegs;
        if (n_segs >= 2 && !test_hooks().chain_whole && (few_lanes || test_hooks().chain_segments)) {
            SegSummary *summ = reinterpret_cast<SegSummary *>(w + L.summ);
            const dim3 seg_grid((pair_waves * n_segs + kSegWaves - 1) / kSegWaves), seg_block(64 * kSegWaves);
            hipLaunchKernelGGL(k_k1p_chain_seg, seg_grid, seg_block, 0, s, p, n_slices, n_segs, status, tn, lbits, lend, est, summ);
            hipLaunchKernelGGL(k_k1p_chain_fix, seg_grid, seg_block, 0, s, p, n_slices, n_segs, status, tn, lbits, lend, init_states, est, summ,
                               final_states, test_hooks().chain_force_redo);
        } else {
            hipLaunchKernelGGL(k_k1p_ctxchain, whole, block, 0, s, p, n_slices, groups, chain_lanes, status, tn, lbits, lend, init_states, est, final_states);
        }
    }
    // the waves of a workgroup share the two tables (12 KiB, static); each has its own state rows: as many waves as fit (1 .. 4)
    const uint32_t per_wave = ((n_states + 8) / 4) * 256;
    const uint32_t replay_waves = per_wave * 4 <= 48 * 1024 ? 4 : per_wave * 2 <= 48 * 1024 ? 2 : 1;
    const uint32_t replay_lds = replay_waves * per_wave;
    auto replay = tile_codes ? k_k1p_replay<true> : k_k1p_replay<false>;
    if (replay_lds > 48 * 1024) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(replay), hipFuncAttributeMaxDynamicSharedMemorySize, int(replay_lds));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(replay, dim3((pl->total_chunks + 64 * replay_waves - 1) / (64 * replay_waves)), dim3(64 * replay_waves),
                       replay_lds, s, p, pl->total_chunks, reinterpret_cast<const uint32_t *>(est), res, status, stretch, max_stretch);
    return hipGetLastError();
}

// Phases B-D: resolved codes -> bytes.  `w` is workspace for stretches, entries, totals, digit sums.
// `have` != nullptr: the stretch summaries (phase B1) have been made already, by k_k1p_replay.
static hipError_t launch_code(hipStream_t s, const Plan &p, uint32_t n_slices, const avr_chunk_plan *pl, uint8_t *w,
                              const uint8_t *res, uint32_t max_stretch, uint8_t *out, const uint64_t *out_off,
                              uint32_t *out_len, int32_t *status, const Stretch *have = nullptr, bool tile_codes = false) {
    Stretch *st_own = reinterpret_cast<Stretch *>(w);        w += up256(uint64_t(pl->total_chunks) * sizeof(Stretch));
    const Stretch *st = have ? have : st_own;
    Entry *en = reinterpret_cast<Entry *>(w);                w += up256(uint64_t(pl->total_chunks) * sizeof(Entry));
    SliceTotals *tot = reinterpret_cast<SliceTotals *>(w);   w += up256(n_slices * sizeof(SliceTotals));
    uint32_t *S = reinterpret_cast<uint32_t *>(w);               w += up256(pl->dig_total * 4 + 16);
    const uint32_t chunk_blocks = (pl->total_chunks + 255) / 256;
    if (!have) {
        // codes from the caller, slice-major: B1 reads them lane by lane once and leaves a wave-interleaved copy for phase C
        uint8_t *tile = w;
        hipLaunchKernelGGL(k_k1p_b1, dim3(chunk_blocks), dim3(256), 0, s, p, pl->total_chunks, res, status, st_own, max_stretch, tile);
        res = tile;
        tile_codes = true;
    }
    hipLaunchKernelGGL(k_k1p_b2, dim3(n_slices), dim3(256), 0, s, p, status, st, en, tot, S);
    if (tile_codes) hipLaunchKernelGGL(k_k1p_c<true>, dim3(chunk_blocks), dim3(256), 0, s, p, pl->total_chunks, res, st, en, tot, S);
    else hipLaunchKernelGGL(k_k1p_c<false>, dim3(chunk_blocks), dim3(256), 0, s, p, pl->total_chunks, res, st, en, tot, S);
    const uint32_t force_retry_every = test_hooks().k1p_force_retry_every;   // test build only, see k_k1p_d
    hipLaunchKernelGGL(k_k1p_d, dim3(n_slices), dim3(256), 0, s, p, tot, S, out, out_off, out_len, status, force_retry_every);
    return hipGetLastError();
}

hipError_t launch_densemap(hipStream_t s, const uint32_t *used, uint16_t *table, uint16_t *index, uint32_t *n_dense) {
    hipLaunchKernelGGL(k_k1p_densemap, dim3(1), dim3(1024), 0, s, used, table, index, n_dense);
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
// the code buffer between the two stages of launch_k1p: interleaved by tiles of 64 chunks (TileCodes)
static inline uint64_t codes_bytes(const avr_chunk_plan *pl) {
    const uint64_t tiled = tile_codes_bytes(pl->total_chunks), linear = pl->res_total + 32;
    return tiled > linear ? tiled : linear;
}
size_t k1p_workspace_bytes(size_t n_slices, uint32_t n_states, const avr_chunk_plan *pl) {
    return size_t(up256(codes_bytes(pl)) + resolve_ws_bytes(n_slices, n_states, pl)) + k1p_code_workspace_bytes(n_slices, pl);
}

constexpr uint32_t kK1pCensusStride = 16;

static uint32_t census_stride() {
    return test_hooks().census_stride ? test_hooks().census_stride : kK1pCensusStride;
}

// One pass of the whole path over the slices whose status is AVR_SLICE_OK.
static hipError_t k1p_pass(hipStream_t s, const Plan &p, uint32_t n_slices, const uint8_t *init_states, uint32_t n_states,
                           const avr_chunk_plan *pl, uint8_t *w, uint8_t *res, uint8_t *out, const uint64_t *out_off, uint32_t *out_len,
                           int32_t *status, uint8_t *final_states, const DenseHint *hint, uint32_t stride, bool second_pass,
                           uint32_t *retry_count) {
    const Stretch *st = nullptr;
    // the codes between the two stages stay inside this call: wave-interleaved (TileCodes)
    hipError_t e = launch_resolve(s, p, n_slices, init_states, pl, w, res, status, final_states, kMaxStretch, &st, hint, stride, second_pass,
                                  retry_count, true);
    if (e != hipSuccess) return e;
    w += resolve_ws_bytes(n_slices, n_states, pl);
    e = launch_code(s, p, n_slices, pl, w, res, kMaxStretch, out, out_off, out_len, status, st, true);
    if (e != hipSuccess) return e;
    // slices the scheme declined (status AVR_SLICE_RETRY_SERIAL) are coded by the serial kernel
    return launch_cabac_encode(false, s, p.recs, p.rec_off, p.n_bins, nullptr, n_slices, init_states, n_states, out, out_off,
                               out_len, status, nullptr, AVR_SLICE_RETRY_SERIAL);
}

// The second pass: the slices k_k1p_local set aside (a bin in a context the sampled census missed) once more, with every
// record counted; the finished ones are parked under AVR_SLICE_DONE meanwhile, which every kernel of the path skips.
static hipError_t k1p_second_pass(hipStream_t s, const Plan &p, uint32_t n_slices, const uint8_t *init_states, uint32_t n_states,
                                  const avr_chunk_plan *pl, uint8_t *w, uint8_t *res, uint8_t *out, const uint64_t *out_off,
                                  uint32_t *out_len, int32_t *status, uint8_t *final_states, bool code) {
    const dim3 grid((n_slices + 255) / 256), block(256);
    hipLaunchKernelGGL(k_k1p_swap, grid, block, 0, s, status, n_slices, AVR_SLICE_OK, AVR_SLICE_DONE, AVR_SLICE_RETRY_CENSUS, AVR_SLICE_OK);
    hipError_t e = code ? k1p_pass(s, p, n_slices, init_states, n_states, pl, w, res, out, out_off, out_len, status, final_states, nullptr, 1,
                                   true, nullptr)
                        : launch_resolve(s, p, n_slices, init_states, pl, w, res, status, final_states, kMaxStretch, nullptr, nullptr, 1, true);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_k1p_swap, grid, block, 0, s, status, n_slices, AVR_SLICE_DONE, AVR_SLICE_OK, AVR_SLICE_DONE, AVR_SLICE_OK);
    return hipGetLastError();
}

hipError_t launch_k1p(hipStream_t s, const uint16_t *recs, const uint64_t *rec_off, const uint32_t *n_bins,
                      uint32_t n_slices, const uint8_t *init_states, uint32_t n_states, const avr_chunk_plan *pl,
                      void *workspace, uint8_t *out, const uint64_t *out_off, uint32_t *out_len, int32_t *status,
                      uint8_t *final_states, const DenseHint *hint) {
    if (n_slices == 0) return hipSuccess;
    uint8_t *w = static_cast<uint8_t *>(workspace);
    uint8_t *res = w;                                        w += up256(codes_bytes(pl));
    const Plan p{recs, rec_off, n_bins, pl->res_off, pl->chunk_base, pl->chunk_slice, pl->blk_base, pl->blk_slice,
                 pl->dig_off, 0, n_states, nullptr, nullptr};
    const uint32_t stride = census_stride();
    uint32_t retry = 0;
    if (hint && hint->host_retry) *hint->host_retry = 0;
    hipError_t e = k1p_pass(s, p, n_slices, init_states, n_states, pl, w, res, out, out_off, out_len, status, final_states, hint, stride, false,
                            &retry);
    if (e != hipSuccess || !retry) return e;
    return k1p_second_pass(s, p, n_slices, init_states, n_states, pl, w, res, out, out_off, out_len, status, final_states, true);
}

hipError_t launch_k1p_retry(hipStream_t s, const uint16_t *recs, const uint64_t *rec_off, const uint32_t *n_bins,
                            uint32_t n_slices, const uint8_t *init_states, uint32_t n_states, const avr_chunk_plan *pl,
                            void *workspace, uint8_t *out, const uint64_t *out_off, uint32_t *out_len, int32_t *status,
                            uint8_t *final_states) {
    if (n_slices == 0) return hipSuccess;
    uint8_t *w = static_cast<uint8_t *>(workspace);
    uint8_t *res = w;                                        w += up256(codes_bytes(pl));
    const Plan p{recs, rec_off, n_bins, pl->res_off, pl->chunk_base, pl->chunk_slice, pl->blk_base, pl->blk_slice,
                 pl->dig_off, 0, n_states, nullptr, nullptr};
    return k1p_second_pass(s, p, n_slices, init_states, n_states, pl, w, res, out, out_off, out_len, status, final_states, true);
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
    uint8_t *w = static_cast<uint8_t *>(workspace);
    const uint32_t stride = census_stride();
    uint32_t retry = 0;
    hipError_t e = launch_resolve(s, p, n_slices, init_states, pl, w, codes, status, final_states, kMaxStretch, nullptr, nullptr, stride, false,
                                  &retry);
    if (e != hipSuccess || !retry) return e;
    return k1p_second_pass(s, p, n_slices, init_states, n_states, pl, w, codes, nullptr, nullptr, nullptr, status, final_states, false);
}
// ... and phases B-D from resolved codes (no stretch is declined for its length here: a stretch without an
// LPS is simply walked to its end by one lane); a slice phase D hands back is coded by k_cabac_encode_codes
size_t k1p_code_workspace_bytes(size_t n_slices, const avr_chunk_plan *pl) {
    return size_t(up256(uint64_t(pl->total_chunks) * sizeof(Stretch)) + up256(uint64_t(pl->total_chunks) * sizeof(Entry)) +
                  up256(n_slices * sizeof(SliceTotals)) + up256(pl->dig_total * 4 + 16) + up256(tile_codes_bytes(pl->total_chunks)));
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
