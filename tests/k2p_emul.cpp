// CPU emulation of the three passes of K2p (csrc/avr_k2p.h) -- the very functions the kernels run, chunk by chunk.
// Test build only (tests/test_k2p_emul.py compares with the oracle).
#include <cstdint>
#include <cstring>
#include <vector>

#include "avr_k2p.h"

using namespace avr::k2p;

extern "C" {

// Returns the number of bytes (written up to cap), or SIZE_MAX for a zero-probability bin.  info[0] = chunks,
// info[1] = bytes emitted before finish(), info[2] = largest digit sum seen, info[3] = bins the double-precision form of pass 1 walked before handing over (all of them, normally).
size_t k2p_emul_encode(const uint16_t *recs, size_t n, uint32_t chunk_bins, uint8_t *out, size_t cap, uint32_t *info) {
    const auto div = [](uint64_t r, uint32_t t) { return t ? r / t : 0; };
    const uint32_t cb = chunk_bins ? chunk_bins : kChunk;
    // pass 1: the range recurrence; range and bytes emitted at the start of every chunk
    std::vector<uint64_t> ck_range;
    std::vector<uint32_t> ck_pos;
    uint64_t range = kOne, dummy = 0;
    uint32_t p = 0;
    RangeFP rf = fp_from_u64(kOne);                        // the double-precision form of the recurrence (the kernel's pass 1), in step
    uint32_t vmin = 0xffffffffu;
    const FpConsts K = fp_consts();
    bool fp_live = true;                                   // until it hands the slice to the integer form (a range below 2^39)
    uint32_t fp_bins = 0;
    for (size_t i = 0; i < n; i++) {
        if (fp_live) {
            const uint32_t k = range_step_fp(rf, vmin, fp_operands(recs[i]), K) / 8;
            if (vmin < kTwo39Hi) fp_live = false;
            else {
                uint64_t r3 = range;
                uint32_t p3 = 0;
                range_step(r3, p3, recs[i], div);
                if (fp_to_u64(rf) != r3 || k != p3 || !(rf.H >= 0) || !(rf.L <= kTwo47 && rf.L >= -kTwo47)) return SIZE_MAX - 3;
                fp_bins++;
            }
        }
        if (i % cb == 0) { ck_range.push_back(range); ck_pos.push_back(p); }
        // the branch-free form the kernel's pass 1 uses, checked against bin<false> on the fly
        uint64_t r2 = range;
        uint32_t p2 = p;
        const bool ok2 = range_step(r2, p2, recs[i], div);
        if (!bin<false>(dummy, range, recs[i], div, [&](uint32_t) { p++; })) return ok2 ? SIZE_MAX - 2 : SIZE_MAX;
        if (!ok2 || r2 != range || p2 != p) return SIZE_MAX - 2;
    }
    const uint32_t P = p;
    // pass 2: every chunk from low = 0, its bytes added into the sums
    std::vector<uint32_t> S(size_t(P) + kTail, 0);
    uint32_t top = 0;
    for (size_t c = 0; c < ck_range.size(); c++) {
        uint64_t low = 0, r = ck_range[c];
        uint32_t q = ck_pos[c];
        const size_t i1 = (c + 1) * cb < n ? (c + 1) * cb : n;
        auto add = [&](uint32_t v) { S[q] += v; if (S[q] > top) top = S[q]; q++; };
        for (size_t i = c * cb; i < i1; i++) bin<true>(low, r, recs[i], div, add);
        leftover(low, add);
    }
    // pass 3: carries from the last byte; the eight positions past the last emitted byte hold the final low
    uint32_t carry = 0;
    std::vector<uint8_t> bytes(S.size());
    for (size_t i = S.size(); i-- > 0;) {
        const uint32_t v = S[i] + carry;
        bytes[i] = uint8_t(v);
        carry = v >> 8;
    }
    if (carry) return SIZE_MAX - 1;                        // cannot happen: the code string is below one
    uint8_t tail[9];
    uint32_t cy;
    const uint32_t n_tail = finish(low_from_tail(bytes.data() + P), range, tail, &cy);
    for (size_t i = P; cy && i-- > 0;) { bytes[i]++; cy = bytes[i] == 0; }
    size_t len = 0;
    for (uint32_t i = 0; i < P; i++, len++) if (len < cap) out[len] = bytes[i];
    for (uint32_t i = 0; i < n_tail; i++, len++) if (len < cap) out[len] = tail[i];
    if (info) { info[0] = uint32_t(ck_range.size()); info[1] = P; info[2] = top; info[3] = fp_bins; }
    return len;
}

// The double-precision form of the range recurrence alone against the integer form, bin by bin: returns the number of
// bins on which they agreed before the first difference (n = all), or n + 1 + (bins walked) when the form handed the slice
// over (a new range below 2^39) at that point.
size_t k2p_emul_fp_walk(const uint16_t *recs, size_t n) {
    const auto div = [](uint64_t r, uint32_t t) { return t ? r / t : 0; };
    uint64_t range = kOne;
    RangeFP rf = fp_from_u64(kOne);
    uint32_t vmin = 0xffffffffu;
    const FpConsts K = fp_consts();
    for (size_t i = 0; i < n; i++) {
        uint32_t p = 0;
        range_step(range, p, recs[i], div);
        const uint32_t k = range_step_fp(rf, vmin, fp_operands(recs[i]), K) / 8;
        if (vmin < kTwo39Hi) return n + 1 + i;
        if (fp_to_u64(rf) != range || k != p) return i;
    }
    return n;
}

}  // extern "C"
