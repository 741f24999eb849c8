// TEST BUILD ONLY.  Runs the product's K1p per-lane functions (avrecode-ms_amd/csrc/avr_k1p.h,
// the same code the HIP kernels wrap) sequentially on the CPU, so the parallel algorithm can be
// checked against the oracle in the build container, which has no GPU.  Not part of the product
// library and not a CPU path of it: nothing ships or links this file.
#include <cstdint>
#include <cstring>
#include <vector>

#include "avr_div.h"
#include "avr_k1p.h"
#include "avr_tables.h"

using namespace avr;
using namespace avr::k1p;

namespace {
constexpr CabacTables kT = make_cabac_tables();
struct HostAdder {
    std::vector<uint32_t> &S;
    void store(uint32_t i, uint32_t v) { S[i] = v; stores++; }
    void add(uint32_t i, uint32_t v) { S[i] += v; adds++; }
    void flush() {}
    size_t stores = 0, adds = 0;
};
}  // namespace

extern "C" {

// Phase A, serial: K1 records + initial states -> resolved codes.  Returns 0, or 3 (bad record).
int k1p_emul_resolve(const uint16_t *recs, size_t n, uint8_t *states, size_t n_states, uint8_t *res, size_t *n_res) {
    size_t m = 0;
    bool done = false;
    for (size_t i = 0; i < n; i++) {
        const uint32_t bin = recs[i] & 1, sel = (recs[i] >> 1) & 0x7ff;
        if (done) return 3;
        if (sel < 1024) {
            if (sel >= n_states) return 3;
            const uint32_t s = states[sel];
            res[m++] = uint8_t(code_context(s, bin));
            states[sel] = (bin != (s & 1)) ? kT.mlps_state[127 - s] : kT.mlps_state[128 + s];
        } else if (sel == 1024) res[m++] = uint8_t(kCodeBypass | bin);
        else if (sel == 1025) { res[m++] = uint8_t(code_terminate(bin)); done = bin; }
        else return 3;
    }
    *n_res = m;
    for (size_t k = m; k % 16; k++) res[k] = uint8_t(kCodePad);
    return 0;
}

// Phases B1, B2, C, D on resolved codes.  info (optional, 8 words): n_active, t_total, r_final, bad,
// plain stores, atomic adds, n_digits, -.
size_t k1p_emul_encode_resolved(const uint8_t *res, size_t n, uint8_t *out, size_t cap, uint32_t *info) {
    uint32_t rows[64];
    for (int p = 0; p < 64; p++) rows[p] = kT.packed[2 * p][0];
    CodeEntry codes[256];
    for (uint32_t c = 0; c < 256; c++) codes[c] = code_entry(c, rows);
    const uint32_t n_chunks = n ? uint32_t((n + kChunk - 1) / kChunk) : 1;
    std::vector<Stretch> st(n_chunks);
    std::vector<Entry> en(n_chunks);
    for (uint32_t c = 0; c < n_chunks; c++) b1_stretch(res, uint32_t(n), c, codes, kMaxStretch, &st[c]);
    SliceTotals tot;
    b2_chain(st.data(), n_chunks, en.data(), &tot);
    const uint32_t nd = ref_digits(tot.t_total);
    std::vector<uint32_t> S(nd + 4, 0);
    HostAdder add{S};
    uint32_t active = 0;
    CodeEntryC codes_c[256];
    for (uint32_t c = 0; c < 256; c++) codes_c[c] = code_entry_c(codes[c]);
    for (uint32_t c = 0; c < n_chunks; c++)
        if (st[c].first != kNone) { c_stretch(res, st[c], en[c], c, codes_c, add); active++; }
    const uint32_t len = d_slice(S.data(), tot, out, uint32_t(cap));
    if (info) {
        info[0] = active; info[1] = tot.t_total; info[2] = tot.r_final; info[3] = tot.bad;
        info[4] = uint32_t(add.stores); info[5] = uint32_t(add.adds); info[6] = nd; info[7] = 0;
    }
    return len;
}

// avr_div.h against the CPU's own divide: n random below 2^63 + 1 (and the edges), every divisor 1..255.  Returns the
// number of mismatches.
uint64_t div_emul_check(uint64_t seed, uint64_t rounds) {
    uint64_t bad = 0, x = seed * 0x9e3779b97f4a7c15ull + 1;
    const uint64_t edges[] = {0, 1, 255, 256, 0xffffffffull, 0x100000000ull, 0x1ffffffffull, (1ull << 51) - 1, 1ull << 51, (1ull << 55) + 12345,
                              (1ull << 63) - 1, 1ull << 63};
    for (uint64_t r = 0; r < rounds + sizeof edges / sizeof *edges; r++) {
        uint64_t n;
        if (r < sizeof edges / sizeof *edges) n = edges[r];
        else {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            n = x >> 1;                                            // < 2^63
            if ((r & 7) == 0) n -= n % ((x >> 3) % 255 + 1);       // exact multiples: the case a biased reciprocal gets wrong
            if ((r & 15) == 1) n = (n >> 32 << 32) | 0xffffffffull;
        }
        for (uint32_t d = 1; d < 256; d++)
            if (div_u64_small_f64(n, double(d), 1.0 / double(d)) != n / d) bad++;
    }
    return bad;
}

}  // extern "C"
