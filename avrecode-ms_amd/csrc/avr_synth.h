// Synthetic H.264 CABAC bin streams for the BASELINE.json configurations.
//
// The reference has no generator: its inputs are real MP4s decoded by the libavcodec-hooks
// fork (recode.cpp:120-141), neither of which exists offline.  This header produces, from
// (seed, slice index) alone, the sequence of (bin, selector) pairs such a decode would hand
// to Driver::cabac_decoder::get / get_bypass / get_terminate (recode.cpp:156-167), following
// the H.264 slice_data / residual_block_cabac syntax closely enough that context usage,
// bypass share (~18 %) and bins-per-bit (~1.3) look like a real stream:
//
//   per macroblock   mb_skip_flag (ctxIdx 11-13), mb_type (14-20), mvd (40-53 + bypass
//                    suffix/sign), ref_idx (54-59), coded_block_pattern (73-84),
//                    mb_qp_delta (60-63), then 4x4 luma residual blocks, end_of_slice
//   residual block   coded_block_flag (85+8+inc), significant_coeff_flag (105+29+i),
//   (ctxBlockCat 2)  last_significant_coeff_flag (166+29+i), coeff_abs_level_minus1
//                    (227+20+inc: TU prefix on contexts, Exp-Golomb-0 suffix on bypass),
//                    coeff_sign_flag on bypass
//
// The same code is compiled for the host and for gfx950 so CPU checks and GPU runs see
// identical records.  It is input generation, not part of the coding path.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define AVR_HD __host__ __device__ inline
#else
#define AVR_HD inline
#endif

namespace avr {

struct SynthRng {
    uint64_t s;
    AVR_HD explicit SynthRng(uint64_t seed, uint64_t slice) {
        s = seed ^ (slice * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull);
        next(); next();
    }
    AVR_HD uint32_t next() {                       // splitmix64, high half
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return uint32_t((z ^ (z >> 31)) >> 32);
    }
    // true with probability p16 / 65536
    AVR_HD bool chance(uint32_t p16) { return (next() >> 16) < p16; }
    AVR_HD uint32_t below(uint32_t n) { return uint32_t((uint64_t(next()) * n) >> 32); }
};

struct SynthShape {
    uint32_t n_mbs;          // macroblocks in the slice (0: residual-only stream)
    uint32_t n_blocks;       // residual-only: number of 4x4 blocks
    uint32_t p_skip16;       // P(mb_skip_flag = 1)
    uint32_t p_cbp16;        // P(an 8x8 luma quadrant is coded)
    uint32_t p_cbf16;        // P(coded_block_flag = 1)
    uint32_t nz0_16;         // P(coefficient 0 is nonzero); decays along the zig-zag
    uint32_t decay16;        // per-position multiplier of that probability (x / 65536)
    uint32_t p_level16;      // P(|level| grows by one more) -- geometric level distribution
    uint32_t n_states;       // state bytes the stream declares
};

// Per-slice shape of workload `w`; scale in 1/1000.
AVR_HD SynthShape synth_shape(int w, uint32_t scale_permille, uint64_t seed, uint64_t slice) {
    SynthShape sh{};
    sh.p_skip16 = 16384;     // 0.25
    sh.p_cbp16 = 27525;      // 0.42
    sh.p_cbf16 = 45875;      // 0.70
    sh.nz0_16 = 63570;       // 0.97
    sh.decay16 = 45875;      // 0.70
    sh.p_level16 = 13107;    // 0.20
    sh.n_states = 460;
    uint64_t mbs = 0;
    if (w == 2) {
        mbs = 8160;                                   // 1080p: 120 x 68 macroblocks
    } else if (w == 3) {                              // ragged: log-normal around config 2, sigma 0.5
        SynthRng r(seed ^ 0x3333333333333333ull, slice);
        // sum of 12 uniforms - 6 ~ N(0,1); exp via 2^x with a small table-free approximation
        int32_t acc = -6 * 65536;
        for (int i = 0; i < 12; i++) acc += int32_t(r.next() >> 16);
        // x = 0.5 * z * log2(e) in 16.16 fixed point
        int32_t x = int32_t((int64_t(acc) * 47274) >> 16);      // 0.5*1.442695 = 0.72135 -> 47274/65536
        int32_t ip = x >> 16;                                    // floor
        uint32_t fp = uint32_t(x) & 0xffff;
        // 2^f ~ 1 + f*(0.6565 + 0.3435 f) on [0,1)
        uint64_t pow2f = 65536 + ((uint64_t(fp) * (43025 + ((22511 * uint64_t(fp)) >> 16))) >> 16);
        uint64_t m = 8160 * pow2f;                               // 16.16
        if (ip >= 0) m <<= (ip > 6 ? 6 : ip); else m >>= (-ip > 6 ? 6 : -ip);
        mbs = m >> 16;
        if (mbs < 64) mbs = 64;
    } else if (w == 4) {
        mbs = 4080;                                   // 4K60, 8 slices/frame: 32 640 / 8
        sh.p_skip16 = 22938;                          // 0.35: ~100 Mb/s at 60 fps is fewer bits per MB
        sh.p_cbp16 = 24904;                           // 0.38
    } else {                                          // 5: residual-only, 64 luma 4x4 blocks, QP 26
        sh.n_blocks = 64;
        sh.p_cbf16 = 60293;                           // 0.92
        sh.nz0_16 = 64225;                            // 0.98
        sh.decay16 = 52429;                           // 0.80
        sh.p_level16 = 19661;                         // 0.30
        sh.n_states = 260;                            // highest ctxIdx used is 256
    }
    if (w == 5) {
        uint64_t nb = (uint64_t(sh.n_blocks) * scale_permille + 999) / 1000;
        sh.n_blocks = uint32_t(nb < 1 ? 1 : nb);
    } else {
        uint64_t n = (mbs * scale_permille + 999) / 1000;
        sh.n_mbs = uint32_t(n < 1 ? 1 : n);
    }
    return sh;
}

// Initial state byte of context `ctx` for a slice (2*pStateIdx + valMPS, pStateIdx <= 62):
// a fixed pseudo-table indexed by (ctx, slice mod 52), standing in for the (m,n,SliceQP)
// initialisation of H.264 9.3.1.1 that libavcodec performs before the first bin.
AVR_HD uint8_t synth_init_state(uint32_t ctx, uint64_t seed, uint64_t slice) {
    uint64_t z = seed ^ (uint64_t(ctx) * 0xA24BAED4963EE407ull) ^ ((slice % 52) * 0x9FB21C651E98DF25ull);
    z = (z ^ (z >> 32)) * 0xD6E8FEB86659FD93ull;
    z ^= z >> 29;
    return uint8_t(z % 126);
}

// ---- the syntax walker.  Sink needs: void put(uint32_t bin, uint32_t sel).

template <class Sink>
AVR_HD void synth_residual_block(SynthRng &r, const SynthShape &sh, Sink &sink) {
    const uint32_t CBF = 85 + 8, SIG = 105 + 29, LAST = 166 + 29, ABS = 227 + 20;
    const bool coded = r.chance(sh.p_cbf16);
    const uint32_t cbf_inc = r.next() >> 30;
    sink.put(coded, CBF + cbf_inc);
    if (!coded) return;
    uint32_t nz_mask = 0, p = sh.nz0_16;
    for (int i = 0; i < 16; i++) {
        if (r.chance(p)) nz_mask |= 1u << i;
        p = uint32_t((uint64_t(p) * sh.decay16) >> 16);
    }
    if (!nz_mask) nz_mask = 1;
    int last = 31 - __builtin_clz(nz_mask);
    for (int i = 0; i < 15 && i <= last; i++) {
        const uint32_t sig = (nz_mask >> i) & 1;
        sink.put(sig, SIG + i);
        if (sig) {
            sink.put(i == last, LAST + i);
            if (i == last) break;
        }
    }
    uint32_t eq1 = 0, gt1 = 0;
    for (int i = last; i >= 0; i--) {
        if (!((nz_mask >> i) & 1)) continue;
        uint32_t lvl = 0;                               // coeff_abs_level_minus1
        while (lvl < 40 && r.chance(sh.p_level16)) lvl++;
        const uint32_t inc0 = gt1 ? 0 : (1 + eq1 > 4 ? 4 : 1 + eq1);
        sink.put(lvl > 0, ABS + inc0);
        if (lvl > 0) {
            const uint32_t inc = 5 + (gt1 > 4 ? 4 : gt1);
            const uint32_t ones = lvl < 14 ? lvl : 14;
            for (uint32_t k = 1; k < ones; k++) sink.put(1, ABS + inc);
            if (lvl < 14) sink.put(0, ABS + inc);
            else {                                       // Exp-Golomb order 0 suffix, bypass
                uint32_t v = lvl - 14, k = 0;
                while (v >= (1u << k)) { sink.put(1, 1024); v -= 1u << k; k++; }
                sink.put(0, 1024);
                while (k--) sink.put((v >> k) & 1, 1024);
            }
            gt1++;
        } else {
            eq1++;
        }
        const uint32_t sign = r.next() >> 31;            // coeff_sign_flag
        sink.put(sign, 1024);
    }
}

template <class Sink>
AVR_HD void synth_mvd_component(SynthRng &r, uint32_t base, Sink &sink) {
    uint32_t v = 0;
    while (v < 24 && r.chance(29491)) v++;               // 0.45: |mvd| geometric
    const uint32_t inc0 = (r.next() >> 30) % 3;
    sink.put(v > 0, base + inc0);
    if (v == 0) return;
    const uint32_t ones = v < 9 ? v : 9;                 // UEG3 prefix, uCoff 9
    for (uint32_t k = 1; k < ones; k++) sink.put(1, base + (k < 4 ? 2 + k : 6));
    if (v < 9) sink.put(0, base + (ones < 4 ? 2 + ones : 6));
    else {
        uint32_t s = v - 9, k = 3;
        while (s >= (1u << k)) { sink.put(1, 1024); s -= 1u << k; k++; }
        sink.put(0, 1024);
        while (k--) sink.put((s >> k) & 1, 1024);
    }
    const uint32_t sign = r.next() >> 31;
    sink.put(sign, 1024);
}

template <class Sink>
AVR_HD void synth_slice(int workload, uint32_t scale_permille, uint64_t seed, uint64_t slice, Sink &sink) {
    const SynthShape sh = synth_shape(workload, scale_permille, seed, slice);
    SynthRng r(seed, slice);
    if (sh.n_mbs == 0) {
        for (uint32_t b = 0; b < sh.n_blocks; b++) synth_residual_block(r, sh, sink);
        sink.put(1, 1025);
        return;
    }
    for (uint32_t mb = 0; mb < sh.n_mbs; mb++) {
        // every random draw is sequenced in its own statement: argument evaluation order is
        // unspecified in C++ and must not differ between the host and the device compiler
        const bool skip = r.chance(sh.p_skip16);
        const uint32_t skip_inc = r.below(3);
        sink.put(skip, 11 + skip_inc);                   // mb_skip_flag
        if (!skip) {
            const bool t0 = r.chance(6554);              // mb_type bin 0 (0.10 intra-ish)
            const uint32_t t0_inc = r.below(3);
            sink.put(t0, 14 + t0_inc);
            const bool t1 = r.chance(45875);             // 0.70
            sink.put(t1, 17);
            if (r.chance(32768)) {
                const bool t2 = r.chance(16384);
                const uint32_t t2_inc = r.below(3);
                sink.put(t2, 18 + t2_inc);
            }
            const bool ref = r.chance(9830);             // ref_idx_l0 > 0, 0.15
            const uint32_t ref_inc = r.below(4);
            sink.put(ref, 54 + ref_inc);
            synth_mvd_component(r, 40, sink);
            synth_mvd_component(r, 47, sink);
            uint32_t cbp = 0;
            for (int q = 0; q < 4; q++) {                // coded_block_pattern luma
                const bool c = r.chance(sh.p_cbp16);
                const uint32_t c_inc = r.below(4);
                cbp |= uint32_t(c) << q;
                sink.put(c, 73 + c_inc);
            }
            const bool chroma = r.chance(22938);         // 0.35
            const uint32_t chroma_inc = r.below(4);
            sink.put(chroma, 77 + chroma_inc);
            if (chroma) {
                const bool c2 = r.chance(26214);
                const uint32_t c2_inc = r.below(4);
                sink.put(c2, 81 + c2_inc);
            }
            if (cbp || chroma) {
                const bool dqp = r.chance(3277);         // mb_qp_delta != 0, 0.05
                const uint32_t dqp_inc = r.below(2);
                sink.put(dqp, 60 + dqp_inc);
                for (int q = 0; q < 4; q++)
                    if ((cbp >> q) & 1)
                        for (int b = 0; b < 4; b++) synth_residual_block(r, sh, sink);
            }
        }
        sink.put(mb + 1 == sh.n_mbs, 1025);              // end_of_slice_flag
    }
}

// ---- sinks

// Record sinks implement put_record(uint16_t); the two adapters below turn the walker's
// (bin, selector) pairs into K1 or K2 records.
struct CountSink {
    uint32_t n = 0;
    AVR_HD void put_record(uint16_t) { n++; }
};

template <class Sink>
struct CabacSink {
    Sink &inner;
    AVR_HD explicit CabacSink(Sink &s) : inner(s) {}
    AVR_HD void put(uint32_t bin, uint32_t sel) { inner.put_record(uint16_t(bin | (sel << 1))); }
};

// Adaptive {pos,neg} model of the compress direction (recode.cpp:1064, 823-827, 1037-1052),
// per slice and keyed by selector, turned into K2 range records.  The reference's estimator
// map is per file (recode.cpp:1314, 669-672); starting every synthetic slice at {1,1} keeps
// slices independent, which is also how the host resolves records (SURVEY.md 8(a) note).
template <class Sink>
struct ModelSink {
    Sink &inner;
    uint8_t pos[1026], neg[1026];
    AVR_HD explicit ModelSink(Sink &s) : inner(s) {
        for (int i = 0; i < 1026; i++) { pos[i] = 1; neg[i] = 1; }
    }
    AVR_HD void put(uint32_t bin, uint32_t sel) {
        inner.put_record(uint16_t(bin | (uint32_t(pos[sel]) << 1) | (uint32_t(neg[sel]) << 8)));
        if (bin) pos[sel]++; else neg[sel]++;
        if (uint32_t(pos[sel]) + neg[sel] > 0x60) {
            pos[sel] = uint8_t((pos[sel] + 1) / 2);
            neg[sel] = uint8_t((neg[sel] + 1) / 2);
        }
    }
};

}  // namespace avr
