// TEST INFRASTRUCTURE ONLY (oracle/).  Builds into oracle/_ref/libavr_ref.so.
//
// Thin extern "C" harness around the REFERENCE's own range coder, compiled
// from the header where it lies (-I/root/reference, see oracle/Makefile);
// no reference source is copied into this repository.  It exists to (1) pin
// oracle/avr_oracle.c byte-for-byte, (2) generate tests/golden/ vectors
// (tests/golden/make_golden.py) and (3) serve as bench.py's
// cpu_baseline kind "reference" where available.
//
// What is and is not the reference here:
//   * arithmetic_code<...>::encoder / ::decoder are the reference's, unmodified
//     (arithmetic_code.h compiles as is once <vector> and <stdexcept>, which it
//     uses at :117 and :200 without including, are included first).
//   * cabac_code.h CANNOT be compiled in this image: it includes
//     libavcodec/cabac.h from the un-vendored libavcodec-hooks fork
//     (cabac_code.h:10) and writing a stand-in header is not allowed.  The
//     functions named ref_cabac_* therefore drive the REAL
//     arithmetic_code<uint32_t,uint16_t,0x200> (the typedef at
//     cabac_code.h:18-24) with the CABAC layer (cabac_code.h:30-67) restated
//     below; the table values come from oracle/avr_oracle_tables.h.
//   * ref_model_* reproduces how recode.cpp drives the coder -- a
//     std::function probability callback over a
//     std::map<std::tuple<const void*,int,int>, estimator> (recode.cpp:325,
//     823-827, 1037-1052, 1064-1065, 1081-1082) -- so that the CPU baseline pays
//     the same per-bin costs the reference pays.  recode.cpp itself needs
//     FFmpeg, protoc and libprotobuf and is not buildable here.
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <map>
#include <stdexcept>
#include <tuple>
#include <vector>

#include "arithmetic_code.h"   // from /root/reference

extern "C" {
#include "avr_oracle_tables.h"
}

namespace {

typedef std::vector<uint8_t> bytes_t;
typedef std::back_insert_iterator<bytes_t> out_it;

size_t copy_out(const bytes_t &v, uint8_t *out, size_t cap) {
    std::memcpy(out, v.data(), v.size() < cap ? v.size() : cap);
    return v.size();
}

uint8_t g_lps_range[512], g_mlps_state[256];
struct table_init { table_init() { avr_oracle_build_tables(g_lps_range, g_mlps_state); } } g_table_init;

int floor_log2(uint64_t x) { int i = 0; while (x >>= 1) i++; return i; }

typedef arithmetic_code<uint64_t, uint16_t> test_code;            // test/arithmetic_code.cpp:93
typedef arithmetic_code<uint64_t, uint8_t> recoded_code;          // recode.cpp:322-323
typedef arithmetic_code<uint32_t, uint16_t, 0x200> cabac_code;    // cabac_code.h:18-24

}  // namespace

extern "C" {

// ---- test/arithmetic_code.cpp:93-111 (active branch): p = 1/2 on <uint64,uint16>
size_t ref_half_encode(const uint8_t *bins, size_t n, uint8_t *out, size_t cap) {
    bytes_t v;
    auto enc = make_encoder<test_code>(&v);
    for (size_t i = 0; i < n; i++) enc.put(bins[i] != 0, [](uint64_t range) { return range / 2; });
    enc.finish();
    return copy_out(v, out, cap);
}

void ref_half_decode(const uint8_t *bytes, size_t len, size_t n, uint8_t *bins_out) {
    bytes_t v(bytes, bytes + len);
    auto dec = make_decoder<test_code>(v);
    for (size_t i = 0; i < n; i++) bins_out[i] = (uint8_t)dec.get([](uint64_t range) { return range / 2; });
}

// ---- recoded_code encoder fed (bin,pos,neg) records; probability as recode.cpp:826
size_t ref_range_encode(const uint16_t *recs, size_t n, uint8_t *out, size_t cap, int *status) {
    bytes_t v;
    *status = 0;
    try {
        recoded_code::encoder<out_it, uint8_t> enc{std::back_inserter(v)};   // recode.cpp:1270-1271
        for (size_t i = 0; i < n; i++) {
            uint64_t pos = (recs[i] >> 1) & 0x7f, neg = (recs[i] >> 8) & 0x7f;
            enc.put(recs[i] & 1, [=](uint64_t range) { return (range / (pos + neg)) * pos; });
        }
        enc.finish();
    } catch (const std::runtime_error &) {
        *status = 1;   // "Encoder error: emitted a zero-probability symbol."
    }
    return copy_out(v, out, cap);
}

void ref_range_decode(const uint8_t *bytes, size_t len, const uint16_t *recs, size_t n, uint8_t *bins_out) {
    const char *b = reinterpret_cast<const char *>(bytes);
    recoded_code::decoder<const char *, uint8_t> dec(b, b + len);            // recode.cpp:1429-1430
    for (size_t i = 0; i < n; i++) {
        uint64_t pos = (recs[i] >> 1) & 0x7f, neg = (recs[i] >> 8) & 0x7f;
        bins_out[i] = (uint8_t)dec.get([=](uint64_t range) { return (range / (pos + neg)) * pos; });
    }
}

// ---- the reference range coder in its CABAC instantiation, driven by the restated layer
size_t ref_cabac_encode(const uint16_t *recs, size_t n, uint8_t *states, size_t n_states,
                        uint8_t *out, size_t cap, int *status) {
    bytes_t v;
    *status = 0;
    try {
        cabac_code::encoder<out_it, uint8_t> e(std::back_inserter(v), (cabac_code::fixed_one / 0x200) * 0x1FE);
        for (size_t i = 0; i < n; i++) {
            int bin = recs[i] & 1;
            unsigned sel = (recs[i] >> 1) & 0x7ff;
            if (sel < 1024) {
                if (sel >= n_states) { *status = 3; break; }
                uint8_t *state = &states[sel];
                bool is_lps = bin != (*state & 1);
                e.put(is_lps, [state](uint32_t range) {
                    int normalize = floor_log2(range / 0x100);
                    int approx = int(range >> (normalize - 1));
                    return uint32_t(g_lps_range[(approx & 0x180) + *state]) << normalize;
                });
                *state = is_lps ? g_mlps_state[127 - *state] : g_mlps_state[128 + *state];
            } else if (sel == 1024) {
                e.put(bin, [](uint32_t range) { return range / 2; });
            } else if (sel == 1025) {
                e.put(bin, [](uint32_t range) { return uint32_t(2) << floor_log2(range / 0x100); });
                if (bin) e.finish();
            } else {
                *status = 3;
                break;
            }
        }
        // ~encoder() runs finish() once more (arithmetic_code.h:100)
    } catch (const std::runtime_error &) {
        *status = 1;
    }
    return copy_out(v, out, cap);
}

// ---- "reference-faithful" CPU cost model for the compress direction: the generic coder
// behind std::function + std::map exactly as h264_model drives it.  `keys[i]` is the dense
// context id of bin i (standing in for the address the reference keys on).
size_t ref_model_range_encode(const uint8_t *bins, const uint16_t *keys, size_t n,
                              uint8_t *out, size_t cap) {
    struct estimator { int pos = 1, neg = 1; };
    typedef std::tuple<const void *, int, int> model_key;
    std::map<model_key, estimator> estimators;
    static const uint8_t anchor[2048] = {};
    bytes_t v;
    recoded_code::encoder<out_it, uint8_t> enc{std::back_inserter(v)};
    for (size_t i = 0; i < n; i++) {
        model_key key(&anchor[keys[i] & 2047], 0, 0);
        enc.put(bins[i], [&](uint64_t range) {
            auto *e = &estimators[key];
            int total = e->pos + e->neg;
            return (range / total) * e->pos;
        });
        auto *e = &estimators[key];
        if (bins[i]) e->pos++; else e->neg++;
        if (e->pos + e->neg > 0x60) { e->pos = (e->pos + 1) / 2; e->neg = (e->neg + 1) / 2; }
    }
    enc.finish();
    return copy_out(v, out, cap);
}

// ---- batch form of ref_model_range_encode for bench.py's cpu_baseline ("reference", compress direction): slices
// given as CABAC records (bin, selector) -- the selector is the model key, as in the synthetic workloads' own model
// (one estimator map per slice, started at {1, 1}: csrc/avr_synth.h ModelSink) -- one thread.
void ref_model_range_encode_batch(const uint16_t *cabac_recs, const uint64_t *off, size_t n_slices,
                                  uint8_t *out, const uint64_t *out_off, uint32_t *out_len) {
    std::vector<uint8_t> bins;
    std::vector<uint16_t> keys;
    for (size_t i = 0; i < n_slices; i++) {
        const size_t n = (size_t)(off[i + 1] - off[i]);
        bins.resize(n);
        keys.resize(n);
        for (size_t j = 0; j < n; j++) { bins[j] = cabac_recs[off[i] + j] & 1; keys[j] = cabac_recs[off[i] + j] >> 1; }
        out_len[i] = (uint32_t)ref_model_range_encode(bins.data(), keys.data(), n, out + out_off[i], (size_t)(out_off[i + 1] - out_off[i]));
    }
}

// ---- batch form of ref_cabac_encode, one thread, for bench.py's cpu_baseline ("reference")
void ref_cabac_encode_batch(const uint16_t *recs, const uint64_t *off, size_t n_slices,
                            const uint8_t *init_states, size_t n_states,
                            uint8_t *out, const uint64_t *out_off, uint32_t *out_len) {
    uint8_t states[1024];
    for (size_t i = 0; i < n_slices; i++) {
        std::memset(states, 0, sizeof states);
        std::memcpy(states, init_states + i * n_states, n_states);
        int st;
        out_len[i] = (uint32_t)ref_cabac_encode(recs + off[i], (size_t)(off[i + 1] - off[i]), states, n_states,
                                                out + out_off[i], (size_t)(out_off[i + 1] - out_off[i]), &st);
    }
}

}  // extern "C"
