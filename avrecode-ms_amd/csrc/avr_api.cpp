// C ABI of the MI355X arithmetic re-encode path (include/avrecode_ms_amd.h).
//
// The batch object is the build's counterpart of the reference's per-slice coder objects
// (compressor::cabac_decoder::encoder, recode.cpp:1270; decompressor::cabac_decoder::
// cabac_encoder, recode.cpp:1525): the hook adapter records bins, a batch codes them.
// There is no CPU coding path in this library: without a HIP device every run fails with
// AVR_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "avr_internal.h"
#include "avr_synth.h"
#include "avr_tables.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

int hip_fail(hipError_t e, const char *what) {
    return fail(AVR_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}

#define AVR_HIP(call)                                            \
    do {                                                         \
        hipError_t e_ = (call);                                  \
        if (e_ != hipSuccess) return hip_fail(e_, #call);        \
    } while (0)

constexpr avr::CabacTables kTables = avr::make_cabac_tables();

int select_device(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(AVR_ERR_NO_DEVICE, "no HIP device visible; this library has no CPU coding path");
    if (device < 0 || device >= n) return fail(AVR_ERR_INVALID, "device %d out of range (0..%d)", device, n - 1);
    AVR_HIP(hipSetDevice(device));
    return AVR_OK;
}

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;       // elements
    int reserve(size_t n) {
        if (n <= cap) return AVR_OK;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        size_t want = n + n / 8 + 64;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), want * sizeof(T));
        if (e != hipSuccess) return fail(AVR_ERR_NOMEM, "hipMalloc(%zu bytes): %s", want * sizeof(T), hipGetErrorString(e));
        cap = want;
        return AVR_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

template <class T>
struct PinBuf {
    T *p = nullptr;
    size_t cap = 0;
    int reserve(size_t n) {
        if (n <= cap) return AVR_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        size_t want = n + n / 8 + 64;
        hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&p), want * sizeof(T), hipHostMallocDefault);
        if (e != hipSuccess) return fail(AVR_ERR_NOMEM, "hipHostMalloc(%zu bytes): %s", want * sizeof(T), hipGetErrorString(e));
        cap = want;
        return AVR_OK;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

// Host-side plan shared by the batch API and tests: process slices longest first, 64 per tile.
void plan_tiles(const std::vector<uint32_t> &n_bins, std::vector<uint32_t> &order, std::vector<uint64_t> &tile_off) {
    const size_t n = n_bins.size();
    order.resize(n);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return n_bins[a] > n_bins[b]; });
    const size_t n_tiles = (n + 63) / 64;
    tile_off.assign(n_tiles + 1, 0);
    for (size_t t = 0; t < n_tiles; t++) {
        const uint64_t chunks = (uint64_t(n_bins[order[t * 64]]) + 7) / 8;   // sorted: first lane is the longest
        tile_off[t + 1] = tile_off[t] + chunks * 64;
    }
}

}  // namespace

struct avr_batch {
    int device = 0;
    size_t max_slices = 0, max_bins = 0, total_bins = 0;
    int kind = -1;
    bool recs8 = false;                     // the slices came as one-byte records (AVR_KIND_CABAC8); kind is AVR_KIND_CABAC
    size_t n_states = 0;
    bool ran = false;
    hipStream_t stream = nullptr;
    hipEvent_t ev[5] = {};
    float ms[4] = {0, 0, 0, 0};

    // host side
    PinBuf<uint16_t> h_recs;
    PinBuf<uint8_t> h_states, h_out, h_final;
    PinBuf<uint32_t> h_out_len;
    PinBuf<int32_t> h_status;
    std::vector<uint64_t> rec_off;          // n+1, records, multiples of 8
    std::vector<uint32_t> n_bins;
    std::vector<uint64_t> dense_off;        // n+1, bytes in h_out

    // device side
    DevBuf<uint16_t> d_recs;
    DevBuf<uint8_t> d_recs8;                // one-byte records as they came over PCIe (widened into d_recs on the device)
    DevBuf<uint4> d_tiles;
    DevBuf<uint64_t> d_rec_off, d_tile_off, d_out_off, d_dense_off;
    DevBuf<uint32_t> d_n_bins, d_order, d_out_len;
    DevBuf<int32_t> d_status;
    DevBuf<uint8_t> d_states, d_final, d_out, d_dense;

    // intra-slice parallel path (K1p): plan arrays + workspace
    DevBuf<uint64_t> d_res_off, d_dig_off;
    DevBuf<uint32_t> d_chunk_base, d_chunk_slice, d_blk_base, d_blk_slice;
    DevBuf<uint8_t> d_workspace;
    int last_path = 0;                      // 0 = one lane per slice, 1 = chunked

    // submit / wait: nothing the device reads may be pageable or local to a call
    PinBuf<uint8_t> h_plan;                 // the plan arrays of the run in flight
    size_t plan_used = 0;
    PinBuf<uint32_t> h_ndense;              // [0]: contexts the batch uses, read back behind the kernels
    std::vector<uint64_t> out_off;          // n+1, bytes in d_out
    bool in_flight = false;
    uint32_t dense_hint = 0;                // context rows the previous run of this object needed (0: no run yet)
    uint32_t hint_used = 0;                 // what the run in flight was sized by (0: it asked the device and waited)
    uint32_t info[4] = {0, 0, 0, 0};        // avr_batch_run_info
    avr_chunk_plan plan{};                  // of the chunked run in flight (device arrays of this batch)
};

// One plan array to the device through the batch's pinned arena: the copy is asynchronous for real (a copy from
// pageable memory is staged by the runtime inside the call), and the source stays put until avr_batch_wait.
template <class T>
static int stage_h2d(avr_batch *b, T *dst, const T *src, size_t n) {
    const size_t bytes = n * sizeof(T), at = (b->plan_used + 63) & ~size_t(63);
    if (at + bytes > b->h_plan.cap) return fail(AVR_ERR_NOMEM, "plan staging area too small (%zu + %zu > %zu)", at, bytes, b->h_plan.cap);
    memcpy(b->h_plan.p + at, src, bytes);
    b->plan_used = at + bytes;
    AVR_HIP(hipMemcpyAsync(dst, b->h_plan.p + at, bytes, hipMemcpyHostToDevice, b->stream));
    return AVR_OK;
}
namespace avr {
// the three environment switches of the library (avr_internal.h), read once
const Env &env() {
    static const Env e = [] {
        Env v{0, false, false};
        if (const char *p = getenv("AVR_K1_PATH")) v.k1_path = strcmp(p, "chunked") == 0 ? 2 : strcmp(p, "serial") == 0 ? 1 : 0;
        v.no_dense = getenv("AVR_NO_DENSE") != nullptr;
        v.no_hint = getenv("AVR_BATCH_NO_HINT") != nullptr;
        return v;
    }();
    return e;
}
#ifdef AVR_TEST_HOOKS
TestHooks &test_hooks() { static TestHooks h{}; return h; }
#endif
}  // namespace avr

extern "C" {

#ifdef AVR_TEST_HOOKS
// Test build only (libavrecode_hip_hooks.so): set a hook by name; returns AVR_ERR_INVALID for an unknown name.
int avr_test_hook_set(const char *name, uint32_t value) {
    avr::TestHooks &h = avr::test_hooks();
    if (!name) return fail(AVR_ERR_INVALID, "null hook name");
    if (!strcmp(name, "k1p_force_retry_every")) h.k1p_force_retry_every = value;
    else if (!strcmp(name, "census_stride")) h.census_stride = value;
    else if (!strcmp(name, "chain_lanes")) h.chain_lanes = value;
    else if (!strcmp(name, "k1_form_ref")) h.k1_form_ref = value;
    else if (!strcmp(name, "k1_emit_lds")) h.k1_emit_lds = value;
    else if (!strcmp(name, "k1_path")) h.k1_path = value;
    else if (!strcmp(name, "no_dense")) h.no_dense = value;
    else if (!strcmp(name, "no_hint")) h.no_hint = value;
    else if (!strcmp(name, "k2p_seg_len")) h.k2p_seg_len = value;
    else if (!strcmp(name, "local_waves")) h.local_waves = value;
    else if (!strcmp(name, "k2p_wave")) h.k2p_wave = value;
    else if (!strcmp(name, "k1_waves")) h.k1_waves = value;
    else if (!strcmp(name, "k1_fwd")) h.k1_fwd = value;
    else if (!strcmp(name, "k1_words8")) h.k1_words8 = value;
    else if (!strcmp(name, "chain_nsegs")) h.chain_nsegs = value;
    else if (!strcmp(name, "chain_whole")) h.chain_whole = value;
    else if (!strcmp(name, "chain_segments")) h.chain_segments = value;
    else if (!strcmp(name, "chain_force_redo")) h.chain_force_redo = value;
    else if (!strcmp(name, "reset")) h = avr::TestHooks{};
    else return fail(AVR_ERR_INVALID, "unknown test hook %s", name);
    return AVR_OK;
}
#endif

const char *avr_last_error(void) { return g_err; }
const char *avr_version(void) { return "avrecode-ms_amd 0.1 (gfx950)"; }

int avr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const uint8_t *avr_cabac_lps_range_table(void) { return kTables.lps_range; }
const uint8_t *avr_cabac_mlps_state_table(void) { return kTables.mlps_state; }

// ------------------------------------------------------------------ batch API

avr_batch *avr_batch_create(int device, size_t max_slices, size_t max_bins) {
    if (max_slices == 0 || max_slices > 0x7fffffffu) { fail(AVR_ERR_INVALID, "max_slices out of range"); return nullptr; }
    if (select_device(device) != AVR_OK) return nullptr;
    avr_batch *b = new (std::nothrow) avr_batch;
    if (!b) { fail(AVR_ERR_NOMEM, "out of host memory"); return nullptr; }
    b->device = device;
    b->max_slices = max_slices;
    b->max_bins = max_bins;
    bool ok = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; ok && i < 5; i++) ok = hipEventCreate(&b->ev[i]) == hipSuccess;
    ok = ok && b->h_recs.reserve(max_bins + 16 * max_slices) == AVR_OK;   // records: n + 7 per slice; codes (bytes): n + 31
    if (!ok) {
        if (!g_err[0]) fail(AVR_ERR_HIP, "stream/event creation failed");
        avr_batch_destroy(b);
        return nullptr;
    }
    b->rec_off.reserve(max_slices + 1);
    b->n_bins.reserve(max_slices);
    b->rec_off.push_back(0);
    return b;
}

void avr_batch_destroy(avr_batch *b) {
    if (!b) return;
    (void)hipSetDevice(b->device);
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    b->h_recs.release(); b->h_states.release(); b->h_out.release(); b->h_final.release();
    b->h_out_len.release(); b->h_status.release(); b->h_plan.release(); b->h_ndense.release();
    b->d_recs.release(); b->d_recs8.release(); b->d_tiles.release(); b->d_rec_off.release(); b->d_tile_off.release();
    b->d_out_off.release(); b->d_dense_off.release(); b->d_n_bins.release(); b->d_order.release();
    b->d_out_len.release(); b->d_status.release(); b->d_states.release(); b->d_final.release();
    b->d_out.release(); b->d_dense.release();
    b->d_res_off.release(); b->d_dig_off.release(); b->d_chunk_base.release(); b->d_chunk_slice.release();
    b->d_blk_base.release(); b->d_blk_slice.release(); b->d_workspace.release();
    for (auto &e : b->ev) if (e) (void)hipEventDestroy(e);
    if (b->stream) { avr::forget_part_streams(b->stream); avr::forget_stream(b->stream); (void)hipStreamDestroy(b->stream); }
    delete b;
}

int avr_batch_reset(avr_batch *b) {
    if (!b) return fail(AVR_ERR_INVALID, "null batch");
    if (b->in_flight) return fail(AVR_ERR_INVALID, "batch is in flight; call avr_batch_wait first");
    b->kind = -1; b->recs8 = false; b->n_states = 0; b->ran = false; b->total_bins = 0;
    b->rec_off.assign(1, 0);
    b->n_bins.clear();
    b->dense_off.clear();
    return AVR_OK;
}

// Room for one more slice in the pinned staging buffer (padding written); *where = its first element.
static int reserve_slice(avr_batch *b, int kind, size_t n, const uint8_t *init_states, size_t n_states, void **where) {
    if (!b) return fail(AVR_ERR_INVALID, "null batch");
    if (b->in_flight) return fail(AVR_ERR_INVALID, "batch is in flight; call avr_batch_wait first");
    if (b->ran) return fail(AVR_ERR_INVALID, "batch already ran; call avr_batch_reset first");
    if (kind != AVR_KIND_CABAC && kind != AVR_KIND_RANGE && kind != AVR_KIND_CABAC_CODES && kind != AVR_KIND_CABAC8)
        return fail(AVR_ERR_INVALID, "unknown kind %d", kind);
    // one-byte records are K1 records in another width: the batch is an AVR_KIND_CABAC batch whose staging buffer holds bytes
    const bool recs8 = kind == AVR_KIND_CABAC8;
    if (recs8) {
        if (n_states > AVR_MAX_STATES8) return fail(AVR_ERR_INVALID, "n_states %zu > %d: one-byte records name at most %d contexts", n_states, AVR_MAX_STATES8, AVR_MAX_STATES8);
        kind = AVR_KIND_CABAC;
    }
    if (b->kind >= 0 && (b->kind != kind || b->recs8 != recs8)) return fail(AVR_ERR_INVALID, "a batch holds slices of one kind only");
    if (n > 0xfffffff0u) return fail(AVR_ERR_INVALID, "slice too long");
    if (b->n_bins.size() >= b->max_slices) return fail(AVR_ERR_CAPACITY, "batch holds max_slices=%zu slices", b->max_slices);
    if (b->total_bins + n > b->max_bins) return fail(AVR_ERR_CAPACITY, "batch holds max_bins=%zu records", b->max_bins);
    const size_t idx = b->n_bins.size();
    if (kind == AVR_KIND_CABAC) {
        if (n_states > AVR_MAX_STATES) return fail(AVR_ERR_INVALID, "n_states %zu > %d", n_states, AVR_MAX_STATES);
        if (!init_states && n_states) return fail(AVR_ERR_INVALID, "null init_states");
        if (idx && n_states != b->n_states) return fail(AVR_ERR_INVALID, "all slices of a batch use the same n_states");
        for (size_t i = 0; i < n_states; i++)
            if (init_states[i] > 127) return fail(AVR_ERR_INVALID, "state byte %zu = %u is not 2*pStateIdx+valMPS", i, init_states[i]);
        if (int rc = b->h_states.reserve(b->max_slices * std::max<size_t>(n_states, 1))) return rc;
        b->n_states = n_states;
        if (n_states) memcpy(b->h_states.p + idx * n_states, init_states, n_states);
    }
    const uint64_t off = b->rec_off.back();
    if (kind == AVR_KIND_CABAC_CODES) {
        // Resolved codes live in the same pinned buffer as records would, as bytes: slice i at byte code_off[i] (the
        // res_off of the chunk plan: 16-byte aligned, padded with a group of its own)
        const uint64_t padded = ((uint64_t(n) + 15) & ~uint64_t(15)) + 16;
        if (off + padded > (b->max_bins + 16 * b->max_slices) * sizeof(uint16_t))
            return fail(AVR_ERR_CAPACITY, "batch code buffer full (%zu slices)", b->n_bins.size());
        uint8_t *dst = reinterpret_cast<uint8_t *>(b->h_recs.p) + off;
        memset(dst + n, AVR_CODE_BYPASS(0), padded - n);         // the padding value of a resolved stream (bypass 0, never coded)
        b->rec_off.push_back(off + padded);
        *where = dst;
    } else if (recs8) {
        // one byte per record in the same pinned buffer, slice i at BYTE rec_off[i] (its records will be at record rec_off[i] on the
        // device); what lies between a slice's last record and the next multiple of 8 is made a no-op by the widening kernel
        const uint64_t padded = (uint64_t(n) + 7) & ~uint64_t(7);
        uint8_t *dst = reinterpret_cast<uint8_t *>(b->h_recs.p) + off;
        memset(dst + n, 0, padded - n);
        b->rec_off.push_back(off + padded);
        *where = dst;
    } else {
        const uint64_t padded = (uint64_t(n) + 7) & ~uint64_t(7);
        for (uint64_t i = n; i < padded; i++) b->h_recs.p[off + i] = kind == AVR_KIND_CABAC ? AVR_NOP_CABAC : AVR_NOP_RANGE;
        b->rec_off.push_back(off + padded);
        *where = b->h_recs.p + off;
    }
    b->total_bins += n;
    b->n_bins.push_back(uint32_t(n));
    b->kind = kind;
    b->recs8 = recs8;
    return int(idx);
}

int avr_batch_reserve_slice(avr_batch *b, int kind, size_t n, const uint8_t *init_states, size_t n_states, void **buffer) {
    if (!buffer) return fail(AVR_ERR_INVALID, "null buffer pointer");
    return reserve_slice(b, kind, n, init_states, n_states, buffer);
}

int avr_batch_add_slice_cabac(avr_batch *b, const uint16_t *recs, size_t n, const uint8_t *init_states, size_t n_states) {
    if (!recs && n) return fail(AVR_ERR_INVALID, "null records");
    void *dst = nullptr;
    const int idx = reserve_slice(b, AVR_KIND_CABAC, n, init_states, n_states, &dst);
    if (idx >= 0 && n) memcpy(dst, recs, n * sizeof(uint16_t));
    return idx;
}

int avr_batch_add_slice_range(avr_batch *b, const uint16_t *recs, size_t n) {
    if (!recs && n) return fail(AVR_ERR_INVALID, "null records");
    void *dst = nullptr;
    const int idx = reserve_slice(b, AVR_KIND_RANGE, n, nullptr, 0, &dst);
    if (idx >= 0 && n) memcpy(dst, recs, n * sizeof(uint16_t));
    return idx;
}

int avr_batch_add_slice_cabac8(avr_batch *b, const uint8_t *recs8, size_t n, const uint8_t *init_states, size_t n_states) {
    if (!recs8 && n) return fail(AVR_ERR_INVALID, "null records");
    void *dst = nullptr;
    const int idx = reserve_slice(b, AVR_KIND_CABAC8, n, init_states, n_states, &dst);
    if (idx >= 0 && n) memcpy(dst, recs8, n);
    return idx;
}

int avr_batch_add_slice_codes(avr_batch *b, const uint8_t *codes, size_t n) {
    if (!codes && n) return fail(AVR_ERR_INVALID, "null codes");
    void *dst = nullptr;
    const int idx = reserve_slice(b, AVR_KIND_CABAC_CODES, n, nullptr, 0, &dst);
    if (idx >= 0 && n) memcpy(dst, codes, n);
    return idx;
}

#define AVR_STAGE(dst, src, n) do { if (int rc_ = stage_h2d(b, dst, src, n)) return rc_; } while (0)

static int enqueue_lengths(avr_batch *b, uint32_t n32);

// A batch of resolved codes: H2D of one byte per bin, K1p phases B-D (or the one-lane-per-slice coder), lengths back.
static int submit_codes(avr_batch *b, uint32_t n32) {
    const size_t n = n32;
    std::vector<uint64_t> &out_off = b->out_off;
    std::vector<uint64_t> dig_off(n + 1, 0);
    std::vector<uint32_t> chunk_base(n + 1, 0), chunk_slice;
    out_off.assign(n + 1, 0);
    for (size_t i = 0; i < n; i++) {
        const uint64_t nb = b->n_bins[i];
        out_off[i + 1] = out_off[i] + ((nb + 16 + 7) & ~uint64_t(7));
        dig_off[i + 1] = dig_off[i] + nb / 2 + 8;
        const uint32_t nc = uint32_t(std::max<uint64_t>(1, (nb + AVR_CHUNK_BINS - 1) / AVR_CHUNK_BINS));
        chunk_base[i + 1] = chunk_base[i] + nc;
        chunk_slice.insert(chunk_slice.end(), nc, uint32_t(i));
    }
    // few, long slices: the intra-slice parallel kernels; many short ones: one lane per slice (same rule as for records)
    bool chunked = n <= 32768 && b->total_bins / n >= 8192;
    if (avr::k1_path()) chunked = avr::k1_path() == 2;
    std::vector<uint32_t> order;
    if (!chunked) {
        std::vector<uint64_t> tile_off;
        plan_tiles(b->n_bins, order, tile_off);                  // longest first: the lanes of a wave finish together
    }
    const uint64_t total_codes = b->rec_off.back(), total_out = out_off.back();
    int rc;
    if ((rc = b->d_recs.reserve((total_codes + 64) / 2 + 1)) || (rc = b->d_res_off.reserve(n + 1)) || (rc = b->d_dig_off.reserve(n + 1)) ||
        (rc = b->d_chunk_base.reserve(n + 1)) || (rc = b->d_chunk_slice.reserve(chunk_slice.size())) ||
        (rc = b->d_out_off.reserve(n + 1)) || (rc = b->d_dense_off.reserve(n + 1)) || (rc = b->d_n_bins.reserve(n)) ||
        (rc = b->d_order.reserve(n)) || (rc = b->d_out_len.reserve(n)) || (rc = b->d_status.reserve(n)) || (rc = b->d_out.reserve(total_out)) ||
        (rc = b->h_out_len.reserve(n)) || (rc = b->h_status.reserve(n)) ||
        (rc = b->h_plan.reserve(64 * 8 + (n + 1) * 36 + chunk_slice.size() * 4)))
        return rc;
    avr_chunk_plan plan{b->d_res_off.p, b->d_chunk_base.p, b->d_chunk_slice.p, nullptr, nullptr, b->d_dig_off.p,
                        total_codes, dig_off.back(), chunk_base.back(), 0};
    const size_t ws = avr::k1p_code_workspace_bytes(n, &plan);
    if ((rc = b->d_workspace.reserve(ws + 256))) return rc;
    hipStream_t s = b->stream;
    b->plan_used = 0;
    b->hint_used = 0;
    AVR_HIP(hipEventRecord(b->ev[0], s));
    AVR_HIP(hipMemcpyAsync(b->d_recs.p, b->h_recs.p, total_codes, hipMemcpyHostToDevice, s));
    AVR_STAGE(b->d_res_off.p, b->rec_off.data(), n + 1);
    AVR_STAGE(b->d_dig_off.p, dig_off.data(), n + 1);
    AVR_STAGE(b->d_chunk_base.p, chunk_base.data(), n + 1);
    AVR_STAGE(b->d_chunk_slice.p, chunk_slice.data(), chunk_slice.size());
    AVR_STAGE(b->d_out_off.p, out_off.data(), n + 1);
    AVR_STAGE(b->d_n_bins.p, b->n_bins.data(), n);
    if (!chunked) AVR_STAGE(b->d_order.p, order.data(), n);
    AVR_HIP(hipMemsetAsync(b->d_status.p, 0, n * sizeof(int32_t), s));
    AVR_HIP(hipEventRecord(b->ev[1], s));
    AVR_HIP(hipEventRecord(b->ev[2], s));
    const uint8_t *d_codes = reinterpret_cast<const uint8_t *>(b->d_recs.p);
    if (chunked) {
        uint8_t *wsp = reinterpret_cast<uint8_t *>((reinterpret_cast<uintptr_t>(b->d_workspace.p) + 255) & ~uintptr_t(255));
        AVR_HIP(avr::launch_k1p_code(s, d_codes, b->d_n_bins.p, n32, &plan, wsp, b->d_out.p, b->d_out_off.p, b->d_out_len.p,
                                     b->d_status.p));
    } else {
        AVR_HIP(avr::launch_cabac_encode_codes(s, d_codes, b->d_res_off.p, b->d_n_bins.p, b->d_order.p, n32, b->d_out.p,
                                               b->d_out_off.p, b->d_out_len.p, b->d_status.p));
    }
    b->last_path = chunked;
    return enqueue_lengths(b, n32);
}

// Everything of a run up to the lengths on their way back.  `use_hint`: size the kernels by the context count of this
// object's previous run instead of asking the device and waiting (avr::DenseHint); avr_batch_wait checks the guess.
static int submit_impl(avr_batch *b, bool use_hint) {
    const size_t n = b->n_bins.size();
    const uint32_t n32 = uint32_t(n);
    if (b->kind == AVR_KIND_CABAC_CODES) return submit_codes(b, n32);
    const size_t ns = b->n_states;
    const bool cabac = b->kind == AVR_KIND_CABAC;

    std::vector<uint32_t> order;
    std::vector<uint64_t> tile_off;
    std::vector<uint64_t> &out_off = b->out_off;
    out_off.assign(n + 1, 0);
    plan_tiles(b->n_bins, order, tile_off);
    // worst case is 8 bits per bin for either coder ("output sizing") + stop bytes
    for (size_t i = 0; i < n; i++) out_off[i + 1] = out_off[i] + ((uint64_t(b->n_bins[i]) + 16 + 7) & ~uint64_t(7));
    const uint64_t total_recs = b->rec_off.back(), total_chunks = tile_off.back(), total_out = out_off.back();
    const size_t n_tiles = tile_off.size() - 1;
    // One lane per slice needs tens of thousands of slices to fill the chip; a batch of few, long
    // slices (a clip with one slice per frame) goes through the intra-slice parallel kernels.
    bool chunked = n <= 32768 && b->total_bins / n >= 8192;
    if (avr::k1_path()) chunked = avr::k1_path() == 2;
    b->last_path = chunked;

    int rc;
    if ((rc = b->d_recs.reserve(total_recs)) || (b->recs8 && (rc = b->d_recs8.reserve(total_recs + 16))) || (!chunked && (rc = b->d_tiles.reserve(total_chunks))) ||
        (rc = b->d_rec_off.reserve(n + 1)) || (rc = b->d_tile_off.reserve(n_tiles + 1)) ||
        (rc = b->d_out_off.reserve(n + 1)) || (rc = b->d_dense_off.reserve(n + 1)) ||
        (rc = b->d_n_bins.reserve(n)) || (rc = b->d_order.reserve(n)) || (rc = b->d_out_len.reserve(n)) ||
        (rc = b->d_status.reserve(n)) || (rc = b->d_out.reserve(total_out)) ||
        (rc = b->h_out_len.reserve(n)) || (rc = b->h_status.reserve(n)) || (rc = b->h_ndense.reserve(16)))
        return rc;
    if (cabac && ((rc = b->d_states.reserve(n * std::max<size_t>(ns, 1))) || (rc = b->d_final.reserve(n * std::max<size_t>(ns, 1))) ||
                  (rc = b->h_final.reserve(n * std::max<size_t>(ns, 1)))))
        return rc;
    const size_t n_chunks_max = size_t(b->total_bins / AVR_CHUNK_BINS) + n + 1, n_blks_max = size_t(b->total_bins / AVR_SORT_BLOCK_BINS) + n + 1;
    if ((rc = b->h_plan.reserve(64 * 16 + (n + 1) * 64 + (n_tiles + 1) * 8 + (chunked ? (n_chunks_max + n_blks_max) * 4 : 0)))) return rc;

    hipStream_t s = b->stream;
    b->plan_used = 0;
    const avr::DenseHint hint{use_hint ? std::min<uint32_t>(b->dense_hint, uint32_t(ns)) : 0u, b->h_ndense.p, b->h_ndense.p + 1};
    b->hint_used = hint.rows;
    b->h_ndense.p[0] = b->h_ndense.p[1] = 0;
    AVR_HIP(hipEventRecord(b->ev[0], s));
    if (b->recs8) AVR_HIP(hipMemcpyAsync(b->d_recs8.p, b->h_recs.p, total_recs, hipMemcpyHostToDevice, s));        // one byte a record
    else AVR_HIP(hipMemcpyAsync(b->d_recs.p, b->h_recs.p, total_recs * sizeof(uint16_t), hipMemcpyHostToDevice, s));
    AVR_STAGE(b->d_rec_off.p, b->rec_off.data(), n + 1);
    AVR_STAGE(b->d_out_off.p, out_off.data(), n + 1);
    AVR_STAGE(b->d_n_bins.p, b->n_bins.data(), n);
    if (!chunked) {
        AVR_STAGE(b->d_tile_off.p, tile_off.data(), n_tiles + 1);
        AVR_STAGE(b->d_order.p, order.data(), n);
    }
    if (cabac && ns) AVR_HIP(hipMemcpyAsync(b->d_states.p, b->h_states.p, n * ns, hipMemcpyHostToDevice, s));
    AVR_HIP(hipEventRecord(b->ev[1], s));
    AVR_HIP(hipMemsetAsync(b->d_status.p, 0, n * sizeof(int32_t), s));
    if (b->recs8)                                                // ... widened into the records the kernels read: from here on an AVR_KIND_CABAC batch
        AVR_HIP(avr::launch_expand_records8(s, b->d_recs8.p, b->d_rec_off.p, b->d_n_bins.p, n32, uint32_t(ns), total_recs, b->d_recs.p));
    // Both K1 paths renumber the batch onto the contexts it uses themselves (the intra-slice parallel kernels inside
    // their census pass, the one-lane-per-slice kernel through launch_cabac_encode): records and states go in as they are.
    if (chunked && !cabac) {
        // K2 for few, long slices: the range recurrence per slice, everything else per chunk (avr_k2p.hip)
        std::vector<uint32_t> chunk_base(n + 1, 0), chunk_slice;
        for (size_t i = 0; i < n; i++) {
            const uint32_t nc = uint32_t(std::max<uint64_t>(1, (uint64_t(b->n_bins[i]) + AVR_CHUNK_BINS - 1) / AVR_CHUNK_BINS));
            chunk_base[i + 1] = chunk_base[i] + nc;
            chunk_slice.insert(chunk_slice.end(), nc, uint32_t(i));
        }
        const size_t ws = avr::k2p_workspace_bytes(n, chunk_base.back(), total_out);
        if ((rc = b->d_chunk_base.reserve(n + 1)) || (rc = b->d_chunk_slice.reserve(chunk_slice.size())) || (rc = b->d_workspace.reserve(ws + 256)))
            return rc;
        AVR_STAGE(b->d_chunk_base.p, chunk_base.data(), n + 1);
        AVR_STAGE(b->d_chunk_slice.p, chunk_slice.data(), chunk_slice.size());
        AVR_HIP(hipEventRecord(b->ev[2], s));
        uint8_t *wsp = reinterpret_cast<uint8_t *>((reinterpret_cast<uintptr_t>(b->d_workspace.p) + 255) & ~uintptr_t(255));
        AVR_HIP(avr::launch_k2p(s, b->d_recs.p, b->d_rec_off.p, b->d_n_bins.p, n32, b->d_chunk_base.p, b->d_chunk_slice.p,
                                chunk_base.back(), total_out, wsp, b->d_out.p, b->d_out_off.p, b->d_out_len.p, b->d_status.p));
    } else if (chunked) {
        std::vector<uint64_t> res_off(n + 1, 0), dig_off(n + 1, 0);
        std::vector<uint32_t> chunk_base(n + 1, 0), blk_base(n + 1, 0), chunk_slice, blk_slice;
        for (size_t i = 0; i < n; i++) {
            const uint64_t nb = b->n_bins[i];
            res_off[i + 1] = res_off[i] + ((nb + 15) & ~uint64_t(15)) + 16;
            dig_off[i + 1] = dig_off[i] + nb / 2 + 8;
            const uint32_t nc = uint32_t(std::max<uint64_t>(1, (nb + AVR_CHUNK_BINS - 1) / AVR_CHUNK_BINS));
            const uint32_t nk = uint32_t(std::max<uint64_t>(1, (nb + AVR_SORT_BLOCK_BINS - 1) / AVR_SORT_BLOCK_BINS));
            chunk_base[i + 1] = chunk_base[i] + nc;
            blk_base[i + 1] = blk_base[i] + nk;
            chunk_slice.insert(chunk_slice.end(), nc, uint32_t(i));
            blk_slice.insert(blk_slice.end(), nk, uint32_t(i));
        }
        if ((rc = b->d_res_off.reserve(n + 1)) || (rc = b->d_dig_off.reserve(n + 1)) || (rc = b->d_chunk_base.reserve(n + 1)) ||
            (rc = b->d_blk_base.reserve(n + 1)) || (rc = b->d_chunk_slice.reserve(chunk_slice.size())) ||
            (rc = b->d_blk_slice.reserve(blk_slice.size())))
            return rc;
        avr_chunk_plan plan{b->d_res_off.p, b->d_chunk_base.p, b->d_chunk_slice.p, b->d_blk_base.p, b->d_blk_slice.p, b->d_dig_off.p,
                            res_off.back(), dig_off.back(), chunk_base.back(), blk_base.back()};
        const size_t ws = avr::k1p_workspace_bytes(n, uint32_t(ns), &plan);
        if ((rc = b->d_workspace.reserve(ws + 256))) return rc;
        b->plan = plan;
        AVR_STAGE(b->d_res_off.p, res_off.data(), n + 1);
        AVR_STAGE(b->d_dig_off.p, dig_off.data(), n + 1);
        AVR_STAGE(b->d_chunk_base.p, chunk_base.data(), n + 1);
        AVR_STAGE(b->d_blk_base.p, blk_base.data(), n + 1);
        AVR_STAGE(b->d_chunk_slice.p, chunk_slice.data(), chunk_slice.size());
        AVR_STAGE(b->d_blk_slice.p, blk_slice.data(), blk_slice.size());
        AVR_HIP(hipEventRecord(b->ev[2], s));
        uint8_t *wsp = reinterpret_cast<uint8_t *>((reinterpret_cast<uintptr_t>(b->d_workspace.p) + 255) & ~uintptr_t(255));
        AVR_HIP(avr::launch_k1p(s, b->d_recs.p, b->d_rec_off.p, b->d_n_bins.p, n32, b->d_states.p, uint32_t(ns), &plan, wsp,
                                b->d_out.p, b->d_out_off.p, b->d_out_len.p, b->d_status.p, b->d_final.p, &hint));
    } else {
        AVR_HIP(avr::launch_pack_tiles(s, b->kind, uint32_t(ns), b->d_recs.p, b->d_rec_off.p, b->d_n_bins.p, b->d_order.p, n32,
                                       b->d_tile_off.p, b->d_tiles.p, b->d_status.p));
        AVR_HIP(hipEventRecord(b->ev[2], s));
        if (cabac)
            AVR_HIP(avr::launch_cabac_encode(true, s, b->d_tiles.p, b->d_tile_off.p, b->d_n_bins.p, b->d_order.p, n32, b->d_states.p,
                                             uint32_t(ns), b->d_out.p, b->d_out_off.p, b->d_out_len.p, b->d_status.p, b->d_final.p,
                                             AVR_SLICE_OK, true, &hint));
        else
            AVR_HIP(avr::launch_range_encode(true, s, b->d_tiles.p, b->d_tile_off.p, b->d_n_bins.p, b->d_order.p, n32, b->d_out.p,
                                             b->d_out_off.p, b->d_out_len.p, b->d_status.p));
    }
    return enqueue_lengths(b, n32);
}

// lengths, statuses and final states on their way to the host behind the kernels
static int enqueue_lengths(avr_batch *b, uint32_t n32) {
    const size_t n = n32, ns = b->n_states;
    const bool with_states = b->kind == AVR_KIND_CABAC && ns;
    hipStream_t s = b->stream;
    AVR_HIP(hipEventRecord(b->ev[3], s));
    AVR_HIP(hipMemcpyAsync(b->h_out_len.p, b->d_out_len.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    AVR_HIP(hipMemcpyAsync(b->h_status.p, b->d_status.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    if (with_states) AVR_HIP(hipMemcpyAsync(b->h_final.p, b->d_final.p, n * ns, hipMemcpyDeviceToHost, s));
    return AVR_OK;
}

int avr_batch_submit(avr_batch *b) {
    if (!b) return fail(AVR_ERR_INVALID, "null batch");
    if (b->in_flight) return fail(AVR_ERR_INVALID, "batch is in flight; call avr_batch_wait first");
    if (int rc = select_device(b->device)) return rc;
    b->ran = false;                                              // a batch may be submitted again after avr_batch_wait: same slices, coded anew
    const size_t n = b->n_bins.size();
    b->dense_off.assign(n + 1, 0);
    if (n == 0) { b->in_flight = true; return AVR_OK; }
    const bool use_hint = b->dense_hint > 0 && !avr::no_hint();
    if (int rc = submit_impl(b, use_hint)) { (void)hipStreamSynchronize(b->stream); return rc; }
    b->in_flight = true;
    return AVR_OK;
}

// Wait for the kernels, then: the coded bytes gathered densely on the device, one D2H copy.
int avr_batch_wait(avr_batch *b) {
    if (!b) return fail(AVR_ERR_INVALID, "null batch");
    if (!b->in_flight) return b->ran ? AVR_OK : fail(AVR_ERR_INVALID, "batch has not been submitted");
    if (int rc = select_device(b->device)) return rc;
    b->in_flight = false;
    const size_t n = b->n_bins.size();
    if (n == 0) { b->ran = true; return AVR_OK; }
    const uint32_t n32 = uint32_t(n);
    hipStream_t s = b->stream;
    int rc;
    AVR_HIP(hipStreamSynchronize(s));
    b->info[0] = uint32_t(b->last_path); b->info[1] = b->info[2] = b->info[3] = 0;
    if (b->kind == AVR_KIND_CABAC) {
        // The run was sized by a guess of the context count.  The one-lane-per-slice kernel is exact whatever the
        // guess; the intra-slice parallel kernels only if the batch needs no more rows than guessed: else once more,
        // asking the device this time.
        b->info[1] = b->hint_used; b->info[3] = 0;
        if (b->hint_used && b->last_path == 1 && b->h_ndense.p[0] > b->hint_used) {
            if ((rc = submit_impl(b, false))) { (void)hipStreamSynchronize(s); return rc; }
            AVR_HIP(hipStreamSynchronize(s));
            b->info[3] = 1;
        }
        if (b->last_path == 1 && b->h_ndense.p[1]) {             // slices with a context the sampled census missed: their second pass
            uint8_t *wsp = reinterpret_cast<uint8_t *>((reinterpret_cast<uintptr_t>(b->d_workspace.p) + 255) & ~uintptr_t(255));
            AVR_HIP(avr::launch_k1p_retry(s, b->d_recs.p, b->d_rec_off.p, b->d_n_bins.p, n32, b->d_states.p, uint32_t(b->n_states), &b->plan,
                                          wsp, b->d_out.p, b->d_out_off.p, b->d_out_len.p, b->d_status.p, b->d_final.p));
            if ((rc = enqueue_lengths(b, n32))) return rc;
            AVR_HIP(hipStreamSynchronize(s));
            b->info[3] |= 2;
        }
        b->info[2] = b->h_ndense.p[0];
        const uint32_t seen = b->h_ndense.p[0];
        if (seen) b->dense_hint = std::min<uint32_t>(uint32_t(b->n_states), seen + 8);   // a little room: the next batch of a stream rarely needs more
    }
    for (size_t i = 0; i < n; i++) {
        const uint64_t cap = b->out_off[i + 1] - b->out_off[i];
        b->dense_off[i + 1] = b->dense_off[i] + std::min<uint64_t>(b->h_out_len.p[i], cap);
    }
    const uint64_t dense_total = b->dense_off.back();
    if ((rc = b->d_dense.reserve(dense_total + 1)) || (rc = b->h_out.reserve(dense_total + 1))) return rc;
    b->plan_used = 0;                                            // the run's plan arrays are consumed: the arena is free again
    AVR_STAGE(b->d_dense_off.p, b->dense_off.data(), n + 1);
    AVR_HIP(avr::launch_compact(s, b->d_out.p, b->d_out_off.p, b->d_out_len.p, b->d_dense_off.p, n32, b->d_dense.p));
    if (dense_total) AVR_HIP(hipMemcpyAsync(b->h_out.p, b->d_dense.p, dense_total, hipMemcpyDeviceToHost, s));
    AVR_HIP(hipEventRecord(b->ev[4], s));
    AVR_HIP(hipStreamSynchronize(s));
    for (int i = 0; i < 4; i++) (void)hipEventElapsedTime(&b->ms[i], b->ev[i], b->ev[i + 1]);
    b->ran = true;                                               // only now: the getters hand out h_out / h_status
    return AVR_OK;
}

int avr_batch_run(avr_batch *b) {
    if (!b) return fail(AVR_ERR_INVALID, "null batch");
    if (b->ran) return fail(AVR_ERR_INVALID, "batch already ran");
    if (int rc = avr_batch_submit(b)) return rc;
    return avr_batch_wait(b);
}

int avr_batch_get(avr_batch *b, size_t slice, const uint8_t **bytes, size_t *len, int *status) {
    if (!b || !b->ran) return fail(AVR_ERR_INVALID, "batch has not run");
    if (slice >= b->n_bins.size()) return fail(AVR_ERR_INVALID, "slice %zu out of range", slice);
    if (bytes) *bytes = b->h_out.p + b->dense_off[slice];
    if (len) *len = size_t(b->dense_off[slice + 1] - b->dense_off[slice]);
    if (status) *status = b->h_status.p[slice];
    return AVR_OK;
}

int avr_batch_get_states(avr_batch *b, size_t slice, const uint8_t **states, size_t *n_states) {
    if (!b || !b->ran) return fail(AVR_ERR_INVALID, "batch has not run");
    if (b->kind != AVR_KIND_CABAC) return fail(AVR_ERR_INVALID, "not a CABAC batch");
    if (slice >= b->n_bins.size()) return fail(AVR_ERR_INVALID, "slice %zu out of range", slice);
    if (states) *states = b->h_final.p + slice * b->n_states;
    if (n_states) *n_states = b->n_states;
    return AVR_OK;
}

int avr_batch_run_info(avr_batch *b, uint32_t info[4]) {
    if (!b || !b->ran || !info) return fail(AVR_ERR_INVALID, "batch has not run");
    memcpy(info, b->info, sizeof b->info);
    return AVR_OK;
}

int avr_batch_timings(avr_batch *b, float ms[4]) {
    if (!b || !b->ran || !ms) return fail(AVR_ERR_INVALID, "batch has not run");
    memcpy(ms, b->ms, sizeof b->ms);
    return AVR_OK;
}

// ------------------------------------------------------------------ one batch over several GPUs

struct avr_multi {
    std::vector<int> devices;
    size_t max_slices = 0, max_bins = 0, total_bins = 0;
    int kind = -1;
    bool recs8 = false;                     // the slices came as one-byte records (AVR_KIND_CABAC8); kind is AVR_KIND_CABAC
    size_t n_states = 0;
    bool ran = false;
    std::vector<uint8_t> store;              // the slices' records / codes, back to back (bytes)
    std::vector<uint64_t> off;               // n+1 byte offsets into store
    std::vector<uint32_t> n_bins;
    std::vector<uint8_t> states;             // n x n_states
    std::vector<avr_batch *> sub;            // one per entry of devices
    std::vector<int> place, local;           // slice -> entry of devices, index inside that sub-batch
    std::vector<uint64_t> load;
};

avr_multi *avr_multi_create(const int *devices, size_t n_devices, size_t max_slices, size_t max_bins) {
    if (!devices || n_devices == 0 || n_devices > 64) { fail(AVR_ERR_INVALID, "need 1..64 devices"); return nullptr; }
    if (max_slices == 0 || max_slices > 0x7fffffffu) { fail(AVR_ERR_INVALID, "max_slices out of range"); return nullptr; }
    for (size_t i = 0; i < n_devices; i++)
        if (select_device(devices[i]) != AVR_OK) return nullptr;
    avr_multi *m = new (std::nothrow) avr_multi;
    if (!m) { fail(AVR_ERR_NOMEM, "out of host memory"); return nullptr; }
    m->devices.assign(devices, devices + n_devices);
    m->max_slices = max_slices;
    m->max_bins = max_bins;
    m->off.push_back(0);
    m->sub.assign(n_devices, nullptr);
    return m;
}

void avr_multi_destroy(avr_multi *m) {
    if (!m) return;
    for (avr_batch *b : m->sub) avr_batch_destroy(b);
    delete m;
}

static int multi_add(avr_multi *m, int kind, const void *data, size_t n, size_t elem, const uint8_t *init_states, size_t n_states) {
    if (!m) return fail(AVR_ERR_INVALID, "null batch");
    if (!data && n) return fail(AVR_ERR_INVALID, "null records");
    if (m->ran) return fail(AVR_ERR_INVALID, "batch already ran");
    if (m->kind >= 0 && m->kind != kind) return fail(AVR_ERR_INVALID, "a batch holds slices of one kind only");
    if (n > 0xfffffff0u) return fail(AVR_ERR_INVALID, "slice too long");
    if (m->n_bins.size() >= m->max_slices) return fail(AVR_ERR_CAPACITY, "batch holds max_slices=%zu slices", m->max_slices);
    if (m->total_bins + n > m->max_bins) return fail(AVR_ERR_CAPACITY, "batch holds max_bins=%zu records", m->max_bins);
    if (kind == AVR_KIND_CABAC) {
        if (n_states > AVR_MAX_STATES) return fail(AVR_ERR_INVALID, "n_states %zu > %d", n_states, AVR_MAX_STATES);
        if (!init_states && n_states) return fail(AVR_ERR_INVALID, "null init_states");
        if (!m->n_bins.empty() && n_states != m->n_states) return fail(AVR_ERR_INVALID, "all slices of a batch use the same n_states");
        m->n_states = n_states;
        m->states.insert(m->states.end(), init_states, init_states + n_states);
    }
    const uint8_t *p = static_cast<const uint8_t *>(data);
    m->store.insert(m->store.end(), p, p + n * elem);
    m->off.push_back(m->store.size());
    m->n_bins.push_back(uint32_t(n));
    m->total_bins += n;
    m->kind = kind;
    return int(m->n_bins.size()) - 1;
}

int avr_multi_add_slice_cabac(avr_multi *m, const uint16_t *recs, size_t n, const uint8_t *init_states, size_t n_states) {
    return multi_add(m, AVR_KIND_CABAC, recs, n, 2, init_states, n_states);
}
int avr_multi_add_slice_range(avr_multi *m, const uint16_t *recs, size_t n) { return multi_add(m, AVR_KIND_RANGE, recs, n, 2, nullptr, 0); }
int avr_multi_add_slice_codes(avr_multi *m, const uint8_t *codes, size_t n) { return multi_add(m, AVR_KIND_CABAC_CODES, codes, n, 1, nullptr, 0); }

int avr_multi_run(avr_multi *m) {
    if (!m) return fail(AVR_ERR_INVALID, "null batch");
    if (m->ran) return fail(AVR_ERR_INVALID, "batch already ran");
    const size_t n = m->n_bins.size(), nd = m->devices.size();
    // greedy LPT: longest slice first, to the device with the least bins so far (ties: the lower entry)
    std::vector<uint32_t> by_len(n);
    std::iota(by_len.begin(), by_len.end(), 0u);
    std::stable_sort(by_len.begin(), by_len.end(), [&](uint32_t a, uint32_t b) { return m->n_bins[a] > m->n_bins[b]; });
    m->place.assign(n, 0);
    m->local.assign(n, 0);
    m->load.assign(nd, 0);
    std::vector<std::vector<uint32_t>> mine(nd);
    for (uint32_t i : by_len) {
        const size_t d = size_t(std::min_element(m->load.begin(), m->load.end()) - m->load.begin());
        m->place[i] = int(d);
        m->load[d] += m->n_bins[i];
        mine[d].push_back(i);
    }
    for (size_t d = 0; d < nd; d++) std::sort(mine[d].begin(), mine[d].end());       // each device keeps the caller's order
    std::vector<int> rc(nd, AVR_OK);
    std::vector<std::string> err(nd);
    auto work = [&](size_t d) {
        if (m->sub[d]) { avr_batch_destroy(m->sub[d]); m->sub[d] = nullptr; }    // a run that failed on another device left this one behind
        if (mine[d].empty()) return;
        avr_batch *b = avr_batch_create(m->devices[d], mine[d].size(), size_t(m->load[d]) + 8);
        if (!b) { rc[d] = AVR_ERR_HIP; err[d] = avr_last_error(); return; }
        m->sub[d] = b;
        for (size_t j = 0; j < mine[d].size() && rc[d] >= 0; j++) {
            const uint32_t i = mine[d][j];
            const uint8_t *p = m->store.data() + m->off[i];
            int r;
            if (m->kind == AVR_KIND_CABAC)
                r = avr_batch_add_slice_cabac(b, reinterpret_cast<const uint16_t *>(p), m->n_bins[i], m->states.data() + size_t(i) * m->n_states, m->n_states);
            else if (m->kind == AVR_KIND_RANGE) r = avr_batch_add_slice_range(b, reinterpret_cast<const uint16_t *>(p), m->n_bins[i]);
            else r = avr_batch_add_slice_codes(b, p, m->n_bins[i]);
            if (r < 0) { rc[d] = r; err[d] = avr_last_error(); }
            else m->local[i] = r;
        }
        if (rc[d] >= 0 && (rc[d] = avr_batch_run(b)) < 0) err[d] = avr_last_error();      // the error text is per thread: carry it over
    };
    std::vector<std::thread> threads;
    for (size_t d = 1; d < nd; d++) threads.emplace_back(work, d);
    work(0);
    for (std::thread &t : threads) t.join();
    for (size_t d = 0; d < nd; d++)
        if (rc[d] < 0) return fail(rc[d], "device entry %zu (device %d): %s", d, m->devices[d], err[d].c_str());
    m->ran = true;
    return AVR_OK;
}

int avr_multi_get(avr_multi *m, size_t slice, const uint8_t **bytes, size_t *len, int *status) {
    if (!m || !m->ran) return fail(AVR_ERR_INVALID, "batch has not run");
    if (slice >= m->n_bins.size()) return fail(AVR_ERR_INVALID, "slice %zu out of range", slice);
    return avr_batch_get(m->sub[size_t(m->place[slice])], size_t(m->local[slice]), bytes, len, status);
}

int avr_multi_placement(avr_multi *m, size_t slice) {
    if (!m || !m->ran) return fail(AVR_ERR_INVALID, "batch has not run");
    if (slice >= m->n_bins.size()) return fail(AVR_ERR_INVALID, "slice %zu out of range", slice);
    return m->place[slice];
}

int avr_multi_load(avr_multi *m, uint64_t *bins_per_device) {
    if (!m || !m->ran || !bins_per_device) return fail(AVR_ERR_INVALID, "batch has not run");
    for (size_t d = 0; d < m->load.size(); d++) bins_per_device[d] = m->load[d];
    return AVR_OK;
}

// ------------------------------------------------------------------ device-resident API

static int check_common(const void *a, const void *b, const void *c, size_t n_slices) {
    if (n_slices > 0x7fffffffu) return fail(AVR_ERR_INVALID, "n_slices out of range");
    if (n_slices && (!a || !b || !c)) return fail(AVR_ERR_INVALID, "null device pointer");
    return AVR_OK;
}

int avr_pack_tiles_device(int device, void *stream, int kind, size_t n_states, const uint16_t *recs, const uint64_t *rec_off,
                          const uint32_t *n_bins, const uint32_t *order, size_t n_slices, const uint64_t *tile_off, void *tiles,
                          int32_t *status) {
    if (int rc = check_common(rec_off, n_bins, tile_off, n_slices)) return rc;
    if (kind != AVR_KIND_CABAC && kind != AVR_KIND_RANGE) return fail(AVR_ERR_INVALID, "kind %d", kind);
    if (n_states > AVR_MAX_STATES) return fail(AVR_ERR_INVALID, "n_states %zu > %d", n_states, AVR_MAX_STATES);
    if (n_slices && !status) return fail(AVR_ERR_INVALID, "null status");
    if (int rc = select_device(device)) return rc;
    AVR_HIP(avr::launch_pack_tiles(static_cast<hipStream_t>(stream), kind, uint32_t(n_states), recs, rec_off, n_bins, order,
                                   uint32_t(n_slices), tile_off, tiles, status));
    return AVR_OK;
}

int avr_cabac_encode_tiles_device(int device, void *stream, const void *tiles, const uint64_t *tile_off, const uint32_t *n_bins,
                                  const uint32_t *order, size_t n_slices, const uint8_t *init_states, size_t n_states,
                                  uint8_t *out, const uint64_t *out_off, uint32_t *out_len, int32_t *status, uint8_t *final_states) {
    if (int rc = check_common(tile_off, n_bins, out_off, n_slices)) return rc;
    if (n_states > AVR_MAX_STATES) return fail(AVR_ERR_INVALID, "n_states %zu > %d", n_states, AVR_MAX_STATES);
    if (int rc = select_device(device)) return rc;
    AVR_HIP(avr::launch_cabac_encode(true, static_cast<hipStream_t>(stream), tiles, tile_off, n_bins, order, uint32_t(n_slices),
                                     init_states, uint32_t(n_states), out, out_off, out_len, status, final_states));
    return AVR_OK;
}

int avr_range_encode_tiles_device(int device, void *stream, const void *tiles, const uint64_t *tile_off, const uint32_t *n_bins,
                                  const uint32_t *order, size_t n_slices, uint8_t *out, const uint64_t *out_off,
                                  uint32_t *out_len, int32_t *status) {
    if (int rc = check_common(tile_off, n_bins, out_off, n_slices)) return rc;
    if (int rc = select_device(device)) return rc;
    AVR_HIP(avr::launch_range_encode(true, static_cast<hipStream_t>(stream), tiles, tile_off, n_bins, order, uint32_t(n_slices),
                                     out, out_off, out_len, status));
    return AVR_OK;
}

size_t avr_cabac_chunked_workspace_bytes(size_t n_slices, size_t n_states, const avr_chunk_plan *plan) {
    if (!plan || n_states > AVR_MAX_STATES) return 0;
    return avr::k1p_workspace_bytes(n_slices, uint32_t(n_states), plan);
}

int avr_cabac_encode_chunked_device(int device, void *stream, const uint16_t *recs, const uint64_t *rec_off, const uint32_t *n_bins,
                                    size_t n_slices, const uint8_t *init_states, size_t n_states, const avr_chunk_plan *plan,
                                    void *workspace, size_t workspace_bytes, uint8_t *out, const uint64_t *out_off,
                                    uint32_t *out_len, int32_t *status, uint8_t *final_states) {
    if (int rc = check_common(rec_off, n_bins, out_off, n_slices)) return rc;
    if (n_states > AVR_MAX_STATES) return fail(AVR_ERR_INVALID, "n_states %zu > %d", n_states, AVR_MAX_STATES);
    if (!plan || (n_slices && (!plan->res_off || !plan->chunk_base || !plan->chunk_slice || !plan->blk_base || !plan->blk_slice ||
                               !plan->dig_off || !workspace || !status)))
        return fail(AVR_ERR_INVALID, "null plan pointer");
    if (workspace_bytes < avr::k1p_workspace_bytes(n_slices, uint32_t(n_states), plan))
        return fail(AVR_ERR_CAPACITY, "workspace of %zu bytes is smaller than avr_cabac_chunked_workspace_bytes()", workspace_bytes);
    if (int rc = select_device(device)) return rc;
    AVR_HIP(avr::launch_k1p(static_cast<hipStream_t>(stream), recs, rec_off, n_bins, uint32_t(n_slices), init_states,
                            uint32_t(n_states), plan, workspace, out, out_off, out_len, status, final_states));
    return AVR_OK;
}

static int check_chunked(const uint64_t *rec_off, const uint32_t *n_bins, const uint64_t *out_off, size_t n_slices, size_t n_states,
                         const avr_chunk_plan *plan, void *workspace, size_t workspace_bytes, const int32_t *status) {
    if (int rc = check_common(rec_off, n_bins, out_off, n_slices)) return rc;
    if (n_states > AVR_MAX_STATES) return fail(AVR_ERR_INVALID, "n_states %zu > %d", n_states, AVR_MAX_STATES);
    if (!plan || (n_slices && (!plan->res_off || !plan->chunk_base || !plan->chunk_slice || !plan->blk_base || !plan->blk_slice ||
                               !plan->dig_off || !workspace || !status)))
        return fail(AVR_ERR_INVALID, "null plan pointer");
    if (workspace_bytes < avr::k1p_workspace_bytes(n_slices, uint32_t(n_states), plan))
        return fail(AVR_ERR_CAPACITY, "workspace of %zu bytes is smaller than avr_cabac_chunked_workspace_bytes()", workspace_bytes);
    return AVR_OK;
}

int avr_cabac_encode_chunked_device_hinted(int device, void *stream, const uint16_t *recs, const uint64_t *rec_off, const uint32_t *n_bins,
                                           size_t n_slices, const uint8_t *init_states, size_t n_states, const avr_chunk_plan *plan,
                                           void *workspace, size_t workspace_bytes, uint8_t *out, const uint64_t *out_off,
                                           uint32_t *out_len, int32_t *status, uint8_t *final_states, uint32_t rows_hint, uint32_t *counts) {
    if (!counts) return fail(AVR_ERR_INVALID, "null counts");
    if (int rc = check_chunked(rec_off, n_bins, out_off, n_slices, n_states, plan, workspace, workspace_bytes, status)) return rc;
    if (int rc = select_device(device)) return rc;
    counts[0] = counts[1] = 0;
    if (n_slices == 0) return AVR_OK;
    const avr::DenseHint hint{std::min<uint32_t>(rows_hint, uint32_t(n_states)), counts, counts + 1};
    AVR_HIP(avr::launch_k1p(static_cast<hipStream_t>(stream), recs, rec_off, n_bins, uint32_t(n_slices), init_states,
                            uint32_t(n_states), plan, workspace, out, out_off, out_len, status, final_states, &hint));
    return AVR_OK;
}

int avr_cabac_encode_chunked_second_pass_device(int device, void *stream, const uint16_t *recs, const uint64_t *rec_off, const uint32_t *n_bins,
                                                size_t n_slices, const uint8_t *init_states, size_t n_states, const avr_chunk_plan *plan,
                                                void *workspace, size_t workspace_bytes, uint8_t *out, const uint64_t *out_off,
                                                uint32_t *out_len, int32_t *status, uint8_t *final_states) {
    if (int rc = check_chunked(rec_off, n_bins, out_off, n_slices, n_states, plan, workspace, workspace_bytes, status)) return rc;
    if (int rc = select_device(device)) return rc;
    if (n_slices == 0) return AVR_OK;
    AVR_HIP(avr::launch_k1p_retry(static_cast<hipStream_t>(stream), recs, rec_off, n_bins, uint32_t(n_slices), init_states,
                                  uint32_t(n_states), plan, workspace, out, out_off, out_len, status, final_states));
    return AVR_OK;
}

int avr_cabac_encode_tiles_device_hinted(int device, void *stream, const void *tiles, const uint64_t *tile_off, const uint32_t *n_bins,
                                         const uint32_t *order, size_t n_slices, const uint8_t *init_states, size_t n_states,
                                         uint8_t *out, const uint64_t *out_off, uint32_t *out_len, int32_t *status, uint8_t *final_states,
                                         uint32_t rows_hint, uint32_t *counts) {
    if (!counts) return fail(AVR_ERR_INVALID, "null counts");
    if (int rc = check_common(tile_off, n_bins, out_off, n_slices)) return rc;
    if (n_states > AVR_MAX_STATES) return fail(AVR_ERR_INVALID, "n_states %zu > %d", n_states, AVR_MAX_STATES);
    if (int rc = select_device(device)) return rc;
    counts[0] = counts[1] = 0;
    const avr::DenseHint hint{std::min<uint32_t>(rows_hint, uint32_t(n_states)), counts, nullptr};
    AVR_HIP(avr::launch_cabac_encode(true, static_cast<hipStream_t>(stream), tiles, tile_off, n_bins, order, uint32_t(n_slices),
                                     init_states, uint32_t(n_states), out, out_off, out_len, status, final_states, AVR_SLICE_OK, true, &hint));
    return AVR_OK;
}

extern "C++" {
// The streams the parts of avr_cabac_encode_chunked_device_parts run on, with their events: one set per (device, caller's stream), made on
// first use and kept (work on the caller's stream is ordered, so consecutive calls share it); released by avr::forget_part_streams.
namespace {
struct PartStreams { int dev; hipStream_t main; hipStream_t side[AVR_MAX_PARTS - 1]; hipEvent_t fork, done[AVR_MAX_PARTS - 1]; };
std::vector<PartStreams *> g_parts;
std::mutex g_parts_mu;
hipError_t part_streams(hipStream_t s, PartStreams **out) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(g_parts_mu);
    for (PartStreams *x : g_parts)
        if (x->dev == dev && x->main == s) { *out = x; return hipSuccess; }
    PartStreams *x = new PartStreams{};
    x->dev = dev; x->main = s;
    e = hipEventCreateWithFlags(&x->fork, hipEventDisableTiming);
    for (int i = 0; i < AVR_MAX_PARTS - 1 && e == hipSuccess; i++) {
        e = hipStreamCreateWithFlags(&x->side[i], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&x->done[i], hipEventDisableTiming);
    }
    if (e != hipSuccess) { delete x; return e; }                 // (what was made of it is left to the process's end: an out-of-resources path)
    g_parts.push_back(x);
    *out = x;
    return hipSuccess;
}
}  // namespace
namespace avr {
void forget_part_streams(hipStream_t s) {
    std::lock_guard<std::mutex> lock(g_parts_mu);
    for (size_t i = 0; i < g_parts.size();) {
        PartStreams *x = g_parts[i];
        if (x->main != s) { i++; continue; }
        for (int k = 0; k < AVR_MAX_PARTS - 1; k++) {
            (void)hipStreamSynchronize(x->side[k]);
            avr::forget_stream(x->side[k]);                      // the library's own per-stream scratch of that stream
            (void)hipEventDestroy(x->done[k]);
            (void)hipStreamDestroy(x->side[k]);
        }
        (void)hipEventDestroy(x->fork);
        delete x;
        g_parts.erase(g_parts.begin() + long(i));
    }
}
}  // namespace avr
}  // extern "C++"

int avr_cabac_encode_chunked_device_parts(int device, void *stream, const uint16_t *recs, size_t n_states, uint8_t *out,
                                          const avr_chunked_part *parts, size_t n_parts) {
    if (!parts || n_parts == 0 || n_parts > AVR_MAX_PARTS) return fail(AVR_ERR_INVALID, "1 .. %d parts", AVR_MAX_PARTS);
    for (size_t i = 0; i < n_parts; i++) {
        const avr_chunked_part &q = parts[i];
        if (!q.counts) return fail(AVR_ERR_INVALID, "part %zu: null counts", i);
        if (int rc = check_chunked(q.rec_off, q.n_bins, q.out_off, q.n_slices, n_states, q.plan, q.workspace, q.workspace_bytes, q.status)) return rc;
    }
    if (int rc = select_device(device)) return rc;
    hipStream_t main = static_cast<hipStream_t>(stream);
    PartStreams *ps = nullptr;
    if (n_parts > 1) {
        AVR_HIP(part_streams(main, &ps));
        AVR_HIP(hipEventRecord(ps->fork, main));
    }
    for (size_t i = 0; i < n_parts; i++) {                       // part 0 on the caller's stream, the others beside it
        const avr_chunked_part &q = parts[i];
        hipStream_t s = i == 0 ? main : ps->side[i - 1];
        if (i) AVR_HIP(hipStreamWaitEvent(s, ps->fork, 0));
        q.counts[0] = q.counts[1] = 0;
        if (q.n_slices) {
            const avr::DenseHint hint{std::min<uint32_t>(q.rows_hint, uint32_t(n_states)), q.counts, q.counts + 1, uint32_t(n_parts)};
            AVR_HIP(avr::launch_k1p(s, recs, q.rec_off, q.n_bins, uint32_t(q.n_slices), q.init_states, uint32_t(n_states), q.plan,
                                    q.workspace, out, q.out_off, q.out_len, q.status, q.final_states, &hint));
        }
        if (i) AVR_HIP(hipEventRecord(ps->done[i - 1], s));
    }
    for (size_t i = 1; i < n_parts; i++) AVR_HIP(hipStreamWaitEvent(main, ps->done[i - 1], 0));
    return AVR_OK;
}

size_t avr_range_chunked_workspace_bytes(size_t n_slices, const avr_chunk_plan *plan, uint64_t out_total) {
    if (!plan) return 0;
    return avr::k2p_workspace_bytes(n_slices, plan->total_chunks, out_total);
}

int avr_range_encode_chunked_device(int device, void *stream, const uint16_t *recs, const uint64_t *rec_off, const uint32_t *n_bins,
                                    size_t n_slices, const avr_chunk_plan *plan, void *workspace, size_t workspace_bytes,
                                    uint8_t *out, const uint64_t *out_off, uint64_t out_total, uint32_t *out_len, int32_t *status) {
    if (int rc = check_common(rec_off, n_bins, out_off, n_slices)) return rc;
    if (!plan || (n_slices && (!plan->chunk_base || !plan->chunk_slice || !workspace || !status || !out || !out_len)))
        return fail(AVR_ERR_INVALID, "null plan / workspace / output pointer");
    if (workspace_bytes < avr::k2p_workspace_bytes(n_slices, plan->total_chunks, out_total))
        return fail(AVR_ERR_CAPACITY, "workspace of %zu bytes is smaller than avr_range_chunked_workspace_bytes()", workspace_bytes);
    if (int rc = select_device(device)) return rc;
    AVR_HIP(avr::launch_k2p(static_cast<hipStream_t>(stream), recs, rec_off, n_bins, uint32_t(n_slices), plan->chunk_base,
                            plan->chunk_slice, plan->total_chunks, out_total, workspace, out, out_off, out_len, status));
    return AVR_OK;
}

static int check_plan(const avr_chunk_plan *plan, size_t n_slices, bool need_blocks) {
    if (!plan || (n_slices && (!plan->res_off || !plan->chunk_base || !plan->chunk_slice || !plan->dig_off ||
                               (need_blocks && (!plan->blk_base || !plan->blk_slice)))))
        return fail(AVR_ERR_INVALID, "null plan pointer");
    return AVR_OK;
}

size_t avr_cabac_resolve_workspace_bytes(size_t n_slices, size_t n_states, const avr_chunk_plan *plan) {
    if (!plan || n_states > AVR_MAX_STATES) return 0;
    return avr::k1p_resolve_workspace_bytes(n_slices, uint32_t(n_states), plan);
}

int avr_cabac_resolve_device(int device, void *stream, const uint16_t *recs, const uint64_t *rec_off, const uint32_t *n_bins,
                             size_t n_slices, const uint8_t *init_states, size_t n_states, const avr_chunk_plan *plan,
                             void *workspace, size_t workspace_bytes, uint8_t *codes, int32_t *status, uint8_t *final_states) {
    if (int rc = check_common(rec_off, n_bins, codes, n_slices)) return rc;
    if (n_states > AVR_MAX_STATES) return fail(AVR_ERR_INVALID, "n_states %zu > %d", n_states, AVR_MAX_STATES);
    if (int rc = check_plan(plan, n_slices, true)) return rc;
    if (n_slices && (!workspace || !status)) return fail(AVR_ERR_INVALID, "null workspace / status");
    if (workspace_bytes < avr::k1p_resolve_workspace_bytes(n_slices, uint32_t(n_states), plan))
        return fail(AVR_ERR_CAPACITY, "workspace smaller than avr_cabac_resolve_workspace_bytes()");
    if (int rc = select_device(device)) return rc;
    AVR_HIP(avr::launch_k1p_resolve(static_cast<hipStream_t>(stream), recs, rec_off, n_bins, uint32_t(n_slices), init_states,
                                    uint32_t(n_states), plan, workspace, codes, status, final_states));
    return AVR_OK;
}

size_t avr_cabac_resolved_workspace_bytes(size_t n_slices, const avr_chunk_plan *plan) {
    return plan ? avr::k1p_code_workspace_bytes(n_slices, plan) : 0;
}

int avr_cabac_encode_resolved_device(int device, void *stream, const uint8_t *codes, const uint32_t *n_bins, size_t n_slices,
                                     const avr_chunk_plan *plan, void *workspace, size_t workspace_bytes, uint8_t *out,
                                     const uint64_t *out_off, uint32_t *out_len, int32_t *status) {
    if (int rc = check_common(codes, n_bins, out_off, n_slices)) return rc;
    if (int rc = check_plan(plan, n_slices, false)) return rc;
    if (n_slices && (!workspace || !status)) return fail(AVR_ERR_INVALID, "null workspace / status");
    if (workspace_bytes < avr::k1p_code_workspace_bytes(n_slices, plan))
        return fail(AVR_ERR_CAPACITY, "workspace smaller than avr_cabac_resolved_workspace_bytes()");
    if (int rc = select_device(device)) return rc;
    AVR_HIP(avr::launch_k1p_code(static_cast<hipStream_t>(stream), codes, n_bins, uint32_t(n_slices), plan, workspace, out, out_off,
                                 out_len, status));
    return AVR_OK;
}

int avr_cabac_encode_codes_device(int device, void *stream, const uint8_t *codes, const uint64_t *res_off, const uint32_t *n_bins,
                                  const uint32_t *order, size_t n_slices, uint8_t *out, const uint64_t *out_off,
                                  uint32_t *out_len, int32_t *status) {
    if (int rc = check_common(codes, n_bins, out_off, n_slices)) return rc;
    if (n_slices && (!res_off || !status || !out_len)) return fail(AVR_ERR_INVALID, "null device pointer");
    if (int rc = select_device(device)) return rc;
    AVR_HIP(avr::launch_cabac_encode_codes(static_cast<hipStream_t>(stream), codes, res_off, n_bins, order, uint32_t(n_slices), out,
                                           out_off, out_len, status));
    return AVR_OK;
}

int avr_cabac_encode_slices_device(int device, void *stream, const uint16_t *recs, const uint64_t *rec_off, const uint32_t *n_bins,
                                   const uint32_t *order, size_t n_slices, const uint8_t *init_states, size_t n_states,
                                   uint8_t *out, const uint64_t *out_off, uint32_t *out_len, int32_t *status, uint8_t *final_states) {
    if (int rc = check_common(rec_off, n_bins, out_off, n_slices)) return rc;
    if (n_states > AVR_MAX_STATES) return fail(AVR_ERR_INVALID, "n_states %zu > %d", n_states, AVR_MAX_STATES);
    if (int rc = select_device(device)) return rc;
    AVR_HIP(avr::launch_cabac_encode(false, static_cast<hipStream_t>(stream), recs, rec_off, n_bins, order, uint32_t(n_slices),
                                     init_states, uint32_t(n_states), out, out_off, out_len, status, final_states));
    return AVR_OK;
}

int avr_range_encode_slices_device(int device, void *stream, const uint16_t *recs, const uint64_t *rec_off, const uint32_t *n_bins,
                                   const uint32_t *order, size_t n_slices, uint8_t *out, const uint64_t *out_off,
                                   uint32_t *out_len, int32_t *status) {
    if (int rc = check_common(rec_off, n_bins, out_off, n_slices)) return rc;
    if (int rc = select_device(device)) return rc;
    AVR_HIP(avr::launch_range_encode(false, static_cast<hipStream_t>(stream), recs, rec_off, n_bins, order, uint32_t(n_slices),
                                     out, out_off, out_len, status));
    return AVR_OK;
}

// ------------------------------------------------------------------ dense context ids

int avr_context_census_device(int device, void *stream, const uint16_t *recs, uint64_t n_records, uint32_t *bitmap) {
    if ((n_records && !recs) || !bitmap || (n_records & 7)) return fail(AVR_ERR_INVALID, "bad argument (n_records must be a multiple of 8)");
    if (int rc = select_device(device)) return rc;
    AVR_HIP(avr::launch_context_census(static_cast<hipStream_t>(stream), recs, n_records, bitmap));
    return AVR_OK;
}

int avr_context_remap_device(int device, void *stream, uint16_t *recs, uint64_t n_records, const uint16_t *table) {
    if ((n_records && !recs) || !table || (n_records & 7)) return fail(AVR_ERR_INVALID, "bad argument (n_records must be a multiple of 8)");
    if (int rc = select_device(device)) return rc;
    AVR_HIP(avr::launch_context_remap(static_cast<hipStream_t>(stream), recs, n_records, table));
    return AVR_OK;
}

int avr_states_permute_device(int device, void *stream, const uint8_t *src, size_t n_src, uint8_t *dst, size_t n_dst,
                              const uint16_t *index, size_t n_index, size_t n_slices, int scatter) {
    if (n_slices && n_index && (!src || !dst || !index)) return fail(AVR_ERR_INVALID, "null pointer");
    if (n_src > AVR_MAX_STATES || n_dst > AVR_MAX_STATES || n_index > AVR_MAX_STATES) return fail(AVR_ERR_INVALID, "more than %d states", AVR_MAX_STATES);
    if (int rc = select_device(device)) return rc;
    AVR_HIP(avr::launch_states_permute(static_cast<hipStream_t>(stream), src, uint32_t(n_src), dst, uint32_t(n_dst), index,
                                       uint32_t(n_index), n_slices, scatter));
    return AVR_OK;
}

// ------------------------------------------------------------------ synthetic streams

int avr_synth_config_init(avr_synth_config *cfg, int workload, uint32_t scale_permille, uint64_t first_slice) {
    if (!cfg) return fail(AVR_ERR_INVALID, "null config");
    if (workload < 2 || workload > 5) return fail(AVR_ERR_INVALID, "workload %d: expected 2..5 (BASELINE.json configs)", workload);
    if (scale_permille == 0) return fail(AVR_ERR_INVALID, "scale_permille must be > 0");
    cfg->workload = workload;
    cfg->scale_permille = scale_permille;
    cfg->seed = 0x5EED0000ull + uint64_t(workload);     // SURVEY.md 8(d) seeds
    cfg->first_slice = first_slice;
    cfg->n_states = avr::synth_shape(workload, scale_permille, cfg->seed, 0).n_states;
    return AVR_OK;
}

namespace {
struct HostRecordSink {
    uint16_t *dst;
    uint32_t n = 0;
    void put_record(uint16_t r) { dst[n++] = r; }
};
}  // namespace

int avr_synth_count_host(const avr_synth_config *cfg, int kind, size_t n_slices, uint32_t *n_bins) {
    if (!cfg || (!n_bins && n_slices)) return fail(AVR_ERR_INVALID, "null argument");
    (void)kind;
    for (size_t i = 0; i < n_slices; i++) {
        avr::CountSink cs;
        avr::CabacSink<avr::CountSink> sink(cs);
        avr::synth_slice(cfg->workload, cfg->scale_permille, cfg->seed, cfg->first_slice + i, sink);
        n_bins[i] = cs.n;
    }
    return AVR_OK;
}

int avr_synth_generate_host(const avr_synth_config *cfg, int kind, size_t n_slices, const uint64_t *rec_off, uint16_t *recs,
                            uint8_t *init_states) {
    if (!cfg || ((!rec_off || !recs) && n_slices)) return fail(AVR_ERR_INVALID, "null argument");
    for (size_t i = 0; i < n_slices; i++) {
        HostRecordSink rs{recs + rec_off[i]};
        if (kind == AVR_KIND_CABAC) {
            avr::CabacSink<HostRecordSink> sink(rs);
            avr::synth_slice(cfg->workload, cfg->scale_permille, cfg->seed, cfg->first_slice + i, sink);
            if (init_states)
                for (uint32_t c = 0; c < cfg->n_states; c++)
                    init_states[i * cfg->n_states + c] = avr::synth_init_state(c, cfg->seed, cfg->first_slice + i);
        } else {
            avr::ModelSink<HostRecordSink> sink(rs);
            avr::synth_slice(cfg->workload, cfg->scale_permille, cfg->seed, cfg->first_slice + i, sink);
        }
    }
    return AVR_OK;
}

int avr_synth_count_device(int device, void *stream, const avr_synth_config *cfg, int kind, size_t n_slices, uint32_t *n_bins) {
    if (!cfg || (!n_bins && n_slices) || n_slices > 0x7fffffffu) return fail(AVR_ERR_INVALID, "bad argument");
    if (int rc = select_device(device)) return rc;
    AVR_HIP(avr::launch_synth_count(static_cast<hipStream_t>(stream), cfg->workload, cfg->scale_permille, cfg->seed, cfg->first_slice,
                                    kind, uint32_t(n_slices), n_bins));
    return AVR_OK;
}

int avr_synth_generate_slices_device(int device, void *stream, const avr_synth_config *cfg, int kind, size_t n_slices,
                                     const uint64_t *rec_off, uint16_t *recs, uint8_t *init_states) {
    if (!cfg || ((!rec_off || !recs) && n_slices) || n_slices > 0x7fffffffu) return fail(AVR_ERR_INVALID, "bad argument");
    if (int rc = select_device(device)) return rc;
    AVR_HIP(avr::launch_synth_slices(static_cast<hipStream_t>(stream), cfg->workload, cfg->scale_permille, cfg->seed, cfg->first_slice,
                                     kind, uint32_t(n_slices), rec_off, recs, init_states, cfg->n_states));
    return AVR_OK;
}

int avr_synth_generate_tiles_device(int device, void *stream, const avr_synth_config *cfg, int kind, size_t n_slices,
                                    const uint32_t *order, const uint64_t *tile_off, void *tiles, uint8_t *init_states) {
    if (!cfg || ((!tile_off || !tiles) && n_slices) || n_slices > 0x7fffffffu) return fail(AVR_ERR_INVALID, "bad argument");
    if (int rc = select_device(device)) return rc;
    AVR_HIP(avr::launch_synth_tiles(static_cast<hipStream_t>(stream), cfg->workload, cfg->scale_permille, cfg->seed, cfg->first_slice,
                                    kind, uint32_t(n_slices), order, tile_off, tiles, init_states, cfg->n_states));
    return AVR_OK;
}

// ------------------------------------------------------------------ host epilogue helpers

size_t avr_drop_stop_byte(const uint8_t *buf, size_t len) {
    // decompressor::cabac_decoder::finish, recode.cpp:1508-1512
    return (len > 0 && buf[len - 1] == 0x80) ? len - 1 : len;
}

size_t avr_tail_patch(uint8_t *buf, size_t len, int length_parity, uint8_t last_byte) {
    // decompressor::run, recode.cpp:1354-1360
    if (length_parity == -1) return len;
    if (length_parity != int(len & 1)) { buf[len] = last_byte; return len + 1; }
    if (len > 0) buf[len - 1] = last_byte;
    return len;
}

}  // extern "C"
