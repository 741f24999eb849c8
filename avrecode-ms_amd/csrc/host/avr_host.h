// Host side of the arithmetic re-encode path: what stays on the CPU around the GPU batches.
//
// The reference (pbluc/avrecode-ms, /root/reference) is C++17, so this layer is C++ and keeps
// the reference's names and semantics; errors are C++ exceptions exactly where the reference
// throws (std::runtime_error / std::invalid_argument), and the C wrappers in avr_host_c.cpp turn
// them into return codes.
//
//   h264_model           recode.cpp:625-1066, in avr_model.h: estimators, model keys, frame store,
//                        the nonzero-count side channel of the significance maps
//   range_decoder        recoded_code::decoder, arithmetic_code.h:209-298 (K3: stays on the CPU,
//                        it is interleaved with libavcodec's syntax parsing)
//   cabac_bin_decoder    the CABAC decoding engine of H.264 9.3.3.2, the part of libavcodec the
//                        compress direction calls (ff_get_cabac / _bypass / _terminate,
//                        recode.cpp:1183,1189,1195); written from the standard
//   compress_recorder    compressor::cabac_decoder minus the coder (recode.cpp:1141-1275):
//                        turns every decoded bin into a K2 range record (h264_symbol::execute,
//                        :1075-1103) and updates the model
//   decompress_recorder  decompressor::cabac_decoder minus the coder (recode.cpp:1418-1527):
//                        get / get_bypass / get_terminate answer from the range decoder, update
//                        *state as cabac_code.h:43-47 does, and record K1 CABAC records
//   Recoded              the .recode container (recode.proto:1-19), proto2 wire format by hand
//   surrogate markers, recognize/tail rules: recode.cpp:1534-1580, 1354-1360
#pragma once
#include <cstdint>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "../avr_tables.h"
#include "avr_model.h"
#include "../../../include/avrecode_ms_amd.h"

namespace avr {
namespace host {

// ---------------------------------------------------------------------------------------------
// recoded_code::decoder<const char*, uint8_t>  (arithmetic_code<uint64_t, uint8_t>, recode.cpp:322-323)
class range_decoder {
  public:
    range_decoder(const uint8_t *in, const uint8_t *end) : in_(in), end_(end) {         // arithmetic_code.h:218-230
        next_digit_ = consume_digit_aligned();
        low_ = next_digit_ / 2;                          // digit_alignment == 2 (:251-252)
        range_ = 256 / 2;
        while (range_ < kFixedOne) renormalize_and_consume_digit();
    }
    uint64_t range() const { return range_; }
    int get(uint64_t range_of_1) {                       // :232-248, probability already evaluated
        const uint64_t range_of_0 = range_ - range_of_1;
        const int symbol = low_ >= range_of_0;
        if (symbol) { low_ -= range_of_0; range_ = range_of_1; }
        else range_ = range_of_0;
        if (range_ < kMinRange)
            while (range_ < kFixedOne / 256) renormalize_and_consume_digit();
        return symbol;
    }

  private:
    static constexpr uint64_t kFixedOne = uint64_t(1) << 63;          // :54-55
    static constexpr uint64_t kMinRange = (kFixedOne / 256) / 16;     // :61-62
    void renormalize_and_consume_digit() {               // :259-275: the stream is read one bit late
        const uint32_t in_digit = consume_digit_aligned();
        const uint32_t digit = ((next_digit_ << 7) | (in_digit >> 1)) & 0xffu;
        next_digit_ = in_digit;
        low_ = low_ * 256 + digit;
        range_ *= 256;
    }
    uint32_t consume_digit_aligned() { return in_ != end_ ? *in_++ : 0; }   // :279-288: zeros past the end
    const uint8_t *in_, *end_;
    uint32_t next_digit_;
    uint64_t low_, range_;
};

// ---------------------------------------------------------------------------------------------
// CABAC decoding engine, ITU-T H.264 9.3.3.2 (Figures 9-2, 9-3, 9-5, 9-6): what ff_get_cabac,
// ff_get_cabac_bypass and ff_get_cabac_terminate compute (recode.cpp:1183,1189,1195), including the
// update of *state (2*pStateIdx + valMPS).
class cabac_bin_decoder {
  public:
    cabac_bin_decoder(const uint8_t *buf, size_t size) : buf_(buf), nbits_(size * 8) {
        for (int i = 0; i < 9; i++) offset_ = (offset_ << 1) | read_bit();                // 9.3.1.2
    }
    int get(uint8_t *state) {                             // Figure 9-3
        const CabacTables &t = tables();
        int p = *state >> 1, mps = *state & 1, bin;
        const uint32_t rlps = t.range_lps[p][(range_ >> 6) & 3];
        range_ -= rlps;
        if (offset_ >= range_) {
            bin = !mps; offset_ -= range_; range_ = rlps;
            *state = t.mlps_state[127 - *state];
        } else {
            bin = mps;
            *state = t.mlps_state[128 + *state];
        }
        (void)p;
        while (range_ < 256) { range_ <<= 1; offset_ = (offset_ << 1) | read_bit(); }
        return bin;
    }
    int get_bypass() {                                    // Figure 9-5
        offset_ = (offset_ << 1) | read_bit();
        if (offset_ >= range_) { offset_ -= range_; return 1; }
        return 0;
    }
    int get_terminate() {                                 // Figure 9-6
        range_ -= 2;
        if (offset_ >= range_) return 1;
        while (range_ < 256) { range_ <<= 1; offset_ = (offset_ << 1) | read_bit(); }
        return 0;
    }
    static const CabacTables &tables() { static const CabacTables t = make_cabac_tables(); return t; }
    size_t bit_position() const { return pos_; }         // bits read so far: the 9 of the initialisation + one per renormalisation shift

  private:
    uint32_t read_bit() {
        uint32_t b = 0;
        if (pos_ < nbits_) b = (buf_[pos_ >> 3] >> (7 - (pos_ & 7))) & 1;
        pos_++;
        return b;
    }
    const uint8_t *buf_;
    size_t nbits_, pos_ = 0;
    uint32_t range_ = 510, offset_ = 0;
};

// ---------------------------------------------------------------------------------------------
// Context identity.  libavcodec hands get() a pointer into its per-slice cabac_state[] and the reference
// keys its model on that ADDRESS (model_key holds a const void *, recode.cpp:325; the estimators live for the
// whole file, :1065, :669-672, and libavcodec's array stays where it is, so one address is one context
// from the first slice to the last).  Nothing else is known about the pointer -- no callback announces the
// array -- so the recorders number the addresses as they first appear: one table per file, shared by the
// slices' recorders.  Ids are dense, which is also what the device wants of a selector (fewer state bytes
// per lane).  Look-up: addresses within 2 KiB of the first one seen (any 1024-byte array holding it) go
// through a flat table, anything else through a hash map.
class context_ids {
  public:
    int id_of(const uint8_t *state) {
        const uintptr_t at = reinterpret_cast<uintptr_t>(state) - window_lo_;
        if (at < kWindow && near_[at] >= 0) return near_[at];
        return assign(state);
    }
    const uint8_t *pointer_of(int id) const { return pointers_.at(size_t(id)); }
    int size() const { return int(pointers_.size()); }

  private:
    static constexpr uintptr_t kWindow = 4096;
    int assign(const uint8_t *state) {
        if (pointers_.empty()) {
            window_lo_ = reinterpret_cast<uintptr_t>(state) - kWindow / 2;
            near_.assign(kWindow, int16_t(-1));
        }
        const uintptr_t at = reinterpret_cast<uintptr_t>(state) - window_lo_;
        if (at >= kWindow) {
            auto it = far_.find(state);
            if (it != far_.end()) return it->second;
        }
        if (pointers_.size() >= AVR_MAX_STATES)
            throw std::invalid_argument("more than " + std::to_string(AVR_MAX_STATES) + " distinct CABAC state addresses in one file");
        const int id = int(pointers_.size());
        pointers_.push_back(state);
        if (at < kWindow) near_[at] = int16_t(id); else far_[state] = id;
        return id;
    }
    uintptr_t window_lo_ = 0;
    std::vector<int16_t> near_;
    std::unordered_map<const uint8_t *, int> far_;
    std::vector<const uint8_t *> pointers_;
};

// ---------------------------------------------------------------------------------------------
// compress direction: every decoded bin becomes a range record (what encoder.put would have read)
class compress_recorder {
  public:
    explicit compress_recorder(h264_model *model) : model_(model) { model_->reset(); }  // recode.cpp:1162-1163

    // compressor::cabac_decoder::execute_symbol (recode.cpp:1167-1180): the bins of a significance
    // map are held back until the block's nonzero count is known (QUEUE_MODE), everything else is
    // coded where it stands
    void execute_symbol(int symbol, int context) {
        if (finished_) throw std::runtime_error("compress_recorder: bin after the end of the slice");
        if (queueing_ == PIP_SIGNIFICANCE_MAP || queueing_ == PIP_SIGNIFICANCE_EOB || !queue_.empty()) {
            queue_.push_back({symbol, context});
            model_->update_state_tracking(symbol);
        } else {
            execute(symbol, context);
        }
    }
    void begin_coding_type(CodingType ct, int zigzag_index, int param0, int param1) {   // :1201-1209
        const bool begin_queue = model_->begin_coding_type(ct, zigzag_index, param0, param1);
        if (begin_queue && (ct == PIP_SIGNIFICANCE_MAP || ct == PIP_SIGNIFICANCE_EOB)) {
            if (queueing_ != PIP_UNKNOWN || !queue_.empty()) throw std::runtime_error("compress_recorder: nested queues are not supported");   // :1243-1245
            queueing_ = ct;
        }
    }
    void end_coding_type(CodingType ct) {                                               // :1210-1236
        // One deliberate difference from recode.cpp:1214-1218.  The keys of the count's bits include the macroblock's "has an
        // 8x8 block" flag (meta.is_8x8, :889), and model->end_coding_type sets that flag for the block just ended (:962).  The
        // reference's compressor calls it BEFORE finished_queueing, its decompressor decodes the count at begin_coding_type,
        // i.e. before (:1483-1492): for the first 8x8 block of a macroblock the two sides of the reference use different
        // estimators and its decoder loses the stream (the path is never run there: the hooks are "Not called", :182-210).
        // A container has to decode, so this side codes the count with the flag as the decoder will see it -- as it was
        // when the block began.
        const bool had_8x8 = model_->current_block_flag_8x8();
        model_->end_coding_type(ct);
        if (ct != PIP_SIGNIFICANCE_MAP && ct != PIP_SIGNIFICANCE_EOB) return;
        if (queueing_ == PIP_UNKNOWN) throw std::runtime_error("compress_recorder: end of a coding type that was not begun");         // :1250
        queueing_ = PIP_UNKNOWN;
        const bool has_8x8 = model_->current_block_flag_8x8();
        model_->set_current_block_flag_8x8(had_8x8);
        model_->finished_queueing(ct, [&](const model_key &key, int *symbol) {          // the nonzero count first ...
            record(*symbol, key);
            model_->update_state_for_model_key(*symbol, key);
        });
        model_->set_current_block_flag_8x8(has_8x8);
        model_->reset_mb_significance_state_tracking();                                 // ... then the map (:1254-1265)
        for (const queued &q : queue_) execute(q.symbol, q.context);
        queue_.clear();
        model_->coding_type = PIP_UNKNOWN;
    }
    bool finished() const { return finished_; }
    const std::vector<uint16_t> &records() const { return recs_; }

  private:
    // h264_symbol::execute (recode.cpp:1075-1103) with the coder call replaced by a record
    void execute(int symbol, int context) {
        if (model_->coding_type != PIP_SIGNIFICANCE_EOB) record(symbol, model_->get_model_key(context));   // :1080
        model_->update_state(symbol, context);                                          // :1094
        if (context == kKeyTerminate && symbol) finished_ = true;                       // :1099-1102
    }
    void record(int symbol, const model_key &key) {                                     // :823-827 inputs
        const h264_model::estimator *e = model_->lookup(key);
        recs_.push_back(uint16_t((symbol & 1) | (e->pos << 1) | (e->neg << 8)));
    }
    struct queued { int symbol, context; };
    h264_model *model_;
    std::vector<uint16_t> recs_;
    std::vector<queued> queue_;                                                          // symbol_buffer, :1273
    CodingType queueing_ = PIP_UNKNOWN;
    bool finished_ = false;
};

// decompress direction: the hook surface of decompressor::cabac_decoder (recode.cpp:1442-1481)
class decompress_recorder {
  public:
    // cabac: the block's recoded bytes (recode.cpp:1429-1430); ids: the file's table of context addresses
    decompress_recorder(h264_model *model, const uint8_t *cabac, size_t cabac_size, context_ids *ids)
        : model_(model), decoder_(cabac, cabac + cabac_size), ids_(ids) {
        model_->reset();                                                                // :1428
        memset(seen_, 0, sizeof seen_);
        memset(init_states_, 0, sizeof init_states_);
    }
    int get(uint8_t *state) {                                                           // :1442-1456
        const int context = ids_->id_of(state);
        int symbol;
        if (model_->coding_type == PIP_SIGNIFICANCE_EOB) symbol = std::get<1>(model_->get_model_key(context));   // not coded: implied by the count
        else symbol = decoder_.get(model_->probability_for_state(decoder_.range(), context));
        if (!seen_[context]) { seen_[context] = 1; init_states_[context] = *state; n_states_ = std::max(n_states_, context + 1); }
        recs_.push_back(uint16_t(symbol | (context << 1)));                             // cabac_encoder.put, deferred
        codes_.push_back(uint8_t(AVR_CODE_CONTEXT(*state, symbol)));                    // the same call with *state resolved
        const CabacTables &t = cabac_bin_decoder::tables();                             // cabac_code.h:43-47
        *state = symbol != (*state & 1) ? t.mlps_state[127 - *state] : t.mlps_state[128 + *state];
        model_->update_state(symbol, context);                                          // :1454
        return symbol;
    }
    int get_bypass() {                                                                  // :1458-1467
        const int symbol = decoder_.get(model_->probability_for_state(decoder_.range(), kKeyBypass));
        model_->update_state(symbol, kKeyBypass);
        recs_.push_back(uint16_t(symbol | (AVR_SEL_BYPASS << 1)));
        codes_.push_back(uint8_t(AVR_CODE_BYPASS(symbol)));
        return symbol;
    }
    int get_terminate() {                                                               // :1469-1481
        const int symbol = decoder_.get(model_->probability_for_state(decoder_.range(), kKeyTerminate));
        model_->update_state(symbol, kKeyTerminate);
        recs_.push_back(uint16_t(symbol | (AVR_SEL_TERMINATE << 1)));
        codes_.push_back(uint8_t(AVR_CODE_TERMINATE(symbol)));
        if (symbol) finished_ = true;
        return symbol;
    }
    void begin_coding_type(CodingType ct, int zigzag_index, int param0, int param1) {   // :1483-1499
        const bool begin_queue = model_->begin_coding_type(ct, zigzag_index, param0, param1);
        if (begin_queue && ct)
            model_->finished_queueing(ct, [&](const model_key &key, int *symbol) {      // the block's nonzero count comes first
                *symbol = decoder_.get(model_->probability_for_model_key(decoder_.range(), key));
                model_->update_state_for_model_key(*symbol, key);
            });
    }
    void end_coding_type(CodingType ct) { model_->end_coding_type(ct); }                // :1500-1505
    bool finished() const { return finished_; }
    const std::vector<uint16_t> &records() const { return recs_; }
    // the same bins as resolved codes (AVR_CODE_*): what avr_batch_add_slice_codes takes; half the bytes, no state arrays
    const std::vector<uint8_t> &codes() const { return codes_; }
    const uint8_t *init_states() const { return init_states_; }   // *state as it was at each context's first bin in this slice
    int n_states() const { return n_states_; }                     // highest context id touched + 1

  private:
    h264_model *model_;
    range_decoder decoder_;
    context_ids *ids_;
    std::vector<uint16_t> recs_;
    std::vector<uint8_t> codes_;
    uint8_t seen_[AVR_MAX_STATES], init_states_[AVR_MAX_STATES];
    int n_states_ = 0;
    bool finished_ = false;
};

// ---------------------------------------------------------------------------------------------
// .recode container: proto2 message Recoded (recode.proto:1-19), written and parsed by hand.
// Fields are emitted in field-number order, as protobuf's C++ serializer does, and a field that
// was set is emitted even when empty (set_literal with a zero-length gap, recode.cpp:1288).
struct Block {                                          // recode.proto:10-17
    bool has_size = false;           int64_t size = 0;
    bool has_literal = false;        std::string literal;
    bool has_skip_coded = false;     bool skip_coded = false;
    bool has_cabac = false;          std::string cabac;
    bool has_length_parity = false;  bool length_parity = false;
    bool has_last_byte = false;      std::string last_byte;
};

struct Recoded {
    std::vector<Block> block;                           // recode.proto:18

    static void put_varint(std::string &o, uint64_t v) {
        while (v >= 0x80) { o.push_back(char(v | 0x80)); v >>= 7; }
        o.push_back(char(v));
    }
    static void put_bytes(std::string &o, int field, const std::string &s) {
        put_varint(o, uint64_t(field) << 3 | 2);
        put_varint(o, s.size());
        o += s;
    }
    std::string SerializeAsString() const {             // recode.cpp:1131
        std::string out;
        for (const Block &b : block) {
            std::string m;
            if (b.has_size) { put_varint(m, 1 << 3 | 0); put_varint(m, uint64_t(b.size)); }
            if (b.has_literal) put_bytes(m, 2, b.literal);
            if (b.has_skip_coded) { put_varint(m, 3 << 3 | 0); put_varint(m, b.skip_coded); }
            if (b.has_cabac) put_bytes(m, 4, b.cabac);
            if (b.has_length_parity) { put_varint(m, 5 << 3 | 0); put_varint(m, b.length_parity); }
            if (b.has_last_byte) put_bytes(m, 6, b.last_byte);
            put_bytes(out, 2, m);
        }
        return out;
    }

    static bool get_varint(const uint8_t *&p, const uint8_t *end, uint64_t *v) {
        uint64_t r = 0;
        for (int shift = 0; shift < 64 && p < end; shift += 7) {
            const uint8_t c = *p++;
            r |= uint64_t(c & 0x7f) << shift;
            if (!(c & 0x80)) { *v = r; return true; }
        }
        return false;
    }
    static bool skip_field(const uint8_t *&p, const uint8_t *end, int wire) {
        uint64_t v;
        switch (wire) {
            case 0: return get_varint(p, end, &v);
            case 1: if (end - p < 8) return false; p += 8; return true;
            case 2: if (!get_varint(p, end, &v) || uint64_t(end - p) < v) return false; p += v; return true;
            case 5: if (end - p < 4) return false; p += 4; return true;
            default: return false;
        }
    }
    static bool parse_block(const uint8_t *p, const uint8_t *end, Block *b) {
        while (p < end) {
            uint64_t tag, v;
            if (!get_varint(p, end, &tag)) return false;
            const int field = int(tag >> 3), wire = int(tag & 7);
            const bool bytes_field = field == 2 || field == 4 || field == 6;
            if (field >= 1 && field <= 6 && wire == (bytes_field ? 2 : 0)) {
                if (!get_varint(p, end, &v)) return false;
                if (bytes_field) {
                    if (uint64_t(end - p) < v) return false;
                    std::string s(reinterpret_cast<const char *>(p), size_t(v));
                    p += v;
                    if (field == 2) { b->has_literal = true; b->literal = s; }
                    else if (field == 4) { b->has_cabac = true; b->cabac = s; }
                    else { b->has_last_byte = true; b->last_byte = s; }
                } else if (field == 1) { b->has_size = true; b->size = int64_t(v); }
                else if (field == 3) { b->has_skip_coded = true; b->skip_coded = v != 0; }
                else { b->has_length_parity = true; b->length_parity = v != 0; }
            } else if (!skip_field(p, end, wire)) {
                return false;
            }
        }
        return true;
    }
    bool ParseFromArray(const void *data, size_t size) {  // recode.cpp:1338, :1342
        block.clear();
        const uint8_t *p = static_cast<const uint8_t *>(data), *end = p + size;
        while (p < end) {
            uint64_t tag, len;
            if (!get_varint(p, end, &tag)) return false;
            if ((tag >> 3) == 2 && (tag & 7) == 2) {
                if (!get_varint(p, end, &len) || uint64_t(end - p) < len) return false;
                Block b;
                if (!parse_block(p, p + len, &b)) return false;
                block.push_back(std::move(b));
                p += len;
            } else if (!skip_field(p, end, int(tag & 7))) {
                return false;
            }
        }
        return true;
    }
};

// Surrogate payloads of the decompress direction.  The decoder is fed, in place of each coded slice payload,
// a block of the payload's size that starts with a marker identifying the block (checked again when the
// decoder hands the bytes to init_decoder) and is filled up with 'X' (recode.cpp:1534-1551).  The marker is
// the block's sequence number written with SURROGATE_MARKER_BYTES base-255 digits, least significant first,
// each stored + 1: no zero byte, so no start code or emulation prevention can arise inside it.
constexpr int SURROGATE_MARKER_BYTES = 8;               // recode.cpp:33
inline std::string next_surrogate_marker(uint64_t *sequence_number) {
    uint64_t value = *sequence_number;
    *sequence_number += 1;
    std::string marker;
    marker.reserve(SURROGATE_MARKER_BYTES);
    while (marker.size() < size_t(SURROGATE_MARKER_BYTES)) {
        marker.push_back(char(value % 255 + 1));
        value /= 255;
    }
    return marker;
}
inline std::string make_surrogate_block(const std::string &marker, size_t size) {
    if (marker.size() > size) throw std::runtime_error("Invalid coded block size for surrogate: " + std::to_string(size));
    return marker + std::string(size - marker.size(), 'X');
}

}  // namespace host
}  // namespace avr
