// compressor / decompressor / roundtrip: the orchestration of the reference (recode.cpp:1109-1640)
// around the GPU batches.  Block bookkeeping, literal gaps, skip blocks, surrogate payloads and
// the tail patch behave as the reference's do (the .recode bytes and the error messages are the
// contract); what differs is WHEN bins are coded: the recorders collect them per slice and one avr_batch codes all slices of the file at
// the end of run() (legal because nothing reads the coded bytes earlier: recode.cpp:1131,
// 1352-1363).
//
// libavcodec's place is taken by anything that drives `hooks` (the callback table the reference
// installs with codec->hooks = &hooks, recode.cpp:130, 219-235).  The reference's fork of FFmpeg
// is not available offline; tests drive the same table from recorded slices (slice_feeder).
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>

#include "avr_host.h"

namespace avr {
namespace host {

// The callback surface of recode.cpp:149-216, with the same names, order and arity.  `ctx` stands
// for libavcodec's CABACContext*: the hook adapter only uses it as the identity of the decoder
// (recode.cpp:153, 236).  A null return from init_decoder means "hooks disabled for this slice"
// (the reference nulls ctx->coding_hooks and calls ff_reset_cabac_decoder, :1148-1150, 1433-1435).
struct hooks {
    void *opaque;
    struct {
        void *(*init_decoder)(void *opaque, void *ctx, const uint8_t *buf, int size);
        int (*get)(void *opaque, uint8_t *state);
        int (*get_bypass)(void *opaque);
        int (*get_terminate)(void *opaque);
        const uint8_t *(*skip_bytes)(void *opaque, int n);
    } cabac;
    struct {
        void (*frame_spec)(void *opaque, int frame_num, int mb_width, int mb_height);
        void (*mb_xy)(void *opaque, int x, int y);
        void (*begin_sub_mb)(void *opaque, int cat, int scan8index, int max_coeff, int is_dc, int chroma422);
        void (*end_sub_mb)(void *opaque, int cat, int scan8index, int max_coeff, int is_dc, int chroma422);
        void (*begin_coding_type)(void *opaque, int ct, int zigzag_index, int param0, int param1);
        void (*end_coding_type)(void *opaque, int ct);
    } model;
};

// What decodes the stream and calls the hooks: libavcodec in the reference (av_decoder,
// recode.cpp:80-237).  read_packet is the custom AVIO callback (recode.cpp:145-148).
struct stream_decoder {
    virtual ~stream_decoder() {}
    // decode everything, pulling bytes through read_packet(opaque, buffer, size) and calling h
    virtual void decode_video(hooks *h, int (*read_packet)(void *, uint8_t *, int), void *opaque) = 0;
    // Asked by the compressor inside init_decoder, about the payload being offered: will the decoder get through it?
    // libavcodec always does; the build's own syntax parser (avr_h264.h) covers a subset of H.264 and answers by
    // trying.  "No" makes the compressor treat the slice like one whose payload it cannot find (skip_coded block,
    // the bytes stay literal): the container stays lossless whatever the parser can or cannot do.
    virtual bool payload_decodes() { return true; }
    // Told by the compressor before decode_video: payload_decodes() will be asked about every slice.  A decoder whose answer
    // costs a parse of the payload can then work the answers out ahead, for all slices at once (on several threads: slices
    // parse independently of each other; only the hooks have to be called in stream order).
    virtual void expect_payload_questions() {}
};

// AVR_TIMING=1: the phases of a run on stderr (the reference prints nothing of the kind; off by default)
struct phase_timer {
    const char *what;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit phase_timer(const char *w) : what(w) {}
    ~phase_timer() {
        static const bool on = getenv("AVR_TIMING") != nullptr;
        if (on) fprintf(stderr, "[timing] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
};

inline void gpu_check(int rc) { if (rc < 0) throw std::runtime_error(std::string("avr: ") + avr_last_error()); }

struct batch_holder {                                    // RAII around the C ABI's avr_batch
    avr_batch *b;
    batch_holder(int device, size_t max_slices, size_t max_bins) : b(avr_batch_create(device, max_slices, max_bins)) {
        if (!b) throw std::runtime_error(std::string("avr: ") + avr_last_error());
    }
    ~batch_holder() { avr_batch_destroy(b); }
};

template <class Driver>
struct hook_adapter {                                    // av_decoder<Driver>'s stubs, recode.cpp:149-216
    static hooks make(Driver *d) {
        hooks h{};
        h.opaque = d;
        h.cabac.init_decoder = [](void *o, void *ctx, const uint8_t *buf, int size) -> void * {
            Driver *self = static_cast<Driver *>(o);
            auto dec = std::unique_ptr<typename Driver::cabac_decoder>(new typename Driver::cabac_decoder(self, buf, size));
            typename Driver::cabac_decoder *raw = dec->hooked() ? dec.get() : nullptr;
            self->cabac_contexts[ctx] = std::move(dec);  // re-init of the same context replaces it (:153)
            return raw;
        };
        h.cabac.get = [](void *o, uint8_t *state) { return static_cast<typename Driver::cabac_decoder *>(o)->get(state); };
        h.cabac.get_bypass = [](void *o) { return static_cast<typename Driver::cabac_decoder *>(o)->get_bypass(); };
        h.cabac.get_terminate = [](void *o) { return static_cast<typename Driver::cabac_decoder *>(o)->get_terminate(); };
        h.cabac.skip_bytes = [](void *, int) -> const uint8_t * {
            throw std::runtime_error("Not implemented: CABAC decoder doesn't use skip_bytes.");     // :168-170
        };
        h.model.frame_spec = [](void *o, int frame_num, int mb_width, int mb_height) {                  // :173-176
            static_cast<Driver *>(o)->get_model()->update_frame_spec(frame_num, mb_width, mb_height);
        };
        h.model.mb_xy = [](void *o, int x, int y) {                                                     // :177-181
            h264_model *m = static_cast<Driver *>(o)->get_model();
            m->mb_coord.mb_x = x;
            m->mb_coord.mb_y = y;
        };
        h.model.begin_sub_mb = [](void *o, int cat, int scan8index, int max_coeff, int is_dc, int chroma422) {   // :182-189
            h264_model *m = static_cast<Driver *>(o)->get_model();
            m->sub_mb_cat = cat;
            m->mb_coord.scan8_index = scan8index;
            m->sub_mb_size = max_coeff;
            m->sub_mb_is_dc = is_dc;
            m->sub_mb_chroma422 = chroma422;
        };
        h.model.end_sub_mb = [](void *o, int cat, int scan8index, int max_coeff, int is_dc, int chroma422) {     // :190-202
            h264_model *m = static_cast<Driver *>(o)->get_model();
            if (m->sub_mb_cat != cat || m->mb_coord.scan8_index != scan8index || m->sub_mb_size != max_coeff ||
                m->sub_mb_is_dc != is_dc || m->sub_mb_chroma422 != chroma422)
                throw std::runtime_error("end_sub_mb does not match begin_sub_mb");      // asserts in the reference
            m->sub_mb_cat = -1;
            m->mb_coord.scan8_index = -1;
            m->sub_mb_size = -1;
            m->sub_mb_is_dc = 0;
            m->sub_mb_chroma422 = 0;
        };
        // the coding-type hooks go to the one live CABAC decoder (:203-215)
        h.model.begin_coding_type = [](void *o, int ct, int zigzag_index, int param0, int param1) {
            only_decoder(static_cast<Driver *>(o))->begin_coding_type(CodingType(ct), zigzag_index, param0, param1);
        };
        h.model.end_coding_type = [](void *o, int ct) { only_decoder(static_cast<Driver *>(o))->end_coding_type(CodingType(ct)); };
        return h;
    }
    static typename Driver::cabac_decoder *only_decoder(Driver *d) {
        if (d->cabac_contexts.size() != 1) throw std::runtime_error("coding-type hook with " + std::to_string(d->cabac_contexts.size()) + " live CABAC decoders");   // :206, :212
        return d->cabac_contexts.begin()->second.get();
    }
};

// ---------------------------------------------------------------------------------------------
class compressor {                                       // recode.cpp:1109-1316
  public:
    compressor(const std::string &original_bytes, int device = 0) : original_(original_bytes), device_(device) {}

    std::string run(stream_decoder *d) {                 // :1122-1132
        prepare(d);
        { phase_timer t("compress: GPU batch (K2)"); code_pending(); }
        return finish();
    }
    // run() in three steps, so that a caller with several files (recode test <dir>, test.cpp:113-148) can code the slices of all of
    // them in one GPU batch: prepare() = everything the host does up to the coding (parse, hooks, recorders), add_to() / take_from() =
    // this file's slices into and out of a batch the caller owns, finish() = the final literal and the container's bytes.
    void prepare(stream_decoder *d) {
        decoder_ = d;
        d->expect_payload_questions();
        hooks h = hook_adapter<compressor>::make(this);
        phase_timer t("compress: parse + record");
        d->decode_video(&h, [](void *o, uint8_t *buf, int size) { return static_cast<compressor *>(o)->read_packet(buf, size); }, this);
        cabac_contexts.clear();
    }
    size_t pending_slices() const { return pending_.size(); }
    size_t pending_bins() const { size_t bins = 0; for (auto &p : pending_) bins += p.recs.size(); return bins; }
    void add_to(avr_batch *b) {                          // the slice indices the batch hands out are consecutive: the first one is kept
        first_in_batch_ = -1;
        for (auto &p : pending_) {
            const int idx = avr_batch_add_slice_range(b, p.recs.data(), p.recs.size());
            gpu_check(idx);
            if (first_in_batch_ < 0) first_in_batch_ = idx;
            std::vector<uint16_t>().swap(p.recs);        // the batch has its copy
        }
    }
    void take_from(avr_batch *b) {
        for (size_t i = 0; i < pending_.size(); i++) {
            const uint8_t *bytes; size_t len; int status;
            gpu_check(avr_batch_get(b, size_t(first_in_batch_) + i, &bytes, &len, &status));
            if (status == AVR_SLICE_ZERO_PROB) throw std::runtime_error("Encoder error: emitted a zero-probability symbol.");   // arithmetic_code.h:117
            if (status != AVR_SLICE_OK) throw std::runtime_error("avr: slice status " + std::to_string(status));
            Block &blk = out_.block[pending_[i].block];
            blk.has_cabac = true;                        // out->set_cabac, recode.cpp:1101
            blk.cabac.assign(reinterpret_cast<const char *>(bytes), len);
        }
        pending_.clear();
    }
    std::string finish() {
        Block final_literal;
        final_literal.has_literal = true;
        final_literal.literal = original_.substr(prev_coded_block_end_);
        out_.block.push_back(final_literal);
        return out_.SerializeAsString();
    }

    int read_packet(uint8_t *buffer_out, int size) {     // :1134-1139
        size = std::min<int>(size, int(original_.size() - read_offset_));
        memcpy(buffer_out, original_.data() + read_offset_, size);
        read_offset_ += size;
        return size;
    }

    class cabac_decoder {                                // :1141-1275
      public:
        cabac_decoder(compressor *c, const uint8_t *buf, int size) : c_(c), decoder_(buf, size) {
            block_ = c->find_next_coded_block_and_emit_literal(buf, size);
            if (block_ < 0) return;                      // skipped: hooks off for this slice (:1146-1152)
            c->out_.block[block_].has_size = true;       // :1154
            c->out_.block[block_].size = size;
            recorder_.reset(new compress_recorder(&c->model_));   // :1161-1163
        }
        ~cabac_decoder() { if (recorder_) c_->pending_.push_back({block_, recorder_->records()}); }
        bool hooked() const { return block_ >= 0; }
        int get(uint8_t *state) {                        // :1182-1186
            const int symbol = decoder_.get(state);      // ::ff_get_cabac on the private context copy
            recorder_->execute_symbol(symbol, c_->contexts_.id_of(state));   // context identity is the address (recode.cpp:325)
            return symbol;
        }
        int get_bypass() {                               // :1188-1192
            const int symbol = decoder_.get_bypass();
            recorder_->execute_symbol(symbol, kKeyBypass);
            return symbol;
        }
        int get_terminate() {                            // :1194-1199
            const int symbol = decoder_.get_terminate() != 0;
            recorder_->execute_symbol(symbol, kKeyTerminate);
            return symbol;
        }
        void begin_coding_type(CodingType ct, int zigzag_index, int param0, int param1) {   // :1201-1209
            if (recorder_) recorder_->begin_coding_type(ct, zigzag_index, param0, param1);
        }
        void end_coding_type(CodingType ct) { if (recorder_) recorder_->end_coding_type(ct); }   // :1210-1236

      private:
        compressor *c_;
        cabac_bin_decoder decoder_;
        int block_ = -1;
        std::unique_ptr<compress_recorder> recorder_;
    };

    h264_model *get_model() { return &model_; }          // :1276-1278
    const context_ids &contexts() const { return contexts_; }
    std::map<void *, std::unique_ptr<cabac_decoder>> cabac_contexts;     // :236

  private:
    // What the reference's find_next_coded_block_and_emit_literal does (recode.cpp:1282-1304).  libavcodec hands
    // over the slice payload it is about to decode; the file bytes read so far and not yet accounted for are
    // searched for it.  Found (and long enough to carry a surrogate marker later): everything in front of it
    // becomes a literal block -- present even when empty -- and a block for the recoded payload follows, noting
    // the parity of the payload's length and its last byte (what the tail patch needs, :1354-1360); the index
    // of that block is returned.  Not found (the NAL was unescaped on the way) or too short: a skip_coded block
    // records the size, the bytes stay in the literal stream, and -1 says "leave this slice alone".
    int find_next_coded_block_and_emit_literal(const uint8_t *buf, int size) {
        const size_t window_begin = size_t(prev_coded_block_end_), window_end = size_t(read_offset_);
        size_t where = std::string::npos;
        if (size >= SURROGATE_MARKER_BYTES && (!decoder_ || decoder_->payload_decodes())) {
            const void *hit = memmem(original_.data() + window_begin, window_end - window_begin, buf, size_t(size));
            if (hit) where = size_t(static_cast<const char *>(hit) - original_.data());
        }
        if (where == std::string::npos) {
            Block skipped;
            skipped.has_skip_coded = true;
            skipped.skip_coded = true;
            skipped.has_size = true;
            skipped.size = size;
            out_.block.push_back(skipped);
            return -1;
        }
        Block in_front;
        in_front.has_literal = true;
        in_front.literal = original_.substr(window_begin, where - window_begin);
        out_.block.push_back(in_front);
        Block coded;
        coded.has_length_parity = true;
        coded.length_parity = (size % 2) != 0;
        if (size > 1) {
            coded.has_last_byte = true;
            coded.last_byte = std::string(1, char(buf[size - 1]));
        }
        out_.block.push_back(coded);
        prev_coded_block_end_ = int(where) + size;
        return int(out_.block.size()) - 1;
    }

    void code_pending() {                                // one K2 batch for the whole file
        if (pending_.empty()) return;
        batch_holder bh(device_, pending_.size(), pending_bins() + 8);
        add_to(bh.b);
        gpu_check(avr_batch_run(bh.b));
        take_from(bh.b);
    }

    struct pending { int block; std::vector<uint16_t> recs; };
    int first_in_batch_ = -1;
    std::string original_;
    int device_;
    int read_offset_ = 0, prev_coded_block_end_ = 0;
    context_ids contexts_;                               // one numbering of the state addresses for the whole file
    stream_decoder *decoder_ = nullptr;
    h264_model model_;
    Recoded out_;
    std::vector<pending> pending_;
};

// ---------------------------------------------------------------------------------------------
class decompressor {                                     // recode.cpp:1319-1598
    struct block_state {                                 // :1321-1328
        bool coded = false;
        std::string surrogate_marker, out_bytes;
        bool done = false;
        int8_t length_parity = -1;
        uint8_t last_byte = 0;
    };

  public:
    decompressor(const std::string &in_bytes, int device = 0) : device_(device) {
        if (!in_.ParseFromArray(in_bytes.data(), in_bytes.size())) throw std::invalid_argument("Failed to parse the recoded input");
    }

    std::string run(stream_decoder *d) {                 // :1345-1364
        prepare(d);
        { phase_timer t("decompress: GPU batch (K1)"); code_pending(); }
        return finish();
    }
    // run() in three steps, as compressor's: prepare() = the range decoder (K3) and the parse, recording resolved codes; add_to() /
    // take_from() = this file's slices into and out of a batch the caller owns; finish() = tail patches and the file's bytes.
    void prepare(stream_decoder *d) {
        blocks_.clear();
        blocks_.resize(in_.block.size());
        hooks h = hook_adapter<decompressor>::make(this);
        phase_timer t("decompress: K3 + parse");
        d->decode_video(&h, [](void *o, uint8_t *buf, int size) { return static_cast<decompressor *>(o)->read_packet(buf, size); }, this);
        cabac_contexts.clear();
    }
    size_t pending_slices() const { return pending_.size(); }
    size_t pending_bins() const { size_t bins = 0; for (auto &p : pending_) bins += p.codes.size(); return bins; }
    void add_to(avr_batch *b) {
        first_in_batch_ = -1;
        for (auto &p : pending_) {
            const int idx = avr_batch_add_slice_codes(b, p.codes.data(), p.codes.size());
            gpu_check(idx);
            if (first_in_batch_ < 0) first_in_batch_ = idx;
            std::vector<uint8_t>().swap(p.codes);
        }
    }
    void take_from(avr_batch *b) {
        for (size_t i = 0; i < pending_.size(); i++) {
            const uint8_t *bytes; size_t len; int status;
            gpu_check(avr_batch_get(b, size_t(first_in_batch_) + i, &bytes, &len, &status));
            if (status != AVR_SLICE_OK) throw std::runtime_error("avr: slice status " + std::to_string(status));
            len = avr_drop_stop_byte(bytes, len);        // cabac_decoder::finish, recode.cpp:1508-1512
            blocks_[pending_[i].index].out_bytes.assign(reinterpret_cast<const char *>(bytes), len);
            blocks_[pending_[i].index].done = true;
        }
        pending_.clear();
    }
    std::string finish() {
        std::string out;
        for (auto &block : blocks_) {
            if (!block.done) throw std::runtime_error("Not all blocks were decoded.");
            if (block.length_parity != -1) {             // :1354-1361, through the C ABI's helper
                block.out_bytes.push_back('\0');
                const size_t n = avr_tail_patch(reinterpret_cast<uint8_t *>(&block.out_bytes[0]), block.out_bytes.size() - 1,
                                                block.length_parity, block.last_byte);
                block.out_bytes.resize(n);
            }
            out += block.out_bytes;
        }
        return out;
    }

    // The stream the decoder reads (decompressor::read_packet, recode.cpp:1366-1416): the blocks in order, a
    // literal as it is, a coded block as a surrogate payload of the recorded size, a skip_coded block as nothing
    // (its bytes are part of the next literal).  A block is classified when the reader first reaches it.
    int read_packet(uint8_t *buffer_out, int size) {
        int written = 0;
        while (written < size) {
            if (feed_at_ == feed_.size()) {              // the current block is used up: open the next one that has bytes
                if (read_index_ >= int(in_.block.size())) break;
                feed_ = open_block(read_index_++);
                feed_at_ = 0;
                continue;
            }
            const size_t n = std::min(feed_.size() - feed_at_, size_t(size - written));
            memcpy(buffer_out + written, feed_.data() + feed_at_, n);
            feed_at_ += n;
            written += int(n);
        }
        return written;
    }

  private:
    // what block `index` contributes to the stream; sets up its block_state
    std::string open_block(int index) {
        const Block &block = in_.block[size_t(index)];
        block_state &state = blocks_[size_t(index)];
        const int kinds = (block.has_literal ? 1 : 0) + (block.has_cabac ? 1 : 0) + (block.has_skip_coded ? 1 : 0);
        if (kinds != 1) throw std::runtime_error("Invalid input block: must have exactly one type");
        if (block.has_literal) {
            state.out_bytes = block.literal;
            state.done = true;
            return block.literal;
        }
        if (block.has_skip_coded) {
            if (!block.skip_coded) throw std::runtime_error("Unknown input block type");
            state.coded = true;
            state.done = true;
            return std::string();
        }
        if (!block.has_size) throw std::runtime_error("CABAC block requires size field.");
        state.coded = true;
        state.done = false;
        state.surrogate_marker = next_surrogate_marker(&surrogate_marker_sequence_number_);
        if (block.has_length_parity && block.has_last_byte && !block.last_byte.empty()) {
            state.length_parity = block.length_parity ? 1 : 0;
            state.last_byte = uint8_t(block.last_byte[0]);
        }
        return make_surrogate_block(state.surrogate_marker, size_t(block.size));
    }

  public:
    class cabac_decoder {                                // :1418-1527
      public:
        cabac_decoder(decompressor *d, const uint8_t *buf, int size) : d_(d) {
            index_ = d->recognize_coded_block(buf, size);
            const Block &block = d->in_.block[index_];
            if (block.has_cabac) {
                recorder_.reset(new decompress_recorder(&d->model_, reinterpret_cast<const uint8_t *>(block.cabac.data()),
                                                        block.cabac.size(), &d->contexts_));
            } else if (!(block.has_skip_coded && block.skip_coded)) {
                throw std::runtime_error("Expected CABAC block.");
            }
        }
        ~cabac_decoder() { if (recorder_) d_->pending_.push_back({index_, recorder_->codes()}); }
        bool hooked() const { return bool(recorder_); }
        int get(uint8_t *state) { return recorder_->get(state); }
        int get_bypass() { return recorder_->get_bypass(); }
        int get_terminate() { return recorder_->get_terminate(); }
        void begin_coding_type(CodingType ct, int zigzag_index, int param0, int param1) {   // :1483-1499
            if (recorder_) recorder_->begin_coding_type(ct, zigzag_index, param0, param1);
        }
        void end_coding_type(CodingType ct) { if (recorder_) recorder_->end_coding_type(ct); }   // :1500-1505

      private:
        decompressor *d_;
        int index_;
        std::unique_ptr<decompress_recorder> recorder_;
    };

    h264_model *get_model() { return &model_; }          // :1528-1530
    std::map<void *, std::unique_ptr<cabac_decoder>> cabac_contexts;

  private:
    // The decoder announces a slice payload (init_decoder): which block is it?  Coded blocks -- recoded or
    // skipped -- come up in the order the reader produced them, so it is the next coded one among the blocks the
    // reader has opened; its size must be the announced one, and a recoded block must begin with the marker its
    // surrogate was given (decompressor::recognize_coded_block, recode.cpp:1553-1580).
    int recognize_coded_block(const uint8_t *buf, int size) {
        int index = next_coded_block_;
        for (;; index++) {
            if (index >= read_index_) throw std::runtime_error("Coded block expected, but not recorded in the compressed data.");
            if (blocks_[size_t(index)].coded) break;
        }
        next_coded_block_ = index + 1;
        const Block &block = in_.block[size_t(index)];
        if (!block.has_cabac && !block.has_skip_coded) throw std::runtime_error("Internal error: expected coded block.");
        if (block.size != size)
            throw std::runtime_error(block.has_cabac ? "Invalid surrogate block size." : "Invalid skip_coded block size.");
        if (block.has_cabac) {
            const std::string &marker = blocks_[size_t(index)].surrogate_marker;
            if (size_t(size) < marker.size() || memcmp(buf, marker.data(), marker.size()) != 0)
                throw std::runtime_error("Invalid surrogate marker in coded block.");
        }
        return index;
    }

    // One K1 batch for the whole file, from resolved codes: the recorder has *state in hand at every bin
    // (it is what updates it), so the (symbol, *state) pair of cabac::encoder::put (cabac_code.h:33) is
    // shipped as one byte and the GPU skips working the states out again.
    void code_pending() {
        if (pending_.empty()) return;
        batch_holder bh(device_, pending_.size(), pending_bins() + 16 * pending_.size() + 64);
        add_to(bh.b);
        gpu_check(avr_batch_run(bh.b));
        take_from(bh.b);
    }

    struct pending { int index; std::vector<uint8_t> codes; };
    int first_in_batch_ = -1;
    int device_;
    Recoded in_;
    int read_index_ = 0;                                 // blocks the reader has opened
    std::string feed_;                                   // bytes of the block being read
    size_t feed_at_ = 0;
    std::vector<block_state> blocks_;
    uint64_t surrogate_marker_sequence_number_ = 1;      // :1592
    int next_coded_block_ = 0;
    context_ids contexts_;                               // one numbering of the state addresses for the whole file
    h264_model model_;
    std::vector<pending> pending_;
};

// recode.cpp:1601-1640: compress, decompress, compare.  `make_decoder` supplies a fresh stream
// decoder for each of the two passes (av_decoder is constructed twice there too).
template <class MakeDecoder>
int roundtrip(const std::string &original, MakeDecoder make_decoder, std::string *compressed_out, int device = 0) {
    compressor c(original, device);
    std::unique_ptr<stream_decoder> d1(make_decoder(&c, nullptr));
    const std::string compressed = c.run(d1.get());
    decompressor d(compressed, device);
    std::unique_ptr<stream_decoder> d2(make_decoder(nullptr, &d));
    const std::string decompressed = d.run(d2.get());
    if (compressed_out) *compressed_out = compressed;
    return original == decompressed ? 0 : 1;
}

}  // namespace host
}  // namespace avr
