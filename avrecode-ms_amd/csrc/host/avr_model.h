// h264_model: the adaptive model of the recoded stream (recode.cpp:625-1066), all of it: the
// estimators every bin passes through, and the significance-map side of it -- the frame store
// (framebuffer.h, block.h), the coordinates of the coefficient being coded, the model keys of
// significant_coeff_flag / last_significant_coeff_flag bins (recode.cpp:691-816), and the
// count of nonzero coefficients sent ahead of each block's map (finished_queueing, :851-947).
//
// Keys.  The reference keys its estimators on (address, int, int) (recode.cpp:325): the address of
// the libavcodec state byte for ordinary bins, and the addresses of a few members of the model for
// everything else.  Here the first component is a small integer: a slice's context i is key i
// (its offset in cabac_state[]), and the members are numbered from 1024 up.  Only identity matters
// -- the estimators live in a map (recode.cpp:1065) -- so the coded bytes are the same.
//
// Kept as found: the count bits are keyed on BlockMeta::is_8x8, which compress reads after
// end_coding_type has set it for the block (recode.cpp:958, then :1216) and decompress reads before
// (begin_coding_type, :1486); and a block whose coefficients are all nonzero does not fit the 2 / 4 / 6
// count bits (:868).  The reference's own round trip cannot hold for such blocks; the compressed
// bytes are the interface, so neither is "fixed" here.  (In-source notes say the pinned FFmpeg fork
// does not fire the coding-type hooks at all: recode.cpp:204, :210 "Not called".)
//
// Parity: the reference's model cannot be compiled here (it needs CodingType from the absent
// libavcodec/coding_hooks.h) and none of its tests reaches it, so this part is
// "parity unpinned": it follows recode.cpp as text and is tested for self-consistency
// (compress -> decompress round trips through both recorders, tests/test_host_model.py).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <map>
#include <stdexcept>
#include <tuple>
#include <vector>

#include "../../../include/avrecode_ms_amd.h"

namespace avr {
namespace host {

// Coding types the model distinguishes (the values recode.cpp uses, :685-691, :809; the full
// list lives in the absent libavcodec/coding_hooks.h).
enum CodingType {
    PIP_UNKNOWN = 0,
    PIP_UNREACHABLE,
    PIP_RESIDUALS,
    PIP_SIGNIFICANCE_MAP,
    PIP_SIGNIFICANCE_EOB,
    PIP_SIGNIFICANCE_NZ,
};

// Stand-ins for the addresses the reference keys on (recode.cpp:1056, :805, :812, :891)
constexpr int kKeyBypass = AVR_SEL_BYPASS, kKeyTerminate = AVR_SEL_TERMINATE;
constexpr int kKeySignificance = 1026;       // &significance_context
constexpr int kKeyEob = 1027;                // &fake_context
constexpr int kKeyNumNonzeroBit = 1028;      // &STATE_FOR_NUM_NONZERO_BIT[i], i = 0..5

typedef std::tuple<int, int, int> model_key;           // (what, int, int), recode.cpp:325

// ---------------------------------------------------------------------------------------------
// Where a coefficient lives: macroblock, 4x4 block in libavcodec's numbering (0..15 luma, 16..31
// Cb, 32..47 Cr in 8x8-quadrant order, 48..50 the three DC blocks), position in coding order.
struct CoefficientCoord {                               // recode.cpp:425-430
    int mb_x = 0, mb_y = 0, scan8_index = 0, zigzag_index = 0;
};

constexpr int kSubBlocks = 3 * (16 + 1);                // block.h:5

// The block left of / above 4x4 block `scan8_index` (< 48), and whether it lies in the neighbouring
// macroblock.  This is what the reference reads out of its reverse_scan_8 table (recode.cpp:292-320):
// within a colour plane the 16 blocks form a 4x4 grid, block b at column 2*b[2] + b[0], row
// 2*b[3] + b[1] (bits of b), and stepping off the grid lands on the far column / row of the
// neighbouring macroblock's same plane.
struct sub_mb_neighbor { int scan8_index; bool in_left_mb, in_up_mb; };
inline sub_mb_neighbor neighbor_block(int scan8_index, bool above) {
    const int plane = scan8_index >> 4, b = scan8_index & 15;
    int col = ((b >> 2) & 1) * 2 + (b & 1), row = ((b >> 3) & 1) * 2 + ((b >> 1) & 1);
    bool left_mb = false, up_mb = false;
    if (above) { if (row == 0) { row = 3; up_mb = true; } else row--; }
    else       { if (col == 0) { col = 3; left_mb = true; } else col--; }
    return {plane * 16 + (row >> 1) * 8 + (col >> 1) * 4 + (row & 1) * 2 + (col & 1), left_mb, up_mb};
}

// recode.cpp:432-480: the sub-macroblock whose nonzero count conditions this one's
inline bool get_neighbor_sub_mb(bool above, int sub_mb_size, const CoefficientCoord &in, CoefficientCoord *out) {
    *out = in;
    if (in.scan8_index >= 16 * 3) {                      // DC blocks: the same block of the neighbouring macroblock
        if (above) { if (in.mb_y <= 0) return false; out->mb_y = in.mb_y - 1; }
        else       { if (in.mb_x <= 0) return false; out->mb_x = in.mb_x - 1; }
        return true;
    }
    const sub_mb_neighbor nb = neighbor_block(in.scan8_index, above);
    if (nb.in_left_mb) { if (in.mb_x == 0) return false; out->mb_x = in.mb_x - 1; }
    if (nb.in_up_mb)   { if (in.mb_y == 0) return false; out->mb_y = in.mb_y - 1; }
    out->scan8_index = sub_mb_size >= 32 ? nb.scan8_index & ~3 : nb.scan8_index;   // 8x8: first of its four blocks
    return true;
}

// framebuffer.h + block.h, reduced to what the model reads and writes: per macroblock the 0/1
// significance of every coefficient, and per block the nonzero count.
class frame_store {
  public:
    struct mb_meta { bool is_8x8 = false, coded = false; uint8_t num_nonzeros[kSubBlocks] = {}; };   // block.h:21-23
    void init(uint32_t width, uint32_t height) {         // framebuffer.h:47-66
        width_ = width; height_ = height;
        residual_.assign(size_t(width) * height * kSubBlocks * 16, 0);
        meta_.assign(size_t(width) * height, mb_meta());
    }
    void bzero() {                                       // framebuffer.h:32-35
        std::fill(residual_.begin(), residual_.end(), 0);
        std::fill(meta_.begin(), meta_.end(), mb_meta());
    }
    void set_frame_num(int n) { frame_num_ = n; }
    bool is_same_frame(int n) const { return frame_num_ == n && width_ != 0 && height_ != 0; }
    uint32_t width() const { return width_; }
    uint32_t height() const { return height_; }
    uint16_t *residual_at(int x, int y) { return &residual_[index(x, y) * kSubBlocks * 16]; }
    const uint16_t *residual_at(int x, int y) const { return &residual_[index(x, y) * kSubBlocks * 16]; }
    mb_meta &meta_at(int x, int y) { return meta_[index(x, y)]; }
    const mb_meta &meta_at(int x, int y) const { return meta_[index(x, y)]; }

  private:
    size_t index(int x, int y) const {
        if (x < 0 || y < 0 || uint32_t(x) >= width_ || uint32_t(y) >= height_)
            throw std::out_of_range("h264_model: macroblock outside the frame (frame_spec / mb_xy not called?)");
        return size_t(x) + size_t(y) * width_;
    }
    std::vector<uint16_t> residual_;
    std::vector<mb_meta> meta_;
    uint32_t width_ = 0, height_ = 0;
    int frame_num_ = 0;
};

// ---------------------------------------------------------------------------------------------
class h264_model {
  public:
    CodingType coding_type = PIP_UNKNOWN;              // recode.cpp:627
    struct estimator { int pos = 1, neg = 1; };        // recode.cpp:1064
    frame_store frames[2];                             // :630-631
    int cur_frame = 0;
    CoefficientCoord mb_coord;                         // :1057-1062
    int nonzeros_observed = 0;
    int sub_mb_cat = -1, sub_mb_size = -1, sub_mb_is_dc = 0, sub_mb_chroma422 = 0;

    // recode.cpp:669-672: reset() forgets nothing that was learned
    void reset() {}

    // ---- keys (recode.cpp:683-822)
    model_key get_model_key(int context) const {
        switch (coding_type) {
            case PIP_SIGNIFICANCE_NZ:
            case PIP_UNKNOWN:
            case PIP_UNREACHABLE:
            case PIP_RESIDUALS:
                return model_key(context, 0, 0);
            case PIP_SIGNIFICANCE_MAP: {
                // Position class of the coefficient.  4x4 blocks: the position itself; 8x8 blocks and
                // 4:2:2 chroma DC: ctxIdxInc of significant_coeff_flag, ITU-T H.264 9.3.3.1.3
                // (Table 9-43, frame column; Min(numDecod / NumC8x8, 2)).
                static const uint8_t inc_8x8_frame[63] = {
                    0, 1, 2, 3, 4, 5, 5, 4, 4, 3, 3, 4, 4, 4, 5, 5, 4, 4, 4, 4, 3,
                    3, 6, 7, 7, 7, 8, 9, 10, 9, 8, 7, 7, 6, 11, 12, 13, 11, 6, 7, 8, 9,
                    14, 10, 9, 8, 6, 11, 12, 13, 11, 6, 9, 14, 10, 9, 11, 12, 13, 11, 14, 10, 12};
                // ctxIdxOffset + ctxBlockCatOffset of significant_coeff_flag (frame) per ctxBlockCat 0..13,
                // Tables 9-34 and 9-40
                static const int cat_base[14] = {105 + 0, 105 + 15, 105 + 29, 105 + 44, 105 + 47, 402, 484 + 0,
                                                 484 + 15, 484 + 29, 660, 528 + 0, 528 + 15, 528 + 29, 718};
                int position_class = mb_coord.zigzag_index;
                if (sub_mb_is_dc && sub_mb_chroma422) {
                    if (position_class >= 7) throw std::runtime_error("h264_model: 4:2:2 chroma DC position out of range");
                    position_class = std::min(position_class / 2, 2);
                } else if (sub_mb_size > 32) {
                    if (position_class >= 63) throw std::runtime_error("h264_model: 8x8 position out of range");
                    position_class = inc_8x8_frame[position_class];
                }
                if (sub_mb_cat < 0 || sub_mb_cat >= 14) throw std::runtime_error("h264_model: block category out of range");
                // (the reference also fetches the left / above / previous-frame coefficients here and
                // then leaves them out of the key, recode.cpp:714-803)
                const int num_nonzeros = current_meta().num_nonzeros[mb_coord.scan8_index];
                return model_key(kKeySignificance, 64 * num_nonzeros + nonzeros_observed,
                                 sub_mb_is_dc + position_class * 2 + 16 * 2 * cat_base[sub_mb_cat]);      // :805-807
            }
            case PIP_SIGNIFICANCE_EOB: {
                const int num_nonzeros = current_meta().num_nonzeros[mb_coord.scan8_index];
                return model_key(kKeyEob, num_nonzeros == nonzeros_observed, 0);                          // :809-816
            }
        }
        throw std::logic_error("h264_model: unreachable coding type");                                    // :820-821
    }

    estimator *lookup(const model_key &key) {
        if (std::get<1>(key) == 0 && std::get<2>(key) == 0 && unsigned(std::get<0>(key)) < 1026) return &flat_[std::get<0>(key)];
        return &estimators_[key];
    }
    uint64_t probability_for_model_key(uint64_t range, const model_key &key) {          // recode.cpp:823-827
        const estimator *e = lookup(key);
        const int total = e->pos + e->neg;
        return (range / uint64_t(total)) * uint64_t(e->pos);
    }
    uint64_t probability_for_state(uint64_t range, int context) {                      // recode.cpp:828-830
        return probability_for_model_key(range, get_model_key(context));
    }

    // ---- frame store (recode.cpp:831-850).  This function and update_state_tracking below follow the reference branch for
    // branch, on purpose: they ARE the model's state machine -- when a picture's store is cleared, which of significant_coeff_flag
    // and last_significant_coeff_flag a bin is, when the last coefficient is implied -- and the recoded bytes depend on every
    // branch of them, so there is one way to write them that decodes the reference's files.  What is this build's own is the data
    // they run on: integer keys, a flat estimator array, a frame store of plain vectors, neighbour geometry as bit arithmetic.
    void update_frame_spec(int frame_num, int mb_width, int mb_height) {
        const uint32_t w = uint32_t(mb_width), h = uint32_t(mb_height);
        if (frames[cur_frame].width() == w && frames[cur_frame].height() == h && frames[cur_frame].is_same_frame(frame_num)) return;
        cur_frame = !cur_frame;
        if (frames[cur_frame].width() != w || frames[cur_frame].height() != h) {
            frames[cur_frame].init(w, h);
            if (frames[!cur_frame].width() != w || frames[!cur_frame].height() != h) frames[!cur_frame].init(w, h);
        } else {
            frames[cur_frame].bzero();
        }
        frames[cur_frame].set_frame_num(frame_num);
    }

    // ---- the nonzero count of a block travels ahead of its significance map (recode.cpp:851-947):
    // put_or_get(key, &bit) codes (compress) or decodes (decompress) one bit of it, least significant
    // first; as many bits as the block size needs.
    template <class F>
    void finished_queueing(CodingType ct, F &&put_or_get) {
        if (ct != PIP_SIGNIFICANCE_MAP) return;
        const CodingType last = coding_type;
        coding_type = PIP_SIGNIFICANCE_NZ;
        frame_store::mb_meta &meta = current_meta();
        const int scan8 = mb_coord.scan8_index;
        int bits[6];
        for (int i = 0; i < 6; i++) bits[i] = (meta.num_nonzeros[scan8] >> i) & 1;
        const uint32_t n_bits = sub_mb_size > 16 ? 6 : sub_mb_size > 4 ? 4 : 2;
        CoefficientCoord nb;
        uint32_t left_nonzero = 0, above_nonzero = 0;
        const bool has_left = get_neighbor_sub_mb(false, sub_mb_size, mb_coord, &nb);
        if (has_left) left_nonzero = frames[cur_frame].meta_at(nb.mb_x, nb.mb_y).num_nonzeros[nb.scan8_index];
        if (get_neighbor_sub_mb(true, sub_mb_size, mb_coord, &nb))
            above_nonzero = frames[cur_frame].meta_at(nb.mb_x, nb.mb_y).num_nonzeros[nb.scan8_index];
        const uint32_t previous = frames[!cur_frame].meta_at(mb_coord.mb_x, mb_coord.mb_y).num_nonzeros[scan8];
        uint32_t so_far = 0;
        for (uint32_t i = 0; i < n_bits; i++) {
            const uint32_t bit = 1u << i;
            const int left_class = has_left ? int(left_nonzero >= bit) : 2;
            const int above_class = above_nonzero ? int(above_nonzero >= bit) : 2;   // sic: tests the count, not has_above (:884)
            put_or_get(model_key(kKeyNumNonzeroBit + int(i),
                                 int(so_far + 64 * (previous >= bit) + 128 * left_class + 384 * above_class),
                                 int(meta.is_8x8) + sub_mb_is_dc * 2 + sub_mb_chroma422 + sub_mb_cat * 4),
                       &bits[i]);
            if (bits[i]) so_far |= bit;
        }
        meta.num_nonzeros[scan8] = 0;
        for (int i = 0; i < 6; i++) meta.num_nonzeros[scan8] |= uint8_t(bits[i] << i);
        coding_type = last;
    }

    void end_coding_type(CodingType ct) {                // recode.cpp:948-967
        if (ct == PIP_SIGNIFICANCE_MAP) {
            const uint16_t *res = frames[cur_frame].residual_at(mb_coord.mb_x, mb_coord.mb_y) + mb_coord.scan8_index * 16;
            uint8_t num_nonzeros = 0;
            for (int i = 0; i < sub_mb_size; i++) num_nonzeros += res[i] != 0;
            frame_store::mb_meta &meta = current_meta();
            meta.is_8x8 = meta.is_8x8 || sub_mb_size > 32;
            meta.coded = true;
            meta.num_nonzeros[mb_coord.scan8_index] = num_nonzeros;
        }
        coding_type = PIP_UNKNOWN;
    }
    bool begin_coding_type(CodingType ct, int zigzag_index, int, int) {                 // recode.cpp:968-991
        coding_type = ct;
        if (ct != PIP_SIGNIFICANCE_MAP) return false;
        if (zigzag_index != 0) throw std::runtime_error("h264_model: a significance map starts at position 0");
        current_meta().num_nonzeros[mb_coord.scan8_index] = 0;
        nonzeros_observed = 0;
        mb_coord.zigzag_index = 0;
        return true;                                     // the caller queues the map's bins
    }
    void reset_mb_significance_state_tracking() {        // recode.cpp:992-996
        mb_coord.zigzag_index = 0;
        nonzeros_observed = 0;
        coding_type = PIP_SIGNIFICANCE_MAP;
    }
    // recode.cpp:997-1033: follow significant_coeff_flag / last_significant_coeff_flag through a block
    void update_state_tracking(int symbol) {
        switch (coding_type) {
            case PIP_SIGNIFICANCE_MAP: {
                uint16_t *res = frames[cur_frame].residual_at(mb_coord.mb_x, mb_coord.mb_y) + mb_coord.scan8_index * 16;
                res[mb_coord.zigzag_index] = uint16_t(symbol);
                nonzeros_observed += symbol;
                if (mb_coord.zigzag_index + 1 == sub_mb_size) {
                    coding_type = PIP_UNREACHABLE;
                    mb_coord.zigzag_index = 0;
                } else if (symbol) {
                    coding_type = PIP_SIGNIFICANCE_EOB;
                } else if (++mb_coord.zigzag_index + 1 == sub_mb_size) {
                    res[mb_coord.zigzag_index] = 1;      // no end of block so far: the last coefficient must be one
                    ++nonzeros_observed;
                    coding_type = PIP_UNREACHABLE;
                    mb_coord.zigzag_index = 0;
                }
                break;
            }
            case PIP_SIGNIFICANCE_EOB:
                if (symbol) {
                    mb_coord.zigzag_index = 0;
                    coding_type = PIP_UNREACHABLE;
                } else if (mb_coord.zigzag_index + 2 == sub_mb_size) {
                    frames[cur_frame].residual_at(mb_coord.mb_x, mb_coord.mb_y)[mb_coord.scan8_index * 16 + mb_coord.zigzag_index + 1] = 1;
                    coding_type = PIP_UNREACHABLE;
                } else {
                    coding_type = PIP_SIGNIFICANCE_MAP;
                    ++mb_coord.zigzag_index;
                }
                break;
            case PIP_SIGNIFICANCE_NZ:
            case PIP_RESIDUALS:
            case PIP_UNKNOWN:
                break;
            case PIP_UNREACHABLE:                        // assert(false) in the reference (:1029)
                throw std::runtime_error("h264_model: a bin after the end of a significance map");
        }
    }

    void update_state_for_model_key(int symbol, const model_key &key) {                 // recode.cpp:1037-1054
        estimator *e = lookup(key);
        if (symbol) e->pos++; else e->neg++;
        if ((coding_type != PIP_SIGNIFICANCE_MAP && e->pos + e->neg > 0x60) ||
            (coding_type == PIP_SIGNIFICANCE_MAP && e->pos + e->neg > 0x50)) {
            e->pos = (e->pos + 1) / 2;
            e->neg = (e->neg + 1) / 2;
        }
        update_state_tracking(symbol);
    }
    void update_state(int symbol, int context) { update_state_for_model_key(symbol, get_model_key(context)); }   // :1034-1036
    // BlockMeta::is_8x8 of the macroblock being coded (block.h:22), for compress_recorder::end_coding_type
    bool current_block_flag_8x8() const { return current_meta().is_8x8; }
    void set_current_block_flag_8x8(bool v) { current_meta().is_8x8 = v; }

  private:
    frame_store::mb_meta &current_meta() { return frames[cur_frame].meta_at(mb_coord.mb_x, mb_coord.mb_y); }
    const frame_store::mb_meta &current_meta() const { return frames[cur_frame].meta_at(mb_coord.mb_x, mb_coord.mb_y); }
    estimator flat_[1026];
    std::map<model_key, estimator> estimators_;        // recode.cpp:1065
};

}  // namespace host
}  // namespace avr
