// Bin-stream source without FFmpeg: what libavcodec-hooks does for the reference -- walk the
// file's NAL units, parse parameter sets and slice headers, and for every CABAC slice call the hook table
// (recode.cpp:219-235) once per bin while parsing the slice_data() syntax -- written from ITU-T H.264 (7.3 syntax,
// 9.3 CABAC parsing).  It is a SYNTAX parser: no prediction, no inverse transform, no picture buffer; it keeps of
// each macroblock only what the context-index derivations of 9.3.3.1.1 look at in its neighbours.
//
//   h264_stream_decoder : stream_decoder   the `av_decoder` of this build (recode.cpp:80-237): MP4 (avcC) or Annex B
//   slice_parser<Bins>                      slice_data() over a bin source: the hook table, or the build's own CABAC
//                                           decoding engine (used to check that a payload parses before the
//                                           compressor commits to it: payload_decodes())
//
// Supported: frame pictures (no field / MBAFF coding), I / P / B slices, one slice group, 4:2:0 / 4:2:2 / 4:4:4
// (not separate planes), 8x8 transform, cabac_init_idc 0 (verified on real streams) and 1 / 2 (tables unverified, see
// avr_h264_tables.h).  Not supported, by design: field / MBAFF pictures, slice groups, and I_PCM macroblocks -- an I_PCM sends
// raw samples through the decoder's skip_bytes hook, which the reference itself answers with an exception
// (recode.cpp:168-170, "CABAC decoder doesn't use skip_bytes"): where the reference gives up on the whole file, this build
// leaves the one slice literal.  Anything unsupported is reported as "not hooked": the slice's bytes stay in the literal
// stream of the container, exactly as the reference treats a slice whose payload it cannot find (recode.cpp:1146-1152);
// `recode probe` counts such slices per reason.
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <atomic>
#include <thread>
#include <map>
#include <string>
#include <vector>

#include "avr_h264_tables.h"
#include "avr_recode.h"

namespace avr {
namespace h264 {

struct unsupported : std::runtime_error { using std::runtime_error::runtime_error; };   // a stream feature outside the list above
struct bad_stream : std::runtime_error { using std::runtime_error::runtime_error; };    // the syntax does not add up

// ---------------------------------------------------------------------------------------------- bits
class bit_reader {                                       // RBSP bits, MSB first (7.2)
  public:
    bit_reader(const uint8_t *p, size_t n_bytes) : p_(p), n_(n_bytes * 8) {}
    size_t pos() const { return at_; }
    size_t size() const { return n_; }
    uint32_t u(int bits) {
        uint32_t v = 0;
        for (int i = 0; i < bits; i++) v = (v << 1) | bit();
        return v;
    }
    uint32_t ue() {                                      // 9.1; codes of 32 or more leading zeros do not fit 32 bits
        int zeros = 0;
        while (bit() == 0) if (++zeros > 31) throw bad_stream("Exp-Golomb code too long");
        return zeros ? ((1u << zeros) - 1u) + u(zeros) : 0u;
    }
    // ue(v) of a syntax element with a range: everything the parser turns into an int, an index or a size goes through here
    // (the reference leaves this to libavcodec, which checks every one of them)
    int ue_max(uint32_t max, const char *what) {
        const uint32_t v = ue();
        if (v > max) throw bad_stream(std::string(what) + " out of range");
        return int(v);
    }
    int32_t se() { const uint32_t k = ue(); return (k & 1u) ? int32_t((k + 1) / 2) : -int32_t(k / 2); }
    bool more_rbsp_data() const {                        // 7.2: anything before the last 1 bit
        size_t last = n_;
        while (last > at_ && !((p_[(last - 1) >> 3] >> (7 - ((last - 1) & 7))) & 1)) last--;
        return last > at_ + 1;
    }
    void align() { at_ = (at_ + 7) & ~size_t(7); }

  private:
    uint32_t bit() {
        if (at_ >= n_) throw bad_stream("read past the end of the NAL unit");
        const uint32_t b = (p_[at_ >> 3] >> (7 - (at_ & 7))) & 1u;
        at_++;
        return b;
    }
    const uint8_t *p_;
    size_t n_, at_ = 0;
};

inline std::vector<uint8_t> unescape(const uint8_t *p, size_t n) {    // 7.4.1: drop emulation_prevention_three_byte
    std::vector<uint8_t> out;
    out.reserve(n);
    int zeros = 0;
    for (size_t i = 0; i < n; i++) {
        if (zeros >= 2 && p[i] == 3) { zeros = 0; continue; }
        out.push_back(p[i]);
        zeros = p[i] == 0 ? zeros + 1 : 0;
    }
    return out;
}

// ---------------------------------------------------------------------------------------------- parameter sets
struct sps_t {
    bool valid = false;
    int profile_idc = 0, chroma_format_idc = 1, separate_colour_plane = 0, bit_depth_luma = 8, bit_depth_chroma = 8;
    int log2_max_frame_num = 4, poc_type = 0, log2_max_poc_lsb = 4, delta_pic_order_always_zero = 0;
    int frame_mbs_only = 1, mbaff = 0, direct_8x8_inference = 0, width_mbs = 0, height_map_units = 0;
};
struct pps_t {
    bool valid = false;
    int sps_id = 0, cabac = 0, bottom_field_pic_order_present = 0, slice_groups = 1, refs_l0 = 1, refs_l1 = 1;
    int weighted_pred = 0, weighted_bipred_idc = 0, pic_init_qp = 26, deblocking_control_present = 0;
    int redundant_pic_cnt_present = 0, transform_8x8_mode = 0;
};

inline void skip_scaling_list(bit_reader &r, int size) {             // 7.3.2.1.1.1
    int last = 8, next = 8;
    for (int j = 0; j < size; j++) {
        if (next != 0) next = (last + r.se() + 256) % 256;
        last = next == 0 ? last : next;
    }
}

// Level 6.2 allows 139 264 macroblocks a frame and 16 384 samples a side (Table A-1); twice that is this parser's limit.
constexpr int kMaxPicMbsSide = 2048, kMaxPicMbs = 1 << 18;

inline void parse_sps(const std::vector<uint8_t> &rbsp, sps_t sps[32]) {     // 7.3.2.1.1 (after the NAL header byte)
    bit_reader r(rbsp.data(), rbsp.size());
    sps_t s;
    s.profile_idc = int(r.u(8));
    r.u(8);
    r.u(8);
    const uint32_t id = r.ue();
    if (id > 31) throw bad_stream("seq_parameter_set_id out of range");
    const int p = s.profile_idc;
    if (p == 100 || p == 110 || p == 122 || p == 244 || p == 44 || p == 83 || p == 86 || p == 118 || p == 128 || p == 138 || p == 139 ||
        p == 134 || p == 135) {
        s.chroma_format_idc = r.ue_max(3, "chroma_format_idc");
        if (s.chroma_format_idc == 3) s.separate_colour_plane = int(r.u(1));
        s.bit_depth_luma = 8 + r.ue_max(6, "bit_depth_luma_minus8");
        s.bit_depth_chroma = 8 + r.ue_max(6, "bit_depth_chroma_minus8");
        r.u(1);                                          // qpprime_y_zero_transform_bypass_flag
        if (r.u(1))                                      // seq_scaling_matrix_present_flag
            for (int i = 0; i < (s.chroma_format_idc != 3 ? 8 : 12); i++)
                if (r.u(1)) skip_scaling_list(r, i < 6 ? 16 : 64);
    }
    s.log2_max_frame_num = 4 + r.ue_max(12, "log2_max_frame_num_minus4");
    s.poc_type = r.ue_max(2, "pic_order_cnt_type");
    if (s.poc_type == 0) s.log2_max_poc_lsb = 4 + r.ue_max(12, "log2_max_pic_order_cnt_lsb_minus4");
    else if (s.poc_type == 1) {
        s.delta_pic_order_always_zero = int(r.u(1));
        r.se();
        r.se();
        const uint32_t n = r.ue();
        if (n > 255) throw bad_stream("num_ref_frames_in_pic_order_cnt_cycle out of range");
        for (uint32_t i = 0; i < n; i++) r.se();
    }
    r.ue();                                              // max_num_ref_frames
    r.u(1);                                              // gaps_in_frame_num_value_allowed_flag
    s.width_mbs = 1 + r.ue_max(kMaxPicMbsSide - 1, "pic_width_in_mbs_minus1");
    s.height_map_units = 1 + r.ue_max(kMaxPicMbsSide - 1, "pic_height_in_map_units_minus1");
    if (int64_t(s.width_mbs) * s.height_map_units > kMaxPicMbs) throw bad_stream("picture size out of range");
    s.frame_mbs_only = int(r.u(1));
    if (!s.frame_mbs_only) s.mbaff = int(r.u(1));
    s.direct_8x8_inference = int(r.u(1));
    s.valid = true;
    sps[id] = s;
}

inline void parse_pps(const std::vector<uint8_t> &rbsp, pps_t pps[256]) {    // 7.3.2.2
    bit_reader r(rbsp.data(), rbsp.size());
    pps_t p;
    const uint32_t id = r.ue();
    if (id > 255) throw bad_stream("pic_parameter_set_id out of range");
    p.sps_id = r.ue_max(31, "seq_parameter_set_id");
    p.cabac = int(r.u(1));
    p.bottom_field_pic_order_present = int(r.u(1));
    p.slice_groups = 1 + r.ue_max(7, "num_slice_groups_minus1");
    if (p.slice_groups == 1) {                           // with slice groups the rest is not needed: such slices are not hooked
        p.refs_l0 = 1 + r.ue_max(31, "num_ref_idx_l0_default_active_minus1");
        p.refs_l1 = 1 + r.ue_max(31, "num_ref_idx_l1_default_active_minus1");
        p.weighted_pred = int(r.u(1));
        p.weighted_bipred_idc = int(r.u(2));
        p.pic_init_qp = 26 + r.se();
        if (p.pic_init_qp < -36 || p.pic_init_qp > 51) throw bad_stream("pic_init_qp_minus26 out of range");
        r.se();                                          // pic_init_qs_minus26
        r.se();                                          // chroma_qp_index_offset
        p.deblocking_control_present = int(r.u(1));
        r.u(1);                                          // constrained_intra_pred_flag
        p.redundant_pic_cnt_present = int(r.u(1));
        if (r.more_rbsp_data()) p.transform_8x8_mode = int(r.u(1));
    }
    p.valid = true;
    pps[id] = p;
}

// ---------------------------------------------------------------------------------------------- slice header
enum { SLICE_P = 0, SLICE_B = 1, SLICE_I = 2 };
struct slice_header {
    int first_mb = 0, type = SLICE_I, frame_num = 0, qp = 26, cabac_init_idc = 0, refs[2] = {0, 0};
    int width_mbs = 0, height_mbs = 0, chroma_array_type = 1, transform_8x8_mode = 0, direct_8x8_inference = 0;
    size_t data_offset = 0;                              // first byte of slice_data() in the RBSP
    // x264 before build 151 derived the coded_block_flag context of a 4:4:4 8x8 block next to a macroblock WITHOUT
    // the 8x8 transform as if that macroblock were unavailable (1 for an intra macroblock, 0 otherwise) where
    // 9.3.3.1.1.9 says 0.  Streams out of those builds are what they are; like libavcodec (which reads the build number
    // from x264's SEI and does the same) the parser follows the encoder.  Set by h264_stream_decoder.
    bool x264_old_444_cbf = false;
};

// 7.3.3; throws `unsupported` for what the parser below does not do
inline slice_header parse_slice_header(const std::vector<uint8_t> &rbsp, int nal_unit_type, int nal_ref_idc, const sps_t sps_tab[32],
                                       const pps_t pps_tab[256]) {
    bit_reader r(rbsp.data(), rbsp.size());
    slice_header h;
    h.first_mb = r.ue_max(uint32_t(kMaxPicMbs) - 1, "first_mb_in_slice");
    const uint32_t st = r.ue();
    if (st > 9) throw bad_stream("slice_type out of range");
    h.type = int(st % 5);
    if (h.type > SLICE_I) throw unsupported("SP / SI slice");
    const uint32_t pps_id = r.ue();
    if (pps_id > 255 || !pps_tab[pps_id].valid) throw bad_stream("slice refers to a picture parameter set that was not sent");
    const pps_t &pps = pps_tab[pps_id];
    if (!sps_tab[pps.sps_id].valid) throw bad_stream("picture parameter set refers to a sequence parameter set that was not sent");
    const sps_t &sps = sps_tab[pps.sps_id];
    if (!pps.cabac) throw unsupported("CAVLC slice");
    if (pps.slice_groups != 1) throw unsupported("slice groups");
    if (sps.separate_colour_plane) throw unsupported("separate colour planes");
    if (!sps.frame_mbs_only) throw unsupported("field / MBAFF coding");
    h.frame_num = int(r.u(sps.log2_max_frame_num));
    if (nal_unit_type == 5) r.ue();                      // idr_pic_id
    if (sps.poc_type == 0) {
        r.u(sps.log2_max_poc_lsb);
        if (pps.bottom_field_pic_order_present) r.se();
    } else if (sps.poc_type == 1 && !sps.delta_pic_order_always_zero) {
        r.se();
        if (pps.bottom_field_pic_order_present) r.se();
    }
    if (pps.redundant_pic_cnt_present) r.ue();
    if (h.type == SLICE_B) r.u(1);                       // direct_spatial_mv_pred_flag
    h.refs[0] = pps.refs_l0;
    h.refs[1] = pps.refs_l1;
    if (h.type != SLICE_I) {
        if (r.u(1)) {                                    // num_ref_idx_active_override_flag
            h.refs[0] = 1 + r.ue_max(31, "num_ref_idx_l0_active_minus1");
            if (h.type == SLICE_B) h.refs[1] = 1 + r.ue_max(31, "num_ref_idx_l1_active_minus1");
        }
        if (h.refs[0] > 32 || h.refs[1] > 32) throw bad_stream("num_ref_idx_active out of range");
        for (int list = 0; list < (h.type == SLICE_B ? 2 : 1); list++)      // ref_pic_list_modification(), 7.3.3.1
            if (r.u(1))
                for (;;) {
                    const uint32_t idc = r.ue();
                    if (idc == 3) break;
                    if (idc > 3) throw unsupported("ref_pic_list_modification of an extension");
                    r.ue();
                }
        if ((pps.weighted_pred && h.type == SLICE_P) || (pps.weighted_bipred_idc == 1 && h.type == SLICE_B)) {     // pred_weight_table(), 7.3.3.2
            r.ue();
            if (sps.chroma_format_idc != 0) r.ue();
            for (int list = 0; list < (h.type == SLICE_B ? 2 : 1); list++)
                for (int i = 0; i < h.refs[list]; i++) {
                    if (r.u(1)) { r.se(); r.se(); }
                    if (sps.chroma_format_idc != 0 && r.u(1)) { r.se(); r.se(); r.se(); r.se(); }
                }
        }
    }
    if (nal_ref_idc != 0) {                              // dec_ref_pic_marking(), 7.3.3.3
        if (nal_unit_type == 5) { r.u(1); r.u(1); }
        else if (r.u(1))
            for (;;) {
                const uint32_t op = r.ue();
                if (op == 0) break;
                if (op > 6) throw bad_stream("memory_management_control_operation out of range");
                if (op == 1 || op == 3) r.ue();
                if (op == 2) r.ue();
                if (op == 3 || op == 6) r.ue();
                if (op == 4) r.ue();
            }
    }
    if (h.type != SLICE_I) h.cabac_init_idc = r.ue_max(2, "cabac_init_idc");
    h.qp = pps.pic_init_qp + r.se();
    if (h.qp < -36 || h.qp > 51) throw bad_stream("slice_qp_delta out of range");
    if (pps.deblocking_control_present && r.ue() != 1) { r.se(); r.se(); }
    r.align();                                           // cabac_alignment_one_bit
    h.data_offset = r.pos() / 8;
    h.width_mbs = sps.width_mbs;
    h.height_mbs = sps.height_map_units;                 // frame_mbs_only
    h.chroma_array_type = sps.chroma_format_idc;
    h.transform_8x8_mode = pps.transform_8x8_mode;
    h.direct_8x8_inference = sps.direct_8x8_inference;
    if (int64_t(h.first_mb) >= int64_t(h.width_mbs) * h.height_mbs) throw bad_stream("first_mb_in_slice outside the picture");
    return h;
}

// ---------------------------------------------------------------------------------------------- bin sources
// the hook table of recode.cpp:219-235: every bin is a call into the driver's cabac_decoder
struct hook_bins {
    host::hooks *h;
    void *dec;
    int get(uint8_t *state) { return h->cabac.get(dec, state); }
    int bypass() { return h->cabac.get_bypass(dec); }
    int terminate() { return h->cabac.get_terminate(dec); }
};
// the build's own CABAC decoding engine (H.264 9.3.3.2) on the payload
struct engine_bins {
    host::cabac_bin_decoder d;
    engine_bins(const uint8_t *buf, size_t size) : d(buf, size) {}
    int get(uint8_t *state) { return d.get(state); }
    int bypass() { return d.get_bypass(); }
    int terminate() { return d.get_terminate(); }
};

// ---------------------------------------------------------------------------------------------- slice data
// What is kept of a macroblock for its right and lower neighbours (9.3.3.1.1.x).
struct mb_info {
    int slice = -1;                                      // the slice it was parsed in (-1: not yet): availability, 6.4.x
    uint8_t intra = 0, i_nxn = 0, skip = 0, direct = 0;  // direct: B_Skip or B_Direct_16x16
    uint8_t transform8x8 = 0, chroma_pred_mode = 0, cbp_luma = 0, cbp_chroma = 0, qp_delta_nonzero = 0;
    uint8_t dc_cbf[3] = {0, 0, 0};                       // coded_block_flag of the Intra16x16 luma DC / the Cb, Cr DC blocks
    uint8_t cbf[3][16];                                  // coded_block_flag per 4x4 block (y*4+x), per plane (chroma AC: its 2x2 / 2x4 grid)
    int8_t ref[2][4];                                    // refIdx > 0 test per 8x8 (y*2+x): -1 none, else value (direct: 0)
    uint8_t mvd[2][16][2];                               // |mvd| per 4x4 block, capped at 64
    mb_info() { memset(cbf, 0, sizeof cbf); memset(ref, 0, sizeof ref); memset(mvd, 0, sizeof mvd); }
};

struct model_hooks {                                     // the six model callbacks of recode.cpp:172-216
    host::hooks *h = nullptr;
    // frame_spec and mb_xy are the two the reference's fork does fire ("Called", recode.cpp:173, :177).  The other four are
    // annotated "Not called" there (:182, :190, :204, :210); `residual` makes this parser fire them around every residual
    // block -- begin_sub_mb before its coded_block_flag, PIP_SIGNIFICANCE_MAP around its significance map, end_sub_mb after
    // its levels -- which is what brings h264_model's significance-map keys and the recorders' queueing (recode.cpp:683-822,
    // 851-1033, 1201-1262, 1483-1505) to life on real streams (h264_stream_decoder::residual_hooks).
    bool residual = false;
    // The reference sends a block's nonzero count ahead of its map in 2, 4 or 6 bits for blocks of up to 4, 16 or more
    // coefficients (recode.cpp:865: serialized_bits) -- a count of 4, 16 or 64 does not fit, and its decompressor would then
    // infer the wrong end of block.  (The path is dead code there, "Not called"; nothing ever ran into it.)  This build keeps the
    // reference's widths and makes the parser refuse such a slice instead when it dry-runs a payload for the compressor
    // (h264_stream_decoder::payload_decodes): the slice stays a literal block, the file stays lossless.
    bool refuse_full_blocks = false;
    static int count_limit(int max_coeff) { return max_coeff > 16 ? 64 : max_coeff > 4 ? 16 : 4; }
    void frame_spec(int frame_num, int w, int hh) const { if (h && h->model.frame_spec) h->model.frame_spec(h->opaque, frame_num, w, hh); }
    void mb_xy(int x, int y) const { if (h && h->model.mb_xy) h->model.mb_xy(h->opaque, x, y); }
    void begin_sub_mb(int cat, int n, int max_coeff, int dc, int c422) const {
        if (residual && h && h->model.begin_sub_mb) h->model.begin_sub_mb(h->opaque, cat, n, max_coeff, dc, c422);
    }
    void end_sub_mb(int cat, int n, int max_coeff, int dc, int c422) const {
        if (residual && h && h->model.end_sub_mb) h->model.end_sub_mb(h->opaque, cat, n, max_coeff, dc, c422);
    }
    void begin_significance_map() const {
        if (residual && h && h->model.begin_coding_type) h->model.begin_coding_type(h->opaque, int(host::PIP_SIGNIFICANCE_MAP), 0, 0, 0);
    }
    void end_significance_map() const {
        if (residual && h && h->model.end_coding_type) h->model.end_coding_type(h->opaque, int(host::PIP_SIGNIFICANCE_MAP));
    }
};

template <class Bins>
class slice_parser {
  public:
    // states: the decoder's 1024 context variables (libavcodec's cabac_state[], 2 * pStateIdx + valMPS); mbs: one
    // mb_info per macroblock of the picture, kept across the slices of a picture; slice_no: a number unique per slice
    slice_parser(Bins &bins, const slice_header &h, uint8_t *states, std::vector<mb_info> &mbs, int slice_no, model_hooks model)
        : b_(bins), h_(h), st_(states), mbs_(mbs), slice_no_(slice_no), model_(model) {
        for (int c = 0; c < 1024; c++) st_[c] = init_state(c, h.type == SLICE_I, h.qp, h.cabac_init_idc);
        const int cat = h.chroma_array_type;
        chroma_w_ = cat == 3 ? 4 : cat == 0 ? 0 : 2;     // chroma plane size in 4x4 blocks
        chroma_h_ = cat == 3 ? 4 : cat == 2 ? 4 : cat == 0 ? 0 : 2;
    }

    // slice_data(), 7.3.4 (CABAC, no MBAFF).  Returns the number of macroblocks parsed; ends on end_of_slice_flag = 1.
    int run() {
        model_.frame_spec(h_.frame_num, h_.width_mbs, h_.height_mbs);
        const int total = h_.width_mbs * h_.height_mbs;
        int addr = h_.first_mb, count = 0;
        prev_qp_delta_nonzero_ = false;
        for (;;) {
            if (addr < 0 || addr >= total || size_t(addr) >= mbs_.size()) throw bad_stream("macroblock address past the end of the picture");
            mb_x_ = addr % h_.width_mbs;
            mb_y_ = addr / h_.width_mbs;
            cur_ = &mbs_[size_t(addr)];
            *cur_ = mb_info();
            left_ = mb_x_ > 0 && mbs_[size_t(addr - 1)].slice == slice_no_ ? &mbs_[size_t(addr - 1)] : nullptr;
            up_ = mb_y_ > 0 && mbs_[size_t(addr - h_.width_mbs)].slice == slice_no_ ? &mbs_[size_t(addr - h_.width_mbs)] : nullptr;
            model_.mb_xy(mb_x_, mb_y_);
            bool skipped = false;
            if (h_.type != SLICE_I) skipped = mb_skip_flag();
            if (skipped) {
                cur_->skip = 1;
                cur_->direct = h_.type == SLICE_B;
                prev_qp_delta_nonzero_ = false;
            } else {
                macroblock_layer();
            }
            cur_->slice = slice_no_;
            count++;
            if (b_.terminate()) return count;            // end_of_slice_flag
            addr++;
        }
    }

  private:
    int get(int ctx) { return b_.get(&st_[ctx]); }

    // ---- 9.3.3.1.1.1
    bool mb_skip_flag() {
        const int inc = (left_ && !left_->skip) + (up_ && !up_->skip);
        return get((h_.type == SLICE_P ? 11 : 24) + inc) != 0;
    }

    // ---- macroblock_layer(), 7.3.5
    void macroblock_layer() {
        int mb_type;                                     // as in Tables 7-11 / 7-13 / 7-14, intra types offset as there
        bool intra = false;
        int i_type = 0;                                  // Table 7-11 number when intra
        if (h_.type == SLICE_I) { intra = true; i_type = intra_mb_type(3, true); }
        else if (h_.type == SLICE_P) {
            if (get(14) == 0) {
                if (get(15) == 0) mb_type = 3 * get(16);             // P_L0_16x16 / P_8x8
                else mb_type = 2 - get(17);                          // P_L0_L0_8x16 / P_L0_L0_16x8
            } else { intra = true; i_type = intra_mb_type(17, false); mb_type = 5 + i_type; }
        } else {
            mb_type = b_mb_type();
            if (mb_type == 23) { intra = true; i_type = intra_mb_type(32, false); }
        }
        if (intra) { intra_macroblock(i_type); return; }
        (void)mb_type;
        inter_macroblock(mb_type);
    }

    // mb_type of an intra macroblock (Table 9-36 binarisation): prefix contexts start at `base`
    int intra_mb_type(int base, bool i_slice) {
        int s = base;
        if (i_slice) {
            const int inc = (left_ && !left_->i_nxn) + (up_ && !up_->i_nxn);      // mb_type of the neighbour is not I_NxN (9.3.3.1.1.3)
            if (get(base + inc) == 0) return 0;          // I_NxN
            s = base + 2;
        } else if (get(base) == 0) return 0;
        if (b_.terminate()) throw unsupported("I_PCM macroblock");
        int t = 1;
        t += 12 * get(s + 1);                            // coded_block_pattern luma != 0
        if (get(s + 2)) t += 4 + 4 * get(s + 2 + (i_slice ? 1 : 0));             // chroma 1 / 2
        t += 2 * get(s + 3 + (i_slice ? 1 : 0));
        t += get(s + 3 + (i_slice ? 2 : 0));
        return t;                                        // 1..24: I_16x16_<pred>_<chroma>_<luma>
    }

    int b_mb_type() {                                    // Table 9-37 (B slices), contexts 27..35
        const int inc = (left_ && !left_->direct) + (up_ && !up_->direct);
        if (!get(27 + inc)) return 0;                    // B_Direct_16x16
        if (!get(27 + 3)) return 1 + get(27 + 5);        // B_L0_16x16, B_L1_16x16
        int bits = get(27 + 4) << 3;
        bits |= get(27 + 5) << 2;
        bits |= get(27 + 5) << 1;
        bits |= get(27 + 5);
        if (bits < 8) return bits + 3;
        if (bits == 13) return 23;                       // intra
        if (bits == 14) return 11;                       // B_L1_L0_8x16
        if (bits == 15) return 22;                       // B_8x8
        bits = (bits << 1) | get(27 + 5);
        return bits - 4;
    }

    // ---- intra
    void intra_macroblock(int i_type) {
        cur_->intra = 1;
        const bool nxn = i_type == 0;
        cur_->i_nxn = nxn;
        if (nxn) {
            if (h_.transform_8x8_mode) cur_->transform8x8 = uint8_t(transform_size_8x8_flag());
            const int n = cur_->transform8x8 ? 4 : 16;
            for (int i = 0; i < n; i++)                  // prev_intra_pred_mode_flag, rem_intra_pred_mode
                if (!get(68)) { get(69); get(69); get(69); }
        }
        if (h_.chroma_array_type == 1 || h_.chroma_array_type == 2) intra_chroma_pred_mode();
        int cbp_luma, cbp_chroma;
        if (nxn) coded_block_pattern(&cbp_luma, &cbp_chroma);
        else {
            cbp_luma = i_type > 12 ? 15 : 0;
            cbp_chroma = ((i_type - 1) / 4) % 3;
        }
        cur_->cbp_luma = uint8_t(cbp_luma);
        cur_->cbp_chroma = uint8_t(cbp_chroma);
        if (cbp_luma || cbp_chroma || !nxn) {
            mb_qp_delta();
            residual(!nxn);
        } else prev_qp_delta_nonzero_ = false;
    }

    int transform_size_8x8_flag() { return get(399 + (left_ ? left_->transform8x8 : 0) + (up_ ? up_->transform8x8 : 0)); }

    void intra_chroma_pred_mode() {                      // 9.3.3.1.1.8
        const int inc = (left_ && left_->intra && left_->chroma_pred_mode != 0) + (up_ && up_->intra && up_->chroma_pred_mode != 0);
        int mode = 0;
        if (get(64 + inc)) { mode = 1; if (get(64 + 3)) { mode = 2; if (get(64 + 3)) mode = 3; } }
        cur_->chroma_pred_mode = uint8_t(mode);
    }

    // ---- coded_block_pattern, 9.3.3.1.1.4: condTermFlagN = 0 when N is unavailable, I_PCM, or has the bit set
    void coded_block_pattern(int *luma, int *chroma) {
        auto bit_a = [&](int b8) { return left_ ? (left_->skip ? 0 : (left_->cbp_luma >> b8) & 1) : 1; };     // 1 = "set" = condTerm 0
        auto bit_b = [&](int b8) { return up_ ? (up_->skip ? 0 : (up_->cbp_luma >> b8) & 1) : 1; };
        int cbp = 0;
        cbp |= get(73 + !bit_a(1) + 2 * !bit_b(2));
        cbp |= get(73 + !(cbp & 1) + 2 * !bit_b(3)) << 1;
        cbp |= get(73 + !bit_a(3) + 2 * !(cbp & 1)) << 2;
        cbp |= get(73 + !((cbp >> 2) & 1) + 2 * !((cbp >> 1) & 1)) << 3;
        *luma = cbp;
        *chroma = 0;
        if (h_.chroma_array_type == 1 || h_.chroma_array_type == 2) {
            const int ca = left_ && !left_->skip ? left_->cbp_chroma : 0, cb = up_ && !up_->skip ? up_->cbp_chroma : 0;
            if (get(77 + (ca > 0) + 2 * (cb > 0))) *chroma = 1 + get(77 + 4 + (ca == 2) + 2 * (cb == 2));
        }
    }

    void mb_qp_delta() {                                 // 9.3.3.1.1.5, unary (Table 9-34: ctxIdxOffset 60)
        int ctx = 60 + (prev_qp_delta_nonzero_ ? 1 : 0), n = 0;
        while (get(ctx)) {
            ctx = 60 + 2 + (n > 0);
            if (++n > 200) throw bad_stream("mb_qp_delta too long");
        }
        prev_qp_delta_nonzero_ = n != 0;
        cur_->qp_delta_nonzero = n != 0;
    }

    // ---- inter
    static int p_partitions(int mb_type) { return mb_type == 0 ? 1 : mb_type == 3 ? 4 : 2; }

    void inter_macroblock(int mb_type) {
        // per 8x8 quadrant: which lists it predicts from (bit 0: L0, bit 1: L1), 0 = direct; sub-partition shape
        int pred[4] = {0, 0, 0, 0}, sub_shape[4] = {0, 0, 0, 0};     // shape: 0 8x8, 1 8x4, 2 4x8, 3 4x4
        int part_shape = 0;                                          // 0 16x16, 1 16x8, 2 8x16, 3 8x8
        bool direct16 = false, sub_less_than_8x8 = false, any_direct_sub = false;
        if (h_.type == SLICE_P) {
            part_shape = mb_type;                                    // 0, 1 (16x8), 2 (8x16), 3 (8x8)
            for (int q = 0; q < 4; q++) pred[q] = 1;
            if (mb_type == 3)
                for (int q = 0; q < 4; q++) {
                    int t;                                           // sub_mb_type, contexts 21..23
                    if (get(21)) t = 0;
                    else if (!get(22)) t = 1;
                    else t = get(23) ? 2 : 3;
                    sub_shape[q] = t;
                    if (t) sub_less_than_8x8 = true;
                }
        } else {
            static const uint8_t b_shape[23] = {0, 0, 0, 0, 1, 2, 1, 2, 1, 2, 1, 2, 1, 2, 1, 2, 1, 2, 1, 2, 1, 2, 3};
            // prediction lists of the two partitions of B mb_type 4..21 (Table 7-14): bit 0 L0, bit 1 L1
            static const uint8_t b_pred[23][2] = {{0, 0}, {1, 1}, {2, 2}, {3, 3}, {1, 1}, {1, 1}, {2, 2}, {2, 2}, {1, 2}, {1, 2}, {2, 1}, {2, 1},
                                                  {1, 3}, {1, 3}, {2, 3}, {2, 3}, {3, 1}, {3, 1}, {3, 2}, {3, 2}, {3, 3}, {3, 3}, {0, 0}};
            part_shape = b_shape[mb_type];
            if (mb_type == 0) direct16 = true;
            else if (mb_type == 22) {
                for (int q = 0; q < 4; q++) {
                    const int t = b_sub_mb_type();                   // Table 7-18
                    static const uint8_t sp[13] = {0, 1, 2, 3, 1, 1, 2, 2, 3, 3, 1, 2, 3}, ss[13] = {0, 0, 0, 0, 1, 2, 1, 2, 1, 2, 3, 3, 3};
                    pred[q] = sp[t];
                    sub_shape[q] = ss[t];
                    if (t == 0) any_direct_sub = true;
                    else if (ss[t]) sub_less_than_8x8 = true;
                }
            } else if (part_shape == 0) for (int q = 0; q < 4; q++) pred[q] = b_pred[mb_type][0];
            else if (part_shape == 1) { pred[0] = pred[1] = b_pred[mb_type][0]; pred[2] = pred[3] = b_pred[mb_type][1]; }
            else { pred[0] = pred[2] = b_pred[mb_type][0]; pred[1] = pred[3] = b_pred[mb_type][1]; }
        }
        cur_->direct = direct16;
        // ref_idx_l0, ref_idx_l1 (all of one list first), then mvd_l0, mvd_l1: 7.3.5.1 / 7.3.5.2
        for (int list = 0; list < 2; list++)
            for (int q = 0; q < 4; q++) cur_->ref[list][q] = (pred[q] >> list) & 1 ? 0 : -1;
        if (!direct16) {
            for (int list = 0; list < (h_.type == SLICE_B ? 2 : 1); list++) {
                if (h_.refs[list] <= 1) continue;
                for (int q = 0; q < 4; q++) {
                    if (!partition_leader(part_shape, q) || !((pred[q] >> list) & 1)) continue;
                    const int r = ref_idx(list, q);
                    for (int k = 0; k < 4; k++) if (same_partition(part_shape, q, k) && ((pred[k] >> list) & 1)) cur_->ref[list][k] = int8_t(r);
                }
            }
            for (int list = 0; list < (h_.type == SLICE_B ? 2 : 1); list++)
                for (int q = 0; q < 4; q++) {
                    if (!((pred[q] >> list) & 1)) continue;
                    if (part_shape != 3) {
                        if (!partition_leader(part_shape, q)) continue;
                        const int w = part_shape == 2 ? 2 : 4, hh = part_shape == 1 ? 2 : 4;          // in 4x4 blocks
                        mvd_block(list, (q & 1) * 2, (q >> 1) * 2, w, hh);
                    } else {
                        const int x0 = (q & 1) * 2, y0 = (q >> 1) * 2;
                        switch (sub_shape[q]) {
                            case 0: mvd_block(list, x0, y0, 2, 2); break;
                            case 1: mvd_block(list, x0, y0, 2, 1); mvd_block(list, x0, y0 + 1, 2, 1); break;
                            case 2: mvd_block(list, x0, y0, 1, 2); mvd_block(list, x0 + 1, y0, 1, 2); break;
                            default: for (int k = 0; k < 4; k++) mvd_block(list, x0 + (k & 1), y0 + (k >> 1), 1, 1);
                        }
                    }
                }
        }
        int cbp_luma, cbp_chroma;
        coded_block_pattern(&cbp_luma, &cbp_chroma);
        cur_->cbp_luma = uint8_t(cbp_luma);
        cur_->cbp_chroma = uint8_t(cbp_chroma);
        if (cbp_luma && h_.transform_8x8_mode && !sub_less_than_8x8 && ((!direct16 && !any_direct_sub) || h_.direct_8x8_inference))
            cur_->transform8x8 = uint8_t(transform_size_8x8_flag());
        if (cbp_luma || cbp_chroma) {
            mb_qp_delta();
            residual(false);
        } else prev_qp_delta_nonzero_ = false;
    }

    static bool partition_leader(int shape, int q) { return shape == 0 ? q == 0 : shape == 1 ? (q == 0 || q == 2) : shape == 2 ? (q == 0 || q == 1) : true; }
    static bool same_partition(int shape, int q, int k) {
        return shape == 0 ? true : shape == 1 ? (q >> 1) == (k >> 1) : shape == 2 ? (q & 1) == (k & 1) : q == k;
    }

    int b_sub_mb_type() {                                // contexts 36..39
        if (!get(36)) return 0;
        if (!get(37)) return 1 + get(39);
        int t = 3;
        if (get(38)) {
            if (get(39)) return 11 + get(39);
            t += 4;
        }
        t += 2 * get(39);
        t += get(39);
        return t;
    }

    // ref_idx of the partition whose first quadrant is q (9.3.3.1.1.6): refIdxZeroFlag / predFlag of the blocks left of and above it
    int ref_idx(int list, int q) {
        auto greater0 = [&](const mb_info *m, int qq) { return m && !m->skip && !m->intra && !m->direct && m->ref[list][qq] > 0; };
        const int x = q & 1, y = q >> 1;
        const bool a = x ? cur_->ref[list][q - 1] > 0 : greater0(left_, y * 2 + 1);      // a direct quadrant has ref -1
        const bool b = y ? cur_->ref[list][q - 2] > 0 : greater0(up_, 2 + x);
        int ctx = (a ? 1 : 0) + (b ? 2 : 0), ref = 0;
        while (get(54 + ctx)) {
            ref++;
            ctx = (ctx >> 2) + 4;
            if (ref >= 32) throw bad_stream("ref_idx out of range");
        }
        return ref;
    }

    // mvd of one partition covering w x h 4x4 blocks at (x, y) (9.3.3.1.1.7): absMvdComp of the blocks left of and above its corner
    void mvd_block(int list, int x, int y, int w, int hh) {
        int amvd[2];
        for (int comp = 0; comp < 2; comp++) {
            const int a = x ? cur_->mvd[list][y * 4 + x - 1][comp] : left_ ? left_->mvd[list][y * 4 + 3][comp] : 0;
            const int bb = y ? cur_->mvd[list][(y - 1) * 4 + x][comp] : up_ ? up_->mvd[list][12 + x][comp] : 0;
            const int sum = a + bb, base = comp ? 47 : 40;
            int v = 0;
            if (get(base + (sum > 2) + (sum > 32))) {
                v = 1;
                int ctx = 3;
                while (v < 9 && get(base + ctx)) { if (ctx < 6) ctx++; v++; }
                if (v >= 9) {                            // UEG3 suffix, bypass
                    int k = 3;
                    while (b_.bypass()) { v += 1 << k; if (++k > 24) throw bad_stream("mvd too long"); }
                    while (k--) v += b_.bypass() << k;
                }
                b_.bypass();                             // sign
            }
            amvd[comp] = v > 64 ? 64 : v;
        }
        for (int j = 0; j < hh; j++)
            for (int i = 0; i < w; i++) { cur_->mvd[list][(y + j) * 4 + x + i][0] = uint8_t(amvd[0]); cur_->mvd[list][(y + j) * 4 + x + i][1] = uint8_t(amvd[1]); }
    }

    // ---- residual(), 7.3.5.3
    void residual(bool i16x16) {
        residual_plane(0, i16x16);                                       // residual_luma
        const int cat = h_.chroma_array_type;
        if (cat == 1 || cat == 2) {
            const int n_dc = 4 * (cat == 2 ? 2 : 1);
            if (cur_->cbp_chroma & 3)
                for (int c = 0; c < 2; c++) cur_->dc_cbf[c + 1] = uint8_t(block(3, c + 1, -1, true, n_dc, 49 + c));      // CHROMA_DC_BLOCK_INDEX + c
            if (cur_->cbp_chroma & 2)
                for (int c = 0; c < 2; c++)
                    for (int blk = 0; blk < n_dc; blk++) {
                        const int x = blk & 1, y = blk >> 1;
                        cur_->cbf[c + 1][y * 4 + x] = uint8_t(block(4, c + 1, y * 4 + x, false, 15, 16 * (c + 1) + blk));
                    }
        } else if (cat == 3) {
            residual_plane(1, i16x16);                                   // Cb, Cr coded like luma with their own categories
            residual_plane(2, i16x16);
        }
    }

    // luma, or a 4:4:4 chroma plane: DC / AC of Intra16x16, 4x4 or 8x8 blocks per 8x8 quadrant with its cbp bit
    void residual_plane(int plane, bool i16x16) {
        const int cat_dc = plane == 0 ? 0 : plane == 1 ? 6 : 10, cat_ac = cat_dc + 1, cat_4x4 = cat_dc + 2, cat_8x8 = plane == 0 ? 5 : plane == 1 ? 9 : 13;
        if (i16x16) cur_->dc_cbf[plane] = uint8_t(block(cat_dc, plane, -1, true, 16, 48 + plane));       // LUMA_DC_BLOCK_INDEX + plane
        for (int q = 0; q < 4; q++) {
            if (!((cur_->cbp_luma >> q) & 1)) continue;
            const int x0 = (q & 1) * 2, y0 = (q >> 1) * 2;
            if (cur_->transform8x8) {
                // the coded_block_flag of an 8x8 block is only sent with 4:4:4 (7.3.5.3.3); otherwise it is inferred 1
                const int cbf = block(cat_8x8, plane, y0 * 4 + x0, false, 64, 16 * plane + 4 * q, h_.chroma_array_type == 3);
                for (int k = 0; k < 4; k++) cur_->cbf[plane][(y0 + (k >> 1)) * 4 + x0 + (k & 1)] = uint8_t(cbf);
            } else {
                for (int k = 0; k < 4; k++) {
                    const int x = x0 + (k & 1), y = y0 + (k >> 1);
                    cur_->cbf[plane][y * 4 + x] = uint8_t(block(i16x16 ? cat_ac : cat_4x4, plane, y * 4 + x, false, i16x16 ? 15 : 16, 16 * plane + 4 * q + k));
                }
            }
        }
    }

    // One residual block: its coded_block_flag (inferred 1 when `sent` is false), then residual_block_cabac().  `n` is the block's
    // number the way libavcodec counts them for its decode_cabac_residual calls -- 4 * (8x8 quadrant) + (4x4 inside it) for
    // luma, + 16 per colour plane, 48 + plane for an Intra16x16 DC block, 49 + c for a chroma DC block -- which is what the
    // model hooks are given as `scan8index` (recode.cpp:182) and what h264_model keeps the nonzero counts by.
    int block(int cat, int plane, int blk, bool dc, int max_coeff, int n, bool sent = true) {
        const int c422 = dc && cat == 3 && h_.chroma_array_type == 2;
        model_.begin_sub_mb(cat, n, max_coeff, dc, c422);
        const int cbf = sent ? coded_block_flag(cat, plane, blk, dc) : 1;
        if (cbf) coefficients(cat, max_coeff);
        model_.end_sub_mb(cat, n, max_coeff, dc, c422);
        return cbf;
    }

    // 9.3.3.1.1.9: transBlockN's coded_block_flag, with the rules for N unavailable / skipped / without the cbp bit
    int coded_block_flag(int cat, int plane, int blk, bool dc) {
        const bool is8x8 = cat == 5 || cat == 9 || cat == 13;
        auto of = [&](const mb_info *m, int b) -> int {
            if (!m) return cur_->intra ? 1 : 0;                          // not available: 1 for an intra macroblock, else 0
            if (is8x8 && !m->transform8x8)                               // transBlockN of an 8x8 block is an 8x8 block, or not available
                return h_.x264_old_444_cbf && cur_->intra ? 1 : 0;       //   (old x264: as if the macroblock -- skipped ones too -- were not there)
            if (m->skip) return 0;
            if (dc) return m->dc_cbf[plane];
            return m->cbf[plane][b];
        };
        int a, bb;
        if (dc) { a = of(left_, 0); bb = of(up_, 0); }
        else {
            const int x = blk & 3, y = blk >> 2;
            const bool chroma_grid = plane > 0 && h_.chroma_array_type != 3;
            const int w = chroma_grid ? chroma_w_ : 4, hh = chroma_grid ? chroma_h_ : 4;
            a = x ? cur_->cbf[plane][y * 4 + x - 1] : of(left_, y * 4 + w - 1);
            bb = y ? cur_->cbf[plane][(y - 1) * 4 + x] : of(up_, (hh - 1) * 4 + x);
        }
        return get(kCbfBase[cat] + a + 2 * bb);
    }

    // residual_block_cabac() after the coded_block_flag: significance map, levels, signs (7.3.5.3.3, 9.3.3.1.3)
    void coefficients(int cat, int max_coeff) {
        const bool is8x8 = cat == 5 || cat == 9 || cat == 13;
        const int sig = kSigBase[cat], last = kLastBase[cat], abs_base = kAbsBase[cat];
        int n = 0;
        int i = 0;
        model_.begin_significance_map();
        for (; i < max_coeff - 1; i++) {
            int inc_s, inc_l;
            if (is8x8) { inc_s = kSig8x8[i]; inc_l = kLast8x8[i]; }
            else if (cat == 3) { const int c8 = h_.chroma_array_type == 2 ? 2 : 1; inc_s = inc_l = (i / c8) < 2 ? i / c8 : 2; }
            else inc_s = inc_l = i;
            if (get(sig + inc_s)) {
                n++;
                if (get(last + inc_l)) break;
            }
        }
        if (i == max_coeff - 1) n++;                                     // the last coefficient is significant by inference
        if (model_.refuse_full_blocks && n >= model_hooks::count_limit(max_coeff))
            throw unsupported("a block with as many nonzero coefficients as the reference's count field cannot hold (recode.cpp:865)");
        model_.end_significance_map();
        int gt1 = 0, eq1 = 0;
        for (int k = 0; k < n; k++) {                                    // coeff_abs_level_minus1 (prefix TU 14, suffix EG0), then the sign
            const int ctx0 = abs_base + (gt1 ? 0 : (1 + eq1 < 4 ? 1 + eq1 : 4));
            if (!get(ctx0)) eq1++;
            else {
                const int cap = 4 - (cat == 3 ? 1 : 0);
                const int ctx1 = abs_base + 5 + (gt1 < cap ? gt1 : cap);
                int v = 1;
                while (v < 14 && get(ctx1)) v++;
                if (v >= 14) {
                    int kk = 0;
                    while (b_.bypass()) { if (++kk > 24) throw bad_stream("coeff_abs_level too long"); }
                    while (kk--) b_.bypass();
                }
                gt1++;
            }
            b_.bypass();                                                 // coeff_sign_flag
        }
    }

    Bins &b_;
    const slice_header &h_;
    uint8_t *st_;
    std::vector<mb_info> &mbs_;
    int slice_no_;
    model_hooks model_;
    int chroma_w_ = 2, chroma_h_ = 2;
    int mb_x_ = 0, mb_y_ = 0;
    mb_info *cur_ = nullptr;
    const mb_info *left_ = nullptr, *up_ = nullptr;
    bool prev_qp_delta_nonzero_ = false;
};

// ---------------------------------------------------------------------------------------------- the file
struct nal_ref { size_t offset, size; };                 // a NAL unit's bytes (header byte first) inside the stream

inline uint32_t be32(const uint8_t *p) { return uint32_t(p[0]) << 24 | uint32_t(p[1]) << 16 | uint32_t(p[2]) << 8 | p[3]; }

// Annex B: NAL units between start codes
inline std::vector<nal_ref> annexb_nals(const std::vector<uint8_t> &d) {
    std::vector<nal_ref> out;
    size_t i = 0, n = d.size(), start = size_t(-1);
    while (i + 3 <= n) {
        if (d[i] == 0 && d[i + 1] == 0 && d[i + 2] == 1) {
            if (start != size_t(-1)) { size_t e = i; while (e > start && d[e - 1] == 0) e--; out.push_back({start, e - start}); }
            start = i + 3;
            i += 3;
        } else i++;
    }
    if (start != size_t(-1) && start < n) { size_t e = n; while (e > start && d[e - 1] == 0) e--; out.push_back({start, e - start}); }
    return out;
}

// MP4: the first avc1 track's parameter sets (avcC) and samples (stsz / stsc / stco|co64), as NAL units in decode order
inline bool mp4_nals(const std::vector<uint8_t> &d, std::vector<nal_ref> *out) {
    struct box { size_t at, body, end; uint32_t type; };
    auto children = [&](size_t from, size_t to) {
        std::vector<box> v;
        size_t p = from;
        while (p + 8 <= to) {
            uint64_t size = be32(&d[p]);
            const uint32_t type = be32(&d[p + 4]);
            size_t hdr = 8;
            if (size == 1) { if (p + 16 > to) break; size = uint64_t(be32(&d[p + 8])) << 32 | be32(&d[p + 12]); hdr = 16; }
            else if (size == 0) size = to - p;
            if (size < hdr || p + size > to) break;
            v.push_back({p, p + hdr, p + size_t(size), type});
            p += size_t(size);
        }
        return v;
    };
    auto fourcc = [](const char *s) { return uint32_t(uint8_t(s[0])) << 24 | uint32_t(uint8_t(s[1])) << 16 | uint32_t(uint8_t(s[2])) << 8 | uint8_t(s[3]); };
    auto find = [&](const std::vector<box> &v, const char *t) -> const box * { for (const box &b : v) if (b.type == fourcc(t)) return &b; return nullptr; };
    const std::vector<box> top = children(0, d.size());
    if (!find(top, "ftyp") && !find(top, "moov")) return false;
    const box *moov = find(top, "moov");
    if (!moov) throw bad_stream("MP4 without a moov box");
    for (const box &trak : children(moov->body, moov->end)) {
        if (trak.type != fourcc("trak")) continue;
        const std::vector<box> tk = children(trak.body, trak.end);
        const box *mdia = find(tk, "mdia");
        if (!mdia) continue;
        const std::vector<box> md = children(mdia->body, mdia->end);
        const box *minf = find(md, "minf");
        if (!minf) continue;
        const std::vector<box> mi = children(minf->body, minf->end);
        const box *stbl = find(mi, "stbl");
        if (!stbl) continue;
        const std::vector<box> sb = children(stbl->body, stbl->end);
        const box *stsd = find(sb, "stsd"), *stsz = find(sb, "stsz"), *stsc = find(sb, "stsc"), *stco = find(sb, "stco"), *co64 = find(sb, "co64");
        if (!stsd || !stsz || !stsc || (!stco && !co64)) continue;
        // sample entry: avc1 with an avcC child (after 8 bytes of stsd header and the 78-byte visual sample entry)
        const std::vector<box> entries = children(stsd->body + 8, stsd->end);
        const box *avc1 = find(entries, "avc1");
        if (!avc1 || avc1->body + 78 > avc1->end) continue;
        const box *avcc = nullptr;
        const std::vector<box> in_entry = children(avc1->body + 78, avc1->end);
        avcc = find(in_entry, "avcC");
        if (!avcc || avcc->body + 7 > avcc->end) continue;
        const size_t length_size = size_t(d[avcc->body + 4] & 3) + 1;
        size_t p = avcc->body + 5;
        for (int pass = 0; pass < 2; pass++) {           // SPS then PPS
            const int count = pass == 0 ? d[p] & 31 : d[p];
            p++;
            for (int i = 0; i < count; i++) {
                if (p + 2 > avcc->end) throw bad_stream("truncated avcC");
                const size_t len = size_t(d[p]) << 8 | d[p + 1];
                if (p + 2 + len > avcc->end) throw bad_stream("truncated avcC");
                out->push_back({p + 2, len});
                p += 2 + len;
            }
        }
        // sample sizes
        const size_t z = stsz->body;
        const uint32_t uniform = be32(&d[z + 4]), n_samples = be32(&d[z + 8]);
        if (!uniform && z + 12 + size_t(n_samples) * 4 > stsz->end) throw bad_stream("truncated stsz");
        // chunk offsets
        std::vector<uint64_t> chunk_off;
        if (stco) { const uint32_t n = be32(&d[stco->body + 4]); for (uint32_t i = 0; i < n; i++) chunk_off.push_back(be32(&d[stco->body + 8 + size_t(i) * 4])); }
        else { const uint32_t n = be32(&d[co64->body + 4]); for (uint32_t i = 0; i < n; i++) chunk_off.push_back(uint64_t(be32(&d[co64->body + 8 + size_t(i) * 8])) << 32 | be32(&d[co64->body + 12 + size_t(i) * 8])); }
        // samples per chunk
        const uint32_t n_stsc = be32(&d[stsc->body + 4]);
        uint32_t sample = 0;
        for (uint32_t e = 0; e < n_stsc && sample < n_samples; e++) {
            const uint32_t first = be32(&d[stsc->body + 8 + size_t(e) * 12]), per = be32(&d[stsc->body + 12 + size_t(e) * 12]);
            const uint32_t next_first = e + 1 < n_stsc ? be32(&d[stsc->body + 8 + size_t(e + 1) * 12]) : uint32_t(chunk_off.size()) + 1;
            for (uint32_t c = first; c < next_first && c <= chunk_off.size() && sample < n_samples; c++) {
                uint64_t at = chunk_off[c - 1];
                for (uint32_t k = 0; k < per && sample < n_samples; k++, sample++) {
                    const uint32_t size = uniform ? uniform : be32(&d[z + 12 + size_t(sample) * 4]);
                    if (at + size > d.size()) throw bad_stream("MP4 sample outside the file");
                    size_t q = size_t(at);                // the sample: length-prefixed NAL units
                    const size_t end = size_t(at) + size;
                    while (q + length_size <= end) {
                        size_t len = 0;
                        for (size_t b = 0; b < length_size; b++) len = len << 8 | d[q + b];
                        q += length_size;
                        if (len == 0 || q + len > end) break;
                        out->push_back({q, len});
                        q += len;
                    }
                    at += size;
                }
            }
        }
        return true;
    }
    throw bad_stream("MP4 without an H.264 (avc1) track");
}

// ---------------------------------------------------------------------------------------------- the decoder
// Plays libavcodec-hooks for compressor / decompressor (host/avr_recode.h): the whole stream is pulled through
// read_packet, then every CABAC slice is offered to init_decoder with its payload (the bytes from the first byte of
// slice_data() to the end of the unescaped NAL unit, what ff_init_cabac_decoder is given); when the driver hooks the
// slice, its bins are requested one by one in syntax order.
class h264_stream_decoder : public host::stream_decoder {
  public:
    struct stats_t {
        size_t slices = 0, hooked = 0, unsupported = 0, failed = 0, macroblocks = 0;
        std::string last_reason;
        std::map<std::string, size_t> literal_reasons;   // why slices were left literal: the message of what stopped the parser, counted
    } stats;
    // Fire begin / end_sub_mb and begin / end_coding_type around residual blocks (model_hooks::residual): all eleven hooks of
    // recode.cpp:219-235 are then live, h264_model keys significance-map bins by position and nonzero count, and the
    // recorders queue them behind the block's nonzero count.  Off by default, as in the reference's fork ("Not called").
    // Both directions of a file must agree on it (the container does not say which model wrote it).
    bool residual_hooks = false;
    // Threads for the payload dry runs (pass 2 of decode_video): 0 = AVR_PARSE_THREADS if that is a number (clamped to 1 .. 16), else the
    // host's cores (at most 16).  A caller that already runs one decoder per thread (recode test <dir>) sets 1.
    unsigned dry_run_threads = 0;

    void expect_payload_questions() override { answers_ahead_ = true; }

    void decode_video(host::hooks *h, int (*read_packet)(void *, uint8_t *, int), void *opaque) override {
        std::vector<uint8_t> data, chunk(1 << 16);
        for (;;) {
            const int got = read_packet(opaque, chunk.data(), int(chunk.size()));
            if (got <= 0) break;
            data.insert(data.end(), chunk.begin(), chunk.begin() + got);
        }
        std::vector<nal_ref> nals;
        if (!mp4_nals(data, &nals)) nals = annexb_nals(data);
        // Pass 1, in stream order: parameter sets and slice headers (a header is read against the parameter sets in force where
        // it stands).  Cheap: a few dozen bits per NAL unit.
        std::vector<slice_job> jobs;
        for (const nal_ref &n : nals) {
            if (n.size < 2) continue;
            const uint8_t header = data[n.offset];
            const int type = header & 31, ref_idc = (header >> 5) & 3;
            if (type != 1 && type != 5 && type != 6 && type != 7 && type != 8) continue;
            std::vector<uint8_t> rbsp = unescape(&data[n.offset + 1], n.size - 1);
            if (type == 6) { note_encoder(rbsp); continue; }
            if (type == 7 || type == 8) {
                try { if (type == 7) parse_sps(rbsp, sps_); else parse_pps(rbsp, pps_); }
                catch (const bad_stream &e) { stats.failed++; stats.last_reason = e.what(); }      // its slices will not find it
                continue;
            }
            slice_job j;
            try {                                        // a header this build does not take: the slice is not offered to the hooks
                j.sh = parse_slice_header(rbsp, type, ref_idc, sps_, pps_);
                if (j.sh.data_offset >= rbsp.size()) throw bad_stream("slice without data");
                j.sh.x264_old_444_cbf = j.sh.chroma_array_type == 3 && x264_build_ >= 0 && x264_build_ < 151;
                j.header_ok = true;
            } catch (const unsupported &e) { j.unsupported_header = true; j.reason = e.what(); }
            catch (const bad_stream &e) { j.reason = std::string("header: ") + e.what(); }
            j.rbsp = std::move(rbsp);
            jobs.push_back(std::move(j));
        }
        // Pass 2, when the compressor has said it will ask (expect_payload_questions): does each payload parse to its end?  One parse of
        // the payload per slice with the build's own CABAC engine, slices independent of each other -- on all the host's cores.
        if (answers_ahead_) {
            host::phase_timer timer("parse: payload dry-runs");
            std::atomic<size_t> next{0};
            auto work = [&]() {
                std::vector<mb_info> scratch;            // one per thread, kept from slice to slice (a fresh half megabyte per slice is an mmap and its page faults each time)
                for (size_t i; (i = next.fetch_add(1)) < jobs.size();)
                    if (jobs[i].header_ok) jobs[i].decodes = dry_run(jobs[i], residual_hooks, &jobs[i].reason, &scratch) ? 1 : 0;
            };
            unsigned want = dry_run_threads;
            if (!want) {                                         // (the CLI's environment, like AVR_DEVICE; anything that is not a number is ignored)
                const char *forced = getenv("AVR_PARSE_THREADS");
                char *end = nullptr;
                const long v = forced ? strtol(forced, &end, 10) : 0;
                if (forced && end != forced && *end == 0) want = unsigned(std::min<long>(std::max<long>(v, 1), 16));
            }
            if (!want) { const unsigned hw = std::thread::hardware_concurrency(); want = std::min<unsigned>(hw ? hw : 4, 16); }
            const size_t n_threads = std::max<size_t>(1, std::min<size_t>(want, jobs.size()));
            std::vector<std::thread> pool;
            for (size_t t = 1; t < n_threads; t++) pool.emplace_back(work);
            work();
            for (std::thread &t : pool) t.join();
        }
        // Pass 3, in stream order: the hooks.
        host::phase_timer timer("parse: hooks in stream order");
        for (slice_job &j : jobs) {
            slice(h, j);
            std::vector<uint8_t>().swap(j.rbsp);                 // one slice's unescaped bytes at a time from here on, not a second copy of the file
        }
    }

    // compressor asks before it commits to the slice just offered: does the payload parse to its end?  (On the way back
    // the payload is a surrogate and nobody asks: the block kind says whether the slice was coded.)
    bool payload_decodes() override {
        if (!offered_) return false;
        if (offered_->decodes < 0) offered_->decodes = dry_run(*offered_, residual_hooks, &offered_->reason, &dry_scratch_) ? 1 : 0;   // not worked out ahead
        if (!offered_->decodes) { stats.last_reason = offered_->reason; stats.literal_reasons[offered_->reason]++; }
        return offered_->decodes != 0;
    }

  private:
    // The encoder's flush (9.3.4.5) ends on the rbsp_stop_one_bit, and that bit is the last one the decoding engine has
    // pulled in when it decodes end_of_slice_flag = 1 (9 bits at initialisation, one per renormalisation shift).  The
    // payload parsed to its end iff that bit is a one and lies in the payload's LAST byte.  What follows it inside that
    // byte is not required to be zero: x264 puts a bit of its own there -- the reason the container keeps every block's
    // last byte (recode.cpp:1291-1294, :1354-1360) -- and a payload with more bytes behind the stop bit
    // (cabac_zero_words) would not come back through that one-byte patch, so it is left alone.
    static bool ends_cleanly(size_t bits_read, const uint8_t *p, size_t size) {
        if (bits_read == 0 || bits_read > size * 8) return false;
        const size_t stop = bits_read - 1;
        return ((p[stop >> 3] >> (7 - (stop & 7))) & 1) && (stop >> 3) == size - 1;
    }

    struct slice_job {
        std::vector<uint8_t> rbsp;                       // the unescaped NAL unit behind its header byte
        slice_header sh;
        bool header_ok = false, unsupported_header = false;
        int decodes = -1;                                // payload_decodes(): -1 not worked out yet, 0 / 1
        std::string reason;                              // why not
    };

    // One parse of a slice's payload with the build's own CABAC engine, no hooks: true iff it ends on end_of_slice_flag in the
    // payload's last byte.  Touches nothing of the decoder: slices can be dry-run side by side.
    static bool dry_run(const slice_job &j, bool residual_hooks, std::string *why, std::vector<mb_info> *scratch_mbs) {
        try {
            const uint8_t *payload = j.rbsp.data() + j.sh.data_offset;
            const size_t size = j.rbsp.size() - j.sh.data_offset;
            engine_bins bins(payload, size);
            uint8_t states[1024];
            std::vector<mb_info> &scratch = *scratch_mbs;                 // (every entry the parser reads it has written in this slice: `slice` numbers tell)
            const size_t n_mbs = size_t(j.sh.width_mbs) * j.sh.height_mbs;
            if (scratch.size() != n_mbs) scratch.assign(n_mbs, mb_info());
            else for (mb_info &m : scratch) m.slice = -1;
            model_hooks dry;                             // no hooks fire; with the residual hooks on, the counts must fit their fields
            dry.refuse_full_blocks = residual_hooks;
            slice_parser<engine_bins> p(bins, j.sh, states, scratch, 0, dry);
            p.run();
            if (ends_cleanly(bins.d.bit_position(), payload, size)) return true;
            *why = "the payload does not end on end_of_slice_flag in its last byte";
            return false;
        } catch (const std::exception &e) {
            *why = e.what();
            return false;
        }
    }

    void slice(host::hooks *h, slice_job &j) {
        stats.slices++;
        if (!j.header_ok) {
            if (j.unsupported_header) stats.unsupported++; else stats.failed++;
            stats.last_reason = j.reason.rfind("header: ", 0) == 0 ? j.reason.substr(8) : j.reason;
            stats.literal_reasons[j.reason]++;
            return;
        }
        slice_header &sh = j.sh;
        // What frame_spec is told (recode.cpp:173): a number that is the same for the slices of a picture and differs from one
        // picture to the next.  frame_num itself repeats across a non-reference picture and the one after it, and h264_model
        // clears its per-picture store only when the number changes (update_frame_spec, recode.cpp:831-850): with the residual
        // hooks on, what an earlier picture left in a block would be counted into this picture's nonzero counts.
        if (sh.first_mb == 0 || pictures_ == 0) pictures_++;
        sh.frame_num = pictures_;
        const size_t n_mbs = size_t(sh.width_mbs) * sh.height_mbs;
        if (mbs_.size() != n_mbs) mbs_.assign(n_mbs, mb_info());
        const uint8_t *payload = j.rbsp.data() + sh.data_offset;
        const size_t size = j.rbsp.size() - sh.data_offset;
        offered_ = &j;
        void *dec = h->cabac.init_decoder(h->opaque, &cabac_context_identity_, payload, int(size));
        offered_ = nullptr;
        if (!dec) return;                                // not hooked: nothing of this slice is needed later (no reconstruction)
        stats.hooked++;
        hook_bins bins{h, dec};
        slice_parser<hook_bins> p(bins, sh, cabac_state_, mbs_, ++slice_counter_, model_hooks{h, residual_hooks});
        stats.macroblocks += size_t(p.run());
    }

    // SEI: x264 signs its streams ("x264 - core <build> ...", user_data_unregistered); the build number decides one
    // context derivation for 4:4:4 streams (slice_header::x264_old_444_cbf)
    void note_encoder(const std::vector<uint8_t> &sei) {
        static const char tag[] = "x264 - core ";
        const size_t n = sizeof tag - 1;
        for (size_t i = 0; i + n < sei.size(); i++)
            if (!memcmp(&sei[i], tag, n)) {
                int build = 0;
                for (size_t j = i + n; j < sei.size() && sei[j] >= '0' && sei[j] <= '9' && build < 100000; j++) build = build * 10 + (sei[j] - '0');
                x264_build_ = build;
                return;
            }
    }

    int x264_build_ = -1;                                // -1: not an x264 stream, or not signed
    sps_t sps_[32];
    pps_t pps_[256];
    std::vector<mb_info> mbs_;
    uint8_t cabac_state_[1024];                          // the addresses get() hands to the hooks, as libavcodec's sl->cabac_state
    int cabac_context_identity_ = 0;                     // stands for the one CABACContext of a single-threaded decode (recode.cpp:153)
    int slice_counter_ = 0, pictures_ = 0;
    slice_job *offered_ = nullptr;                       // the slice init_decoder is being called for
    std::vector<mb_info> dry_scratch_;
    bool answers_ahead_ = false;
};

}  // namespace h264
}  // namespace avr
