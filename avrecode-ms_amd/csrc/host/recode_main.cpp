// recode -- the reference's command line (recode.cpp:1642-1677, test.cpp:113-148) on the MI355X build:
//
//     recode [compress|decompress|roundtrip|test] <input> [output]
//
// Same commands, same messages, same exit codes (0 = done, 1 = usage / exception / roundtrip mismatch).  The
// stream is decoded by this build's own H.264 syntax parser (avr_h264.h) where the reference links a patched
// FFmpeg; the bins go through the same eleven hooks into compressor / decompressor (avr_recode.h), which code all
// slices of a file in one GPU batch (K2 on the way in, K3 on the CPU + K1 on the way out).
//
// One more command, `recode probe <input>`: parse every slice with the build's own CABAC engine and report how many
// parse to their end, and for the others why not, counted per reason (no GPU, nothing written) -- what tests/test_h264.py
// pins the parser and its tables with, and what shows on any file how much of it the parser leaves literal.
//
// Environment: AVR_DEVICE = HIP device index (default 0); AVR_MODEL_HOOKS=1 = the stream decoder also fires begin / end_sub_mb
// and begin / end_coding_type around residual blocks (all eleven hooks of recode.cpp:219-235 live: the significance-map side of
// h264_model), for compress AND decompress of the same file -- the reference's fork leaves those four uncalled, so off by default.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <filesystem>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <sstream>
#include <typeinfo>

#include "avr_h264.h"

namespace {

using namespace avr;

int device() { const char *d = getenv("AVR_DEVICE"); return d ? atoi(d) : 0; }
bool model_hooks() { const char *d = getenv("AVR_MODEL_HOOKS"); return d && atoi(d) != 0; }

std::string slurp(const std::string &path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + path);
    std::stringstream s;
    s << f.rdbuf();
    return s.str();
}

std::string compress_bytes(const std::string &original) {            // compressor::run, recode.cpp:1122-1132
    host::compressor c(original, device());
    h264::h264_stream_decoder d;
    d.residual_hooks = model_hooks();
    return c.run(&d);
}

std::string decompress_bytes(const std::string &recoded) {           // decompressor::run, recode.cpp:1345-1364
    host::decompressor d(recoded, device());
    h264::h264_stream_decoder dec;
    dec.residual_hooks = model_hooks();
    return d.run(&dec);
}

// What libavformat's av_dump_format prints for the reference (recode.cpp:113-117) and test.cpp:81-104 scrapes: only the
// lines and fields the tester looks for, from what the MP4 says about itself.
void dump_stream_info(const std::string &path, const std::string &bytes, int index) {
    double duration = 0;
    unsigned samples = 0, width = 0, height = 0;
    const std::vector<uint8_t> d(bytes.begin(), bytes.end());
    const uint8_t mvhd[4] = {'m', 'v', 'h', 'd'}, stsz[4] = {'s', 't', 's', 'z'}, avc1[4] = {'a', 'v', 'c', '1'};
    for (size_t i = 4; i + 40 < d.size(); i++) {
        if (!memcmp(&d[i], mvhd, 4) && duration == 0) {
            const int version = d[i + 4];
            const size_t p = i + 8 + (version ? 16 : 8);
            const uint32_t timescale = h264::be32(&d[p]);
            const uint64_t dur = version ? (uint64_t(h264::be32(&d[p + 4])) << 32 | h264::be32(&d[p + 8])) : h264::be32(&d[p + 4]);
            if (timescale) duration = double(dur) / timescale;
        } else if (!memcmp(&d[i], avc1, 4) && !width && i + 4 + 78 + 8 <= d.size() && !memcmp(&d[i + 4 + 78 + 4], "avcC", 4)) {
            width = unsigned(d[i + 4 + 24]) << 8 | d[i + 4 + 25];       // the sample entry (not the brand of the same name in ftyp)
            height = unsigned(d[i + 4 + 26]) << 8 | d[i + 4 + 27];
        } else if (!memcmp(&d[i], stsz, 4) && !samples && width) samples = h264::be32(&d[i + 12]);
    }
    const int total = int(duration * 100 + 0.5);
    char dur[32];
    snprintf(dur, sizeof dur, "%02d:%02d:%02d.%02d", total / 360000, total / 6000 % 60, total / 100 % 60, total % 100);
    const double kbps = duration > 0 ? bytes.size() * 8.0 / duration / 1000.0 : 0, fps = duration > 0 ? samples / duration : 0;
    std::cerr << "Input #" << index << ", h264, from '" << path << "':" << std::endl;
    std::cerr << "  Duration: " << (duration > 0 ? dur : "N/A") << ", start: 0.000000, bitrate: " << int(kbps) << " kb/s" << std::endl;
    std::cerr << "    Stream #" << index << ":0(und): Video: h264, " << width << "x" << height << ", " << int(kbps) << " kb/s, " << std::setprecision(4)
              << fps << " fps" << std::endl;
}

// recode.cpp:1601-1640
int roundtrip(const std::string &input_filename, std::ostream *out, int *compression_time = nullptr, int *decompression_time = nullptr,
              const int input_index = 0) {
    const std::string original = slurp(input_filename);
    dump_stream_info(input_filename, original, input_index);
    const auto c1 = std::chrono::high_resolution_clock::now();
    const std::string compressed = compress_bytes(original);
    const auto c2 = std::chrono::high_resolution_clock::now();
    const std::string decompressed = decompress_bytes(compressed);
    const auto d2 = std::chrono::high_resolution_clock::now();
    if (compression_time && decompression_time) {
        *compression_time = int(std::chrono::duration_cast<std::chrono::milliseconds>(c2 - c1).count());
        *decompression_time = int(std::chrono::duration_cast<std::chrono::milliseconds>(d2 - c2).count());
    }
    if (original == decompressed) {
        if (out) (*out) << compressed;
        const double ratio = compressed.size() * 1.0 / original.size();
        host::Recoded proto;
        proto.ParseFromArray(compressed.data(), compressed.size());
        size_t block_bytes = 0;
        for (const host::Block &b : proto.block) block_bytes += b.literal.size() + b.cabac.size();
        const double overhead = (compressed.size() - block_bytes) * 1.0 / compressed.size();
        std::cerr << "Compress-decompress roundtrip succeeded:" << std::endl;
        std::cerr << " compression ratio: " << ratio * 100. << "%" << std::endl;
        std::cerr << " protobuf overhead: " << overhead * 100. << "%" << std::endl;
        return 0;
    }
    std::cerr << "Compress-decompress roundtrip failed." << std::endl;
    return 1;
}

// test.cpp:113-148 (perf_test_driver) and :20-110 (the metrics file), without the freopen / re-parsing of the log: the same
// files come out (output/<name>, output/log.txt, output/metrics.csv with the columns of test.cpp:32)
void perf_test_driver(const std::string &directory_path) {
    namespace fs = std::filesystem;
    std::vector<fs::path> files;
    for (const auto &entry : fs::directory_iterator(directory_path))
        if (fs::is_regular_file(entry.path())) files.push_back(entry.path());
    std::sort(files.begin(), files.end());
    fs::create_directory(directory_path + "/output");
    std::ofstream log(directory_path + "/output/log.txt"), csv(directory_path + "/output/metrics.csv");
    csv << "File,Duration,Initial size (MB),Compressed size (MB),Compression rate (%),Space saving (%),Total time (ms),Compression time (ms),"
           "Compression speed (MB/s),Decompression time (ms),Decompression speed (MB/s),Video stream,Frames per second"
        << std::endl;
    int fail_count = 0;
    for (size_t i = 0; i < files.size(); i++) {
        std::cout << i + 1 << "/" << files.size() << "..." << std::endl;
        std::stringstream captured;
        std::streambuf *saved = std::cerr.rdbuf(captured.rdbuf());       // the run's messages go to the log (test.cpp:129)
        int ctime = 0, dtime = 0, rc = 1;
        try {
            std::ofstream output_file(directory_path + "/output/" + files[i].filename().string(), std::ios::binary);
            rc = roundtrip(files[i].string(), output_file.is_open() ? &output_file : nullptr, &ctime, &dtime, int(i));
        } catch (const std::exception &e) {
            std::cerr << "Exception (" << typeid(e).name() << "): " << e.what() << std::endl;
        }
        std::cerr << std::endl;
        std::cerr.rdbuf(saved);
        const std::string text = captured.str();
        log << text;
        if (rc != 0) { fail_count++; continue; }
        auto field = [&](const std::string &from, const std::string &to) {      // what test.cpp:81-104 extracts from the same lines
            const size_t a = text.find(from);
            if (a == std::string::npos) return std::string();
            const size_t b = text.find(to, a + from.size());
            return text.substr(a + from.size(), b == std::string::npos ? std::string::npos : b - a - from.size());
        };
        const double rate = atof(field("compression ratio: ", "%").c_str());
        const double original_size = double(fs::file_size(files[i])) / 1000000.0;
        csv << "\"" << files[i].string() << "\"," << field("Duration: ", ",") << "," << original_size << "," << original_size * (rate / 100) << ","
            << rate << "," << 100 - rate << "," << ctime + dtime << "," << ctime << "," << original_size / (ctime / 1000.0) << "," << dtime << ","
            << original_size / (dtime / 1000.0) << "," << field("Video: ", ",") << "," << field("kb/s, ", " fps") << std::endl;
    }
    if (fail_count > 0)
        std::cout << "Compress-decompress roundtrip failed on " << fail_count << " / " << files.size() << " files" << std::endl;
}

int probe(const std::string &input_filename) {
    const std::string bytes = slurp(input_filename);
    h264::h264_stream_decoder dec;
    dec.expect_payload_questions();                      // every slice is asked about: the answers are worked out ahead, on all cores
    struct counts { h264::h264_stream_decoder *d; size_t ok = 0, bad = 0; std::string why; } c{&dec};
    host::hooks h{};
    h.opaque = &c;
    h.cabac.init_decoder = [](void *o, void *, const uint8_t *, int) -> void * {
        counts *c = static_cast<counts *>(o);
        if (c->d->payload_decodes()) c->ok++; else { c->bad++; c->why = c->d->stats.last_reason; }
        return nullptr;
    };
    struct reader { const std::string *s; size_t at; } r{&bytes, 0};
    dec.decode_video(&h, [](void *o, uint8_t *b, int n) {
        reader *r = static_cast<reader *>(o);
        const size_t k = std::min<size_t>(size_t(n), r->s->size() - r->at);
        memcpy(b, r->s->data() + r->at, k);
        r->at += k;
        return int(k); }, &r);
    std::cout << "{\"slices\": " << dec.stats.slices << ", \"parse_to_the_end\": " << c.ok << ", \"fail\": " << c.bad << ", \"unsupported\": "
              << dec.stats.unsupported << ", \"header_failures\": " << dec.stats.failed << ", \"literal_reasons\": {";
    bool first = true;                                                // why slices would stay literal blocks, counted per reason
    for (const auto &kv : dec.stats.literal_reasons) {
        std::string key;
        for (char ch : kv.first) { if (ch == '"' || ch == '\\') key += '\\'; key += ch; }
        std::cout << (first ? "" : ", ") << "\"" << key << "\": " << kv.second;
        first = false;
    }
    std::cout << "}}" << std::endl;
    if (c.bad || dec.stats.unsupported || dec.stats.failed) std::cerr << "last reason: " << (c.why.empty() ? dec.stats.last_reason : c.why) << std::endl;
    return 0;
}

}  // namespace

int main(int argc, char **argv) {
    if (argc < 3 || argc > 4) {
        std::cerr << "Usage: " << argv[0] << " [compress|decompress|roundtrip|test] <input> [output]" << std::endl;
        return 1;
    }
    const std::string command = argv[1], input_filename = argv[2];
    std::ofstream out_file;
    if (argc > 3) out_file.open(argv[3], std::ios::binary);
    try {
        if (command == "compress") {
            (out_file.is_open() ? out_file : std::cout) << compress_bytes(slurp(input_filename));
        } else if (command == "decompress") {
            (out_file.is_open() ? out_file : std::cout) << decompress_bytes(slurp(input_filename));
        } else if (command == "roundtrip") {
            return roundtrip(input_filename, out_file.is_open() ? &out_file : nullptr);
        } else if (command == "test") {
            perf_test_driver(input_filename);
            return 0;
        } else if (command == "probe") {
            return probe(input_filename);
        } else {
            throw std::invalid_argument("Unknown command: " + command);
        }
    } catch (const std::exception &e) {
        std::cerr << "Exception (" << typeid(e).name() << "): " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
