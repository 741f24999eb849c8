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
#include <thread>
#include <typeinfo>
#include <atomic>
#include <memory>

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
void dump_stream_info(const std::string &path, const std::string &bytes, int index, std::ostream &err = std::cerr) {
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
    err << "Input #" << index << ", h264, from '" << path << "':" << std::endl;
    err << "  Duration: " << (duration > 0 ? dur : "N/A") << ", start: 0.000000, bitrate: " << int(kbps) << " kb/s" << std::endl;
    err << "    Stream #" << index << ":0(und): Video: h264, " << width << "x" << height << ", " << int(kbps) << " kb/s, " << std::setprecision(4)
        << fps << " fps" << std::endl;
}

// what roundtrip() says about a file that came back as it went in (recode.cpp:1620-1634)
void report_roundtrip(const std::string &original, const std::string &compressed, std::ostream &err) {
    const double ratio = compressed.size() * 1.0 / original.size();
    host::Recoded proto;
    proto.ParseFromArray(compressed.data(), compressed.size());
    size_t block_bytes = 0;
    for (const host::Block &b : proto.block) block_bytes += b.literal.size() + b.cabac.size();
    const double overhead = (compressed.size() - block_bytes) * 1.0 / compressed.size();
    err << "Compress-decompress roundtrip succeeded:" << std::endl;
    err << " compression ratio: " << ratio * 100. << "%" << std::endl;
    err << " protobuf overhead: " << overhead * 100. << "%" << std::endl;
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
        report_roundtrip(original, compressed, std::cerr);
        return 0;
    }
    std::cerr << "Compress-decompress roundtrip failed." << std::endl;
    return 1;
}

// AVR_TEST_SEQUENTIAL=1: the reference's loop as rounds 1-3 had it (test.cpp:113-148) -- a file at a time, roundtrip() on each, a GPU
// batch per file and direction -- kept for the comparison (profiles/r04_cli_timing.txt) and as a check that both write the same files
void perf_test_driver_sequential(const std::string &directory_path) {
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


// test.cpp:113-148 (perf_test_driver) and :20-110 (the metrics file), without the freopen / re-parsing of the log: the same
// files come out (output/<name>, output/log.txt, output/metrics.csv with the columns of test.cpp:32).
//
// The reference walks the directory one file after another, compress then decompress.  Here the files of a window (AVR_TEST_WINDOW,
// default 64) go through the two directions TOGETHER: the host side of each direction -- the parse, the hooks, the recorders, and on the way
// back the range decoder (K3) -- runs one file per thread (a file's estimators are its own, recode.cpp:1065: files are independent; inside a
// file both passes stay in stream order), and the slices of all the window's files are coded in ONE avr_batch per direction (K2 on the way
// in, K1 from resolved codes on the way out): one HIP context, two batch objects and their pinned buffers for the whole directory, and
// batches of thousands of slices where a single clip offers a few hundred.  AVR_TEST_SEQUENTIAL=1 keeps the reference's loop (a file at a
// time, a batch per file and direction) -- same output files, for comparison.  Per-file times in metrics.csv: the file's own host time plus
// its share (by bins) of the window's batch.
struct file_job {
    std::filesystem::path path;
    std::string original, compressed, decompressed, log_text;
    std::unique_ptr<host::compressor> c;
    std::unique_ptr<host::decompressor> d;
    double c_host_ms = 0, d_host_ms = 0, c_gpu_ms = 0, d_gpu_ms = 0;
    size_t c_bins = 0, d_bins = 0;
    bool failed = false;
};

template <class F>
void for_each_parallel(size_t n, unsigned threads, F &&f) {          // f(i) for i < n, on `threads` threads; f does not throw
    std::atomic<size_t> next{0};
    auto work = [&]() { for (size_t i; (i = next.fetch_add(1)) < n;) f(i); };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < threads && t < n; t++) pool.emplace_back(work);
    work();
    for (std::thread &t : pool) t.join();
}

double ms_since(std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }

// one avr_batch for everything the window's files have pending in one direction; grown when a window needs more than the last one did
struct shared_batch {
    avr_batch *b = nullptr;
    size_t slices = 0, bins = 0;
    ~shared_batch() { if (b) avr_batch_destroy(b); }
    avr_batch *get(size_t want_slices, size_t want_bins) {
        if (!b || want_slices > slices || want_bins > bins) {
            if (b) avr_batch_destroy(b);
            slices = std::max(want_slices, slices); bins = std::max(want_bins, bins);
            b = avr_batch_create(device(), slices, bins);
            if (!b) throw std::runtime_error(std::string("avr: ") + avr_last_error());
        } else host::gpu_check(avr_batch_reset(b));
        return b;
    }
};

void log_exception(file_job &j, const std::exception &e) {
    j.failed = true;
    j.log_text += std::string("Exception (") + typeid(e).name() + "): " + e.what() + "\n";
}

void perf_test_driver(const std::string &directory_path) {
    if (getenv("AVR_TEST_SEQUENTIAL")) { perf_test_driver_sequential(directory_path); return; }
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
    const char *w = getenv("AVR_TEST_WINDOW");
    const size_t window = std::max<long>(1, w ? atol(w) : 64);
    const unsigned hw = std::thread::hardware_concurrency();
    const unsigned threads = std::min<unsigned>(hw ? hw : 4, 32);
    shared_batch batch_in, batch_out;
    for (size_t base = 0; base < files.size(); base += window) {
        const size_t n = std::min(window, files.size() - base);
        std::vector<file_job> jobs(n);
        std::cout << base + 1 << "-" << base + n << "/" << files.size() << "..." << std::endl;
        // ---- compress, host side: one file per thread
        for_each_parallel(n, threads, [&](size_t i) {
            file_job &j = jobs[i];
            j.path = files[base + i];
            const auto t0 = std::chrono::steady_clock::now();
            try {
                j.original = slurp(j.path.string());
                std::stringstream captured;                          // what the reference's run prints for the file (test.cpp:129 sends it to the log)
                dump_stream_info(j.path.string(), j.original, int(base + i), captured);
                j.log_text = captured.str();
                j.c.reset(new host::compressor(j.original, device()));
                h264::h264_stream_decoder dec;
                dec.residual_hooks = model_hooks();
                dec.dry_run_threads = 1;                             // the cores are taken by the files
                j.c->prepare(&dec);
                j.c_bins = j.c->pending_bins();
            } catch (const std::exception &e) { log_exception(j, e); }
            j.c_host_ms = ms_since(t0);
        });
        // ---- compress, GPU: the slices of every file of the window in one K2 batch
        try {
            size_t slices = 0, bins = 0;
            for (file_job &j : jobs) if (!j.failed) { slices += j.c->pending_slices(); bins += j.c_bins; }
            if (slices) {
                const auto t0 = std::chrono::steady_clock::now();
                avr_batch *b = batch_in.get(slices, bins + 8);
                for (file_job &j : jobs) if (!j.failed) j.c->add_to(b);
                host::gpu_check(avr_batch_run(b));
                for (file_job &j : jobs) if (!j.failed) { try { j.c->take_from(b); } catch (const std::exception &e) { log_exception(j, e); } }
                const double ms = ms_since(t0);
                for (file_job &j : jobs) j.c_gpu_ms = bins ? ms * double(j.c_bins) / double(bins) : 0;
            }
        } catch (const std::exception &e) { for (file_job &j : jobs) if (!j.failed) log_exception(j, e); }
        // ---- the container's bytes, then decompress, host side (K3 + parse): one file per thread again
        for_each_parallel(n, threads, [&](size_t i) {
            file_job &j = jobs[i];
            if (j.failed) return;
            auto t0 = std::chrono::steady_clock::now();
            try {
                j.compressed = j.c->finish();
                j.c.reset();
                j.c_host_ms += ms_since(t0);
                t0 = std::chrono::steady_clock::now();
                j.d.reset(new host::decompressor(j.compressed, device()));
                h264::h264_stream_decoder dec;
                dec.residual_hooks = model_hooks();
                j.d->prepare(&dec);
                j.d_bins = j.d->pending_bins();
            } catch (const std::exception &e) { log_exception(j, e); }
            j.d_host_ms = ms_since(t0);
        });
        // ---- decompress, GPU: one K1 batch (resolved codes)
        try {
            size_t slices = 0, bins = 0;
            for (file_job &j : jobs) if (!j.failed) { slices += j.d->pending_slices(); bins += j.d_bins; }
            if (slices) {
                const auto t0 = std::chrono::steady_clock::now();
                avr_batch *b = batch_out.get(slices, bins + 16 * slices + 64);
                for (file_job &j : jobs) if (!j.failed) j.d->add_to(b);
                host::gpu_check(avr_batch_run(b));
                for (file_job &j : jobs) if (!j.failed) { try { j.d->take_from(b); } catch (const std::exception &e) { log_exception(j, e); } }
                const double ms = ms_since(t0);
                for (file_job &j : jobs) j.d_gpu_ms = bins ? ms * double(j.d_bins) / double(bins) : 0;
            }
        } catch (const std::exception &e) { for (file_job &j : jobs) if (!j.failed) log_exception(j, e); }
        // ---- the files' bytes back, the comparison (recode.cpp:1618), the output files
        for_each_parallel(n, threads, [&](size_t i) {
            file_job &j = jobs[i];
            if (j.failed) return;
            const auto t0 = std::chrono::steady_clock::now();
            try {
                j.decompressed = j.d->finish();
                j.d.reset();
                j.d_host_ms += ms_since(t0);
                std::stringstream msg;
                if (j.original == j.decompressed) {
                    std::ofstream output_file(directory_path + "/output/" + j.path.filename().string(), std::ios::binary);
                    if (output_file.is_open()) output_file << j.compressed;
                    report_roundtrip(j.original, j.compressed, msg);
                } else {
                    msg << "Compress-decompress roundtrip failed." << std::endl;
                    j.failed = true;
                }
                j.log_text += msg.str();
            } catch (const std::exception &e) { log_exception(j, e); }
        });
        // ---- log.txt and metrics.csv, in the order of the files
        for (file_job &j : jobs) {
            const std::string text = j.log_text + "\n";
            log << text;
            if (j.failed) { fail_count++; continue; }
            auto field = [&](const std::string &from, const std::string &to) {      // what test.cpp:81-104 extracts from the same lines
                const size_t a = text.find(from);
                if (a == std::string::npos) return std::string();
                const size_t b = text.find(to, a + from.size());
                return text.substr(a + from.size(), b == std::string::npos ? std::string::npos : b - a - from.size());
            };
            const int ctime = std::max(1, int(j.c_host_ms + j.c_gpu_ms + 0.5)), dtime = std::max(1, int(j.d_host_ms + j.d_gpu_ms + 0.5));
            const double rate = atof(field("compression ratio: ", "%").c_str());
            const double original_size = double(fs::file_size(j.path)) / 1000000.0;
            csv << "\"" << j.path.string() << "\"," << field("Duration: ", ",") << "," << original_size << "," << original_size * (rate / 100) << ","
                << rate << "," << 100 - rate << "," << ctime + dtime << "," << ctime << "," << original_size / (ctime / 1000.0) << "," << dtime << ","
                << original_size / (dtime / 1000.0) << "," << field("Video: ", ",") << "," << field("kb/s, ", " fps") << std::endl;
        }
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
