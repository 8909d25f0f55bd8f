// rx_link.cpp -- the reference's per-buffer RX loop around the drop-in class.
//
// Reproduces TXRX::rx_single_link (ref: cpp/USRP_server_link_threads.cpp:605-702)
// with the pieces it depends on reduced to their shape:
//   * a software RX thread that hands over pinned 1 M-sample buffers
//     (ref: hardware_manager::software_rx_thread, cpp/USRP_hardware_manager.cpp:1331-1395),
//   * a pool of pinned output buffers of buffer_len*data_mem_mult samples
//     (ref: preallocator<float2>, headers/USRP_server_memory_management.hpp:103-273,
//      sized as in cpp/USRP_server_link_threads.cpp:143-150),
//   * a consumer that stands in for Sync_server::tcp_streamer (drops the packet).
// The demodulator is include/USRP_demodulator.hpp exactly as the server would
// compile it.  Reports the PCIe-INCLUSIVE rate of the host-pointer entry
// (process() is synchronous like the reference: H2D, kernels, D2H, sync).
//
//   hipcc -O2 -std=c++17 -Iinclude tools/rx_link.cpp -Lgpu_sdr_amd -lgsdr \
//         -Wl,-rpath,$PWD/gpu_sdr_amd -lpthread -o /tmp/rx_link
//   rx_link [n_tones=256] [decim=100] [buffers=200] [pipe]
// With "pipe" the loop uses the pipelined submit()/wait() pair (upload, kernels
// and download of successive buffers overlap) instead of the synchronous process().
//
//   rx_link file <config.txt> <in.c64> <out.c64> [pipe]
// The same loop over a recorded stream (tests/test_gpu_rxlink.py): config.txt holds
// "key value..." lines (mode DIRECT|TONES|CHIRP|NOISE|NODSP, rate, buffer_len, decim,
// pf_average, fft_tones, freq ..., chirp_f ..., swipe_s ..., chirp_t ...); in.c64 holds
// whole buffers of complex64; every packet's payload is appended to out.c64 and the
// valid lengths are printed, as the streamer would put them into the packet headers.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <mutex>
#include <queue>
#include <random>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "USRP_demodulator.hpp"
#include "USRP_buffer_generator.hpp"

template <typename T>
class BlockingQueue {  // stands in for the boost::lockfree queues of the reference
   public:
    void push(T v) {
        { std::lock_guard<std::mutex> l(m_); q_.push(v); }
        c_.notify_one();
    }
    T pop() {
        std::unique_lock<std::mutex> l(m_);
        c_.wait(l, [&] { return !q_.empty(); });
        T v = q_.front();
        q_.pop();
        return v;
    }
   private:
    std::mutex m_;
    std::condition_variable c_;
    std::queue<T> q_;
};

static w_type wave_from_string(const std::string &s) {
    if (s == "TONES") return TONES;
    if (s == "CHIRP") return CHIRP;
    if (s == "NOISE") return NOISE;
    if (s == "DIRECT") return DIRECT;
    return NODSP;
}

// rx_single_link over a recorded stream; see the header comment
// rx_link tx <config.txt> <out.c64> <n_buffers>: `new TX_buffer_generator(&param)` + get(), the loop of the
// reference's TX side (ref: cpp/USRP_server_link_threads.cpp, tx_single_link), written to a file
static int tx_mode(int argc, char **argv) {
    if (argc < 5) {
        std::fprintf(stderr, "usage: rx_link tx <config.txt> <out.c64> <n_buffers>\n");
        return 2;
    }
    param p;
    p.mode = TX;
    p.rate = 0; p.gain = 0; p.bw = 0; p.tone = 0; p.samples = 0; p.delay = 0; p.burst_on = p.burst_off = 0;
    p.buffer_len = 0; p.tuning_mode = false; p.decim = 0; p.data_mem_mult = 1; p.fft_tones = 0; p.pf_average = 4;
    w_type mode = NODSP;
    int channels = -1;
    std::ifstream cfg(argv[2]);
    if (!cfg) { std::fprintf(stderr, "cannot read %s\n", argv[2]); return 2; }
    std::string line;
    while (std::getline(cfg, line)) {
        std::istringstream is(line);
        std::string key;
        if (!(is >> key)) continue;
        if (key == "mode") { std::string m; is >> m; mode = wave_from_string(m); }
        else if (key == "channels") is >> channels;
        else if (key == "rate") is >> p.rate;
        else if (key == "buffer_len") is >> p.buffer_len;
        else if (key == "freq") { int v; while (is >> v) p.freq.push_back(v); }
        else if (key == "ampl") { float v; while (is >> v) p.ampl.push_back(v); }
        else if (key == "chirp_f") { int v; while (is >> v) p.chirp_f.push_back(v); }
        else if (key == "swipe_s") { int v; while (is >> v) p.swipe_s.push_back(v); }
        else if (key == "chirp_t") { float v; while (is >> v) p.chirp_t.push_back(v); }
    }
    if (channels < 0) channels = mode == CHIRP ? 1 : (int)p.freq.size();
    for (int k = 0; k < channels; ++k) p.wave_type.push_back(mode);
    const long long n_buffers = std::atoll(argv[4]);
    FILE *fout = std::fopen(argv[3], "wb");
    if (!fout || p.buffer_len == 0) { std::fprintf(stderr, "cannot open the output file\n"); return 2; }
    RX_buffer_demodulator::device_index() = 0;
    TX_buffer_generator *generator = new TX_buffer_generator(&p);
    // the reference's loop (ref: cpp/USRP_server_link_threads.cpp:568-584): memory for a packet only when the
    // parameters need it (`dynamic`: everything but TONES); for TONES tx_vector goes in UNINITIALISED -- a poisoned
    // pointer here, so that a generator that wrote through it would fault -- and get() replaces it
    const bool dynamic = p.dynamic_buffer();
    float2 *pool = nullptr;
    if (dynamic && hipHostMalloc((void **)&pool, p.buffer_len * sizeof(float2)) != hipSuccess) return 1;
    const auto t0 = std::chrono::steady_clock::now();
    for (long long k = 0; k < n_buffers; ++k) {
        float2 *tx_vector = reinterpret_cast<float2 *>((uintptr_t)0x10);
        if (dynamic) tx_vector = pool;                 // memory->get()
        generator->get(&tx_vector);
        std::fwrite(tx_vector, sizeof(float2), p.buffer_len, fout);
    }
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    generator->close();
    delete generator;
    std::fclose(fout);
    if (pool) (void)hipHostFree(pool);
    std::printf("{\"harness\": \"tx_single_link\", \"buffers\": %lld, \"buffer_len\": %d, \"msamples_per_s_incl_file\": %.1f}\n",
                n_buffers, (int)p.buffer_len, (double)n_buffers * (double)p.buffer_len / sec / 1e6);
    return 0;
}

static int file_mode(int argc, char **argv) {
    if (argc < 5) {
        std::fprintf(stderr, "usage: rx_link file <config.txt> <in.c64> <out.c64> [pipe]\n");
        return 2;
    }
    const bool pipelined = argc > 5 && std::string(argv[5]) == "pipe";
    param p;
    p.mode = RX;
    p.rate = 0; p.gain = 0; p.bw = 0; p.tone = 0; p.samples = 0; p.delay = 0; p.burst_on = p.burst_off = 0;
    p.buffer_len = 0; p.tuning_mode = false; p.decim = 0; p.data_mem_mult = 1; p.fft_tones = 0; p.pf_average = 4;
    w_type mode = NODSP;
    int channels = -1;
    std::ifstream cfg(argv[2]);
    if (!cfg) { std::fprintf(stderr, "cannot read %s\n", argv[2]); return 2; }
    std::string line;
    while (std::getline(cfg, line)) {
        std::istringstream is(line);
        std::string key;
        if (!(is >> key)) continue;
        if (key == "mode") { std::string m; is >> m; mode = wave_from_string(m); }
        else if (key == "channels") is >> channels;
        else if (key == "rate") is >> p.rate;
        else if (key == "buffer_len") is >> p.buffer_len;
        else if (key == "decim") is >> p.decim;
        else if (key == "pf_average") is >> p.pf_average;
        else if (key == "fft_tones") is >> p.fft_tones;
        else if (key == "freq") { int v; while (is >> v) p.freq.push_back(v); }
        else if (key == "chirp_f") { int v; while (is >> v) p.chirp_f.push_back(v); }
        else if (key == "swipe_s") { int v; while (is >> v) p.swipe_s.push_back(v); }
        else if (key == "chirp_t") { float v; while (is >> v) p.chirp_t.push_back(v); }
    }
    if (channels < 0) channels = mode == CHIRP || mode == NOISE ? 1 : (int)p.freq.size();
    for (int k = 0; k < channels; ++k) p.wave_type.push_back(mode);
    const size_t L = p.buffer_len;
    FILE *fin = std::fopen(argv[3], "rb"), *fout = std::fopen(argv[4], "wb");
    if (!fin || !fout || L == 0) { std::fprintf(stderr, "cannot open the stream files\n"); return 2; }
    std::fseek(fin, 0, SEEK_END);
    const long long n_buffers = std::ftell(fin) / (long long)(L * sizeof(float2));
    std::fseek(fin, 0, SEEK_SET);
    p.samples = (size_t)n_buffers * L;

    RX_buffer_demodulator::device_index() = 0;
    RX_buffer_demodulator *demodulator = new RX_buffer_demodulator(&p);  // link_threads.cpp:121
    const size_t out_len = (size_t)gsdr_demod_out_capacity_of(demodulator);
    const int pool = GSDR_PIPELINE_DEPTH + 1;
    std::vector<float2 *> in_pool(pool), out_pool(pool);
    for (int i = 0; i < pool; ++i)
        if (hipHostMalloc((void **)&in_pool[i], L * sizeof(float2)) != hipSuccess ||
            hipHostMalloc((void **)&out_pool[i], out_len * sizeof(float2)) != hipSuccess) {
            std::fprintf(stderr, "pinned allocation failed\n");
            return 1;
        }
    std::vector<int> lengths;
    std::queue<int> in_flight;
    auto retire = [&] {
        const int slot = in_flight.front();
        in_flight.pop();
        const int n = demodulator->wait();
        lengths.push_back(n);
        std::fwrite(out_pool[slot], sizeof(float2), (size_t)n, fout);
    };
    size_t recv_samples = 0;
    for (long long k = 0; recv_samples < p.samples; ++k) {                       // :647
        const int slot = (int)(k % pool);
        if (pipelined && (int)in_flight.size() == GSDR_PIPELINE_DEPTH) retire();  // frees slot k % pool
        if (std::fread(in_pool[slot], sizeof(float2), L, fin) != L) { std::fprintf(stderr, "short read\n"); return 1; }
        recv_samples += L;                                                        // :660
        float2 *in = in_pool[slot], *out = out_pool[slot];
        if (pipelined) {
            if (!demodulator->submit(&in, &out)) { std::fprintf(stderr, "submit failed\n"); return 1; }
            in_flight.push(slot);
        } else {
            const int n = demodulator->process(&in, &out);                        // :666
            lengths.push_back(n);
            std::fwrite(out, sizeof(float2), (size_t)n, fout);
        }
    }
    while (!in_flight.empty()) retire();
    demodulator->close();
    std::fclose(fin);
    std::fclose(fout);
    std::printf("{\"harness\": \"rx_single_link file%s\", \"channels\": %d, \"buffers\": %lld, \"lengths\": [",
                pipelined ? " (submit/wait)" : "", channels, n_buffers);
    for (size_t i = 0; i < lengths.size(); ++i) std::printf("%s%d", i ? ", " : "", lengths[i]);
    std::printf("]}\n");
    for (int i = 0; i < pool; ++i) { (void)hipHostFree(in_pool[i]); (void)hipHostFree(out_pool[i]); }
    return 0;
}

int main(int argc, char **argv) {
    if (argc > 1 && std::string(argv[1]) == "file") return file_mode(argc, argv);
    if (argc > 1 && std::string(argv[1]) == "tx") return tx_mode(argc, argv);
    const int n_tones = argc > 1 ? std::atoi(argv[1]) : 256;
    const int decim = argc > 2 ? std::atoi(argv[2]) : 100;
    const int n_buffers = argc > 3 ? std::atoi(argv[3]) : 200;
    const bool pipelined = argc > 4 && std::string(argv[4]) == "pipe";
    const size_t L = 1000000;  // DEFAULT_BUFFER_LEN, ref: headers/USRP_server_settings.hpp:102
    const int rate = 200000000;

    param p;
    p.mode = RX;
    p.rate = rate;
    p.buffer_len = L;
    p.decim = decim;
    p.pf_average = 4;
    p.fft_tones = 0;
    p.samples = (size_t)n_buffers * L;
    std::mt19937 rng(20251004);
    std::uniform_int_distribution<int> fd(-rate / 2 + 1, rate / 2 - 1);
    for (int k = 0; k < n_tones; ++k) {
        p.freq.push_back(fd(rng));  // scripts/get_noise.py:91
        p.wave_type.push_back(DIRECT);
    }
    p.data_mem_mult = (size_t)std::max(std::ceil(n_tones / std::max((double)decim, 1.0)), 1.0);  // USRP_files.py:683-684

    RX_buffer_demodulator::device_index() = 0;
    RX_buffer_demodulator *demodulator = new RX_buffer_demodulator(&p);  // link_threads.cpp:121

    // pinned pools (preallocator<float2> uses cudaMallocHost)
    const int pool = 8;
    const size_t out_len = L * std::max<size_t>(p.data_mem_mult, 1);
    std::vector<float2 *> in_pool(pool), out_pool(pool);
    for (int i = 0; i < pool; ++i) {
        if (hipHostMalloc((void **)&in_pool[i], L * sizeof(float2)) != hipSuccess ||
            hipHostMalloc((void **)&out_pool[i], out_len * sizeof(float2)) != hipSuccess) {
            std::fprintf(stderr, "pinned allocation failed\n");
            return 1;
        }
        std::normal_distribution<float> g(0.f, 0.1f);
        for (size_t j = 0; j < L; ++j) in_pool[i][j] = float2{g(rng), g(rng)};
    }

    BlockingQueue<RX_wrapper> rx_queue, stream_queue;
    BlockingQueue<float2 *> in_free, out_free;
    for (int i = 0; i < pool; ++i) { in_free.push(in_pool[i]); out_free.push(out_pool[i]); }

    // software RX thread: "receives" a buffer (already in pinned memory) and queues it
    std::thread rx_thread([&] {
        for (int k = 0; k < n_buffers; ++k) {
            RX_wrapper w{};
            w.buffer = in_free.pop();
            w.usrp_number = 0;
            w.front_end_code = 'B';  // RX on front-end A is tagged 'B', hardware_manager.cpp:1413-1418
            w.packet_number = k;
            w.length = (int)L;
            w.errors = 0;
            rx_queue.push(w);
        }
    });
    // streamer stand-in: recycles the output buffer
    std::atomic<long long> streamed{0};
    std::thread tx_thread([&] {
        for (int k = 0; k < n_buffers; ++k) {
            RX_wrapper w = stream_queue.pop();
            streamed += w.length;
            out_free.push(w.buffer);
        }
    });

    // ---- rx_single_link, link_threads.cpp:647-690 ----
    size_t recv_samples = 0;
    double worst_ms = 0, worst_steady_ms = 0;
    long long worst_at = -1, call_no = 0, calls_above_3ms = 0;
    const auto t0 = std::chrono::steady_clock::now();
    std::queue<std::pair<RX_wrapper, float2 *>> in_flight;  // pipelined mode: submitted, not yet waited for
    auto retire = [&] {
        auto pr = in_flight.front();
        in_flight.pop();
        pr.first.length = demodulator->wait();
        in_free.push(pr.first.buffer);
        pr.first.buffer = pr.second;
        stream_queue.push(pr.first);
    };
    while (recv_samples < p.samples) {
        RX_wrapper rx_buffer = rx_queue.pop();
        rx_buffer.channels = (int)demodulator->parameters->wave_type.size();  // :657
        recv_samples += rx_buffer.length;                                      // :660
        float2 *output_buffer = out_free.pop();                                // :663
        const auto a = std::chrono::steady_clock::now();
        if (pipelined) {
            if ((int)in_flight.size() == GSDR_PIPELINE_DEPTH) retire();
            demodulator->submit(&rx_buffer.buffer, &output_buffer);
            in_flight.push({rx_buffer, output_buffer});
        } else {
            rx_buffer.length = demodulator->process(&rx_buffer.buffer, &output_buffer);  // :666
            in_free.push(rx_buffer.buffer);                                    // :669
            rx_buffer.buffer = output_buffer;                                  // :672
            stream_queue.push(rx_buffer);                                      // :676
        }
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count();
        if (ms > worst_ms) { worst_ms = ms; worst_at = call_no; }
        if (ms > 3.0) calls_above_3ms++;
        if (call_no >= 2 * GSDR_PIPELINE_DEPTH && ms > worst_steady_ms) worst_steady_ms = ms;
        call_no++;
    }
    while (!in_flight.empty()) retire();
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    rx_thread.join();
    tx_thread.join();
    demodulator->close();

    const double msps = (double)recv_samples / sec / 1e6;
    std::printf("{\"harness\": \"rx_single_link%s\", \"tones\": %d, \"decim\": %d, \"buffers\": %d, "
                "\"msamples_per_s_pcie_inclusive\": %.1f, \"ms_per_buffer\": %.3f, \"worst_ms\": %.3f, \"worst_at_call\": %lld, "
                "\"worst_ms_after_first_%d_calls\": %.3f, \"calls_above_3ms\": %lld, "
                "\"realtime_factor_200Msps\": %.2f, \"streamed_samples\": %lld}\n",
                pipelined ? " (submit/wait)" : "", n_tones, decim, n_buffers, msps, sec / n_buffers * 1e3, worst_ms, worst_at,
                2 * GSDR_PIPELINE_DEPTH, worst_steady_ms, calls_above_3ms, msps / 200.0,
                streamed.load());
    for (int i = 0; i < pool; ++i) { (void)hipHostFree(in_pool[i]); (void)hipHostFree(out_pool[i]); }
    return 0;
}
