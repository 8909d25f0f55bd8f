// gsdr_server.cpp -- a minimal GPU_SDR server in software-loopback mode around
// libgsdr: the pyUSRP JSON/TCP command surface end to end, without UHD.
//
// What it keeps of the reference (citations relative to /root/reference):
//   * async (command) socket, default :22001 -- every message = int32 0 + int32
//     len + JSON (cpp/USRP_server_network.cpp:497-501,523-547); reply
//     {"type":"ack"|"nack","payload":...} after parsing ("Message received" /
//     "Cannot convert JSON to params") and "EOM: end of measurement" when the
//     measurement is over (cpp/usrp_server.cpp:80-104);
//   * sync (data) socket, default :61360 -- per packet the 21-byte RX_wrapper
//     header + length*8 bytes of complex64, [sample][channel]
//     (cpp/USRP_server_network.cpp:164-191,244-256);
//   * --sw_loop: the TX generator of a front end feeds its RX demodulator
//     (cpp/USRP_hardware_manager.cpp:1071-1123,1331-1395);
//   * the rx_single_link loop: count INPUT samples, always consume buffer_len,
//     channels = wave_type.size() (cpp/USRP_server_link_threads.cpp:647-690);
//     RX on front end A is tagged 'B', on B 'D' (hardware_manager.cpp:1413-1418).
//   * the thread links: one tx_single_link and one rx_single_link thread per active front-end (A and B may both be
//     active in one command, cpp/USRP_server_link_threads.cpp:121,136,325-336), pinned pools, one tcp_streamer
//     thread draining the stream queue (cpp/USRP_server_network.cpp:195-308); the demodulators run through the
//     overlapped entry, gsdr_demod_submit() / gsdr_demod_wait();
//   * burst mode's buffer length, buffer_len = burst_on * rate (link_threads.cpp:99-102).
// What it drops: UHD, HDF5 writer, reconnect threads, burst timing, delays, logging.
//
//   hipcc -O2 -std=c++17 -Iinclude tools/gsdr_server.cpp -Lgpu_sdr_amd -lgsdr \
//         -Wl,-rpath,$PWD/gpu_sdr_amd -o gpu_sdr_amd/gsdr_server
//   gpu_sdr_amd/gsdr_server [--async 22001] [--data 61360] [--device 0] [--once]
#include <arpa/inet.h>
#include <hip/hip_runtime.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <unistd.h>

#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "gsdr.h"

static bool read_all(int fd, void *buf, size_t n) {
    char *p = (char *)buf;
    while (n) {
        ssize_t r = ::recv(fd, p, n, 0);
        if (r <= 0) return false;
        p += r;
        n -= (size_t)r;
    }
    return true;
}
static bool write_all(int fd, const void *buf, size_t n) {
    const char *p = (const char *)buf;
    while (n) {
        ssize_t r = ::send(fd, p, n, MSG_NOSIGNAL);
        if (r <= 0) return false;
        p += r;
        n -= (size_t)r;
    }
    return true;
}
static int listen_on(int port) {
    int fd = ::socket(AF_INET, SOCK_STREAM, 0);
    int one = 1;
    ::setsockopt(fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
    sockaddr_in a{};
    a.sin_family = AF_INET;
    a.sin_addr.s_addr = htonl(INADDR_ANY);
    a.sin_port = htons((uint16_t)port);
    if (::bind(fd, (sockaddr *)&a, sizeof(a)) != 0 || ::listen(fd, 1) != 0) {
        std::perror("bind/listen");
        std::exit(1);
    }
    return fd;
}
static bool send_reply(int fd, bool ack, const char *payload) {
    char text[512];
    const int n = gsdr_server_reply(ack ? 1 : 0, payload, text, sizeof(text));
    unsigned char head[8];
    gsdr_format_async_header(n, head);
    return write_all(fd, head, 8) && write_all(fd, text, (size_t)n);
}

// ---- the thread-link shape of the reference (cpp/USRP_server_link_threads.cpp) -------------------------------
// TXRX::set (:73-235) creates the demodulators and the pinned pools on the main thread; TXRX::start (:238-431)
// spawns one tx_single_link (:540-604) and one rx_single_link (:605-702) thread per active front-end; a single
// Sync_server::tcp_streamer (cpp/USRP_server_network.cpp:195-308) drains the stream queue to the data socket and
// hands every buffer back to its pool; TXRX::stop (:435-538) joins.  In software loop-back (--sw_loop) the TX
// buffers are the RX buffers (cpp/USRP_hardware_manager.cpp:1071-1123, 1331-1395).
template <typename T>
class BlockingQueue {          // stands in for the boost::lockfree queues + polling sleeps of the reference
   public:
    void push(T v) {
        { std::lock_guard<std::mutex> l(m_); q_.push_back(v); }
        c_.notify_one();
    }
    bool pop(T &v) {           // false: closed and empty
        std::unique_lock<std::mutex> l(m_);
        c_.wait(l, [&] { return !q_.empty() || closed_; });
        if (q_.empty()) return false;
        v = q_.front();
        q_.pop_front();
        return true;
    }
    void close() {
        { std::lock_guard<std::mutex> l(m_); closed_ = true; }
        c_.notify_all();
    }
   private:
    std::mutex m_;
    std::condition_variable c_;
    std::deque<T> q_;
    bool closed_ = false;
};

class PinnedPool {             // preallocator<float2> (headers/USRP_server_memory_management.hpp:103-273): get() / trash()
   public:
    bool init(size_t samples, int count) {
        for (int i = 0; i < count; ++i) {
            gsdr_c64 *p = nullptr;
            if (hipHostMalloc((void **)&p, (samples ? samples : 1) * sizeof(gsdr_c64)) != hipSuccess) return false;
            all_.push_back(p);
            free_.push(p);
        }
        return true;
    }
    gsdr_c64 *get() {
        gsdr_c64 *p = nullptr;
        free_.pop(p);
        return p;
    }
    void trash(gsdr_c64 *p) { free_.push(p); }
    void close() {
        for (gsdr_c64 *p : all_) (void)hipHostFree(p);
        all_.clear();
    }
   private:
    std::vector<gsdr_c64 *> all_;
    BlockingQueue<gsdr_c64 *> free_;
};

struct Packet {                // RX_wrapper + where its buffer goes back to
    gsdr_c64 *buffer;
    PinnedPool *pool;
    gsdr_rx_header h;
};

struct FrontEnd {              // one RX demodulator fed by its TX generator (or silence)
    gsdr_demod *dem = nullptr;
    gsdr_txgen *gen = nullptr;
    gsdr_param_c rx{}, tx{};
    gsdr_antenna_info rxi{}, txi{};
    bool has_tx = false;
    char code = 'B';
    int usrp = 0;
    long long n_buffers = 0;          // ceil(samples / buffer_len): rx_single_link counts INPUT samples (:660)
    PinnedPool in_pool, out_pool;
    BlockingQueue<gsdr_c64 *> rx_queue;   // what the (software) receiver hands to rx_single_link
    std::thread tx_thr, rx_thr;
    std::atomic<bool> ok{true};
};

// tx_single_link (:568-584) + software_tx/rx_thread: generator->get() into a pool buffer, which becomes an RX buffer
static void tx_link(FrontEnd *F, int device) {
    (void)hipSetDevice(device);
    for (long long k = 0; k < F->n_buffers; ++k) {
        gsdr_c64 *buf = F->in_pool.get();
        if (F->gen) {
            if (gsdr_txgen_get(F->gen, buf) != 0) {
                std::fprintf(stderr, "ERROR: TX generator: %s\n", gsdr_last_error(nullptr));
                F->ok = false;
                std::memset(buf, 0, (size_t)F->rx.buffer_len * sizeof(gsdr_c64));
            }
        } else {
            std::memset(buf, 0, (size_t)F->rx.buffer_len * sizeof(gsdr_c64));     // RX without TX: silence
        }
        F->rx_queue.push(buf);
    }
}

// rx_single_link (:605-702): pop an RX buffer, demodulate, hand the result to the streamer.  The demodulator runs
// through the overlapped entry -- submit() / wait(), GSDR_PIPELINE_DEPTH buffers outstanding -- on the pinned pools.
static void rx_link(FrontEnd *F, BlockingQueue<Packet> *stream_queue, int device) {
    (void)hipSetDevice(device);
    struct InFlight { gsdr_c64 *in, *out; };
    std::deque<InFlight> pending;
    int packet = 0;
    auto finish_oldest = [&]() {
        const InFlight f = pending.front();
        pending.pop_front();
        int n = gsdr_demod_wait(F->dem);
        if (n < 0) {
            std::fprintf(stderr, "ERROR: demodulator: %s\n", gsdr_last_error(F->dem));
            F->ok = false;
            n = 0;
        }
        F->in_pool.trash(f.in);                                                   // input_memory->trash (:669)
        Packet p{f.out, &F->out_pool, gsdr_rx_header{F->usrp, F->code, packet++, n, 0, gsdr_demod_channels(F->dem)}};
        stream_queue->push(p);                                                    // :676
    };
    for (long long k = 0; k < F->n_buffers; ++k) {
        gsdr_c64 *in = nullptr;
        if (!F->rx_queue.pop(in)) break;
        gsdr_c64 *out = F->out_pool.get();                                        // output_memory->get() (:663)
        if ((int)pending.size() == GSDR_PIPELINE_DEPTH) finish_oldest();
        if (gsdr_demod_submit(F->dem, in, out) != 0) {
            std::fprintf(stderr, "ERROR: demodulator: %s\n", gsdr_last_error(F->dem));
            F->ok = false;
            F->in_pool.trash(in);
            F->out_pool.trash(out);
            continue;
        }
        pending.push_back(InFlight{in, out});
    }
    while (!pending.empty()) finish_oldest();
}

// Sync_server::tcp_streamer (network.cpp:195-308): header + payload per packet, the buffer back to its pool
static void tcp_streamer(BlockingQueue<Packet> *stream_queue, int data_fd, std::atomic<bool> *ok) {
    Packet p;
    while (stream_queue->pop(p)) {
        unsigned char head[21];
        gsdr_format_rx_header(&p.h, head);
        if (ok->load() && (!write_all(data_fd, head, 21) || !write_all(data_fd, p.buffer, (size_t)p.h.length * 8))) *ok = false;
        p.pool->trash(p.buffer);
    }
}

static bool run_measurement(const gsdr_command *cmd, int data_fd, int device) {
    FrontEnd fe[2];
    std::vector<FrontEnd *> active;
    bool ok = true;
    for (int f = 0; f < 2 && ok; ++f) {   // antennas: 0 A_TXRX, 1 B_TXRX, 2 A_RX2, 3 B_RX2
        gsdr_param_c p[2];
        gsdr_antenna_info info[2];
        gsdr_command_antenna(cmd, f, &p[0], &info[0]);
        gsdr_command_antenna(cmd, f + 2, &p[1], &info[1]);
        int rx = -1, tx = -1;
        for (int k = 0; k < 2; ++k) {
            if (info[k].mode == 1 && rx < 0) rx = k;
            if (info[k].mode == 0 && tx < 0) tx = k;
        }
        if (rx < 0) continue;
        FrontEnd &F = fe[f];
        F.rx = p[rx];
        F.rxi = info[rx];
        F.rx.device_index = device;
        F.code = f == 0 ? 'B' : 'D';                            // hardware_manager.cpp:1413-1418
        F.usrp = gsdr_command_device(cmd);
        // burst mode: the buffer is one burst long (link_threads.cpp:99-102)
        if (F.rxi.burst_on != 0.f) F.rx.buffer_len = (long long)(F.rxi.burst_on * (float)F.rx.rate);
        if (tx >= 0) {
            F.tx = p[tx];
            F.txi = info[tx];
            F.tx.device_index = device;
            if (F.txi.burst_on != 0.f) F.tx.buffer_len = (long long)(F.txi.burst_on * (float)F.tx.rate);
            F.has_tx = true;
        }
        if (F.rx.buffer_len < 1) {
            std::fprintf(stderr, "ERROR: empty RX buffer\n");
            ok = false;
            break;
        }
        F.dem = gsdr_demod_create(&F.rx);                       // TXRX::set: new RX_buffer_demodulator (:121,:136)
        if (!F.dem) {
            std::fprintf(stderr, "ERROR: %s\n", gsdr_last_error(nullptr));
            continue;
        }
        if (gsdr_demod_prepare(F.dem, GSDR_PREPARE_PIPELINE | GSDR_PREPARE_PIPELINE_HOST) != 0)
            std::fprintf(stderr, "WARNING: demodulator: %s\n", gsdr_last_error(F.dem));
        if (F.has_tx && F.tx.n_wave_type > 0) {
            // software loop-back: the TX buffer IS the RX buffer, so the generator makes RX-sized buffers
            gsdr_param_c txp = F.tx;
            txp.buffer_len = F.rx.buffer_len;
            F.gen = gsdr_txgen_create(&txp, F.txi.ampl, F.txi.n_ampl);
            if (!F.gen) std::fprintf(stderr, "WARNING: TX generator: %s (RX runs on silence)\n", gsdr_last_error(nullptr));
        }
        const long long L = F.rx.buffer_len;
        F.n_buffers = F.rxi.samples > 0 ? (F.rxi.samples + L - 1) / L : 0;
        const size_t cap = (size_t)gsdr_demod_out_capacity(F.dem);
        // pools: RX buffers (preallocator(A_rx_buffer_len, RX_QUEUE_LENGTH), :114) and outputs (:143-150)
        if (!F.in_pool.init((size_t)L, GSDR_PIPELINE_DEPTH + 3) || !F.out_pool.init(cap, GSDR_PIPELINE_DEPTH + 5)) {
            std::fprintf(stderr, "ERROR: allocation failed\n");
            ok = false;
            break;
        }
        active.push_back(&F);
    }
    if (ok && !active.empty()) {
        BlockingQueue<Packet> stream_queue;
        std::atomic<bool> net_ok{true};
        std::thread streamer(tcp_streamer, &stream_queue, data_fd, &net_ok);
        for (FrontEnd *F : active) {                            // TXRX::start (:238-431)
            F->tx_thr = std::thread(tx_link, F, device);
            F->rx_thr = std::thread(rx_link, F, &stream_queue, device);
        }
        for (FrontEnd *F : active) {                            // TXRX::stop (:435-538)
            F->tx_thr.join();
            F->rx_thr.join();
            ok = ok && F->ok.load();
        }
        stream_queue.close();
        streamer.join();
        ok = ok && net_ok.load();
    }
    for (FrontEnd &F : fe) {
        if (F.dem) gsdr_demod_close(F.dem);                     // close() in TXRX::stop (:475,:485)
        if (F.gen) gsdr_txgen_close(F.gen);
        F.in_pool.close();
        F.out_pool.close();
    }
    return ok;
}

int main(int argc, char **argv) {
    int async_port = 22001, data_port = 61360, device = 0;   // ref: cpp/USRP_server_settings.cpp:3-4
    bool once = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "--async" && i + 1 < argc) async_port = std::atoi(argv[++i]);
        else if (a == "--data" && i + 1 < argc) data_port = std::atoi(argv[++i]);
        else if (a == "--device" && i + 1 < argc) device = std::atoi(argv[++i]);
        else if (a == "--once") once = true;
        else if (a == "--sw_loop" || a == "--no_net" || a == "--fw") {}  // accepted for script compatibility
    }
    if (hipSetDevice(device) != hipSuccess) {
        std::fprintf(stderr, "ERROR: no GPU %d\n", device);
        return 1;
    }
    const int data_l = listen_on(data_port), async_l = listen_on(async_port);
    std::printf("gsdr_server: data :%d, async :%d, device %d (software loop-back)\n", data_port, async_port, device);
    std::fflush(stdout);
    // the reference blocks on the data connection first (Sync_server::connect), then on the async one
    const int data_fd = ::accept(data_l, nullptr, nullptr);
    const int async_fd = ::accept(async_l, nullptr, nullptr);
    int one = 1;
    ::setsockopt(async_fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
    for (;;) {
        int head[2];
        if (!read_all(async_fd, head, 8)) break;
        if (head[0] != 0) std::fprintf(stderr, "WARNING: Corrupted async header detected!\n");
        if (head[1] < 0 || head[1] > (64 << 20)) break;
        std::string json((size_t)head[1], '\0');
        if (!read_all(async_fd, &json[0], json.size())) break;
        gsdr_command *cmd = gsdr_command_parse(json.data(), (int)json.size());
        if (!cmd) {
            std::fprintf(stderr, "ERROR: %s\n", gsdr_command_error());
            if (!send_reply(async_fd, false, "Cannot convert JSON to params")) break;
            if (once) break;
            continue;
        }
        if (!send_reply(async_fd, true, "Message received")) break;
        const bool ok = run_measurement(cmd, data_fd, device);
        gsdr_command_free(cmd);
        if (!send_reply(async_fd, true, "EOM: end of measurement") || !ok) break;
        if (once) break;
    }
    ::close(async_fd);
    ::close(data_fd);
    ::close(async_l);
    ::close(data_l);
    return 0;
}
