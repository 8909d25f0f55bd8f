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
// What it drops: UHD, HDF5 writer, reconnect threads, burst mode, delays, logging.
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

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
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

struct FrontEnd {            // one RX demodulator fed by its TX generator (or silence)
    gsdr_demod *dem = nullptr;
    gsdr_param_c rx{}, tx{};
    gsdr_antenna_info rxi{}, txi{};
    bool has_tx = false;
    gsdr_chirp_param tx_chirp{};
    gsdr_txgen *tx_tones = nullptr;   // TX tone comb generator (created with the first buffer)
    char code = 'B';
    gsdr_c64 *d_in = nullptr, *d_out = nullptr, *h_out = nullptr;
    long long produced = 0;   // TX sample counter
    long long recv_samples = 0;
    int packet = 0;
};

static bool run_measurement(const gsdr_command *cmd, int data_fd, int device) {
    FrontEnd fe[2];
    int active = 0;
    for (int f = 0; f < 2; ++f) {   // antennas: 0 A_TXRX, 1 B_TXRX, 2 A_RX2, 3 B_RX2
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
        F.code = f == 0 ? 'B' : 'D';
        if (tx >= 0) {
            F.tx = p[tx];
            F.txi = info[tx];
            F.has_tx = true;
            if (F.tx.n_wave_type > 0 && F.tx.wave_type[0] == GSDR_CHIRP && F.tx.n_freq && F.tx.n_chirp_f &&
                F.tx.n_swipe_s && F.tx.n_chirp_t) {
                gsdr_chirp_derive(F.tx.rate, F.tx.freq[0], F.tx.chirp_f[0], F.tx.swipe_s[0], F.tx.chirp_t[0],
                                  &F.tx_chirp);
            }
        }
        F.dem = gsdr_demod_create(&F.rx);
        if (!F.dem) {
            std::fprintf(stderr, "ERROR: %s\n", gsdr_last_error(nullptr));
            continue;
        }
        const size_t L = (size_t)F.rx.buffer_len, cap = (size_t)gsdr_demod_out_capacity(F.dem);
        if (hipMalloc((void **)&F.d_in, L * 8) != hipSuccess || hipMalloc((void **)&F.d_out, cap * 8) != hipSuccess ||
            hipHostMalloc((void **)&F.h_out, cap * 8) != hipSuccess) {
            std::fprintf(stderr, "ERROR: allocation failed\n");
            return false;
        }
        (void)hipMemset(F.d_in, 0, L * 8);
        active++;
    }
    bool ok = true;
    bool more = active > 0;
    while (more && ok) {
        more = false;
        for (FrontEnd &F : fe) {
            if (!F.dem || F.recv_samples >= F.rxi.samples) continue;   // link_threads.cpp:647
            const long long L = F.rx.buffer_len;
            // software TX -> RX loop-back
            if (F.has_tx && F.tx.n_wave_type > 0 && F.tx.wave_type[0] == GSDR_TONES) {
                // (gsdr_command_parse has checked this already; the source reads n entries of both)
                if (F.txi.n_ampl < F.tx.n_wave_type || F.tx.n_freq < F.tx.n_wave_type) {
                    std::fprintf(stderr, "ERROR: TX TONES needs one ampl and one freq per wave_type entry\n");
                    ok = false;
                    break;
                }
                if (!F.tx_tones) {
                    // the tones the reference's tone_gen really produces (bin assignment, kernels.cu:617-635)
                    std::vector<int> tf((size_t)F.tx.n_wave_type);
                    std::vector<float> ta((size_t)F.tx.n_wave_type);
                    const int nt = gsdr_tx_tone_bins(F.tx.rate, F.tx.freq, F.txi.ampl, F.tx.n_wave_type, tf.data(), ta.data());
                    F.tx_tones = gsdr_txgen_tones_create(F.tx.rate, tf.data(), ta.data(), nullptr, nt > 0 ? nt : 0, device);
                    if (!F.tx_tones) {
                        std::fprintf(stderr, "ERROR: TX generator: %s\n", gsdr_last_error(nullptr));
                        ok = false;
                        break;
                    }
                }
                if (gsdr_txgen_tones_fill(F.tx_tones, F.d_in, L, F.produced, nullptr) != 0) {
                    std::fprintf(stderr, "ERROR: TX generator: %s\n", gsdr_last_error(nullptr));
                    ok = false;
                    break;
                }
            } else if (F.has_tx && F.tx.n_wave_type > 0 && F.tx.wave_type[0] == GSDR_CHIRP) {
                gsdr_source_chirp(F.d_in, L, (unsigned long long)F.produced, &F.tx_chirp,
                                  F.txi.n_ampl ? F.txi.ampl[0] : 1.f, nullptr);
            }
            F.produced += L;
            F.recv_samples += L;                                          // :660
            const int n = gsdr_demod_process_device(F.dem, F.d_in, F.d_out, nullptr);  // :666
            if (n < 0) {
                std::fprintf(stderr, "ERROR: demodulator: %s\n", gsdr_last_error(F.dem));
                ok = false;
                break;
            }
            if (hipMemcpy(F.h_out, F.d_out, (size_t)n * 8, hipMemcpyDeviceToHost) != hipSuccess) {
                ok = false;
                break;
            }
            gsdr_rx_header h{gsdr_command_device(cmd), F.code, F.packet++, n, 0, gsdr_demod_channels(F.dem)};
            unsigned char head[21];
            gsdr_format_rx_header(&h, head);
            if (!write_all(data_fd, head, 21) || !write_all(data_fd, F.h_out, (size_t)n * 8)) {
                ok = false;
                break;
            }
            if (F.recv_samples < F.rxi.samples) more = true;
        }
    }
    for (FrontEnd &F : fe) {
        if (F.dem) gsdr_demod_close(F.dem);
        if (F.tx_tones) gsdr_txgen_close(F.tx_tones);
        if (F.d_in) (void)hipFree(F.d_in);
        if (F.d_out) (void)hipFree(F.d_out);
        if (F.h_out) (void)hipHostFree(F.h_out);
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
