/*
 * gsdr.h -- C ABI of libgsdr.so, the MI355X-native RX demodulation engine.
 *
 * Drop-in boundary for the demodulation path of zjc263/GPU_SDR
 * (RX_buffer_demodulator, /root/reference/headers/USRP_demodulator.hpp:13-33).
 * Plain pointers and sizes only; no HIP, torch or C++ types cross this header.
 * The C++ class of the same name that the reference's server code compiles
 * against is provided header-only on top of this ABI in
 * include/USRP_demodulator.hpp; INTEGRATION.md shows the binding.
 *
 * Citations "ref:" are relative to /root/reference.
 */
#ifndef GSDR_H
#define GSDR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSDR_ABI_VERSION 1

/* complex64, layout-identical to CUDA/HIP float2 (x = re, y = im).
 * ref: every buffer in headers/USRP_demodulator.hpp is float2*. */
typedef struct gsdr_c64 { float x, y; } gsdr_c64;

/* ref: headers/USRP_server_settings.hpp:114  enum w_type */
enum gsdr_w_type {
    GSDR_TONES = 0, GSDR_CHIRP = 1, GSDR_NOISE = 2, GSDR_RAMP = 3,
    GSDR_NODSP = 4, GSDR_SWONLY = 5, GSDR_DIRECT = 6
};

/* Flattened view of the fields of `struct param` that the demodulator reads.
 * ref: headers/USRP_server_settings.hpp:130-167 (same names, same meaning).
 * Vectors become (pointer, count) pairs; the pointers are only read during
 * gsdr_demod_create(). */
typedef struct gsdr_param_c {
    int rate;                 /* param::rate  [samples/s]                     */
    long long decim;          /* param::decim (size_t)                        */
    int fft_tones;            /* param::fft_tones                             */
    long long pf_average;     /* param::pf_average (size_t)                   */
    long long buffer_len;     /* param::buffer_len (size_t)                   */
    const int *wave_type;     /* param::wave_type, values of gsdr_w_type      */
    int n_wave_type;
    const int *freq;          /* param::freq [Hz]                             */
    int n_freq;
    const float *chirp_t;     /* param::chirp_t [s]                           */
    int n_chirp_t;
    const int *chirp_f;       /* param::chirp_f [Hz]                          */
    int n_chirp_f;
    const int *swipe_s;       /* param::swipe_s                               */
    int n_swipe_s;
    int device_index;         /* server_settings::GPU_device_index
                                 (ref: USRP_server_settings.hpp:197); -1 =
                                 keep the calling thread's current device     */
} gsdr_param_c;

typedef struct gsdr_demod gsdr_demod;

/* ---- demodulator lifecycle (replaces RX_buffer_demodulator) ------------- */

/* ref: RX_buffer_demodulator::RX_buffer_demodulator, cpp/USRP_demodulator.cpp:7-327.
 * Returns NULL on failure; gsdr_last_error(NULL) then holds the reason.  Where
 * the reference calls exit(-1) (mixed wave types :36-39, more than one CHIRP
 * :31-34, unsupported type :322-325) this returns NULL with the same message. */
gsdr_demod *gsdr_demod_create(const gsdr_param_c *p);

/* ref: RX_buffer_demodulator::process, cpp/USRP_demodulator.cpp:330.
 * in_host : buffer_len complex64 (host, pinned or pageable), not modified.
 * out_host: room for gsdr_demod_out_capacity() complex64.
 * Synchronous like the reference (returns after the stream has drained).
 * Returns the number of valid complex samples written to out_host
 * ([sample][channel] interleaved), or -1 on a device error. */
int gsdr_demod_process(gsdr_demod *h, const gsdr_c64 *in_host, gsdr_c64 *out_host);

/* Same contract with DEVICE pointers, enqueued on `hip_stream` (a hipStream_t
 * passed as void*; NULL = HIP's null stream, as in every HIP call) and NOT
 * synchronised: the caller orders it against the producer of in_dev and the
 * consumer of out_dev through that stream.
 * The returned length is known on the host before the kernels finish.  This
 * is the entry the synthetic in-HBM source and bench.py use.
 * Consecutive calls may use different streams, and may be mixed with the
 * submit entries below: the state a call inherits (FIR carry, NCO index, raw
 * windows) is ordered on the device -- a call on another stream than its
 * predecessor first waits for it (one event; nothing on the usual path).
 * For that the handle remembers the stream of a call until the NEXT call on
 * the handle has been made: a stream passed here must stay alive that long
 * (or until gsdr_demod_close()); HIP itself faults on a destroyed stream. */
int gsdr_demod_process_device(gsdr_demod *h, const gsdr_c64 *in_dev,
                              gsdr_c64 *out_dev, void *hip_stream);

/* Pipelined host-pointer entry (an extension: the reference's process() is
 * synchronous and serialises H2D, compute and D2H of successive buffers, ref:
 * cpp/USRP_demodulator.cpp:393,462,555).  gsdr_demod_submit() enqueues the
 * upload on a copy stream, the kernels on the compute stream and the download
 * on a third stream, linked by events, and returns at once; up to
 * GSDR_PIPELINE_DEPTH buffers may be outstanding.  gsdr_demod_wait() blocks
 * until the OLDEST outstanding buffer is complete in its out_host and returns
 * its valid length (-1: device error, -2: nothing outstanding).  in_host and
 * out_host must stay valid until that wait returns and should be pinned
 * (hipHostMalloc), otherwise the copies degrade to synchronous ones.
 * Do not mix with gsdr_demod_process() while buffers are outstanding. */
#define GSDR_PIPELINE_DEPTH 4
int gsdr_demod_submit(gsdr_demod *h, const gsdr_c64 *in_host, gsdr_c64 *out_host);
int gsdr_demod_wait(gsdr_demod *h);
/* The same pipeline for a device-resident source (a synthetic generator, a GPU-direct
 * receiver): in_dev must be complete when the call is made, out_dev must be a different
 * buffer for every outstanding call; gsdr_demod_wait() returns when out_dev is complete.
 * Both submit entries run the DIRECT and TONES/NOISE kernels of consecutive buffers on
 * three streams in turn: the next buffer starts on the compute units the last workgroups of this one
 * leave idle (GSDR_PIPE_OVERLAP=0: strictly one after the other). */
int gsdr_demod_submit_device(gsdr_demod *h, const gsdr_c64 *in_dev, gsdr_c64 *out_dev);

/* Creates now what the entries above would otherwise create on first use (device
 * staging buffers of the host-pointer entries, the streams, events and per-slot
 * buffers of the pipelined ones), so that the first buffers of a measurement are not
 * late: the reference's constructor allocates everything up front as well (ref:
 * cpp/USRP_demodulator.cpp:59-119), and include/USRP_demodulator.hpp calls this from
 * its constructor.  `what` is a bit set; returns 0 or -1. */
#define GSDR_PREPARE_HOST 1           /* gsdr_demod_process                  */
#define GSDR_PREPARE_PIPELINE 2       /* gsdr_demod_submit_device / _wait    */
#define GSDR_PREPARE_PIPELINE_HOST 4  /* gsdr_demod_submit / _wait           */
/* GSDR_PREPARE_REHEARSE: also runs a throw-away twin of this demodulator (same parameters) through a
 * few buffers of zeros on every entry, then closes it.  The first launches of a kernel, the first
 * asynchronous copies from a pinned buffer and the first uses of a stream cost the process 5 - 7 ms
 * each, once (measured: calls 0, 1 and 6 of the first handle's submit() loop; later handles of the
 * process show none): the rehearsal pays them at set-up time, so that the first packets are not late.
 * The state of `h` itself (carry, NCO phase, call count) is untouched. */
#define GSDR_PREPARE_REHEARSE 8
int gsdr_demod_prepare(gsdr_demod *h, int what);

/* ref: RX_buffer_demodulator::close, cpp/USRP_demodulator.cpp:333 (+ :466-698).
 * Frees every device allocation and the stream, then the handle itself. */
void gsdr_demod_close(gsdr_demod *h);

/* Last error text of a handle, or of the calling thread's last failed
 * gsdr_demod_create() when h == NULL.  Never NULL. */
const char *gsdr_last_error(const gsdr_demod *h);

/* ---- introspection ------------------------------------------------------ */
int gsdr_abi_version(void);
/* "abi 1; arch gfx950; timing_build 0".  timing_build 1 marks an ablation build
 * (-DGSDR_TIMING_BUILD, scratch/ only) whose kernels can be told to skip work;
 * the shipped library is always 0 and bench.py refuses anything else. */
const char *gsdr_build_info(void);
/* Diagnostic: the GSDR_PFB_* environment switches (A/B runs, kernel variants in the tests) are read
 * once and cached -- a launch does not walk the environment.  This makes their next use read them again.  Handles
 * created earlier keep the kernel NAME they reported; which kernel a call runs follows the new values. */
void gsdr_reload_env(void);
/* One-line JSON object describing the engine this handle resolved to: mode,
 * dominant kernel, kernel family, row tiles per workgroup, pipeline streams,
 * and every GSDR_* environment variable that was set in the process (the
 * tuning knobs are read at create time).  Returns the text length (truncated
 * to cap-1 characters + NUL). */
int gsdr_demod_describe(const gsdr_demod *h, char *buf, int cap);
/* gsdr_w_type actually dispatched (ref: USRP_demodulator.cpp:19-25,56). */
int gsdr_demod_mode(const gsdr_demod *h);
/* parameters->wave_type.size(), what rx_single_link stores in
 * RX_wrapper.channels (ref: cpp/USRP_server_link_threads.cpp:657). */
int gsdr_demod_channels(const gsdr_demod *h);
/* Upper bound of the value process() can return for this configuration. */
long long gsdr_demod_out_capacity(const gsdr_demod *h);
/* RX_buffer_demodulator::fcut (ref: USRP_demodulator.cpp:131). */
float gsdr_demod_fcut(const gsdr_demod *h);
/* Copies the FIR taps / PFB window / VNA profile in use (real part) into w
 * (up to cap floats); returns its length. */
int gsdr_demod_get_window(const gsdr_demod *h, float *w, int cap);
/* TONES: copies the tone->FFT-bin table (ref: USRP_demodulator.cpp:726-733). */
int gsdr_demod_get_bins(const gsdr_demod *h, int *bins, int cap);

/* Per-kernel device timing of the dominant kernel with hipEvents recorded on
 * the launch stream.  enable != 0 starts a fresh accumulation; enable = n > 1
 * times every n-th launch only (a pair of events costs a few microseconds of
 * stream time, which a throughput measurement should not carry on every step). */
void gsdr_demod_profile_enable(gsdr_demod *h, int enable);
/* Synchronises the recorded events; returns the number of timed launches and
 * their summed duration in milliseconds. */
int gsdr_demod_profile_read(gsdr_demod *h, double *total_ms);
/* Name of the dominant kernel for the active mode (as rocprofv3 reports it). */
const char *gsdr_demod_kernel_name(const gsdr_demod *h);

/* ---- host-side pieces of the path (usable without a GPU) ---------------- */
/* ref: make_sinc_window, cpp/kernels.cu:258-310 (real part; imag is 0). */
void gsdr_make_sinc_window(int length, float fc, float *w);
/* ref: make_flat_window, cpp/kernels.cu:208-253. */
void gsdr_make_flat_window(int length, int side, float *w);

/* ref: class buffer_helper, cpp/USRP_server_memory_management.cpp:104-156,
 * headers/USRP_server_memory_management.hpp:77-101 (same field names). */
typedef struct gsdr_buffer_helper {
    int n_tones, eff_length, buffer_len, average, n_eff_tones;
    int new_0, copy_size, current_batch, spare_samples, spare_begin;
} gsdr_buffer_helper;
void gsdr_buffer_helper_init(gsdr_buffer_helper *b, int n_tones, int buffer_len,
                             int average, int n_eff_tones);
void gsdr_buffer_helper_update(gsdr_buffer_helper *b);

/* ref: class VNA_decimator_helper, cpp/USRP_server_memory_management.cpp:30-56. */
typedef struct gsdr_vna_helper {
    int valid_size, new0, total_len, spare_begin, ppt, buffer_len;
} gsdr_vna_helper;
void gsdr_vna_helper_init(gsdr_vna_helper *v, int ppt, int buffer_len);
void gsdr_vna_helper_update(gsdr_vna_helper *v);

/* ref: upload_multitone_parameters, cpp/USRP_demodulator.cpp:722-733.
 * bins[u] = -1 when no bin matches (the reference leaves it uninitialised). */
void gsdr_pfb_tone_bins(int rate, int fft_tones, const int *freq, int n, int *bins);
/* ref: cpp/USRP_demodulator.cpp:706 */
int gsdr_pfb_batching(long long buffer_len, int fft_tones, long long pf_average);
/* The radix stages of the in-LDS transform of an fft_tones-point frame (TONES / NOISE as one launch per buffer:
 * polyphase filter + FFT inside the LDS + bin selection): writes the radices into radices[0..15] in stage order and
 * returns their number, or -1 when the length has no such plan (more than 8192 points, a prime factor above 127).
 * Such lengths up to 4096 points still run inside the LDS, through Bluestein's identity at the padded length
 * 2^ceil(log2(2 fft_tones - 1)); longer ones through the global-memory FFT stages -- for TONES and NOISE alike
 * (round 3: every TONES frame length takes filter + FFT + selection, as the reference's cufftPlanMany takes any
 * fft_tones).  Host only.  Replaces the reference's cufftPlanMany for the PFB
 * (ref: cpp/USRP_demodulator.cpp:149-152, :292-295). */
int gsdr_pfb_lds_stages(int fft_tones, int *radices);

/* ref: struct chirp_parameter, headers/kernels.cuh:58-64 and its derivation
 * in cpp/USRP_demodulator.cpp:192-214. */
typedef struct gsdr_chirp_param {
    unsigned long long num_steps, length;
    unsigned int chirpness;
    int f0;
} gsdr_chirp_param;
void gsdr_chirp_derive(int rate, int freq0, int chirp_f, int swipe_s,
                       float chirp_t, gsdr_chirp_param *cp);
/* The TX side's own derivation (ref: cpp/USRP_buffer_generator.cpp:107-129): as above, but a step shorter than
 * one sample also resets num_steps to chirp_t * rate before the slope is taken from it (:118-125). */
void gsdr_chirp_derive_tx(int rate, int freq0, int chirp_f, int swipe_s,
                          float chirp_t, gsdr_chirp_param *cp);

/* ---- the pyUSRP command surface (host only) ------------------------------
 * One JSON command per measurement arrives on the async socket, framed by an
 * 8-byte header; results leave on the data socket as 21-byte header + complex64
 * payload (wire formats: SURVEY.md section 9). */
typedef struct gsdr_command gsdr_command;   /* a parsed and checked usrp_param */

/* ref: string2param (cpp/USRP_JSON_interpreter.cpp:19-257) followed by chk_param
 * (:268-439): every key of A_TXRX, B_TXRX, A_RX2, B_RX2 is mandatory; buffer_len
 * 0 or outside [50000, 6000000] -> 1000000; PFB modes force pf_average >= 1 and
 * fft_tones >= 2; |freq|, |chirp_f| <= rate for TONES/CHIRP.  NULL = the command
 * must be nack'ed; gsdr_command_error() says why. */
gsdr_command *gsdr_command_parse(const char *json, int len);
void gsdr_command_free(gsdr_command *c);
const char *gsdr_command_error(void);
int gsdr_command_device(const gsdr_command *c);            /* usrp_param::usrp_number */

/* the fields of `struct param` that gsdr_param_c does not carry
 * (ref: headers/USRP_server_settings.hpp:130-167) */
typedef struct gsdr_antenna_info {
    int mode;                 /* ant_mode: 0 TX, 1 RX, 2 OFF */
    double rf;                /* param::tone                     */
    int gain, bw, tuning_mode;
    long long samples;
    double delay;
    float burst_on, burst_off;
    long long data_mem_mult;
    const float *ampl;
    int n_ampl;
} gsdr_antenna_info;
/* antenna: 0 A_TXRX, 1 B_TXRX, 2 A_RX2, 3 B_RX2 (usrp_param member order).  The
 * pointers inside *p and *info stay valid until gsdr_command_free(). */
int gsdr_command_antenna(const gsdr_command *c, int antenna, gsdr_param_c *p, gsdr_antenna_info *info);

/* ref: server_ack / server_nack (cpp/USRP_JSON_interpreter.cpp:441-457), in the
 * layout boost::property_tree::write_json produces.  Returns the text length. */
int gsdr_server_reply(int is_ack, const char *payload, char *buf, int cap);
/* ref: Async_server::format_header, cpp/USRP_server_network.cpp:497-501: {0, len} */
void gsdr_format_async_header(int message_len, unsigned char out[8]);
/* ref: RX_wrapper without its pointer (headers/USRP_server_settings.hpp:216-224)
 * and Sync_server::format_net_buffer (cpp/USRP_server_network.cpp:164-191) */
typedef struct gsdr_rx_header {
    int usrp_number;
    char front_end_code;
    int packet_number, length, errors, channels;
} gsdr_rx_header;
void gsdr_format_rx_header(const gsdr_rx_header *h, unsigned char out[21]);

/* ---- synthetic in-memory IQ source (replaces the UHD hardware manager for
 * benchmarking; shaped after software_rx_thread, ref:
 * cpp/USRP_hardware_manager.cpp:1331-1395) ------------------------------- */
/* Fills out_dev[0..n) on the device with
 *   sum_k ampl[k]*exp(i*(2*pi*freq[k]*((start+j) mod rate)/rate + phase[k]))
 *   + sigma*(g1 + i*g2)         (g: counter-based unit gaussians from `seed`)
 * freq/ampl/phase are HOST arrays of n_tones entries.  Returns after the
 * kernel has finished on `hip_stream` (setup helper, not part of the hot path). */
int gsdr_source_tones(gsdr_c64 *out_dev, long long n, long long start, int rate,
                      const int *freq, const float *ampl, const float *phase,
                      int n_tones, float sigma, unsigned long long seed,
                      void *hip_stream);
/* The tone set the reference's TX tone generator actually produces (host only; ref:
 * tone_gen, cpp/kernels.cu:617-635): tones are ASSIGNED to bins of a length-`rate`
 * spectrum (index f, or rate + f for f <= 0), so of tones on one bin the last wins, and
 * an index outside [0, rate) -- a 0 Hz tone lands on `rate` -- is dropped (the reference
 * writes it past its allocation; the inverse FFT never sees it).  Writes the surviving
 * tones (signed Hz, amplitude) to out_freq/out_ampl (room for n each), returns their
 * count.  Feed the result to gsdr_source_tones with phase 0 for the TX buffer
 * (ref: TX_buffer_generator TONES, cpp/USRP_buffer_generator.cpp:60-95,226-229). */
int gsdr_tx_tone_bins(int rate, const int *freq, const float *ampl, int n,
                      int *out_freq, float *out_ampl);

/* TX tone comb generator at scale (row f3).  ref: tone_gen (cpp/kernels.cu:589-684) builds one period of
 * `rate` samples by an inverse FFT when the generator is set up and TX_buffer_generator::get_from_tones
 * (cpp/USRP_buffer_generator.cpp:226-229) serves slices of it; this generator synthesises every buffer on
 * demand from exact integer phases (no 1.6 GB period at 200 Msps), faster than the stream needs it
 * (2048 tones: 17 x real time at 200 Msps).  freq/ampl are the tones gsdr_tx_tone_bins() returned (signed Hz,
 * one per bin), phase their initial phases in radians (NULL = 0).  x[s] = sum_k ampl_k e^(i phase_k)
 * e^(+2 pi i freq_k s / rate).  gsdr_txgen_tones_fill writes samples start .. start + n - 1 (start taken
 * mod rate) into device memory on hip_stream; asynchronous.  Returns 0 / -1 (gsdr_last_error(NULL)). */
typedef struct gsdr_txgen gsdr_txgen;
gsdr_txgen *gsdr_txgen_tones_create(int rate, const int *freq, const float *ampl, const float *phase, int n_tones,
                                    int device_index);
int gsdr_txgen_tones_fill(gsdr_txgen *g, gsdr_c64 *out_dev, long long n, long long start, void *hip_stream);
void gsdr_txgen_close(gsdr_txgen *g);

/* The reference's TX_buffer_generator as one object (ref: headers/USRP_buffer_generator.hpp:47-66,
 * cpp/USRP_buffer_generator.cpp:10-160): created from the TX parameters, every get() hands out the next
 * buffer_len samples.  TONES: the tone comb above (tone set as gsdr_tx_tone_bins(); the sample index wraps at
 * rate * ceil(buffer_len / rate), :60-75); CHIRP: the chirp_gen law (cpp/kernels.cu:335-372) scaled by ampl[0],
 * with the TX side's own num_steps reset (:111-115), the running index wrapping at num_steps * length.  Where
 * the reference exits (mixed or several CHIRP wave types, NODSP/SWONLY/RAMP/DIRECT)
 * create returns NULL with the same text in gsdr_last_error(NULL).  gsdr_txgen_get: to host memory
 * (synchronous, like the reference's get); gsdr_txgen_get_device: to device memory on hip_stream. */
gsdr_txgen *gsdr_txgen_create(const gsdr_param_c *p, const float *ampl, int n_ampl);
int gsdr_txgen_get(gsdr_txgen *g, gsdr_c64 *out_host);
/* TONES as the reference hands them out (ref: get_from_tones, cpp/USRP_buffer_generator.cpp:226-229): the next
 * buffer_len samples as a pointer INTO the generator's own host memory -- one period of the comb plus one buffer
 * (rate * ceil(buffer_len / rate) + buffer_len samples, :60-95), valid and unchanged until gsdr_txgen_close().  The
 * caller's buffer is not touched: tx_single_link passes an unallocated pointer for TONES and queues what it gets
 * back (ref: cpp/USRP_server_link_threads.cpp:568-584).  The period is generated on the GPU the first time it is
 * needed, or by gsdr_txgen_prepare_host() (what the reference's constructor does with its inverse FFT of length
 * `rate`: 1.6 GB of host memory at 200 Msps, there as here).  NULL on failure (gsdr_last_error(NULL)). */
const gsdr_c64 *gsdr_txgen_get_ptr(gsdr_txgen *g);
int gsdr_txgen_prepare_host(gsdr_txgen *g);
/* GSDR_TONES or GSDR_CHIRP (a NOISE request is TONES: the reference's `case NOISE` falls through, :52-58) */
int gsdr_txgen_mode(const gsdr_txgen *g);
int gsdr_txgen_get_device(gsdr_txgen *g, gsdr_c64 *out_dev, void *hip_stream);
long long gsdr_txgen_buffer_len(const gsdr_txgen *g);
/* TX chirp law (ref: chirp_gen, cpp/kernels.cu:335-372) written to device. */
int gsdr_source_chirp(gsdr_c64 *out_dev, long long n, unsigned long long last_index,
                      const gsdr_chirp_param *cp, float scale, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* GSDR_H */
