/*
 * gsdr_oracle.h -- CPU restatement of the GPU_SDR RX demodulation path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (gpu_sdr_amd/, include/,
 * libgsdr.so) may include, link or call this.  Allowed users: tests/,
 * __graft_entry__.smoke() and the cpu_baseline leg of bench.py.
 *
 * PARITY STATUS: "parity unpinned".  The reference ships no tests, fixtures or
 * golden vectors for this path, and its sources need
 * CUDA/cuBLAS/cuFFT/boost/UHD headers that this image lacks, so it cannot be
 * built here without writing stand-ins for them (not allowed).  This oracle is
 * therefore a restatement of the reference *source text*, function by
 * function, each citing the file:line it follows; it is additionally checked
 * against closed forms and against the handful of numeric probes recorded in
 * SURVEY.md section 9 (tests/golden/survey_probes.json).
 *
 * All citations are relative to /root/reference.
 */
#ifndef GSDR_ORACLE_H
#define GSDR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y; } oc64; /* float2 / complex64, x=re y=im */

/* ---- window builders (host code in the reference) ---------------------- */
/* cpp/kernels.cu:258-310 ; writes the real part (imag is 0 in the reference) */
void oracle_make_sinc_window(int length, float fc, float *w);
/* cpp/kernels.cu:208-253 */
void oracle_make_flat_window(int length, int side, float *w);

/* ---- integer state machines ------------------------------------------- */
/* cpp/USRP_server_memory_management.cpp:104-156 */
typedef struct {
    int n_tones, eff_length, buffer_len, average, n_eff_tones;
    int new_0, copy_size, current_batch, spare_samples, spare_begin;
} oracle_buffer_helper;
void oracle_buffer_helper_init(oracle_buffer_helper *h, int n_tones,
                               int buffer_len, int average, int n_eff_tones);
void oracle_buffer_helper_update(oracle_buffer_helper *h);

/* cpp/USRP_server_memory_management.cpp:30-56 */
typedef struct {
    int valid_size, new0, total_len, spare_begin, ppt, buffer_len;
} oracle_vna_helper;
void oracle_vna_helper_init(oracle_vna_helper *h, int ppt, int buffer_len);
void oracle_vna_helper_update(oracle_vna_helper *h);

/* ---- parameter derivations -------------------------------------------- */
/* cpp/USRP_demodulator.cpp:722-733 ; bins[u] = -1 when no bin matched (the
 * reference leaves the entry uninitialised). */
void oracle_pfb_tone_bins(int rate, int fft_tones, const int *freq, int n,
                          int *bins);
/* cpp/USRP_demodulator.cpp:706 */
int oracle_pfb_batching(long buffer_len, int fft_tones, long pf_average);

/* cpp/USRP_demodulator.cpp:192-214 + headers/kernels.cuh:58-64 */
typedef struct {
    unsigned long num_steps, length;
    unsigned int chirpness;
    int f0;
} oracle_chirp_param;
void oracle_chirp_params(int rate, int freq0, int chirp_f, int swipe_s,
                         float chirp_t, oracle_chirp_param *cp);
/* TX generator's derivation, cpp/USRP_buffer_generator.cpp:107-129 */
void oracle_chirp_params_tx(int rate, int freq0, int chirp_f, int swipe_s,
                            float chirp_t, oracle_chirp_param *cp);

/* ---- DIRECT (per-tone DDC) -------------------------------------------- */
/* cpp/kernels.cu:45-86 : out[ch*L + j], tone-major, float-rounded */
void oracle_direct_mix(const int *freq, int n_tones, int rate, size_t idx,
                       size_t L, const oc64 *in, oc64 *out);

typedef struct oracle_direct oracle_direct;
/* cpp/USRP_demodulator.cpp:59-119 ; decim==0 disables the FIR */
oracle_direct *oracle_direct_create(const int *freq, int n_tones, int rate,
                                    long decim, long pf_average, long buffer_len);
/* cpp/USRP_demodulator.cpp:400-464 ; out is sample-major [sample][tone];
 * returns the number of valid complex samples. */
long oracle_direct_process(oracle_direct *d, const oc64 *in, oc64 *out);
void oracle_direct_destroy(oracle_direct *d);
const float *oracle_direct_taps(const oracle_direct *d);

/* ---- TONES (polyphase filter bank + tone select, decim==0) ------------- */
typedef struct oracle_pfb oracle_pfb;
/* cpp/USRP_demodulator.cpp:121-175,702-768 */
oracle_pfb *oracle_pfb_create(const int *freq, int n_tones, int rate,
                              int fft_tones, long pf_average, long buffer_len);
/* NOISE (full spectrum, decim==0): cpp/USRP_demodulator.cpp:264-313, 568-649;
 * processed and destroyed with oracle_pfb_process / oracle_pfb_destroy */
oracle_pfb *oracle_noise_create(int fft_tones, long pf_average, long buffer_len);
/* cpp/USRP_demodulator.cpp:486-565 (decim==0 branch) */
long oracle_pfb_process(oracle_pfb *p, const oc64 *in, oc64 *out);
void oracle_pfb_destroy(oracle_pfb *p);
const int *oracle_pfb_bins(const oracle_pfb *p);

/* ---- CHIRP (VNA) ------------------------------------------------------- */
/* cpp/kernels.cu:389-427 */
void oracle_chirp_demod(const oracle_chirp_param *cp, unsigned long last_index,
                        size_t L, const oc64 *in, oc64 *out);
/* cpp/kernels.cu:335-372 (TX law; used to build loop-back test inputs) */
void oracle_chirp_gen(const oracle_chirp_param *cp, unsigned long last_index,
                      size_t L, float scale, oc64 *out);

/* cpp/kernels.cu:589-684 (tone_gen) + cpp/USRP_buffer_generator.cpp:60-95,226-229
 * (the TX side serves slices of a length-`rate` buffer, wrapping): samples
 * [start, start+L) of the periodic buffer the UNNORMALISED inverse DFT of the bin
 * vector gives.  Bin placement as the reference does it: index = f if f > 0 else
 * rate + f; ASSIGNED, so of equal indices the last tone wins; an index outside
 * [0, rate) -- a 0 Hz tone (index = rate), |f| >= rate -- is a write outside the
 * vector in the reference (undefined there): the FFT never sees that tone, it is
 * dropped here.  Returns the number of bins in use. */
int oracle_tone_gen(const int *freq, const float *ampl, int n_tones, int rate,
                    float scale, long start, size_t L, oc64 *out);

typedef struct oracle_chirp oracle_chirp;
/* cpp/USRP_demodulator.cpp:177-262 */
oracle_chirp *oracle_chirp_create(int rate, int freq0, int chirp_f, int swipe_s,
                                  float chirp_t, long decim, long buffer_len);
/* cpp/USRP_demodulator.cpp:342-397 */
long oracle_chirp_process(oracle_chirp *c, const oc64 *in, oc64 *out);
void oracle_chirp_destroy(oracle_chirp *c);

/* number of OpenMP threads the oracle will use (1 when built without -fopenmp) */
int oracle_num_threads(void);
void oracle_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
