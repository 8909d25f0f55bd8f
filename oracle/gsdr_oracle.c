/*
 * gsdr_oracle.c -- CPU restatement of the GPU_SDR RX demodulation path.
 *
 * TEST INFRASTRUCTURE ONLY (see gsdr_oracle.h).  "parity unpinned": the
 * reference has no fixtures and cannot be built in this image; every function
 * below restates the cited reference source, with fp64 accumulation where the
 * reference delegates to cuBLAS/cuFFT (whose summation order is unspecified),
 * so that it can arbitrate fp32 implementations at the 1e-5 level.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off [-fopenmp]).
 * Citations are relative to /root/reference.
 */
#include "gsdr_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* headers/kernels.cuh:34 */
#define ORACLE_PI_F 3.14159265358979f

static int g_threads = 0;

int oracle_num_threads(void) {
#ifdef _OPENMP
    return g_threads > 0 ? g_threads : omp_get_max_threads();
#else
    return 1;
#endif
}

void oracle_set_num_threads(int n) { g_threads = n; }

/* ======================================================================== */
/* windows                                                                  */
/* ======================================================================== */

/* cpp/kernels.cu:258-310.  nvcc resolves sin(float)/cos(float) in host code
 * to the float overloads, so the sinc and the hamming cosine are float; the
 * hamming factor 0.54-0.46*cos(..) is double and the product is rounded back
 * to float.  (length-1)/2 is an integer division: even lengths are
 * asymmetric.  The accumulator `scale` is float. */
void oracle_make_sinc_window(int length, float fc, float *w) {
    float scale = 0;
    for (int i = 0; i < length; i++) {
        int sinc_index = i - (length - 1) / 2;
        float v;
        if (sinc_index != 0) {
            float arg = 2.f * ORACLE_PI_F * fc * sinc_index;
            v = (2.f * fc) * sinf(arg) / arg;
        } else {
            v = 2.f * fc;
        }
        float ham_arg = 2.f * ORACLE_PI_F * i / (length - 1);
        v = (float)(v * (0.54 - 0.46 * cosf(ham_arg)));
        w[i] = v;
        scale += v;
    }
    for (int i = 0; i < length; i++) w[i] /= scale;
}

/* cpp/kernels.cu:208-253.  The second zeroing loop (:223-226) is overwritten
 * by the fill loop (:227-233), which runs over [side, length). */
void oracle_make_flat_window(int length, int side, float *w) {
    float scale = 0;
    for (int i = 0; i < side; i++) w[i] = 0;
    for (int i = length - side; i < length; i++)
        if (i >= 0) w[i] = 0;
    for (int i = 0; i < length - side; i++) {
        w[i + side] = 1.f;
        scale += w[i + side];
    }
    for (int i = 0; i < length; i++) w[i] /= scale;
}

/* ======================================================================== */
/* integer helpers                                                          */
/* ======================================================================== */

/* cpp/USRP_server_memory_management.cpp:145-156 */
static int bh_simulate_batching(const oracle_buffer_helper *h) {
    int offset = 0, batching = 0;
    while (offset + h->average * h->n_tones < h->eff_length) {
        offset += h->n_tones;
        batching++;
    }
    return batching;
}

/* cpp/USRP_server_memory_management.cpp:104-123 */
void oracle_buffer_helper_init(oracle_buffer_helper *h, int n_tones,
                               int buffer_len, int average, int n_eff_tones) {
    h->n_tones = n_tones;
    h->buffer_len = buffer_len;
    h->average = average;
    h->n_eff_tones = n_eff_tones;
    h->eff_length = buffer_len;
    h->current_batch = bh_simulate_batching(h);
    h->spare_samples = h->eff_length - h->current_batch * n_tones;
    h->spare_begin = h->eff_length - h->spare_samples;
    h->new_0 = 0;
    h->copy_size = n_eff_tones * h->current_batch;
}

/* cpp/USRP_server_memory_management.cpp:125-142 */
void oracle_buffer_helper_update(oracle_buffer_helper *h) {
    h->new_0 = h->spare_samples;
    h->eff_length = h->spare_samples + h->buffer_len;
    h->current_batch = bh_simulate_batching(h);
    h->copy_size = h->n_eff_tones * h->current_batch;
    h->spare_samples = h->eff_length - h->current_batch * h->n_tones;
    h->spare_begin = h->eff_length - h->spare_samples;
}

/* cpp/USRP_server_memory_management.cpp:30-43 */
void oracle_vna_helper_init(oracle_vna_helper *h, int ppt, int buffer_len) {
    h->ppt = ppt;
    h->buffer_len = buffer_len;
    h->total_len = buffer_len;
    h->valid_size = h->total_len / ppt;
    h->new0 = h->total_len - ppt * h->valid_size;
    h->spare_begin = h->total_len - h->new0;
}

/* cpp/USRP_server_memory_management.cpp:45-56 */
void oracle_vna_helper_update(oracle_vna_helper *h) {
    h->total_len = h->buffer_len + h->new0;
    h->valid_size = h->total_len / h->ppt;
    h->new0 = h->total_len - h->ppt * h->valid_size;
    h->spare_begin = h->total_len - h->new0;
}

/* ======================================================================== */
/* parameter derivations                                                    */
/* ======================================================================== */

/* cpp/USRP_demodulator.cpp:722-733, literal double arithmetic and loop order:
 * every bin whose open interval (c_i - bin, c_i + bin) holds the tone assigns
 * it, so the LAST matching i wins. */
void oracle_pfb_tone_bins(int rate, int fft_tones, const int *freq, int n,
                          int *bins) {
    double bin_size = (double)rate / (double)fft_tones;
    for (int u = 0; u < n; u++) bins[u] = -1;
    for (size_t i = 0; i < (size_t)fft_tones; i++) {
        double axis = i * bin_size - bin_size * (fft_tones / 2);
        for (int u = 0; u < n; u++) {
            if ((freq[u] < axis + bin_size) && (freq[u] > axis - bin_size))
                bins[u] = (int)((i + (size_t)(fft_tones / 2)) % (size_t)fft_tones);
        }
    }
}

/* cpp/USRP_demodulator.cpp:706 (float arithmetic inside std::ceil) */
int oracle_pfb_batching(long buffer_len, int fft_tones, long pf_average) {
    return (int)(ceilf((float)buffer_len / (float)fft_tones) + pf_average + 5);
}

/* double -> unsigned int / int the way x86-64 gcc compiles the reference's
 * implicit conversions (cvttsd2si): out-of-range doubles are undefined in
 * C++; we pin the x86 result so the oracle is deterministic. */
static unsigned int d2u32_x86(double v) {
    if (!(v > -9.2e18 && v < 9.2e18)) return 0u;
    return (unsigned int)(long long)v;
}
static int d2i32_x86(double v) {
    if (!(v > -2147483649.0 && v < 2147483648.0)) return INT_MIN;
    return (int)v;
}

/* cpp/USRP_demodulator.cpp:192-214 */
void oracle_chirp_params(int rate, int freq0, int chirp_f, int swipe_s,
                         float chirp_t, oracle_chirp_param *cp) {
    cp->num_steps = (unsigned long)(long)swipe_s;              /* :192 */
    if (cp->num_steps < 1)                                      /* :193-196 */
        cp->num_steps = (unsigned long)(chirp_t * rate);
    /* :202 float * int / unsigned long, evaluated in float */
    cp->length = (unsigned long)(chirp_t * rate / cp->num_steps);
    if (cp->length < 1) cp->length = 1;                         /* :203-206 */
    double chirpness = ((pow(2, 32) - 1) * (chirp_f - freq0) /
                        ((double)cp->num_steps - 1.)) / (double)rate; /* :210 */
    cp->chirpness = d2u32_x86(chirpness);
    double f0 = (pow(2, 32) - 1) * ((double)freq0 / (double)rate);   /* :214 */
    cp->f0 = d2i32_x86(f0);
}

/* cpp/USRP_buffer_generator.cpp:107-129: the TX generator's derivation.  Differs from the RX
 * one in :118-122: a step shorter than a sample resets num_steps as well, before the slope. */
void oracle_chirp_params_tx(int rate, int freq0, int chirp_f, int swipe_s,
                            float chirp_t, oracle_chirp_param *cp) {
    cp->num_steps = (unsigned long)(long)swipe_s;              /* :107 */
    if (cp->num_steps < 1)                                      /* :108-111 */
        cp->num_steps = (unsigned long)(chirp_t * rate);
    cp->length = (unsigned long)(chirp_t * rate / cp->num_steps);  /* :117 */
    if (cp->length < 1) {                                       /* :118-122 */
        cp->length = 1;
        cp->num_steps = (unsigned long)(chirp_t * rate);
    }
    double chirpness = ((pow(2, 32) - 1) * (chirp_f - freq0) /
                        ((double)cp->num_steps - 1.)) / (double)rate; /* :125 */
    cp->chirpness = d2u32_x86(chirpness);
    double f0 = (pow(2, 32) - 1) * ((double)freq0 / (double)rate);   /* :129 */
    cp->f0 = d2i32_x86(f0);
}

/* ======================================================================== */
/* DIRECT                                                                   */
/* ======================================================================== */

/* cpp/kernels.cu:45-86.  The product is formed in double and stored as
 * float, like the kernel. */
static inline oc64 direct_mix_one(long long tf, long long tp, int rate,
                                  size_t idx, size_t j, oc64 in) {
    long long ii = (long long)((j + idx) % (size_t)rate);       /* :66 */
    long long my_phase = tp + (tf * ii) % rate;                 /* :68 */
    double ph = 2. * (my_phase / (double)rate);                 /* :69 */
    double q = sin(M_PI * ph), i = cos(M_PI * ph);              /* :72 sincospi */
    oc64 o;
    o.y = (float)(in.y * i - in.x * q);                         /* :82 */
    o.x = (float)(in.x * i + in.y * q);                         /* :83 */
    return o;
}

void oracle_direct_mix(const int *freq, int n_tones, int rate, size_t idx,
                       size_t L, const oc64 *in, oc64 *out) {
#pragma omp parallel for schedule(static) num_threads(oracle_num_threads())
    for (int ch = 0; ch < n_tones; ch++)
        for (size_t j = 0; j < L; j++)
            out[(size_t)ch * L + j] = direct_mix_one(freq[ch], 0, rate, idx, j, in[j]);
}

struct oracle_direct {
    int n_tones, rate;
    long decim, f, L, nb;
    int *freq;
    size_t idx;       /* DIRECT_current_index, USRP_demodulator.cpp:88,437-440 */
    float *taps;      /* fir_taps, :99 */
    double *dout;     /* per tone FIR::_dout as (re,im) doubles, fir.hpp:29 */
};

oracle_direct *oracle_direct_create(const int *freq, int n_tones, int rate,
                                    long decim, long pf_average, long buffer_len) {
    oracle_direct *d = (oracle_direct *)calloc(1, sizeof(*d));
    d->n_tones = n_tones;
    d->rate = rate;
    d->decim = decim;
    d->f = pf_average;
    d->L = buffer_len;
    d->freq = (int *)malloc(sizeof(int) * (size_t)n_tones);
    memcpy(d->freq, freq, sizeof(int) * (size_t)n_tones);
    d->idx = 0;
    if (decim > 0) {
        if (buffer_len % decim != 0) { /* fir.cu:20 assert(nt % M == 0) */
            oracle_direct_destroy(d);
            return NULL;
        }
        d->nb = buffer_len / decim;
        d->taps = (float *)malloc(sizeof(float) * (size_t)(decim * pf_average));
        /* USRP_demodulator.cpp:99 ; 0.75/(decim*2) is double, narrowed to the
         * float parameter of make_sinc_window */
        oracle_make_sinc_window((int)(decim * pf_average),
                                (float)(0.75 / (decim * 2)), d->taps);
        /* fir.cu:24-26: _dout is meant to start at zero (the memset there
         * passes the wrong pointer; intended semantics restated). */
        d->dout = (double *)calloc((size_t)n_tones * (size_t)(d->nb + d->f - 1) * 2,
                                   sizeof(double));
    }
    return d;
}

const float *oracle_direct_taps(const oracle_direct *d) { return d->taps; }

void oracle_direct_destroy(oracle_direct *d) {
    if (!d) return;
    free(d->freq);
    free(d->taps);
    free(d->dout);
    free(d);
}

long oracle_direct_process(oracle_direct *d, const oc64 *in, oc64 *out) {
    const long L = d->L, N = d->n_tones;
    if (d->decim <= 0) {
        /* USRP_demodulator.cpp:442-457: transposed raw mix, [sample][tone] */
#pragma omp parallel for schedule(static) num_threads(oracle_num_threads())
        for (long ch = 0; ch < N; ch++)
            for (long j = 0; j < L; j++)
                out[(size_t)j * N + ch] =
                    direct_mix_one(d->freq[ch], 0, d->rate, d->idx, (size_t)j, in[j]);
        d->idx = (d->idx + (size_t)L) % (size_t)d->rate;
        return N * L;
    }
    const long M = d->decim, f = d->f, nb = d->nb, nout = nb + f - 1;
#pragma omp parallel for schedule(static) num_threads(oracle_num_threads())
    for (long ch = 0; ch < N; ch++) {
        double *dout = d->dout + (size_t)ch * (size_t)nout * 2;
        for (long b = 0; b < nb; b++) {
            /* fir.cu:48-54: C[b,j] = sum_m x[b*M+m] * h[j*M+m] */
            for (long m = 0; m < M; m++) {
                oc64 x = direct_mix_one(d->freq[ch], 0, d->rate, d->idx,
                                        (size_t)(b * M + m), in[b * M + m]);
                for (long j = 0; j < f; j++) {
                    double h = d->taps[j * M + m];
                    /* fir.cu:56-61: dout[f-1-j+b] += C[b,j] */
                    dout[2 * (f - 1 - j + b)] += h * x.x;
                    dout[2 * (f - 1 - j + b) + 1] += h * x.y;
                }
            }
        }
        /* fir.cu:79-82 fir_to_dev + USRP_demodulator.cpp:422-433 transpose */
        for (long b = 0; b < nb; b++) {
            out[(size_t)b * N + ch].x = (float)dout[2 * b];
            out[(size_t)b * N + ch].y = (float)dout[2 * b + 1];
        }
        /* fir.cu:64-69 fir_shift */
        for (long k = 0; k < f - 1; k++) {
            dout[2 * k] = dout[2 * (nb + k)];
            dout[2 * k + 1] = dout[2 * (nb + k) + 1];
        }
        memset(dout + 2 * (f - 1), 0, sizeof(double) * 2 * (size_t)nb);
    }
    d->idx = (d->idx + (size_t)L) % (size_t)d->rate; /* :437-440 */
    return N * L / M;                                 /* :459 */
}

/* ======================================================================== */
/* TONES (PFB)                                                              */
/* ======================================================================== */

struct oracle_pfb {
    int n_tones, rate, nfft, batching;
    long avg, L;
    int *bins;
    float *window;     /* real part, avg*nfft */
    oc64 *raw;         /* raw_input, nfft*batching */
    oracle_buffer_helper bh;
};

oracle_pfb *oracle_pfb_create(const int *freq, int n_tones, int rate,
                              int fft_tones, long pf_average, long buffer_len) {
    oracle_pfb *p = (oracle_pfb *)calloc(1, sizeof(*p));
    p->n_tones = n_tones;
    p->rate = rate;
    p->nfft = fft_tones;
    p->avg = pf_average;
    p->L = buffer_len;
    p->bins = (int *)malloc(sizeof(int) * (size_t)n_tones);
    oracle_pfb_tone_bins(rate, fft_tones, freq, n_tones, p->bins);
    p->batching = oracle_pfb_batching(buffer_len, fft_tones, pf_average);
    p->window = (float *)malloc(sizeof(float) * (size_t)(fft_tones * pf_average));
    /* USRP_demodulator.cpp:131-134: fcut = 1./(2*fft_tones) stored in a float */
    float fcut = (float)(1. / (2 * fft_tones));
    oracle_make_sinc_window((int)(fft_tones * pf_average), fcut, p->window);
    p->raw = (oc64 *)calloc((size_t)fft_tones * (size_t)p->batching, sizeof(oc64));
    oracle_buffer_helper_init(&p->bh, fft_tones, (int)buffer_len,
                              (int)pf_average, n_tones);
    return p;
}

/* NOISE, decim == 0: cpp/USRP_demodulator.cpp:264-313 (ctor) and :568-649
 * (process_pfb_spec).  Same polyphase filter + forward FFT as TONES, but every
 * FFT bin is kept: n_eff_tones = fft_tones in buffer_helper (:301) and the
 * returned length is copy_size = fft_tones * current_batch (:638). */
oracle_pfb *oracle_noise_create(int fft_tones, long pf_average, long buffer_len) {
    oracle_pfb *p = (oracle_pfb *)calloc(1, sizeof(*p));
    p->n_tones = fft_tones;
    p->rate = fft_tones;
    p->nfft = fft_tones;
    p->avg = pf_average;
    p->L = buffer_len;
    p->bins = (int *)malloc(sizeof(int) * (size_t)fft_tones);
    for (int u = 0; u < fft_tones; u++) p->bins[u] = u;
    p->batching = oracle_pfb_batching(buffer_len, fft_tones, pf_average);
    p->window = (float *)malloc(sizeof(float) * (size_t)(fft_tones * pf_average));
    float fcut = (float)(1. / (2 * fft_tones));                 /* :274 */
    oracle_make_sinc_window((int)(fft_tones * pf_average), fcut, p->window); /* :277 */
    p->raw = (oc64 *)calloc((size_t)fft_tones * (size_t)p->batching, sizeof(oc64));
    oracle_buffer_helper_init(&p->bh, fft_tones, (int)buffer_len, (int)pf_average, fft_tones);
    return p;
}

const int *oracle_pfb_bins(const oracle_pfb *p) { return p->bins; }

void oracle_pfb_destroy(oracle_pfb *p) {
    if (!p) return;
    free(p->bins);
    free(p->window);
    free(p->raw);
    free(p);
}

long oracle_pfb_process(oracle_pfb *p, const oc64 *in, oc64 *out) {
    const long nfft = p->nfft, avg = p->avg, N = p->n_tones;
    /* USRP_demodulator.cpp:491-495 */
    memcpy(p->raw + p->bh.new_0, in, sizeof(oc64) * (size_t)p->L);
    const long cb = p->bh.current_batch;
    /* kernels.cu:474-516 (window), cufftExecC2C forward (:501) restricted to
     * the selected bins, kernels.cu:531-554 (select): fp64 direct DFT. */
#pragma omp parallel for schedule(static) num_threads(oracle_num_threads())
    for (long u = 0; u < N; u++) {
        long bin = p->bins[u] < 0 ? 0 : p->bins[u];
        double *cs = (double *)malloc(sizeof(double) * 2 * (size_t)nfft);
        for (long k = 0; k < nfft; k++) {
            long long ph = ((long long)bin * k) % nfft;
            double a = -2. * M_PI * (double)ph / (double)nfft;
            cs[2 * k] = cos(a);
            cs[2 * k + 1] = sin(a);
        }
        for (long r = 0; r < cb; r++) {
            double yr = 0, yi = 0;
            for (long k = 0; k < nfft; k++) {
                double ar = 0, ai = 0;
                for (long i = 0; i < avg; i++) {
                    oc64 s = p->raw[(r + i) * nfft + k];
                    double w = p->window[i * nfft + k];
                    ar += s.x * w;
                    ai += s.y * w;
                }
                yr += ar * cs[2 * k] - ai * cs[2 * k + 1];
                yi += ar * cs[2 * k + 1] + ai * cs[2 * k];
            }
            out[r * N + u].x = (float)yr;
            out[r * N + u].y = (float)yi;
        }
        free(cs);
    }
    /* USRP_demodulator.cpp:504-509 carry */
    memmove(p->raw, p->raw + p->bh.spare_begin,
            sizeof(oc64) * (size_t)p->bh.spare_samples);
    long ret = N * cb;                     /* :546 */
    oracle_buffer_helper_update(&p->bh);   /* :552 */
    return ret;
}

/* ======================================================================== */
/* CHIRP                                                                    */
/* ======================================================================== */

/* cpp/kernels.cu:407-419 (identical in chirp_gen :354-365).  C `unsigned
 * long` is 64-bit here as on the reference's Linux/x86-64 target. */
static inline int chirp_index(const oracle_chirp_param *cp,
                              unsigned long last_index, unsigned int offset) {
    unsigned long effective_index =
        (last_index + offset) % (cp->num_steps * cp->length);
    unsigned long frequency_index = effective_index / cp->length;
    unsigned long q_phase = (frequency_index / 2) * (frequency_index + 1) +
                            (frequency_index % 2) * ((frequency_index + 1) / 2);
    unsigned long phase_correction = cp->chirpness * (cp->length * q_phase);
    return (int)(effective_index *
                     (cp->f0 + frequency_index * cp->chirpness) -
                 phase_correction);
}

void oracle_chirp_demod(const oracle_chirp_param *cp, unsigned long last_index,
                        size_t L, const oc64 *in, oc64 *out) {
#pragma omp parallel for schedule(static) num_threads(oracle_num_threads())
    for (size_t o = 0; o < L; o++) {
        int index = chirp_index(cp, last_index, (unsigned int)o);
        oc64 chirp; /* kernels.cu:421-422: double sinpi/cospi stored in float */
        chirp.x = (float)sin(M_PI * ((double)index / 2147483647.5));
        chirp.y = (float)-cos(M_PI * ((double)index / 2147483647.5));
        /* kernels.cu:424-425, float arithmetic */
        out[o].x = chirp.x * in[o].x + chirp.y * in[o].y;
        out[o].y = chirp.x * in[o].y - chirp.y * in[o].x;
    }
}

/* cpp/kernels.cu:335-372 */
void oracle_chirp_gen(const oracle_chirp_param *cp, unsigned long last_index,
                      size_t L, float scale, oc64 *out) {
#pragma omp parallel for schedule(static) num_threads(oracle_num_threads())
    for (size_t o = 0; o < L; o++) {
        int index = chirp_index(cp, last_index, (unsigned int)o);
        out[o].x = (float)(sin(M_PI * ((double)index / 2147483647.5)) * scale);
        out[o].y = (float)(-cos(M_PI * ((double)index / 2147483647.5)) * scale);
    }
}

struct oracle_chirp {
    oracle_chirp_param cp;
    long decim, L;
    int ppt;
    unsigned long last_index;
    int spare_size;
    float *profile;
    oc64 *output; /* 3*L like USRP_demodulator.cpp:225 */
    oracle_vna_helper vh;
};

oracle_chirp *oracle_chirp_create(int rate, int freq0, int chirp_f, int swipe_s,
                                  float chirp_t, long decim, long buffer_len) {
    oracle_chirp *c = (oracle_chirp *)calloc(1, sizeof(*c));
    oracle_chirp_params(rate, freq0, chirp_f, swipe_s, chirp_t, &c->cp);
    c->decim = decim;
    c->L = buffer_len;
    c->last_index = 0;
    c->spare_size = 0;
    c->output = (oc64 *)calloc((size_t)(decim > 0 ? 3 : 1) * (size_t)buffer_len,
                               sizeof(oc64));
    if (decim > 0) {
        c->ppt = (int)(c->cp.length * (unsigned long)decim);   /* :231 */
        if (c->ppt < 1 || c->ppt > buffer_len) { /* reference would misbehave */
            oracle_chirp_destroy(c);
            return NULL;
        }
        oracle_vna_helper_init(&c->vh, c->ppt, (int)buffer_len); /* :235 */
        c->profile = (float *)malloc(sizeof(float) * (size_t)c->ppt);
        oracle_make_flat_window(c->ppt, c->ppt / 10, c->profile); /* :246 */
    }
    return c;
}

void oracle_chirp_destroy(oracle_chirp *c) {
    if (!c) return;
    free(c->profile);
    free(c->output);
    free(c);
}

long oracle_chirp_process(oracle_chirp *c, const oc64 *in, oc64 *out) {
    const long L = c->L;
    /* USRP_demodulator.cpp:352 */
    oracle_chirp_demod(&c->cp, c->last_index, (size_t)L, in,
                       c->output + c->spare_size);
    /* :355 */
    c->last_index = (c->last_index + (unsigned long)L) %
                    (c->cp.num_steps * c->cp.length);
    if (c->decim <= 0) {                                        /* :384-391 */
        memcpy(out, c->output, sizeof(oc64) * (size_t)L);
        return L;
    }
    const long valid = c->vh.valid_size, ppt = c->ppt;
    /* :363 + kernels.cu:852-872: y[v] = sum_p out[v*ppt+p]*profile[p] (fp64) */
#pragma omp parallel for schedule(static) num_threads(oracle_num_threads())
    for (long v = 0; v < valid; v++) {
        double sr = 0, si = 0;
        for (long p = 0; p < ppt; p++) {
            oc64 s = c->output[v * ppt + p];
            sr += (double)s.x * c->profile[p];
            si += (double)s.y * c->profile[p];
        }
        out[v].x = (float)sr;
        out[v].y = (float)si;
    }
    c->spare_size = c->vh.new0;                                 /* :369 */
    if (c->spare_size > 0)                                      /* :373-380 */
        memmove(c->output, c->output + c->vh.spare_begin,
                sizeof(oc64) * (size_t)c->vh.new0);
    oracle_vna_helper_update(&c->vh);                           /* :382 */
    return valid;
}


/* ======================================================================== */
/* TX tone comb (row f3)                                                    */
/* ======================================================================== */

/* cpp/kernels.cu:589-684.  The reference fills a zeroed vector of `rate` bins
 * (:611-635), runs an unnormalised CUFFT_INVERSE of length `rate` over it (:641)
 * and multiplies by `scale` when it is not 1 (:650-651); TX_buffer_generator
 * hands out consecutive slices of that periodic buffer (cpp/USRP_buffer_generator.cpp
 * :60-95, :226-229).  x[n] = scale * sum_bins X[bin] * exp(+2 pi i bin n / rate),
 * evaluated in double with an exact integer phase. */
int oracle_tone_gen(const int *freq, const float *ampl, int n_tones, int rate,
                    float scale, long start, size_t L, oc64 *out) {
    if (rate <= 0 || n_tones < 0) return -1;
    /* bin placement (:619-628): later tones overwrite earlier ones on the same bin */
    int *bin = (int *)malloc(sizeof(int) * (size_t)(n_tones > 0 ? n_tones : 1));
    float *amp = (float *)malloc(sizeof(float) * (size_t)(n_tones > 0 ? n_tones : 1));
    int used = 0;
    for (int i = 0; i < n_tones; i++) {
        long idx = freq[i] > 0 ? (long)freq[i] : (long)rate + (long)freq[i];  /* :622-624 */
        if (idx < 0 || idx >= rate) continue; /* base_vector[idx] out of bounds: never transformed */
        int k;
        for (k = 0; k < used; k++)
            if (bin[k] == (int)idx) break;
        bin[k] = (int)idx;
        amp[k] = ampl[i];                      /* :628 assignment, imaginary part 0 (:631) */
        if (k == used) used++;
    }
    long s0 = start % rate;
    if (s0 < 0) s0 += rate;
#pragma omp parallel for schedule(static)
    for (long j = 0; j < (long)L; j++) {
        const unsigned long long n = (unsigned long long)((s0 + j) % rate);
        double re = 0, im = 0;
        for (int k = 0; k < used; k++) {
            const unsigned long long ph = ((unsigned long long)bin[k] * n) % (unsigned long long)rate;
            const double a = 2.0 * M_PI * ((double)ph / (double)rate);
            re += (double)amp[k] * cos(a);
            im += (double)amp[k] * sin(a);
        }
        if (scale != 1.f) { re *= scale; im *= scale; }
        out[j].x = (float)re;
        out[j].y = (float)im;
    }
    free(bin);
    free(amp);
    return used;
}
