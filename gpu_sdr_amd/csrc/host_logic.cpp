// host_logic.cpp -- the host-only pieces of the RX demodulation path:
// window/tap generation, the integer carry state machines, the tone->bin map
// and the chirp parameter derivation.  No HIP here; everything is callable on
// a machine without a GPU (tests -m "not gpu").
//
// "ref:" citations are relative to /root/reference.
#include "../../include/gsdr.h"

#include <climits>
#include <cmath>
#include <vector>

namespace {
// ref: headers/kernels.cuh:34 -- the reference's float pi literal.
constexpr float kPiF = 3.14159265358979f;
}  // namespace

extern "C" {

int gsdr_abi_version(void) { return GSDR_ABI_VERSION; }

// ref: make_sinc_window, cpp/kernels.cu:258-310.
// Low-pass prototype 2fc*sinc(2*pi*fc*(i-c)) * hamming(i), unit DC gain.
// Quirks kept on purpose: the centre c = (length-1)/2 is an INTEGER division
// (:268), so even lengths are asymmetric; sinc and the hamming cosine are
// evaluated in float (nvcc picks the float overloads in host code), the
// hamming factor in double; the normalising sum is a float accumulator.
void gsdr_make_sinc_window(int length, float fc, float *w) {
    const int centre = (length - 1) / 2;
    const float gain = 2.f * fc;
    float sum = 0.f;
    for (int i = 0; i < length; ++i) {
        const int k = i - centre;
        float s = gain;
        if (k != 0) {
            const float a = 2.f * kPiF * fc * k;
            s = gain * sinf(a) / a;
        }
        const float c = cosf(2.f * kPiF * i / (length - 1));
        const float tap = static_cast<float>(s * (0.54 - 0.46 * c));
        w[i] = tap;
        sum += tap;
    }
    for (int i = 0; i < length; ++i) w[i] /= sum;
}

// ref: make_flat_window, cpp/kernels.cu:208-253.
// Zero on [0, side), 1/(length-side) on [side, length): the reference's
// trailing-zero loop (:223-226) is overwritten by its fill loop (:227-233).
void gsdr_make_flat_window(int length, int side, float *w) {
    float sum = 0.f;
    for (int i = 0; i < length; ++i) {
        w[i] = (i < side) ? 0.f : 1.f;
        if (i >= side) sum += 1.f;
    }
    for (int i = 0; i < length; ++i) w[i] /= sum;
}

// ---- TX tone comb: which spectrum the reference's tone_gen transforms --------
// ref: tone_gen, cpp/kernels.cu:617-635.  Tone i goes to index f if f > 0, else rate + f, of a
// zeroed vector of `rate` bins, by ASSIGNMENT: of tones that meet on one bin the last wins
// (they do not add).  An index outside [0, rate) -- a 0 Hz tone gives `rate` -- is written
// past the allocation there (undefined behaviour); the inverse FFT never sees that tone, so
// it is dropped here.  Output: the surviving tones as signed Hz (bin b > rate/2 -> b - rate,
// the same phasor) in first-seen bin order; returns their count.
int gsdr_tx_tone_bins(int rate, const int *freq, const float *ampl, int n, int *out_freq, float *out_ampl) {
    if (rate <= 0 || n < 0 || !freq || !ampl || !out_freq || !out_ampl) return -1;
    std::vector<long long> bins;
    int used = 0;
    for (int i = 0; i < n; ++i) {
        const long long idx = freq[i] > 0 ? (long long)freq[i] : (long long)rate + freq[i];
        if (idx < 0 || idx >= rate) continue;
        int k = 0;
        for (; k < used; ++k)
            if (bins[k] == idx) break;
        if (k == used) {
            bins.push_back(idx);
            ++used;
        }
        out_freq[k] = (int)(idx > rate / 2 ? idx - rate : idx);
        out_ampl[k] = ampl[i];
    }
    return used;
}

// ---- buffer_helper ---------------------------------------------------------
// ref: cpp/USRP_server_memory_management.cpp:104-156.  The batch count is the
// number of r >= 0 with r*n_tones + average*n_tones < eff_length (strict), the
// closed form of the reference's simulate_batching() loop (:145-156).
static int count_batches(const gsdr_buffer_helper *b) {
    const long long lim = (long long)b->eff_length - (long long)b->average * b->n_tones;
    if (lim <= 0) return 0;
    return (int)((lim + b->n_tones - 1) / b->n_tones);
}

static void refresh(gsdr_buffer_helper *b) {
    b->current_batch = count_batches(b);
    b->copy_size = b->n_eff_tones * b->current_batch;
    b->spare_samples = b->eff_length - b->current_batch * b->n_tones;
    b->spare_begin = b->eff_length - b->spare_samples;
}

void gsdr_buffer_helper_init(gsdr_buffer_helper *b, int n_tones, int buffer_len,
                             int average, int n_eff_tones) {
    b->n_tones = n_tones;
    b->buffer_len = buffer_len;
    b->average = average;
    b->n_eff_tones = n_eff_tones;
    b->eff_length = buffer_len;  // :114
    b->new_0 = 0;                // :121
    refresh(b);
}

void gsdr_buffer_helper_update(gsdr_buffer_helper *b) {
    b->new_0 = b->spare_samples;                      // :128
    b->eff_length = b->spare_samples + b->buffer_len; // :131
    refresh(b);
}

// ---- VNA_decimator_helper --------------------------------------------------
// ref: cpp/USRP_server_memory_management.cpp:30-56.
static void vna_refresh(gsdr_vna_helper *v) {
    v->valid_size = v->total_len / v->ppt;
    v->new0 = v->total_len - v->ppt * v->valid_size;
    v->spare_begin = v->total_len - v->new0;
}

void gsdr_vna_helper_init(gsdr_vna_helper *v, int ppt, int buffer_len) {
    v->ppt = ppt;
    v->buffer_len = buffer_len;
    v->total_len = buffer_len;
    vna_refresh(v);
}

void gsdr_vna_helper_update(gsdr_vna_helper *v) {
    v->total_len = v->buffer_len + v->new0;
    vna_refresh(v);
}

// ---- tone -> FFT bin -------------------------------------------------------
// ref: upload_multitone_parameters, cpp/USRP_demodulator.cpp:722-733.
// The reference scans every bin i and lets the LAST i whose open interval
// (c_i - bin, c_i + bin) contains the tone win, with c_i = i*bin - bin*(nfft/2)
// evaluated in double.  Only bins next to (f - c_0)/bin can match, so we test
// the same double predicate on a +-3 neighbourhood and keep the largest
// matching i: same result, O(n) instead of O(n * nfft).
void gsdr_pfb_tone_bins(int rate, int fft_tones, const int *freq, int n, int *bins) {
    const double bin = (double)rate / (double)fft_tones;
    const int half = fft_tones / 2;
    for (int u = 0; u < n; ++u) {
        bins[u] = -1;
        const double guess = ((double)freq[u] + bin * half) / bin;
        long long lo = (long long)std::floor(guess) - 3, hi = (long long)std::floor(guess) + 3;
        if (lo < 0) lo = 0;
        if (hi > fft_tones - 1) hi = fft_tones - 1;
        for (long long i = lo; i <= hi; ++i) {
            const double centre = (size_t)i * bin - bin * half;
            if ((freq[u] < centre + bin) && (freq[u] > centre - bin))
                bins[u] = (int)((i + half) % fft_tones);
        }
    }
}

// ref: cpp/USRP_demodulator.cpp:706 (the division and ceil are in float).
int gsdr_pfb_batching(long long buffer_len, int fft_tones, long long pf_average) {
    return (int)(std::ceil((float)buffer_len / (float)fft_tones) + pf_average + 5);
}

// ---- chirp parameters ------------------------------------------------------
// ref: cpp/USRP_demodulator.cpp:192-214, headers/kernels.cuh:58-64.
// The reference narrows doubles into `unsigned int chirpness` and `int f0`
// implicitly; for values outside the target range that is undefined in C++ and
// we pin what x86-64 does (cvttsd2si): wrap through 64 bits for chirpness,
// INT_MIN for f0.
void gsdr_chirp_derive(int rate, int freq0, int chirp_f, int swipe_s,
                       float chirp_t, gsdr_chirp_param *cp) {
    unsigned long long steps = (unsigned long long)(long long)swipe_s;  // :192
    if (steps < 1) steps = (unsigned long long)(chirp_t * rate);        // :193-196
    unsigned long long len = (unsigned long long)(chirp_t * rate / steps);  // :202, float maths
    if (len < 1) len = 1;                                               // :203-206
    const double two32m1 = std::pow(2, 32) - 1;
    const double slope = (two32m1 * (chirp_f - freq0) / ((double)steps - 1.)) / (double)rate;  // :210
    const double start = two32m1 * ((double)freq0 / (double)rate);                           // :214
    cp->num_steps = steps;
    cp->length = len;
    cp->chirpness = (slope > -9.2e18 && slope < 9.2e18) ? (unsigned int)(long long)slope : 0u;
    cp->f0 = (start > -2147483649.0 && start < 2147483648.0) ? (int)start : INT_MIN;
}

// ref: TX_buffer_generator, CHIRP case, cpp/USRP_buffer_generator.cpp:107-127.  The same derivation with one
// difference: when a step would be shorter than one sample the TX side also resets num_steps to chirp_t * rate
// (:115-119) BEFORE the slope is computed from it (:122); the RX side keeps the requested num_steps
// (cpp/USRP_demodulator.cpp:203-206).
void gsdr_chirp_derive_tx(int rate, int freq0, int chirp_f, int swipe_s,
                          float chirp_t, gsdr_chirp_param *cp) {
    unsigned long long steps = (unsigned long long)(long long)swipe_s;  // :107
    if (steps < 1) steps = (unsigned long long)(chirp_t * rate);        // :108-111
    unsigned long long len = (unsigned long long)(chirp_t * rate / steps);  // :117, float maths
    if (len < 1) {                                                      // :118-122
        len = 1;
        steps = (unsigned long long)(chirp_t * rate);
    }
    const double two32m1 = std::pow(2, 32) - 1;
    const double slope = (two32m1 * (chirp_f - freq0) / ((double)steps - 1.)) / (double)rate;  // :125
    const double start = two32m1 * ((double)freq0 / (double)rate);                           // :129
    cp->num_steps = steps;
    cp->length = len;
    cp->chirpness = (slope > -9.2e18 && slope < 9.2e18) ? (unsigned int)(long long)slope : 0u;
    cp->f0 = (start > -2147483649.0 && start < 2147483648.0) ? (int)start : INT_MIN;
}

}  // extern "C"
