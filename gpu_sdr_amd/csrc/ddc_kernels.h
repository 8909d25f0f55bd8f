// ddc_kernels.h -- launch interface of the fused DDC kernels (internal).
#pragma once
#include <hip/hip_runtime.h>

namespace gsdr {

// Shape of one DDC launch; passed to the kernels by value.
struct DdcShape {
    int N;                     // tones
    int Npad;                  // tones rounded up to 64 (table pitch)
    int TW;                    // tone waves = Npad / 64
    int M;                     // samples per block (decimation / nfft)
    int nblk;                  // input blocks in this launch
    int nch;                   // chunks the blocks are split into
    int cbase, crem;           // nblk / nch and nblk % nch (chunk_begin(), no device division)
    int g_off;                 // first output kept; out row = G - g_off
    unsigned rate;             // NCO modulus (sample rate, or nfft for TONES)
    unsigned long long idx0;   // NCO index of x[0] (mod rate)
    unsigned long long rate_magic;  // floor((2^64-1)/rate), Barrett reduction mod rate
    double inv_rate;           // 1.0 / rate
    unsigned m_mod_rate;       // M mod rate
    long long total;           // mix_kernel only: number of samples
    long long xlast;           // ddc_flat_kernel: x[0 .. xlast+4) is readable
    int prefetch;              // ddc_flat_kernel: LDS-DMA L2 prefetch of the IQ stream on/off
};

struct DdcLaunch {
    const float2 *x;
    const float *taps_t;    // [M][F]                 (ddc_kernel)
    const float *taps_p;    // [nsub*PK+2][FP] zero padded (ddc_flat_kernel), FP = 1,2,4
    const float2 *btab;
    const double2 *wk;
    const double2 *wrem;
    const unsigned *fmod;
    float2 *out;
    float2 *tails;
    int tails_nch;          // chunk slots `tails` was allocated for (launch_ddc refuses more)
    const float2 *carry_in;
    float2 *carry_out;
    DdcShape sh;
    unsigned lds_bytes;     // dynamic LDS per workgroup (occupancy cap, see launch_flat_fk)
    bool pipe;              // use ddc_flat_kernel (F <= 4, phasor table length K in {12,16,20})
};

// F = tap phases (pf_average, 1..8), K = phasor-table length (16 or 32; 12/16/20
// when a.pipe selects ddc_flat_kernel).
// Enqueues ddc_kernel<F,K> and, when F > 1, ddc_fixup.  `stop` (may be null)
// is recorded right after ddc_kernel, before the fixup.
hipError_t launch_ddc(int F, int K, const DdcLaunch &a, hipStream_t st, hipEvent_t stop);
// Undecimated DIRECT (decim == 0).
hipError_t launch_mix(int K, const DdcLaunch &a, hipStream_t st);

hipError_t launch_ddc_flat_main(int F, int PK, const DdcLaunch &a, hipStream_t st);
const char *ddc_kernel_name();
const char *ddc_flat_kernel_name();
const char *mix_kernel_name();

// ---- chirp ---------------------------------------------------------------
struct ChirpShape {
    unsigned long long num_steps, length, period;  // period = num_steps*length
    unsigned chirpness;
    int f0;
};

// out[o] = in[o] * conj(chirp(index0 + o)), o < n
hipError_t launch_chirp_demod(const float2 *in, float2 *out, long long n,
                              unsigned long long index0, const ChirpShape &cs, hipStream_t st);
// y[v] = sum_p demod(stage[v*ppt+p]) * profile[p], v < valid, where the logical
// stage is [carry (carry_len samples) | in]; index0 is the chirp index of stage[0].
hipError_t launch_chirp_lockin(const float2 *carry, int carry_len, const float2 *in,
                               const float *profile, int ppt, int valid, float2 *out,
                               unsigned long long index0, const ChirpShape &cs, hipStream_t st);
const char *chirp_demod_kernel_name();
const char *chirp_lockin_kernel_name();

// ---- synthetic sources ---------------------------------------------------
hipError_t launch_source_tones(float2 *out, long long n, long long start, unsigned rate,
                               const unsigned *fmod_dev, const float *ampl_dev,
                               const float *phase_dev, int n_tones, float sigma,
                               unsigned long long seed, hipStream_t st);
hipError_t launch_source_chirp(float2 *out, long long n, unsigned long long index0,
                               const ChirpShape &cs, float scale, hipStream_t st);

}  // namespace gsdr
