// ddc_kernels.h -- launch interface of the fused DDC kernels (internal).
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

// No packed FP32 (v_pk_*_f32) in a kernel that may share a SIMD with the matrix-core DDC loop of another
// launch -- another handle on the same GPU, or the neighbouring buffer of this one: a v_pk_*_f32 with a
// high-half broadcast returned stale values in some lanes while another wave of the SIMD ran an MFMA loop
// (rule R3; tools/ubench_pk_hazard.hip; scratch/concurrent_handles.py shows it across
// kernels).  Device helpers used by such a kernel must be always_inline.
#ifndef GSDR_NO_PK        // (scratch/pk_hazard_ab.sh builds a variant with packed FP32 to show the hazard)
#define GSDR_NO_PK __attribute__((target("no-packed-fp32-ops")))
#endif

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
    bool few;               // use ddc_few_kernel (a wave per (chunk, tone): a handful of tones at a long decimation)
};

// F = tap phases (pf_average, 1..8), K = phasor-table length (16 or 32; 12/16/20
// when a.pipe selects ddc_flat_kernel).
// Enqueues ddc_kernel<F,K> and, when F > 1, ddc_fixup.  `stop` (may be null)
// is recorded right after ddc_kernel, before the fixup.
hipError_t launch_ddc(int F, int K, const DdcLaunch &a, hipStream_t st, hipEvent_t stop);
const char *ddc_few_kernel_name();
// Undecimated DIRECT (decim == 0).
hipError_t launch_mix(int K, const DdcLaunch &a, hipStream_t st);

hipError_t launch_ddc_flat_main(int F, int PK, const DdcLaunch &a, hipStream_t st);
const char *ddc_kernel_name();
const char *ddc_flat_kernel_name();
const char *mix_kernel_name(int n_tones);   // at most 32 tones: several sample phases per wave

// ---- DDC on the matrix cores (ddc_mfma.hip) --------------------------------
struct MfmaShape {
    int N;                     // tones
    int NT32;                  // 32-tone tiles the tables hold (= ntg * TT)
    int ntg;                   // tone groups (one wave each) = ceil(ceil(N/32) / TT)
    int ntq;                   // workgroups per row tile = ceil(ntg / 4)
    int ngt;                   // row tiles = ceil(nout / 32)
    int M;                     // samples per block
    int MF;                    // window length in samples (M * F)
    int F;                     // tap phases (blocks of M samples a window spans)
    int nk8;                   // MFMA k-steps of 8 samples = ceil(MF / 8)
    int nout;                  // output rows of this launch
    int woff;                  // row o's window starts at sample (o + woff) * M
    int carry_len;             // samples of history in front of sample 0 (in `head`)
    long long tail0;           // first sample held by `tail`
    long long nx;              // x[0 .. nx) is readable
    unsigned rate;             // NCO modulus
    unsigned long long rate_magic;
    double inv_rate;
    unsigned m_mod_rate;
    unsigned idx_base;         // NCO index of sample woff*M (mod rate)
    int seg_k;                 // blocks of M samples per segment of the maxima table (segment = seg_k * M samples)
    float unscale;             // power of two the taps were divided by
    int rt;                    // ring kernel: row tiles a workgroup does one after the other (0, 1: one)
    int timing_mode;           // GSDR_MFMA_TIMING (wrong results): 1 = no stores, 2 = one block only, 3 = both
};

struct MfmaLaunch {
    const float2 *x;
    const float2 *head;        // row tile 0 reads here: sample s at head[s + carry_len]
    const float2 *tail;        // last row tile reads here: sample s at tail[s - tail0], zeros past nx
    const float *taps;         // [nk8*8 + 8] scaled, zero padded
    const uint4 *bfrag;        // phasor-table operand images, see mfma_build_tables
    const float2 *ptab;        // [ceil(nk8/KS)][NT32*32]  w_n^(hi*PK)
    const float2 *dtab;        // [32][NT32*32]            w_n^(row*M)
    const unsigned *fmod;      // [NT32*32]
    const unsigned *segmax;    // this call's table of segment maxima (float bits) over [carry | buffer], see absmax_kernel
    float2 *out;
    const uint4 *img;          // AsmRing16P: [ngt][nhi] pre-converted ring-slot images of 8 KiB (ddc_convert_kernel)
    MfmaShape sh;
};

struct MfmaPlan {
    int TT, PK;                // tone tiles per wave (1, 2), phasor block (16, 32)
    int ntg, nk8, MF, M;
    unsigned rate;
    bool x16;                  // phasor images in the v_mfma_f32_16x16x32_f16 layout (AsmRing16)
};

void mfma_build_tables(const MfmaPlan &pl, const std::vector<unsigned> &fmod_in, const float *window,
                       std::vector<uint4> &bfrag, std::vector<float2> &ptab,
                       std::vector<float2> &dtab, std::vector<float> &taps,
                       std::vector<unsigned> &fmod, float &unscale);
// The staging pass in front of the matrix-core kernels.  The logical stream of a call is T = [B | A]: A the new
// buffer x[0 .. n), B what the previous call left in front of it (DIRECT: the raw-sample carry, read from the head
// copy the previous pass wrote; TONES: the unconsumed end of the previous raw window, copied to b_dst).  One pass
//   * folds max |finite component| of every segment T[q*seg_len, (q+1)*seg_len) into seg[q] (float bits,
//     atomicMax on a table the previous pass cleared; NaN and Inf patterns are left out: they must not set the
//     scale of the ordinary samples around them) and clears seg_clear[0 .. nseg_alloc) for the next call;
//   * lays out the copies the main kernels read without boundary cases: head_cur[carry_len + i] = x[i], i < head_n;
//     head_next[i - (n - carry_len)] = x[i] for the last carry_len samples; tail[i - tail0] = x[i], i >= tail0.
struct StageLaunch {
    const float2 *x;           // A
    long long n;
    const float2 *b;           // B (may be null when nb == 0)
    long long nb;
    float2 *b_dst;             // copy of B (TONES), or null
    unsigned *seg, *seg_clear;
    int nseg_alloc;            // entries of a table
    long long seg_len;         // samples per segment
    float2 *head_cur;
    long long head_n;
    float2 *head_next;
    int carry_len;
    float2 *tail;
    long long tail0;
};
hipError_t launch_absmax(const StageLaunch &s, hipStream_t st);
// AsmRing16 / AsmRing16P / AsmRing16W8: assembly main loops on the 16x16x32 MFMA (production, pre-converted
// operands, eight-wave workgroups); AsmRing: round 1's loop on the 32x32x16 MFMA; Cxx: compiler-scheduled
// (TT, PK, W apply to it only; the assembly kernels are TT = 1, PK = 32, W = 4).
enum class MfmaKernel { AsmRing, Cxx, AsmRing16, AsmRing16W8, AsmRing16P };
hipError_t launch_ddc_mfma(MfmaKernel kind, int TT, int PK, int W, const MfmaLaunch &a, hipStream_t st);
const char *ddc_mfma_kernel_name(MfmaKernel kind);

// ---- batched FFT of arbitrary length + polyphase filter (NOISE mode, fft_kernels.hip) ----
struct FftPlan {
    int n = 0;                 // transform length
    int m = 0;                 // 0: mixed-radix Stockham stages over n; else Bluestein through length m = 2^k
    int n_radices = 0;
    int radices[32] = {};      // stages of n (m == 0) or of m
    float2 *d_tw = nullptr;    // w_len^k, len = n or m
    float2 *d_chirp = nullptr; // Bluestein: exp(+i pi k^2 / n), k < n
    float2 *d_bhat = nullptr;  // Bluestein: transform of the wrapped chirp, length m
};
int fft_plan_build(FftPlan &pl, int n);
void fft_plan_free(FftPlan &pl);
// [batch][n] in src -> [batch][n] in dst; src and tmp are scratch of batch * max(n, m) each (src is destroyed)
hipError_t fft_forward(const FftPlan &pl, float2 *src, float2 *dst, float2 *tmp, int batch, hipStream_t st);
// frames[r][k] = sum_{i<avg} raw[(r+i)*nfft + k] * window[i*nfft + k], r < frames_n (ref: cpp/kernels.cu:474-516)
hipError_t launch_pfb_filter(const float2 *raw, const float *window, int nfft, int avg, int frames_n, float2 *frames,
                             hipStream_t st);
// out[frame][u] = spectra[frame][sel[u]], u < n_out (ref: tone_select, cpp/kernels.cu:520-554)
hipError_t launch_pfb_select(const float2 *spectra, int nfft, int frames_n, const int *sel, int n_out, float2 *out, hipStream_t st);
const char *fft_kernel_name();
// The whole PFB of a frame in one workgroup (filter, in-LDS transform, bin selection): frames of up to
// kPfbLdsMaxN points whose prime factors do not exceed kPfbLdsMaxPrime.
constexpr int kPfbLdsMaxN = 8192;
constexpr int kPfbLdsMaxPrime = 127;
constexpr int kPfbLdsTwMaxN = 4096;                       // up to here the twiddle table sits in the LDS as well
constexpr int kPfbLdsMaxBytes = (2 * kPfbLdsMaxN + kPfbLdsMaxPrime + 1) * 8;   // two frame buffers + roots: 129 KiB of the 160 KiB
int pfb_lds_plan(int n, int *radices16);                 // number of stages, -1 when the length does not fit
// logical window [carry (new_0 samples) | in (window_len - new_0)]; frames_n complete frames -> out[frame][n_out]
// (sel: bin per output column, nullptr = all nfft bins); W[spare_begin .. +spare_n) -> carry_out
// (round 3) a run of consecutive frames per compute unit when it fits the LDS, else a frame per workgroup; `blue`:
// transform through Bluestein's identity at length blue->m (frame lengths with a prime factor above kPfbLdsMaxPrime;
// tw is then the table of length m, blue->d_tw)
hipError_t launch_pfb_lds(const float2 *carry, int new_0, const float2 *in, const float *window, const float2 *tw,
                          int nfft, int avg, int frames_n, const int *sel, int n_out, float2 *out,
                          float2 *carry_out, int spare_begin, int spare_n, long long window_len, hipStream_t st,
                          const FftPlan *blue = nullptr);
const char *pfb_lds_kernel_name();
const char *pfb_cu_kernel_name();
bool pfb_cu_fits(int nfft, int avg, int len);                      // does one frame fit the run kernel's LDS layout
void fft_env_reload();                                              // the cached GSDR_PFB_* switches are read again
bool pfb_cu_takes(int nfft, int avg, int len, bool bluestein, int frames_per_call);     // ... and is it the kernel launch_pfb_lds() runs (for calls of about that many frames)

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
                               unsigned long long index0, const ChirpShape &cs, hipStream_t st,
                               float2 *partial = nullptr, int partial_cap = 0);   // partial sums of split points (may be null)
hipError_t launch_warm(hipStream_t st);
// TX tone comb: out[s] = sum_k q0[k] w_k^(start + s), s < n; fmod = f mod rate, btab[k][64] = w_k^lo,
// ctab[k][16] = w_k^(64 j), w_k = e^(+2 pi i f_k / rate) (ref: tone_gen, cpp/kernels.cu:589-684)
hipError_t launch_tones_synth(float2 *out, long long n, unsigned long long start, unsigned rate, const unsigned *fmod,
                              const float2 *q0, const float2 *btab, const float2 *ctab, int n_tones, hipStream_t st);
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
