// ddc_pipe.hip -- the production DDC kernel: same mathematics and the same
// lane/wave mapping as ddc_kernels.hip (read its header first), restructured so
// that the inner loop is ONE straight-line body whose scalar loads are
// software-pipelined and whose arithmetic is packed FP32.
//
// Shape: a block (M samples, one decimated output) is cut into nsub = ceil(M/PK)
// sub-blocks of PK samples; the tap table is zero-padded to nsub*PK samples, so
// the last sub-block needs no special code (the host picks PK in {12,16,20} so
// that PK divides M whenever it can: M = 100 or 1000 -> PK = 20, no padding).
// The phasor table B[lo] = w_n^lo has PK entries in VGPRs; the sub-block phasor
// P advances in double by w^PK (by w^(M-(nsub-1)PK) at a block end, so every
// block starts from the exact grid again).
//
// Scalar pipeline.  IQ samples and taps are wave-uniform and enter the VALU as
// SGPR operands.  hipcc waits right behind every s_load it emits, and it copies
// loop-carried SGPR tuples with s_mov at back-edges -- fatal for registers that
// still have a load in flight.  So the pipeline owns a PRIVATE SGPR range:
// the kernel is compiled with amdgpu_num_sgpr(64) (hipcc allocates s0..s63
// only) and s[64:87] are used exclusively by the inline asm below, as two sets
// of one 2-sample group each (x: 4 dwords, taps: 2*FP dwords):
//
//     wait; load set B <- group g+1;  MAC(set A = group g)
//     wait; load set A <- group g+2;  MAC(set B = group g+1)   ...
//
// The s_waitcnt of group g therefore sits behind the 12 packed VALU
// instructions of group g-1 of the same wave (plus whatever the other waves of
// the SIMD issue meanwhile) instead of directly behind the load.
//
// Packed math.  Per sample: u = x*B[lo] is v_pk_mul_f32 + v_pk_fma_f32, and each
// tap phase is one v_pk_fma_f32 on the (re,im) pair: 2+F instructions instead of
// 4+2F (tools/ubench: v_pk_fma_f32 with an SGPR-pair operand sustains 145 TF).
#include <hip/hip_runtime.h>

#include "ddc_kernels.h"

namespace gsdr {

typedef float f2v __attribute__((ext_vector_type(2)));

// ---- private SGPR sets (never allocated by the compiler) ------------------
// set A: x s[64:67] (sample 0: s[64:65], sample 1: s[66:67]), taps s[68:75]
// set B: x s[76:79],                                          taps s[80:87]
#define GSDR_CLOBBER_A "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75"
#define GSDR_CLOBBER_B "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87"

// u = x * b   (x: SGPR pair (re,im), b: VGPR pair):  t = (-xi*by, xi*bx);  u = (xr*bx, xr*by) + t
#define GSDR_MIX(X, BREG)                                                                  \
    "v_pk_mul_f32 %[t], " X ", " BREG " op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[1,0]\n\t"      \
    "v_pk_fma_f32 %[u], " X ", " BREG ", %[t] op_sel_hi:[0,1,1]\n\t"
// S += h * u with h the low / high half of an SGPR pair
#define GSDR_MAC_LO(SREG, T) "v_pk_fma_f32 " SREG ", " T ", %[u], " SREG " op_sel_hi:[0,1,1]\n\t"
#define GSDR_MAC_HI(SREG, T) "v_pk_fma_f32 " SREG ", " T ", %[u], " SREG " op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"

// wait for everything in flight, then start loading the next group into a set
#define GSDR_WAIT "s_waitcnt lgkmcnt(0)\n\t"
#define GSDR_LOAD(XDST, TDST, TOP) \
    "s_load_dwordx4 " XDST ", %[xp], %[xo]\n\t" TOP " " TDST ", %[tp], %[to]\n\t"

#define GSDR_IN_OPS [b0] "v"(b0), [b1] "v"(b1), [xp] "s"(xp), [tp] "s"(tp), [xo] "n"(XO), [to] "n"(TO)

// One pipelined step: wait; issue the load of the NEXT group into the other
// set; multiply-accumulate the two samples of the CURRENT set.
//   F = 4 (and 3): taps of sample s are two SGPR pairs (h0,h1),(h2,h3)
#define GSDR_STEP4(NX, NT, CX0, CX1, CT00, CT01, CT10, CT11, ...)                                \
    asm volatile(GSDR_WAIT GSDR_LOAD(NX, NT, "s_load_dwordx8")                                   \
                 GSDR_MIX(CX0, "%[b0]") GSDR_MAC_LO("%[s0]", CT00) GSDR_MAC_HI("%[s1]", CT00)    \
                 GSDR_MAC_LO("%[s2]", CT01) GSDR_MAC_HI("%[s3]", CT01)                           \
                 GSDR_MIX(CX1, "%[b1]") GSDR_MAC_LO("%[s0]", CT10) GSDR_MAC_HI("%[s1]", CT10)    \
                 GSDR_MAC_LO("%[s2]", CT11) GSDR_MAC_HI("%[s3]", CT11)                           \
                 : [s0] "+v"(S[0]), [s1] "+v"(S[1]), [s2] "+v"(S[2]), [s3] "+v"(S[3]),           \
                   [t] "=&v"(t), [u] "=&v"(u)                                                    \
                 : GSDR_IN_OPS                                                                   \
                 : __VA_ARGS__)
#define GSDR_STEP3(NX, NT, CX0, CX1, CT00, CT01, CT10, CT11, ...)                                \
    asm volatile(GSDR_WAIT GSDR_LOAD(NX, NT, "s_load_dwordx8")                                   \
                 GSDR_MIX(CX0, "%[b0]") GSDR_MAC_LO("%[s0]", CT00) GSDR_MAC_HI("%[s1]", CT00)    \
                 GSDR_MAC_LO("%[s2]", CT01)                                                      \
                 GSDR_MIX(CX1, "%[b1]") GSDR_MAC_LO("%[s0]", CT10) GSDR_MAC_HI("%[s1]", CT10)    \
                 GSDR_MAC_LO("%[s2]", CT11)                                                      \
                 : [s0] "+v"(S[0]), [s1] "+v"(S[1]), [s2] "+v"(S[2]), [t] "=&v"(t), [u] "=&v"(u) \
                 : GSDR_IN_OPS                                                                   \
                 : __VA_ARGS__)
//   F = 2: taps of sample s are one SGPR pair (h0,h1)
#define GSDR_STEP2(NX, NT, CX0, CX1, CT0, CT1, ...)                                              \
    asm volatile(GSDR_WAIT GSDR_LOAD(NX, NT, "s_load_dwordx4")                                   \
                 GSDR_MIX(CX0, "%[b0]") GSDR_MAC_LO("%[s0]", CT0) GSDR_MAC_HI("%[s1]", CT0)      \
                 GSDR_MIX(CX1, "%[b1]") GSDR_MAC_LO("%[s0]", CT1) GSDR_MAC_HI("%[s1]", CT1)      \
                 : [s0] "+v"(S[0]), [s1] "+v"(S[1]), [t] "=&v"(t), [u] "=&v"(u)                  \
                 : GSDR_IN_OPS                                                                   \
                 : __VA_ARGS__)
//   F = 1: the taps of both samples share one SGPR pair (h(sample 0), h(sample 1))
#define GSDR_STEP1(NX, NT, CX0, CX1, CT, ...)                                                    \
    asm volatile(GSDR_WAIT GSDR_LOAD(NX, NT, "s_load_dwordx2")                                   \
                 GSDR_MIX(CX0, "%[b0]") GSDR_MAC_LO("%[s0]", CT)                                 \
                 GSDR_MIX(CX1, "%[b1]") GSDR_MAC_HI("%[s0]", CT)                                 \
                 : [s0] "+v"(S[0]), [t] "=&v"(t), [u] "=&v"(u)                                   \
                 : GSDR_IN_OPS                                                                   \
                 : __VA_ARGS__)

// compute set A while loading set B ...
template <int F, int XO, int TO>
__device__ __forceinline__ void step_a(f2v (&S)[F], f2v b0, f2v b1, const void *xp, const void *tp) {
    f2v t, u;
    if constexpr (F == 4)
        GSDR_STEP4("s[76:79]", "s[80:87]", "s[64:65]", "s[66:67]", "s[68:69]", "s[70:71]", "s[72:73]", "s[74:75]", GSDR_CLOBBER_B);
    else if constexpr (F == 3)
        GSDR_STEP3("s[76:79]", "s[80:87]", "s[64:65]", "s[66:67]", "s[68:69]", "s[70:71]", "s[72:73]", "s[74:75]", GSDR_CLOBBER_B);
    else if constexpr (F == 2)
        GSDR_STEP2("s[76:79]", "s[80:83]", "s[64:65]", "s[66:67]", "s[68:69]", "s[70:71]", GSDR_CLOBBER_B);
    else
        GSDR_STEP1("s[76:79]", "s[80:81]", "s[64:65]", "s[66:67]", "s[68:69]", GSDR_CLOBBER_B);
}
// ... and set B while loading set A
template <int F, int XO, int TO>
__device__ __forceinline__ void step_b(f2v (&S)[F], f2v b0, f2v b1, const void *xp, const void *tp) {
    f2v t, u;
    if constexpr (F == 4)
        GSDR_STEP4("s[64:67]", "s[68:75]", "s[76:77]", "s[78:79]", "s[80:81]", "s[82:83]", "s[84:85]", "s[86:87]", GSDR_CLOBBER_A);
    else if constexpr (F == 3)
        GSDR_STEP3("s[64:67]", "s[68:75]", "s[76:77]", "s[78:79]", "s[80:81]", "s[82:83]", "s[84:85]", "s[86:87]", GSDR_CLOBBER_A);
    else if constexpr (F == 2)
        GSDR_STEP2("s[64:67]", "s[68:71]", "s[76:77]", "s[78:79]", "s[80:81]", "s[82:83]", GSDR_CLOBBER_A);
    else
        GSDR_STEP1("s[64:67]", "s[68:69]", "s[76:77]", "s[78:79]", "s[80:81]", GSDR_CLOBBER_A);
}

// first load of a chunk (into set A) and the final drain
template <int F>
__device__ __forceinline__ void prime_a(const void *xp, const void *tp) {
    if constexpr (F >= 3)
        asm volatile("s_load_dwordx4 s[64:67], %0, 0x0\n\ts_load_dwordx8 s[68:75], %1, 0x0" ::"s"(xp), "s"(tp) : GSDR_CLOBBER_A);
    else if constexpr (F == 2)
        asm volatile("s_load_dwordx4 s[64:67], %0, 0x0\n\ts_load_dwordx4 s[68:71], %1, 0x0" ::"s"(xp), "s"(tp) : GSDR_CLOBBER_A);
    else
        asm volatile("s_load_dwordx4 s[64:67], %0, 0x0\n\ts_load_dwordx2 s[68:69], %1, 0x0" ::"s"(xp), "s"(tp) : GSDR_CLOBBER_A);
}
__device__ __forceinline__ void drain() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Groups G, G+1 (a pair: set A then set B), G+2, ... of a sub-block of PK/2
// groups.  Group g+1 is loaded from xg/tg with immediate offsets; the load
// issued by the LAST group fetches the first group of whatever follows.
template <int F, int FP, int PK, int G>
struct PairSteps {
    static constexpr int NG = PK / 2;
    static __device__ __forceinline__ void run(f2v (&S)[F], const f2v (&B)[PK], const float2 *xg,
                                               const float *tg, const float2 *xnext,
                                               const float *tnext) {
        // group G sits in set A; next = group G+1 (always inside the sub-block)
        step_a<F, (G + 1) * 2 * 8, (G + 1) * 2 * FP * 4>(S, B[2 * G], B[2 * G + 1], xg, tg);
        // group G+1 sits in set B; next = group G+2, or the look-ahead beyond the sub-block
        if constexpr (G + 2 < NG) {
            step_b<F, (G + 2) * 2 * 8, (G + 2) * 2 * FP * 4>(S, B[2 * G + 2], B[2 * G + 3], xg, tg);
            PairSteps<F, FP, PK, G + 2>::run(S, B, xg, tg, xnext, tnext);
        } else {
            step_b<F, 0, 0>(S, B[2 * G + 2], B[2 * G + 3], xnext, tnext);
        }
    }
};

template <int F, int PK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(64))) void ddc_flat_kernel(
    const float2 *__restrict__ x,       // block b starts at x[b*M]; readable up to x[sh.xlast+2)
    const float *__restrict__ taps_p,   // [nsub*PK + 2][FP]: taps_p[m*FP+j] = h[j*M+m], zero padded
    const float2 *__restrict__ btab,    // [PK][Npad]: w_n^lo
    const double2 *__restrict__ wk,     // [Npad]: w_n^PK
    const double2 *__restrict__ wrem,   // [Npad]: w_n^(M-(nsub-1)*PK)
    const unsigned *__restrict__ fmod,  // [Npad]: f_n mod rate
    float2 *__restrict__ out, float2 *__restrict__ tails, float2 *__restrict__ carry_out,
    DdcShape sh) {
    constexpr int FP = (F == 3) ? 4 : F;
    static_assert(PK % 4 == 0, "groups of 2 samples come in pairs");

    const int lane = threadIdx.x & 63;
    // wave-uniform ids must be SGPRs so that x/taps go through the scalar path
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wave = (int)blockIdx.x * 4 + wid;
    if (wave >= sh.TW * sh.nch) return;
    const int tw = wave % sh.TW;
    const int chunk = wave / sh.TW;
    const int n = tw * 64 + lane;
    const int M = sh.M, N = sh.N, Npad = sh.Npad;

    f2v B[PK];
#pragma unroll
    for (int lo = 0; lo < PK; ++lo) {
        const float2 v = btab[(size_t)lo * Npad + n];
        B[lo] = f2v{v.x, v.y};
    }
    const double2 WK = wk[n];
    const double2 WR = wrem[n];

    const int b0 = (int)(((long long)chunk * sh.nblk) / sh.nch);
    const int b1 = (int)(((long long)(chunk + 1) * sh.nblk) / sh.nch);

    // exact NCO phase at the first sample of the chunk
    // ref: kernels.cu:66-69  ii=(j+idx)%rate; phase=(tf*ii)%rate
    const unsigned long long s0 = (sh.idx0 + (unsigned long long)b0 * (unsigned)M) % sh.rate;
    const unsigned long long ph = ((unsigned long long)fmod[n] * s0) % sh.rate;
    double Pr, Pi;
    {
        double s, c;
        sincospi(2.0 * ((double)ph / (double)sh.rate), &s, &c);
        Pr = c;
        Pi = -s;
    }
    f2v A[F];  // A[k]: partial sum of output G = b + k while block b is processed
#pragma unroll
    for (int k = 0; k < F; ++k) A[k] = f2v{0.f, 0.f};

    const int nsub = (M + PK - 1) / PK;
    const float2 *const xlast = x + sh.xlast;  // last address a 2-sample group may be read from

    const float2 *xg = x + (size_t)b0 * M;  // first group of the current sub-block
    const float *tg = taps_p;
    int q = 0, b = b0;
    const int total_sub = (b1 - b0) * nsub;

    if (total_sub > 0) prime_a<F>(xg, tg);

    for (int it = 0; it < total_sub; ++it) {
        f2v S[F];
#pragma unroll
        for (int j = 0; j < F; ++j) S[j] = f2v{0.f, 0.f};
        const bool blk_end = (q == nsub - 1);
        // what follows this sub-block: the next one of the block, or the first of
        // the next block (x is contiguous across blocks, the taps restart)
        const float2 *xnext = blk_end ? x + (size_t)(b + 1) * M : xg + PK;
        const float *tnext = blk_end ? taps_p : tg + PK * FP;
        // the look-ahead past the chunk's data may be any readable address
        const float2 *xla = xnext <= xlast ? xnext : xlast;

        PairSteps<F, FP, PK, 0>::run(S, B, xg, tg, xla, tnext);

        // fold the sub-block into the output partial sums and advance the phasor
        const float pr = (float)Pr, pi = (float)Pi;
#pragma unroll
        for (int j = 0; j < F; ++j) {
            // tap phase j of block b feeds output G = b + F-1-j  (ref: fir.cu:56-61)
            A[F - 1 - j].x += pr * S[j].x - pi * S[j].y;
            A[F - 1 - j].y += pr * S[j].y + pi * S[j].x;
        }
        const double wx = blk_end ? WR.x : WK.x, wy = blk_end ? WR.y : WK.y;
        const double t = Pr * wx - Pi * wy;
        Pi = Pr * wy + Pi * wx;
        Pr = t;

        xg = xnext;
        tg = tnext;
        if (blk_end) {
            // output G = b has now seen every block this chunk can give it
            if (b >= sh.g_off && n < N)
                out[(size_t)(b - sh.g_off) * N + n] = make_float2(A[0].x, A[0].y);
#pragma unroll
            for (int k = 0; k + 1 < F; ++k) A[k] = A[k + 1];
            A[F - 1] = f2v{0.f, 0.f};
            q = 0;
            ++b;
        } else {
            ++q;
        }
    }
    drain();  // retire the look-ahead load before the wave ends

    if (F > 1) {
        float2 *dst = (chunk == sh.nch - 1) ? carry_out
                                            : tails + (size_t)(chunk + 1) * (F - 1) * Npad;
        if (dst) {
#pragma unroll
            for (int k = 0; k + 1 < F; ++k)
                dst[(size_t)k * Npad + n] = make_float2(A[k].x, A[k].y);
        }
    }
}

template <int F, int PK>
static hipError_t launch_flat_fk(const DdcLaunch &a, hipStream_t st) {
    const int waves = a.sh.TW * a.sh.nch;
    hipLaunchKernelGGL((ddc_flat_kernel<F, PK>), dim3((waves + 3) / 4), dim3(256), 0, st, a.x,
                       a.taps_p, a.btab, a.wk, a.wrem, a.fmod, a.out, a.tails, a.carry_out, a.sh);
    return hipGetLastError();
}

template <int F>
static hipError_t launch_flat_f(int PK, const DdcLaunch &a, hipStream_t st) {
    switch (PK) {
        case 12: return launch_flat_fk<F, 12>(a, st);
        case 16: return launch_flat_fk<F, 16>(a, st);
        case 20: return launch_flat_fk<F, 20>(a, st);
        default: return hipErrorInvalidValue;
    }
}

// F in 1..4, PK in {12,16,20}; x[0 .. sh.xlast + 2) must be readable and the tap
// table padded to ceil(M/PK)*PK samples plus one group for the look-ahead.
hipError_t launch_ddc_flat_main(int F, int PK, const DdcLaunch &a, hipStream_t st) {
    if (a.sh.xlast < 0 || a.sh.nblk < 1) return hipErrorInvalidValue;
    switch (F) {
        case 1: return launch_flat_f<1>(PK, a, st);
        case 2: return launch_flat_f<2>(PK, a, st);
        case 3: return launch_flat_f<3>(PK, a, st);
        case 4: return launch_flat_f<4>(PK, a, st);
        default: return hipErrorInvalidValue;
    }
}

const char *ddc_flat_kernel_name() { return "ddc_flat_kernel"; }

}  // namespace gsdr
