// ddc_mfma.hip -- the DDC on the matrix cores (gfx950, v_mfma_f32_32x32x16_f16).
//
// Same arithmetic as ddc_kernel / ddc_flat_kernel,
//     y[n,o] = sum_{t < M*F} h[t] * x[(o+woff)*M + t] * w_n^(idx + (o+woff)*M + t),
// (ref: direct_demodulator_integer + FIR, cpp/kernels.cu:45-86, cpp/fir.cu:48-61;
//  TONES: polyphase_filter + FFT bin, cpp/kernels.cu:718-779)
// arranged as a GEMM whose one operand never changes.  With t = hi*PK + lo,
//     y[n,o] = rot[n,o] * sum_hi P_n[hi] * ( sum_lo b[o,hi,lo] * B_n[lo] ),
//     b = S*h[t]*x[...] (tone independent),  B_n[lo] = w_n^lo,  P_n[hi] = w_n^(hi*PK),
//     rot[n,o] = w_n^(idx + (o+woff)*M) / S.
// The inner sum is a [32 rows o] x [2*PK reals] x [32 tones] real matrix product
// per 32-tone tile: its B operand (the phasor table) is loaded into registers
// once per wave, its A operand is the input stream times the taps.  Both are
// split into fp16 hi + lo and the three leading products are kept
// (hi*hi + hi*lo + lo*hi), which carries ~22 bits: scratch/mfma_sim.py measures
// 1e-7..3e-7 relative error per tone against fp64.  P and rot are applied in
// fp32 on the VALU (2*F/PK packed instructions per tone-sample instead of 2+F).
// S is a power of two that puts max|b| near 2^13: fp16 holds 2^16, so nothing
// overflows and the lo halves stay normal; it comes from absmax_kernel's pass
// over the buffer (max over this and the previous buffer, whose tail is the carry).
//
// Work split: one wave = 32 output rows x 32 tones (x TT tiles in the C++ kernel); the
// four waves of a workgroup take four tone groups of the same rows, and workgroups of
// the same rows sit on one XCD.
//
// One algorithm, five kernels (all parity-tested, tests/test_gpu_parity.py):
//   ddc_mfma_ring16_kernel    production: assembly main loop on v_mfma_f32_16x16x32_f16
//                             (tools/gen_ddc_mfma_ring16.py), converted A operand shared by the four
//                             waves through an LDS ring
//   ddc_mfma_ring16p_kernel   the same loop fed with operands converted once per buffer by
//                             ddc_convert_kernel (tools/gen_ddc_mfma_ring16p.py): long launches, streams
//   ddc_mfma_ring16w8_kernel  the same loop for workgroups of eight waves
//                             (tools/gen_ddc_mfma_ring16w8.py): single launches of one round
//   ddc_mfma_ring_kernel      round 1's production kernel, the ring loop on v_mfma_f32_32x32x16_f16
//                             (tools/gen_ddc_mfma_ring.py; GSDR_MFMA_ASM=2)
//   ddc_mfma_kernel           C++, compiler-scheduled, phases pinned with sched_barrier
//                             (GSDR_MFMA_ASM=0): the readable statement of the algorithm
// (Two further assembly variants of round 1 -- every wave converting its own operand, and a
//  single-launch loop reading buffer and carry in place -- were measured dead ends and are gone.)
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "ddc_device.h"
#include "ddc_mfma_ring_gen.h"
#include "ddc_mfma_ring16_gen.h"
#include "ddc_mfma_ring16w8_gen.h"
#include "ddc_mfma_ring16p_gen.h"

namespace gsdr {

typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float4u __attribute__((ext_vector_type(4), aligned(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));

namespace {

struct Frag {
    half8 hi, lo;
};

// x * h with 0 * anything = 0 (padded taps meet samples outside the window, which may be anything)
__device__ __forceinline__ float mul_legacy(float x, float h) {
    float r;
    asm("v_mul_legacy_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(h));
    return r;
}

// one complex sample times its (scaled) tap, split into fp16 hi and lo
__device__ __forceinline__ void split_pair(float xr, float xi, float hs, half2v &hi, half2v &lo) {
    // no contraction: the residual is (rounded product) - hi, bit for bit what the assembly loops
    // compute (v_mul_legacy_f32, v_cvt_pk_f16_f32, v_fma_mix_f32 x*1.0 - hi), so that a stream may change
    // between the kernels that convert in the loop and the pre-converted path from call to call
#pragma clang fp contract(off)
    const float2v v = {mul_legacy(xr, hs), mul_legacy(xi, hs)};
    hi = __builtin_convertvector(v, half2v);
    const float2v res = v - __builtin_convertvector(hi, float2v);
    lo = __builtin_convertvector(res, half2v);
}

__device__ __forceinline__ Frag make_frag(const float4v xa, const float4v xb, const float4v hs) {
    half2v h0, h1, h2, h3, l0, l1, l2, l3;
    split_pair(xa.x, xa.y, hs.x, h0, l0);
    split_pair(xa.z, xa.w, hs.y, h1, l1);
    split_pair(xb.x, xb.y, hs.z, h2, l2);
    split_pair(xb.z, xb.w, hs.w, h3, l3);
    Frag f;
    const half4v ha = __builtin_shufflevector(h0, h1, 0, 1, 2, 3);
    const half4v hb = __builtin_shufflevector(h2, h3, 0, 1, 2, 3);
    const half4v la = __builtin_shufflevector(l0, l1, 0, 1, 2, 3);
    const half4v lb = __builtin_shufflevector(l2, l3, 0, 1, 2, 3);
    f.hi = __builtin_shufflevector(ha, hb, 0, 1, 2, 3, 4, 5, 6, 7);
    f.lo = __builtin_shufflevector(la, lb, 0, 1, 2, 3, 4, 5, 6, 7);
    return f;
}

}  // namespace

// Bit casts, workgroup barrier and lane exchange for the kernels compiled with
// target("no-packed-fp32-ops"): the HIP header versions (__uint_as_float, __syncthreads,
// __shfl_xor) are not always_inline, cannot be inlined into a function with other target
// features, and end up as real calls (s_swappc_b64) there.
__device__ __forceinline__ float bits_to_float(unsigned u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ unsigned float_to_bits(float f) { return __builtin_bit_cast(unsigned, f); }
__device__ __forceinline__ void workgroup_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ unsigned lane_xor(unsigned v, int d) {
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    return (unsigned)__builtin_amdgcn_ds_bpermute((lane ^ d) << 2, (int)v);
}

// Timing-only builds (WRONG results: no stores / one block only) exist for the ablation tables of
// DESIGN.md only.  They are compiled in by -DGSDR_TIMING_BUILD (scratch/ablate*.sh); the shipped
// library has no such path: the two predicates fold to false.
#ifdef GSDR_TIMING_BUILD
__device__ __forceinline__ bool timing_no_stores(const MfmaShape &sh) { return (sh.timing_mode & 1) != 0; }
__device__ __forceinline__ bool timing_one_block(const MfmaShape &sh) { return (sh.timing_mode & 2) != 0; }
#else
__device__ __forceinline__ bool timing_no_stores(const MfmaShape &) { return false; }
__device__ __forceinline__ bool timing_one_block(const MfmaShape &) { return false; }
#endif

// Diagnostic builds only (-DGSDR_STAMP_BUILD, scratch/stamp_probe.py): every workgroup of the
// ring kernels leaves its start and end time (s_memrealtime, 100 MHz) and its XCC id in a buffer
// of its own.  Nothing in the kernel reads the stamps; the shipped library has none of this.
#ifdef GSDR_STAMP_BUILD
__device__ unsigned long long *g_stamp_buf = nullptr;
__device__ __forceinline__ void stamp(int slot) {
    if (g_stamp_buf && threadIdx.x == 0) {
        unsigned long long t = __builtin_amdgcn_s_memrealtime();
        if (slot == 2) {
            unsigned xcc, hwid;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            t = (xcc & 0xfu) | ((unsigned long long)hwid << 8);
        }
        g_stamp_buf[8 * (size_t)blockIdx.x + slot] = t;
    }
}
#else
__device__ __forceinline__ void stamp(int) {}
#endif

// Wave-wide maximum of unsigned values (DPP butterfly; every lane gets the result).
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    int x = (int)v;
#define GSDR_DPP_MAX(ctrl, row_mask)                                                          \
    {                                                                                         \
        const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, x, ctrl, row_mask, 0xf, false); \
        x = (int)((unsigned)x > o ? (unsigned)x : o);                                         \
    }
    GSDR_DPP_MAX(0xB1, 0xf);    // quad_perm [1,0,3,2]
    GSDR_DPP_MAX(0x4E, 0xf);    // quad_perm [2,3,0,1]
    GSDR_DPP_MAX(0x141, 0xf);   // row_half_mirror
    GSDR_DPP_MAX(0x140, 0xf);   // row_mirror
    GSDR_DPP_MAX(0x142, 0xa);   // row_bcast:15
    GSDR_DPP_MAX(0x143, 0xc);   // row_bcast:31: lane 63 holds the maximum
#undef GSDR_DPP_MAX
    return (unsigned)__builtin_amdgcn_readlane(x, 63);
}

// The scale of ONE OUTPUT ROW (round 3; rounds 1 and 2 had one scale per buffer).  Row o of a launch reads
// T[o*M, o*M + M*F) of the call's logical stream T = [carry | buffer] (TONES: the raw window); absmax_kernel
// left max |finite component| of every segment of seg_k*M samples of T in a.segmax.  The exponent that puts
// the row's largest |x * h'| below 2^14 (fp16 holds 2^16, |h'| <= 1):  |x| < 2^(e-126)  =>  se = 140 - e.
// So what a row's ordinary samples keep of their 22 bits depends on the samples of ITS window alone, as the
// precision of an output of the reference depends on its window alone (ref: cpp/kernels.cu:82-83 is
// elementwise, cpp/fir.cu:48-61 sums one window): a spike, however large, costs the rows that hold it
// nothing relative to their own magnitude and the other rows nothing at all, and a NaN or Inf (left out of
// the maxima) turns into NaN exactly in the rows whose window holds it (x * h' * S overflows or stays NaN
// in the conversion; zero-padded taps multiply with v_mul_legacy_f32, where 0 * anything = 0).
// All kernels derive the same exponent from the same table: a stream may change kernels from call to call.
__device__ __forceinline__ void row_segments(const MfmaShape &sh, int o, int &q0, int &q1) {
    const int oc = o < sh.nout ? o : sh.nout - 1;
    q0 = oc;
    q1 = oc + sh.F - 1;
    if (sh.seg_k > 1) {
        q0 /= sh.seg_k;
        q1 /= sh.seg_k;
    }
}
// The maxima of the (at most eight) segments of row o's window, all loads in flight at once.
__device__ __forceinline__ unsigned row_max_bits(const MfmaLaunch &a, int o) {
    int q0, q1;
    row_segments(a.sh, o, q0, q1);
    const int span = q1 - q0;
    unsigned v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = a.segmax[q0 + (i < span ? i : span)];
    unsigned m = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) m = m > v[i] ? m : v[i];
    return m;
}
__device__ __forceinline__ int scale_exp_of(unsigned m) {
    int se = 140 - (int)((m >> 23) & 0xffu);
    return se > 100 ? 100 : (se < -100 ? -100 : se);
}
__device__ __forceinline__ int row_scale_exp(const MfmaLaunch &a, int o) { return scale_exp_of(row_max_bits(a, o)); }
constexpr int kScaleFromTable = 0x7fffffff;
__device__ __forceinline__ float exp2_bits(int e) { return __builtin_bit_cast(float, (unsigned)(127 + e) << 23); }

// w_n^(idx_base + 32*gt*M): the phasor of tone n (fm = f_n mod rate) at the first row of
// row tile gt.  Computed behind the loop, in the shadow of the dtab loads (as an operand of the
// assembly block, to have its load issued in front of the loop, fm costs a full s_waitcnt there).
__device__ __forceinline__ float2 tile_phasor(const MfmaLaunch &a, int gt, unsigned fm) {
    const MfmaShape &sh = a.sh;
    const unsigned long long s_tile = mod_rate(
        (unsigned long long)sh.idx_base + (unsigned long long)(gt * 32) * sh.m_mod_rate, sh.rate, sh.rate_magic);
    const unsigned long long ph = mod_rate((unsigned long long)fm * s_tile, sh.rate, sh.rate_magic);
    double bre, bim;
    exact_phasor(ph, sh.inv_rate, bre, bim);
    return make_float2((float)bre, (float)bim);
}

// rot[n,o] = w_n^(idx_base + o*M) / S applied to the accumulators, then the stores.
// Every table load is issued before the first one is waited for: with the loads inside the range
// guard of the stores the compiler emits sixteen load -> wait -> store round trips, one after the
// other (4.5 us of a 20 us workgroup on C2, scratch/stamp_c2_phases.sh).  The tables are padded to
// whole tiles, so the loads need no guard.
__device__ __forceinline__ void store_tile(const MfmaLaunch &a, int gt, int n, int hh, int se_self,
                                           const float16v &accr, const float16v &acci) {
    // se_self: the scale exponent of row (lane & 31) of this tile (row_scale_exp); a lane holds 16 rows
    const MfmaShape &sh = a.sh;
    const int Np = sh.NT32 * 32;
    const unsigned fm = a.fmod[n];
    float2 d[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) d[i] = a.dtab[(size_t)((i & 3) + 8 * (i >> 2) + 4 * hh) * Np + n];
    asm volatile("" ::: "memory");          // the loads stay above the phasor arithmetic
    const float2 base = tile_phasor(a, gt, fm);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * hh;
        const int se = __builtin_amdgcn_ds_bpermute(row << 2, se_self);
        const float inv = exp2_bits(-se) * sh.unscale;
        const float dx = d[i].x * inv, dy = d[i].y * inv;
        const float rr = base.x * dx - base.y * dy, ri = base.x * dy + base.y * dx;
        float2 y;
        y.x = accr[i] * rr - acci[i] * ri;
        y.y = accr[i] * ri + acci[i] * rr;
        const int orow = gt * 32 + row;
        if (orow < sh.nout && n < sh.N) a.out[(size_t)orow * sh.N + n] = y;
    }
}

template <int TT>
__device__ __forceinline__ void store_rows(const MfmaLaunch &a, int gt, int n0, int hh, int se_self,
                                           const float16v (&accr)[TT], const float16v (&acci)[TT]) {
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
        const int n = n0 + tt * 32;
        store_tile(a, gt, n, hh, se_self, accr[tt], acci[tt]);
    }
}

// Workgroup barrier that no LDS access may be scheduled across (asm + memory
// clobber: the compiler was seen to move ring reads over a plain s_barrier).
__device__ __forceinline__ void wg_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// One workgroup = W waves on the same 32 output rows, one tone group (TT tiles of
// 32 tones) per wave.  The A operand (input x taps, split into fp16 hi/lo) does
// not depend on the tone: the waves produce it once, W-way split, into an LDS
// ring of two phasor blocks, and every wave reads all of it back as MFMA
// operands.  One barrier per block of PK samples.
//
// The kernel has no boundary cases in its loads: rows that reach before x[0]
// (the carry) or past its end are served from the head / tail copies that
// absmax_kernel lays out ([carry | first rows] and [last rows | zeros]); every
// block is a whole block (taps zero padded to nhi*PK samples).
//
// Ordering inside one iteration is pinned with sched_barrier(0):
//   barrier | LDS reads of the block | convert next block -> LDS, loads of the one
//   after | MFMAs | wait | P*C on the VALU.
// Two hazards were measured on gfx950 (scratch/mfma_diag*.py) that the compiler
// does not guard against, and this order keeps clear of both:
//   * an LDS (or scratch) load returning into a register that an MFMA issued a few
//     instructions earlier reads as A/B operand: rows 16..31 of that MFMA were
//     computed from the NEW contents.  Here the operand registers are rewritten
//     only after the barrier that follows the VALU pass over all 32 results.
//   * VALU reads of a VGPR-form MFMA result behind the compiler's own s_nop count
//     returned stale upper registers (8..15); an explicit wait precedes them.
template <int TT, int PK, int W>
__global__ __launch_bounds__(64 * W, TT == 1 ? 2 : 1) void ddc_mfma_kernel(const MfmaLaunch a) {
    constexpr int KS = PK / 8;    // MFMA k-steps (8 complex samples each) per phasor block
    constexpr int SPW = KS / W;   // k-steps each wave produces per block
    static_assert(KS % W == 0 && SPW >= 1, "every wave produces whole k-steps");
    __shared__ uint4 ring[2][KS][2][64];
    const MfmaShape &sh = a.sh;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // scalar
    const int r = lane & 31, hh = lane >> 5;
    // workgroup -> (row tile, tone-group W-tuple): blockIdx % 8 is the XCD, and all
    // tone groups of one row tile go to the same XCD (one copy of x per L2)
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int gt = (q / sh.ntq) * 8 + xcd;
    if (gt >= sh.ngt) return;     // whole workgroup
    const int tg_raw = (q % sh.ntq) * W + wave;
    const bool active = tg_raw < sh.ntg;      // idle waves still produce and keep the barriers
    const int tg = active ? tg_raw : sh.ntg - 1;

    // ---- constant operand: phasor table fragments, register resident ----
    half8 Bf[TT][KS][2][2];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int sp = 0; sp < 2; ++sp) {
                    const size_t at = ((((size_t)(tg * TT + tt) * KS + ks) * 2 + c) * 2 + sp) * 64 + lane;
                    const uint4 v = a.bfrag[at];
                    Bf[tt][ks][c][sp] = __builtin_bit_cast(half8, v);
                }

    // ---- this lane's row of the A operand and its scale (row_scale_exp) ----
    const int o = gt * 32 + r;
    const int se = row_scale_exp(a, o);
    const float S = exp2_bits(se);
    const int oc = o < sh.nout ? o : sh.nout - 1;
    // head: sample s lives at head[s + carry_len]; tail: at tail[s - tail0]
    const float2 *xbase = gt == 0 ? a.head + sh.carry_len
                                  : (gt == sh.ngt - 1 ? a.tail - sh.tail0 : a.x);
    const float2 *xrow = xbase + ((long long)(oc + sh.woff) * sh.M + 4 * hh);
    const float *tp = a.taps + 4 * hh;

    const int nhi = (sh.nk8 + KS - 1) / KS;
    float4v xa[SPW], xb[SPW], hv[SPW];
    auto gload = [&](int blk) {   // blocks past the last one: clamped, valid addresses, results unused
        const int bc = blk < nhi ? blk : nhi - 1;
#pragma unroll
        for (int j = 0; j < SPW; ++j) {
            const int k = bc * KS + wave + j * W;
            hv[j] = *reinterpret_cast<const float4v *>(tp + 8 * k);
            const float4u *p = reinterpret_cast<const float4u *>(xrow + 8 * k);
            xa[j] = p[0];
            xb[j] = p[1];
        }
    };
    auto produce = [&](int slot) {
#pragma unroll
        for (int j = 0; j < SPW; ++j) {
            const int ks = wave + j * W;
            const Frag f = make_frag(xa[j], xb[j], hv[j] * S);
            ring[slot][ks][0][lane] = __builtin_bit_cast(uint4, f.hi);
            ring[slot][ks][1][lane] = __builtin_bit_cast(uint4, f.lo);
        }
    };

    float16v accr[TT], acci[TT];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
        for (int i = 0; i < 16; ++i) accr[tt][i] = acci[tt][i] = 0.f;

    const int Np = sh.NT32 * 32;
    const int n0 = tg * TT * 32 + r;   // this lane's tone in tile 0
    const float2 *pp = a.ptab + n0;

    gload(0);
    produce(0);
    gload(1);
    for (int hi = 0; hi < nhi; ++hi) {
        const int slot = hi & 1;
        wg_barrier();   // block hi is in the ring; the other slot is free again
        half8 fh[KS], fl[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            fh[ks] = __builtin_bit_cast(half8, ring[slot][ks][0][lane]);
            fl[ks] = __builtin_bit_cast(half8, ring[slot][ks][1][lane]);
        }
        float2 P[TT];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) P[tt] = pp[(size_t)hi * Np + tt * 32];
        produce(slot ^ 1);
        gload(hi + 2);
        __builtin_amdgcn_sched_barrier(0);
        float16v Cr[TT], Ci[TT];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const float16v zero = {0};
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) {
                float16v cr = ks == 0 ? zero : Cr[tt], ci = ks == 0 ? zero : Ci[tt];
                cr = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[ks], Bf[tt][ks][0][0], cr, 0, 0, 0);
                ci = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[ks], Bf[tt][ks][1][0], ci, 0, 0, 0);
                cr = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[ks], Bf[tt][ks][0][1], cr, 0, 0, 0);
                ci = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[ks], Bf[tt][ks][1][1], ci, 0, 0, 0);
                cr = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[ks], Bf[tt][ks][0][0], cr, 0, 0, 0);
                ci = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[ks], Bf[tt][ks][1][0], ci, 0, 0, 0);
                Cr[tt] = cr;
                Ci[tt] = ci;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 15\n\ts_nop 15");   // see the header: results first, then their readers
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                accr[tt][i] = __builtin_fmaf(P[tt].x, Cr[tt][i], accr[tt][i]);
                accr[tt][i] = __builtin_fmaf(-P[tt].y, Ci[tt][i], accr[tt][i]);
                acci[tt][i] = __builtin_fmaf(P[tt].x, Ci[tt][i], acci[tt][i]);
                acci[tt][i] = __builtin_fmaf(P[tt].y, Cr[tt][i], acci[tt][i]);
            }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (!active) return;
    store_rows<TT>(a, gt, n0, hh, se, accr, acci);
}

// Variant with the converted A operand shared through an LDS ring
// (tools/gen_ddc_mfma_ring.py): each wave converts one k-step of every block, all
// four read every k-step back.  A quarter of the loads and conversions of the
// kernel above, one s_barrier per block.
// One row tile of ddc_mfma_ring_kernel: the assembly loop and the stores.  `first`: load the
// phasor images (the second tile of a workgroup keeps them).
__device__ __forceinline__ __attribute__((target("no-packed-fp32-ops"))) void ring_tile(
    const MfmaLaunch &a, uint4 *lds, int gt, int first, int tg, int wave, bool active) {
    constexpr int KS = 4;
    const MfmaShape &sh = a.sh;
    const int Np = sh.NT32 * 32;
    const int nhi = timing_one_block(sh) ? 1 : (sh.nk8 + KS - 1) / KS;
    // Everything per lane is derived again for every tile, from an id the compiler cannot see
    // through: it has twelve registers of its own across the assembly (v0..v11), and what it
    // would carry over from the first tile it would have to park in AGPRs.
    unsigned tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = (int)(tid & 63u);
    const int r = lane & 31, hh = lane >> 5;
    const float S = exp2_bits(row_scale_exp(a, gt * 32 + r));   // this lane converts row r of the tile
    const unsigned to = (unsigned)((4 * hh + 8 * wave) * 4);
    const int n0 = tg * 32 + r;
    const unsigned po = (unsigned)n0 * 8u;
    const unsigned bo = (unsigned)tg * (KS * 4 * 1024u) + (unsigned)lane * 16u;
    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char *)lds;
    const unsigned lane16 = lds_base + (unsigned)lane * 16u;
    const unsigned wr16 = lane16 + (unsigned)wave * 2048u;
    const unsigned accaddr = lds_base + (unsigned)wave * 8192u + (unsigned)lane * 16u;
    const unsigned long long tpb = (unsigned long long)a.taps, ppb = (unsigned long long)a.ptab,
                             bfb = (unsigned long long)a.bfrag;
    const int o = gt * 32 + r;
    const int oc = o < sh.nout ? o : sh.nout - 1;
    const float2 *xbase;
    long long xshift;
    if (gt == 0) {
        xbase = a.head;
        xshift = sh.carry_len;
    } else if (gt == sh.ngt - 1) {
        xbase = a.tail;
        xshift = -sh.tail0;
    } else {
        xbase = a.x;
        xshift = 0;
    }
    // this wave converts k-step `wave` of every block: 8 samples = 64 bytes further on
    const unsigned xo = (unsigned)((((long long)(oc + sh.woff) * sh.M + xshift) + 4 * hh + 8 * wave) * 8);
    const unsigned long long xb = (unsigned long long)xbase;
    asm volatile(GSDR_MFMA_RING_TEXT
                 :
                 : [xo] "v"(xo), [to] "v"(to), [po] "v"(po), [bo] "v"(bo), [lane16] "v"(lane16), [wr16] "v"(wr16),
                   [accaddr] "v"(accaddr), [xb_lo] "s"(__builtin_amdgcn_readfirstlane((int)(unsigned)xb)),
                   [xb_hi] "s"(__builtin_amdgcn_readfirstlane((int)(unsigned)(xb >> 32))),
                   [tp_lo] "s"(__builtin_amdgcn_readfirstlane((int)(unsigned)tpb)),
                   [tp_hi] "s"(__builtin_amdgcn_readfirstlane((int)(unsigned)(tpb >> 32))),
                   [pp_lo] "s"(__builtin_amdgcn_readfirstlane((int)(unsigned)ppb)),
                   [pp_hi] "s"(__builtin_amdgcn_readfirstlane((int)(unsigned)(ppb >> 32))),
                   [bf_lo] "s"(__builtin_amdgcn_readfirstlane((int)(unsigned)bfb)),
                   [bf_hi] "s"(__builtin_amdgcn_readfirstlane((int)(unsigned)(bfb >> 32))),
                   [pstride] "s"(__builtin_amdgcn_readfirstlane((int)((unsigned)Np * 8u))),
                   [nhi] "s"(__builtin_amdgcn_readfirstlane(nhi)),
                   [first] "s"(__builtin_amdgcn_readfirstlane(first)),
                   [scale] "v"(S)
                 : GSDR_MFMA_RING_CLOBBERS);
    if (active && !timing_no_stores(sh)) {
        unsigned tid2 = threadIdx.x;
        asm volatile("" : "+v"(tid2));
        const int lane2 = (int)(tid2 & 63u);
        const int se_self = row_scale_exp(a, gt * 32 + (lane2 & 31));   // again: nothing lives across the assembly
        float16v accr, acci;
        const float4v *acc = reinterpret_cast<const float4v *>(lds) + wave * 512 + lane2;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const float4v vr = acc[qd * 64], vi = acc[(qd + 4) * 64];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                accr[qd * 4 + j] = vr[j];
                acci[qd * 4 + j] = vi[j];
            }
        }
        const int n = tg * 32 + (lane2 & 31);
        store_tile(a, gt, n, lane2 >> 5, se_self, accr, acci);
    }
}

__global__ __launch_bounds__(256, 2) __attribute__((target("no-packed-fp32-ops"))) void ddc_mfma_ring_kernel(
    const MfmaLaunch a) {
    constexpr int W = 4;
    // ring (3 slots of 8 KiB) while the loop runs, then the accumulators (4 waves x 8 KiB)
    __shared__ uint4 lds[2048];
    static_assert(sizeof(uint4) * 2048 >= GSDR_MFMA_RING_BYTES, "ring fits");
    const MfmaShape &sh = a.sh;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // sh.rt = 2: two row tiles per workgroup, one after the other, both on this workgroup's XCD:
    // the phasor images (64 KiB per workgroup) and everything up to the scale are loaded once.
    // (Two calls of the tile code, not a loop: across a loop's back edge the compiler parks what
    // it keeps in AGPRs -- it has twelve registers of its own beside the assembly -- and the
    // kernel drops to one wave per SIMD.)
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int rt = sh.rt > 1 ? 2 : 1;
    const int gt0 = (q / sh.ntq) * rt * 8 + xcd;
    if (gt0 >= sh.ngt) return;
    const int tg_raw = (q % sh.ntq) * W + wave;
    const bool active = tg_raw < sh.ntg;
    const int tg = active ? tg_raw : sh.ntg - 1;
    stamp(0);
    stamp(2);
    ring_tile(a, lds, gt0, 1, tg, wave, active);
    if (rt > 1 && gt0 + 8 < sh.ngt) {
        // the ring of the next tile overlays the accumulators of this one: every wave has read its own
        workgroup_sync();
        ring_tile(a, lds, gt0 + 8, 0, tg, wave, active);
    }
    stamp(1);
}

// ---------------------------------------------------------------------------------------------
// The ring loop re-tiled for v_mfma_f32_16x16x32_f16 (tools/gen_ddc_mfma_ring16.py).  Under the
// package power cap the 16x16x32 shape does the same FLOPs per cycle for less energy, i.e. at a
// higher clock (MI355X_MICROARCH.md, DVFS give-back item 7).  Same ring, same writers; the
// readers take each 16-row x 32-real fragment out of the unchanged slot layout through a
// per-lane base, and the results come out as 2 x 2 tiles of 16 x 16:
//     tile q = 2*rh + th, register j of lane l  <->  row 16*rh + 4*(l >> 4) + j,
//                                                   tone 16*th + (l & 15) of the wave's 32.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ __attribute__((target("no-packed-fp32-ops"))) void store_tile16(
    const MfmaLaunch &a, int gt, int tg, int lane, int se_known, const float16v &accr, const float16v &acci) {
    const MfmaShape &sh = a.sh;
    const int Np = sh.NT32 * 32;
    const int l15 = lane & 15, l4 = lane >> 4;
    // all table loads first (see store_tile); every lane computes the tile phasor of tone lane & 31.
    // se_known: the scale exponent of row lane & 31 of the tile as the loop left it in the LDS, or kScaleFromTable:
    // the segment maxima are loaded again here (the kernel that copies pre-converted operands has no scale of its own)
    const unsigned fm = a.fmod[tg * 32 + (lane & 31)];
    const unsigned mbits = se_known == kScaleFromTable ? row_max_bits(a, gt * 32 + (lane & 31)) : 0u;
    float2 d[16];
#pragma unroll
    for (int th = 0; th < 2; ++th)
#pragma unroll
        for (int rh = 0; rh < 2; ++rh)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                d[4 * (2 * rh + th) + j] = a.dtab[(size_t)(16 * rh + 4 * l4 + j) * Np + tg * 32 + 16 * th + l15];
    asm volatile("" ::: "memory");
    // the tile phasor first (fp64 arithmetic on a value loaded long ago): the segment maxima are still on their way
    const float2 base_self = tile_phasor(a, gt, fm);
    asm volatile("" ::: "memory");
    const int se_self = se_known == kScaleFromTable ? scale_exp_of(mbits) : se_known;
    // 1 / S of the lane's eight rows (row 16*rh + 4*l4 + j lives in lane of the same number): all eight lane
    // exchanges are issued before the first is used
    float inv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = 16 * (i >> 2) + 4 * l4 + (i & 3);
        inv[i] = exp2_bits(-__builtin_amdgcn_ds_bpermute(row << 2, se_self)) * sh.unscale;
    }
#pragma unroll
    for (int th = 0; th < 2; ++th) {
        // lane 16*th + l15 holds the tile phasor of tone 16*th + l15
        const int src = (16 * th + l15) << 2;
        const float bx = bits_to_float((unsigned)__builtin_amdgcn_ds_bpermute(src, (int)float_to_bits(base_self.x)));
        const float by = bits_to_float((unsigned)__builtin_amdgcn_ds_bpermute(src, (int)float_to_bits(base_self.y)));
        const int n = tg * 32 + 16 * th + l15;
#pragma unroll
        for (int rh = 0; rh < 2; ++rh)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = 4 * (2 * rh + th) + j;
                const int row = 16 * rh + 4 * l4 + j;
                const float dx = d[i].x * inv[4 * rh + j], dy = d[i].y * inv[4 * rh + j];
                const float rr = bx * dx - by * dy, ri = bx * dy + by * dx;
                float2 y;
                y.x = accr[i] * rr - acci[i] * ri;
                y.y = accr[i] * ri + acci[i] * rr;
                const int orow = gt * 32 + row;
                if (orow < sh.nout && n < sh.N) a.out[(size_t)orow * sh.N + n] = y;
            }
    }
}

__device__ __forceinline__ __attribute__((target("no-packed-fp32-ops"))) void ring16_tile(
    const MfmaLaunch &a, uint4 *lds, unsigned *lds_scale, int gt, int first, int tg, int wave, bool active) {
    constexpr int KS = 4;
    const MfmaShape &sh = a.sh;
    const int Np = sh.NT32 * 32;
    const int nhi = timing_one_block(sh) ? 1 : (sh.nk8 + KS - 1) / KS;
    unsigned tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = (int)(tid & 63u);
    const int r = lane & 31, hh = lane >> 5;
    // this lane converts row r of the tile: the segments of the maxima table its window covers (the loop's
    // prologue loads them and forms the scale, see scale_loads() in the generator)
    int q0, q1;
    row_segments(sh, gt * 32 + r, q0, q1);
    const unsigned sgo = (unsigned)q0 * 4u, sgn = (unsigned)(q1 - q0);
    const unsigned long long sgb = (unsigned long long)a.segmax;
    const unsigned to = (unsigned)((4 * hh + 8 * wave) * 4);
    // P of the lane's two tones: tone 16*th + (lane & 15); the second one 16 tones = 128 bytes on
    const unsigned po = (unsigned)(tg * 32 + (lane & 15)) * 8u;
    const unsigned bo = (unsigned)tg * (KS * 4 * 1024u) + (unsigned)lane * 16u;
    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char *)lds;
    // readers: fragment (k2, rh, sp) = old k-step 2*k2 + (lane >> 5), old lane (16*rh + (lane & 15)) + 32*((lane >> 4) & 1)
    const unsigned rd16 = lds_base + (unsigned)(lane >> 5) * 2048u + (unsigned)(lane & 15) * 16u +
                          (unsigned)((lane >> 4) & 1) * 512u;
    // writers: unchanged (wave converts old k-step `wave`, lane (row r, half hh))
    const unsigned wr16 = lds_base + (unsigned)lane * 16u + (unsigned)wave * 2048u;
    const unsigned accaddr = lds_base + (unsigned)wave * 8192u + (unsigned)lane * 16u;
    const unsigned seaddr = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char *)lds_scale + tid * 4u;
    const unsigned long long tpb = (unsigned long long)a.taps, ppb = (unsigned long long)a.ptab,
                             bfb = (unsigned long long)a.bfrag;
    const int o = gt * 32 + r;
    const int oc = o < sh.nout ? o : sh.nout - 1;
    const float2 *xbase;
    long long xshift;
    if (gt == 0) {
        xbase = a.head;
        xshift = sh.carry_len;
    } else if (gt == sh.ngt - 1) {
        xbase = a.tail;
        xshift = -sh.tail0;
    } else {
        xbase = a.x;
        xshift = 0;
    }
    const unsigned xo = (unsigned)((((long long)(oc + sh.woff) * sh.M + xshift) + 4 * hh + 8 * wave) * 8);
    const unsigned long long xb = (unsigned long long)xbase;
    stamp(4);
    asm volatile(GSDR_MFMA_RING16_TEXT
                 :
                 : [xo] "v"(xo), [to] "v"(to), [po] "v"(po), [bo] "v"(bo), [lane16] "v"(rd16), [wr16] "v"(wr16),
                   [accaddr] "v"(accaddr), [xb_lo] "s"(__builtin_amdgcn_readfirstlane((int)(unsigned)xb)),
                   [xb_hi] "s"(__builtin_amdgcn_readfirstlane((int)(unsigned)(xb >> 32))),
                   [tp_lo] "s"(__builtin_amdgcn_readfirstlane((int)(unsigned)tpb)),
                   [tp_hi] "s"(__builtin_amdgcn_readfirstlane((int)(unsigned)(tpb >> 32))),
                   [pp_lo] "s"(__builtin_amdgcn_readfirstlane((int)(unsigned)ppb)),
                   [pp_hi] "s"(__builtin_amdgcn_readfirstlane((int)(unsigned)(ppb >> 32))),
                   [bf_lo] "s"(__builtin_amdgcn_readfirstlane((int)(unsigned)bfb)),
                   [bf_hi] "s"(__builtin_amdgcn_readfirstlane((int)(unsigned)(bfb >> 32))),
                   [pstride] "s"(__builtin_amdgcn_readfirstlane((int)((unsigned)Np * 8u))),
                   [nhi] "s"(__builtin_amdgcn_readfirstlane(nhi)),
                   [first] "s"(__builtin_amdgcn_readfirstlane(first)),
                   [sgo] "v"(sgo), [sgn] "v"(sgn), [seaddr] "v"(seaddr),
                   [sg_lo] "s"(__builtin_amdgcn_readfirstlane((int)(unsigned)sgb)),
                   [sg_hi] "s"(__builtin_amdgcn_readfirstlane((int)(unsigned)(sgb >> 32)))
                 : GSDR_MFMA_RING16_CLOBBERS);
    stamp(3);
    if (active && !timing_no_stores(sh)) {
        unsigned tid2 = threadIdx.x;
        asm volatile("" : "+v"(tid2));
        const int lane2 = (int)(tid2 & 63u);
        const int se_self = (int)(lds_scale[tid2] >> 23) - 127;      // the bits of S = 2^se, left by the loop
        float16v accr, acci;
        const float4v *acc = reinterpret_cast<const float4v *>(lds) + wave * 512 + lane2;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const float4v vr = acc[qd * 64], vi = acc[(qd + 4) * 64];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                accr[qd * 4 + j] = vr[j];
                acci[qd * 4 + j] = vi[j];
            }
        }
        store_tile16(a, gt, tg, lane2, se_self, accr, acci);
    }
}

__global__ __launch_bounds__(256, 2) __attribute__((target("no-packed-fp32-ops"))) void ddc_mfma_ring16_kernel(
    const MfmaLaunch a) {
    constexpr int W = 4;
    __shared__ uint4 lds[2048];
    __shared__ unsigned lds_scale[256];      // the bits of every lane's S, left by the loop for the epilogue
    static_assert(sizeof(uint4) * 2048 >= GSDR_MFMA_RING16_BYTES, "ring fits");
    const MfmaShape &sh = a.sh;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int rt = sh.rt > 1 ? 2 : 1;
    const int gt0 = (q / sh.ntq) * rt * 8 + xcd;
    if (gt0 >= sh.ngt) return;
    const int tg_raw = (q % sh.ntq) * W + wave;
    const bool active = tg_raw < sh.ntg;
    const int tg = active ? tg_raw : sh.ntg - 1;
    stamp(0);
    stamp(2);
    ring16_tile(a, lds, lds_scale, gt0, 1, tg, wave, active);
    if (rt > 1 && gt0 + 8 < sh.ngt) {
        workgroup_sync();
        ring16_tile(a, lds, lds_scale, gt0 + 8, 0, tg, wave, active);
    }
    stamp(1);
}

#ifdef GSDR_STAMP_BUILD
extern "C" int gsdr_debug_set_stamp_buffer(void *dev_ptr) {
    unsigned long long *p = (unsigned long long *)dev_ptr;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#endif

// ---------------------------------------------------------------------------------------------
// The 16x16x32 ring loop for workgroups of EIGHT waves (tools/gen_ddc_mfma_ring16w8.py): 32 rows x
// 256 tones per workgroup, one workgroup per compute unit.  The two waves of a SIMD are partners
// inside one workgroup -- the barrier of every block keeps them together and they take priority
// in turns -- so a launch of one round has no tail in which half of the workgroups run alone
// (the 4-wave kernels: 105 us / 145 us on every compute unit of a C3 launch, scratch/stamp_probe.py),
// and the conversion is shared by eight waves: waves 0..3 convert the even blocks, 4..7 the odd ones.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2) __attribute__((target("no-packed-fp32-ops"))) void ddc_mfma_ring16w8_kernel(
    const MfmaLaunch a) {
    constexpr int KS = 4, W = 8;
    // ring (3 slots of 8 KiB) while the loop runs, then the accumulators (8 waves x 8 KiB)
    __shared__ uint4 lds[4096];
    __shared__ unsigned lds_scale[512];      // the bits of every lane's S, left by the loop for the epilogue
    static_assert(sizeof(uint4) * 4096 >= GSDR_MFMA_RING16W8_BYTES, "ring fits");
    const MfmaShape &sh = a.sh;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int ntq8 = (sh.ntg + W - 1) / W;
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int gt = (q / ntq8) * 8 + xcd;
    if (gt >= sh.ngt) return;
    stamp(0);
    stamp(2);
    const int tg_raw = (q % ntq8) * W + wave;
    const bool active = tg_raw < sh.ntg;      // idle waves still convert and keep the barriers
    const int tg = active ? tg_raw : sh.ntg - 1;
    const int Np = sh.NT32 * 32;
    const int nhi = timing_one_block(sh) ? 1 : (sh.nk8 + KS - 1) / KS;
    unsigned tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = (int)(tid & 63u);
    const int r = lane & 31, hh = lane >> 5;
    const int kw = wave & 3;                  // old k-step this wave converts (in its blocks)
    int q0, q1;                               // this lane converts row r of the tile: see ring16_tile
    row_segments(sh, gt * 32 + r, q0, q1);
    const unsigned sgo = (unsigned)q0 * 4u, sgn = (unsigned)(q1 - q0);
    const unsigned long long sgb = (unsigned long long)a.segmax;
    const unsigned seaddr = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char *)lds_scale + tid * 4u;
    const unsigned to = (unsigned)((4 * hh + 8 * kw) * 4);
    const unsigned po = (unsigned)(tg * 32 + (lane & 15)) * 8u;
    const unsigned bo = (unsigned)tg * (KS * 4 * 1024u) + (unsigned)lane * 16u;
    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char *)lds;
    const unsigned rd16 = lds_base + (unsigned)(lane >> 5) * 2048u + (unsigned)(lane & 15) * 16u +
                          (unsigned)((lane >> 4) & 1) * 512u;
    const unsigned wr16 = lds_base + (unsigned)lane * 16u + (unsigned)kw * 2048u;
    const unsigned accaddr = lds_base + (unsigned)wave * 8192u + (unsigned)lane * 16u;
    const unsigned long long tpb = (unsigned long long)a.taps, ppb = (unsigned long long)a.ptab,
                             bfb = (unsigned long long)a.bfrag;
    const int o = gt * 32 + r;
    const int oc = o < sh.nout ? o : sh.nout - 1;
    const float2 *xbase;
    long long xshift;
    if (gt == 0) {
        xbase = a.head;
        xshift = sh.carry_len;
    } else if (gt == sh.ngt - 1) {
        xbase = a.tail;
        xshift = -sh.tail0;
    } else {
        xbase = a.x;
        xshift = 0;
    }
    const unsigned xo = (unsigned)((((long long)(oc + sh.woff) * sh.M + xshift) + 4 * hh + 8 * kw) * 8);
    const unsigned long long xb = (unsigned long long)xbase;
    asm volatile(GSDR_MFMA_RING16W8_TEXT
                 :
                 : [xo] "v"(xo), [to] "v"(to), [po] "v"(po), [bo] "v"(bo), [lane16] "v"(rd16), [wr16] "v"(wr16),
                   [accaddr] "v"(accaddr), [xb_lo] "s"(__builtin_amdgcn_readfirstlane((int)(unsigned)xb)),
                   [xb_hi] "s"(__builtin_amdgcn_readfirstlane((int)(unsigned)(xb >> 32))),
                   [tp_lo] "s"(__builtin_amdgcn_readfirstlane((int)(unsigned)tpb)),
                   [tp_hi] "s"(__builtin_amdgcn_readfirstlane((int)(unsigned)(tpb >> 32))),
                   [pp_lo] "s"(__builtin_amdgcn_readfirstlane((int)(unsigned)ppb)),
                   [pp_hi] "s"(__builtin_amdgcn_readfirstlane((int)(unsigned)(ppb >> 32))),
                   [bf_lo] "s"(__builtin_amdgcn_readfirstlane((int)(unsigned)bfb)),
                   [bf_hi] "s"(__builtin_amdgcn_readfirstlane((int)(unsigned)(bfb >> 32))),
                   [pstride] "s"(__builtin_amdgcn_readfirstlane((int)((unsigned)Np * 8u))),
                   [nhi] "s"(__builtin_amdgcn_readfirstlane(nhi)),
                   [role] "s"(__builtin_amdgcn_readfirstlane(wave >> 2)),
                   [sgo] "v"(sgo), [sgn] "v"(sgn), [seaddr] "v"(seaddr),
                   [sg_lo] "s"(__builtin_amdgcn_readfirstlane((int)(unsigned)sgb)),
                   [sg_hi] "s"(__builtin_amdgcn_readfirstlane((int)(unsigned)(sgb >> 32)))
                 : GSDR_MFMA_RING16W8_CLOBBERS);
    if (active && !timing_no_stores(sh)) {
        unsigned tid2 = threadIdx.x;
        asm volatile("" : "+v"(tid2));
        const int lane2 = (int)(tid2 & 63u);
        const int se_self = (int)(lds_scale[tid2] >> 23) - 127;      // the bits of S = 2^se, left by the loop
        float16v accr, acci;
        const float4v *acc = reinterpret_cast<const float4v *>(lds) + wave * 512 + lane2;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const float4v vr = acc[qd * 64], vi = acc[(qd + 4) * 64];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                accr[qd * 4 + j] = vr[j];
                acci[qd * 4 + j] = vi[j];
            }
        }
        store_tile16(a, gt, tg, lane2, se_self, accr, acci);
    }
    stamp(1);
}

// ---------------------------------------------------------------------------------------------
// Pre-converted operands (tools/gen_ddc_mfma_ring16p.py).  In a launch of many rounds every
// workgroup of a row tile -- N/128 of them -- repeats the same loads and the same conversion to
// build the tone-independent A operand.  ddc_convert_kernel does it ONCE per buffer: one
// workgroup per (row tile, block) writes the 8-KiB image the ring slot holds (same layout, same
// arithmetic: make_frag of the compiler-scheduled kernel); ddc_mfma_ring16p_kernel only copies
// images into its ring by LDS-DMA, three blocks ahead, and spends its vector issue slots on
// the rotation alone.  Pays from a few thousand tones on (the pass costs ~10 us per buffer).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) __attribute__((target("no-packed-fp32-ops"))) void ddc_convert_kernel(const MfmaLaunch a, uint4 *__restrict__ img, int nhi) {
    const MfmaShape &sh = a.sh;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int gt = blockIdx.x / nhi, blk = blockIdx.x - gt * nhi;
    const int o = gt * 32 + r;
    const float S = exp2_bits(row_scale_exp(a, o));
    const int oc = o < sh.nout ? o : sh.nout - 1;
    const float2 *xbase = gt == 0 ? a.head + sh.carry_len : (gt == sh.ngt - 1 ? a.tail - sh.tail0 : a.x);
    const int k = blk * 4 + wave;       // 8-sample k-step of the window
    const float4u *px = reinterpret_cast<const float4u *>(xbase + ((long long)(oc + sh.woff) * sh.M + 4 * hh + 8 * k));
    const float4v xa = px[0], xb = px[1];
    const float4v hv = *reinterpret_cast<const float4v *>(a.taps + 8 * k + 4 * hh);
    const Frag f = make_frag(xa, xb, hv * S);
    uint4 *dst = img + ((size_t)blockIdx.x * 8 + 2 * wave) * 64 + lane;
    dst[0] = __builtin_bit_cast(uint4, f.hi);
    dst[64] = __builtin_bit_cast(uint4, f.lo);
}

__device__ __forceinline__ __attribute__((target("no-packed-fp32-ops"))) void ring16p_tile(
    const MfmaLaunch &a, uint4 *lds, int gt, int first, int tg, int wave, bool active) {
    constexpr int KS = 4;
    const MfmaShape &sh = a.sh;
    const int Np = sh.NT32 * 32;
    const int nhi = (sh.nk8 + KS - 1) / KS;
    unsigned tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = (int)(tid & 63u);
    const unsigned po = (unsigned)(tg * 32 + (lane & 15)) * 8u;
    const unsigned bo = (unsigned)tg * (KS * 4 * 1024u) + (unsigned)lane * 16u;
    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char *)lds;
    const unsigned rd16 = lds_base + (unsigned)(lane >> 5) * 2048u + (unsigned)(lane & 15) * 16u +
                          (unsigned)((lane >> 4) & 1) * 512u;
    // this wave copies the two 1-KiB pieces of old k-step `wave` of every image
    const unsigned io_hi = (unsigned)wave * 2048u + (unsigned)lane * 16u, io_lo = io_hi + 1024u;
    const unsigned wrs = lds_base + (unsigned)wave * 2048u;
    const unsigned accaddr = lds_base + (unsigned)wave * 8192u + (unsigned)lane * 16u;
    const unsigned long long ibb = (unsigned long long)(a.img + (size_t)gt * nhi * 512), ppb = (unsigned long long)a.ptab,
                             bfb = (unsigned long long)a.bfrag;
    asm volatile(GSDR_MFMA_RING16P_TEXT
                 :
                 : [io_hi] "v"(io_hi), [io_lo] "v"(io_lo), [po] "v"(po), [bo] "v"(bo), [lane16] "v"(rd16),
                   [accaddr] "v"(accaddr), [ib_lo] "s"(__builtin_amdgcn_readfirstlane((int)(unsigned)ibb)),
                   [ib_hi] "s"(__builtin_amdgcn_readfirstlane((int)(unsigned)(ibb >> 32))),
                   [wrs] "s"(__builtin_amdgcn_readfirstlane((int)wrs)),
                   [pp_lo] "s"(__builtin_amdgcn_readfirstlane((int)(unsigned)ppb)),
                   [pp_hi] "s"(__builtin_amdgcn_readfirstlane((int)(unsigned)(ppb >> 32))),
                   [bf_lo] "s"(__builtin_amdgcn_readfirstlane((int)(unsigned)bfb)),
                   [bf_hi] "s"(__builtin_amdgcn_readfirstlane((int)(unsigned)(bfb >> 32))),
                   [pstride] "s"(__builtin_amdgcn_readfirstlane((int)((unsigned)Np * 8u))),
                   [nhi] "s"(__builtin_amdgcn_readfirstlane(nhi)),
                   [first] "s"(__builtin_amdgcn_readfirstlane(first))
                 : GSDR_MFMA_RING16P_CLOBBERS);
    if (active) {
        unsigned tid2 = threadIdx.x;
        asm volatile("" : "+v"(tid2));
        const int lane2 = (int)(tid2 & 63u);
        float16v accr, acci;
        const float4v *acc = reinterpret_cast<const float4v *>(lds) + wave * 512 + lane2;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const float4v vr = acc[qd * 64], vi = acc[(qd + 4) * 64];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                accr[qd * 4 + j] = vr[j];
                acci[qd * 4 + j] = vi[j];
            }
        }
        store_tile16(a, gt, tg, lane2, kScaleFromTable, accr, acci);
    }
}

__global__ __launch_bounds__(256, 2) __attribute__((target("no-packed-fp32-ops"))) void ddc_mfma_ring16p_kernel(
    const MfmaLaunch a) {
    constexpr int W = 4;
    // ring (4 slots of 8 KiB) while the loop runs, then the accumulators (4 waves x 8 KiB)
    __shared__ uint4 lds[2048];
    static_assert(sizeof(uint4) * 2048 >= GSDR_MFMA_RING16P_BYTES, "ring fits");
    const MfmaShape &sh = a.sh;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int gt0 = (q / sh.ntq) * 8 + xcd;
    if (gt0 >= sh.ngt) return;
    const int tg_raw = (q % sh.ntq) * W + wave;
    const bool active = tg_raw < sh.ntg;
    const int tg = active ? tg_raw : sh.ntg - 1;
    ring16p_tile(a, lds, gt0, 1, tg, wave, active);
}

// The staging pass (StageLaunch in ddc_kernels.h).  A workgroup takes 2048 consecutive samples of region A (the
// new buffer) or of region B (what the previous call left in front), a wave 512 of them: sixteen bytes per lane and
// load, all four loads of a wave in flight at once -- one memory round trip per wave.  Maxima per segment:
//   * segments of 512 samples or more: a wave's samples lie in at most two of them; two running maxima per lane,
//     a DPP butterfly each, one LDS atomic per wave and segment;
//   * shorter segments: every lane folds its samples into the workgroup's LDS slots itself (ds_max_u32; the
//     segment of a sample by a multiply-high with a host-made magic number);
// then one global atomicMax per segment the workgroup touched (float bits of non-negative numbers order like
// unsigned integers; an entry is hit by the few workgroups whose chunks share the segment).  NaN and Inf patterns
// count as zero: they must not set a scale (see row_scale_exp).  The copies for the main kernels ride along in
// the chunks that touch their ranges.
constexpr int kStageChunk = 2048;                 // samples per workgroup
constexpr int kStageSlots = kStageChunk / 64 + 3; // segments are at least 64 samples long
struct StageShape {
    unsigned seg_len;          // samples per segment (>= 64)
    unsigned seg_magic;        // floor(2^32 / seg_len) + 1: x / seg_len = umulhi(x, magic) for x < 2^16 (short segments)
    unsigned main_blocks;      // workgroups of region A; the rest take region B
};

__device__ __forceinline__ unsigned finite_bits(float v) {
    const unsigned b = __builtin_bit_cast(unsigned, v) & 0x7fffffffu;
    return b < 0x7f800000u ? b : 0u;
}

__global__ __launch_bounds__(256) void absmax_kernel(const StageLaunch s, const StageShape g) {
    // kStageCopies copies of every slot, picked by the lane: 50 lanes of a wave folding into ONE LDS word serialise
    // (C2's 100-sample segments: 1 us of the pass); into eight words they do not
    constexpr int kStageCopies = 8;
    __shared__ unsigned slot[kStageSlots * kStageCopies];
    // the table of the next call
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < s.nseg_alloc; i += (long long)gridDim.x * 256)
        s.seg_clear[i] = 0u;
    for (int i = threadIdx.x; i < kStageSlots * kStageCopies; i += 256) slot[i] = 0u;
    const bool region_a = blockIdx.x < g.main_blocks;
    const float2 *src = region_a ? s.x : s.b;
    const long long cnt = region_a ? s.n : s.nb;
    const long long tbase = region_a ? s.nb : 0;         // position of src[0] in T = [B | A]
    const long long c0 = (long long)(region_a ? blockIdx.x : blockIdx.x - g.main_blocks) * kStageChunk;
    const long long c1 = c0 + kStageChunk < cnt ? c0 + kStageChunk : cnt;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // segment of the chunk's first sample, and where in that segment it sits (one division per workgroup)
    const unsigned long long t0 = (unsigned long long)(tbase + c0);
    const unsigned seg0 = (unsigned)(t0 / g.seg_len);
    const unsigned rem0 = (unsigned)(t0 - (unsigned long long)seg0 * g.seg_len);
    const bool copies = region_a ? ((s.head_cur && c0 < s.head_n) || (s.head_next && c1 > s.n - s.carry_len) ||
                                    (s.tail && c1 > s.tail0))
                                 : s.b_dst != nullptr;
    __syncthreads();
    const long long w0 = c0 + 512 * wave;                 // this wave's samples: [w0, w0 + 512) below c1
    float4u v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long i = w0 + 2 * lane + 128 * k;
        if (i + 1 < c1) {
            v[k] = *reinterpret_cast<const float4u *>(src + i);
        } else if (i < c1) {
            const float2 w = src[i];
            v[k] = float4u{w.x, w.y, 0.f, 0.f};
        } else {
            v[k] = float4u{0.f, 0.f, 0.f, 0.f};
        }
    }
    const bool long_segments = g.seg_len >= 512u;
    unsigned m_lo = 0, m_hi = 0;
    const unsigned wrem = rem0 + 512u * (unsigned)wave;   // offset of the wave's first sample from the start of segment seg0
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long i = w0 + 2 * lane + 128 * k;
        const unsigned b0 = finite_bits(v[k].x), b1 = finite_bits(v[k].y), b2 = finite_bits(v[k].z), b3 = finite_bits(v[k].w);
        const unsigned ma = b0 > b1 ? b0 : b1, mb = b2 > b3 ? b2 : b3;     // the lane's two samples
        const unsigned oa = wrem + 2u * (unsigned)lane + 128u * (unsigned)k, ob = oa + 1u;
        if (long_segments) {
            // the wave's 512 samples lie in at most two segments: the one of its first sample and the next
            const unsigned first = wrem / g.seg_len;      // wave-uniform; wrem < seg_len + 2048
            const unsigned edge = (first + 1u) * g.seg_len;
            if (oa < edge) m_lo = m_lo > ma ? m_lo : ma; else m_hi = m_hi > ma ? m_hi : ma;
            if (ob < edge) m_lo = m_lo > mb ? m_lo : mb; else m_hi = m_hi > mb ? m_hi : mb;
        } else {
            if (ma) atomicMax(&slot[__umulhi(oa, g.seg_magic) * kStageCopies + (lane & (kStageCopies - 1))], ma);
            if (mb) atomicMax(&slot[__umulhi(ob, g.seg_magic) * kStageCopies + (lane & (kStageCopies - 1))], mb);
        }
        if (copies) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const long long ie = i + e;
                if (ie < c1) {
                    const float2 w2 = e ? make_float2(v[k].z, v[k].w) : make_float2(v[k].x, v[k].y);
                    if (region_a) {
                        if (s.head_cur && ie < s.head_n) s.head_cur[s.carry_len + ie] = w2;
                        if (s.head_next && ie >= s.n - s.carry_len) s.head_next[ie - (s.n - s.carry_len)] = w2;
                        if (s.tail && ie >= s.tail0) s.tail[ie - s.tail0] = w2;
                    } else {
                        s.b_dst[ie] = w2;
                    }
                }
            }
        }
    }
    if (long_segments) {
        const unsigned first = wrem / g.seg_len;
        m_lo = wave_max_u32(m_lo);
        m_hi = wave_max_u32(m_hi);
        if (lane == 0) {
            if (m_lo) atomicMax(&slot[first * kStageCopies], m_lo);
            if (m_hi) atomicMax(&slot[(first + 1u) * kStageCopies], m_hi);
        }
    }
    __syncthreads();
    if (threadIdx.x < kStageSlots) {
        unsigned m = 0;
#pragma unroll
        for (int c = 0; c < kStageCopies; ++c) {
            const unsigned t = slot[threadIdx.x * kStageCopies + c];
            m = m > t ? m : t;
        }
        if (m) atomicMax(&s.seg[seg0 + threadIdx.x], m);
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
namespace {

unsigned short to_half_bits(float v) {
    const _Float16 h = (_Float16)v;   // round to nearest even, like v_cvt_pk_f16_f32
    unsigned short b;
    __builtin_memcpy(&b, &h, 2);
    return b;
}
float from_half_bits(unsigned short b) {
    _Float16 h;
    __builtin_memcpy(&h, &b, 2);
    return (float)h;
}

void host_phasor(unsigned long long ph, unsigned rate, double &re, double &im) {
    const double ang = 2.0 * M_PI * ((double)ph / (double)rate);
    re = std::cos(ang);
    im = -std::sin(ang);
}

template <int TT, int PK, int W>
hipError_t launch_tpw(const MfmaLaunch &a, hipStream_t st) {
    const int gt8 = (a.sh.ngt + 7) / 8;
    const long long grid = (long long)gt8 * 8 * a.sh.ntq;
    if (grid < 1 || grid > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL((ddc_mfma_kernel<TT, PK, W>), dim3((unsigned)grid), dim3(64 * W), 0, st, a);
    return hipGetLastError();
}

template <int TT, int PK>
hipError_t launch_tp(int W, const MfmaLaunch &a, hipStream_t st) {
    switch (W) {
        case 2: return launch_tpw<TT, PK, 2>(a, st);
        case 4: if constexpr (PK >= 32) return launch_tpw<TT, PK, 4>(a, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace

void mfma_build_tables(const MfmaPlan &pl, const std::vector<unsigned> &fmod_in, const float *window,
                       std::vector<uint4> &bfrag, std::vector<float2> &ptab,
                       std::vector<float2> &dtab, std::vector<float> &taps,
                       std::vector<unsigned> &fmod, float &unscale) {
    const int KS = pl.PK / 8;
    const int tiles = pl.ntg * pl.TT, Np = tiles * 32;
    const unsigned rate = pl.rate;
    fmod.assign(Np, 0u);
    for (size_t n = 0; n < fmod_in.size() && n < (size_t)Np; ++n) fmod[n] = fmod_in[n];
    // B operand images: lane (r, hh), element j  <->  k = 8*hh + j  <->  sample
    // lo = 8*ks + 4*hh + (j >> 1), component j & 1 (0: pairs with Re b, 1: with Im b).
    //   c = 0 (real part of the product):  [ Wr, -Wi ]
    //   c = 1 (imaginary part):            [ Wi,  Wr ]
    bfrag.assign((size_t)tiles * KS * 4 * 64, uint4{0, 0, 0, 0});
    if (pl.x16) {
        // v_mfma_f32_16x16x32_f16: image f = ((k2*2 + th)*2 + c)*2 + sp; lane l holds tone
        // 16*th + (l & 15) of the tile, element j <-> k = 8*(l >> 4) + j <-> sample
        // lo = 16*k2 + 4*(l >> 4) + (j >> 1), component j & 1 (same pairing with Re/Im b as below)
        for (int T = 0; T < tiles; ++T)
            for (int k2 = 0; k2 < 2; ++k2)
                for (int th = 0; th < 2; ++th)
                    for (int lane = 0; lane < 64; ++lane) {
                        const unsigned long long fm = fmod[(size_t)T * 32 + 16 * th + (lane & 15)];
                        unsigned short img[2][2][8];
                        for (int j = 0; j < 8; ++j) {
                            const int lo = 16 * k2 + 4 * (lane >> 4) + (j >> 1);
                            double wr, wi;
                            host_phasor((fm * (unsigned long long)lo) % rate, rate, wr, wi);
                            const float v[2] = {(j & 1) ? (float)-wi : (float)wr, (j & 1) ? (float)wr : (float)wi};
                            for (int c = 0; c < 2; ++c) {
                                const unsigned short hb = to_half_bits(v[c]);
                                img[c][0][j] = hb;
                                img[c][1][j] = to_half_bits(v[c] - from_half_bits(hb));
                            }
                        }
                        for (int c = 0; c < 2; ++c)
                            for (int sp = 0; sp < 2; ++sp) {
                                uint4 w;
                                __builtin_memcpy(&w, img[c][sp], 16);
                                const int f = ((k2 * 2 + th) * 2 + c) * 2 + sp;
                                bfrag[((size_t)T * 16 + f) * 64 + lane] = w;
                            }
                    }
    } else
    for (int T = 0; T < tiles; ++T)
        for (int ks = 0; ks < KS; ++ks)
            for (int lane = 0; lane < 64; ++lane) {
                const int r = lane & 31, hh = lane >> 5;
                const unsigned long long fm = fmod[(size_t)T * 32 + r];
                unsigned short img[2][2][8];
                for (int j = 0; j < 8; ++j) {
                    const int lo = 8 * ks + 4 * hh + (j >> 1);
                    double wr, wi;
                    host_phasor((fm * (unsigned long long)lo) % rate, rate, wr, wi);
                    const float v[2] = {(j & 1) ? (float)-wi : (float)wr, (j & 1) ? (float)wr : (float)wi};
                    for (int c = 0; c < 2; ++c) {
                        const unsigned short hb = to_half_bits(v[c]);
                        img[c][0][j] = hb;
                        img[c][1][j] = to_half_bits(v[c] - from_half_bits(hb));
                    }
                }
                for (int c = 0; c < 2; ++c)
                    for (int sp = 0; sp < 2; ++sp) {
                        uint4 w;
                        __builtin_memcpy(&w, img[c][sp], 16);
                        bfrag[((((size_t)T * KS + ks) * 2 + c) * 2 + sp) * 64 + lane] = w;
                    }
            }
    const int nhi = (pl.nk8 + KS - 1) / KS;
    ptab.resize((size_t)nhi * Np);
    for (int hi = 0; hi < nhi; ++hi)
        for (int n = 0; n < Np; ++n) {
            double re, im;
            host_phasor(((unsigned long long)fmod[n] * (((unsigned long long)hi * pl.PK) % rate)) % rate, rate, re, im);
            ptab[(size_t)hi * Np + n] = make_float2((float)re, (float)im);
        }
    dtab.resize((size_t)32 * Np);
    for (int row = 0; row < 32; ++row)
        for (int n = 0; n < Np; ++n) {
            double re, im;
            host_phasor(((unsigned long long)fmod[n] * (((unsigned long long)row * pl.M) % rate)) % rate, rate, re, im);
            dtab[(size_t)row * Np + n] = make_float2((float)re, (float)im);
        }
    // taps scaled by a power of two to max |h'| in (1/2, 1], zero padded to whole k-steps
    float hmax = 0.f;
    for (int t = 0; t < pl.MF; ++t) hmax = std::fmax(hmax, std::fabs(window[t]));
    int eh = 0;
    if (hmax > 0.f) {
        (void)std::frexp(hmax, &eh);   // hmax = m * 2^eh, m in [0.5, 1)
    }
    unscale = std::ldexp(1.f, eh);
    taps.assign((size_t)nhi * pl.PK + 8, 0.f);   // whole blocks, zero padded
    for (int t = 0; t < pl.MF; ++t) taps[t] = std::ldexp(window[t], -eh);
}

hipError_t launch_absmax(const StageLaunch &s, hipStream_t st) {
    if (!s.x || s.n < 1 || s.nb < 0 || (s.nb > 0 && !s.b) || s.carry_len < 0 || s.carry_len > s.n || s.head_n < 0 ||
        s.head_n > s.n || s.tail0 < 0 || s.tail0 > s.n || !s.seg || !s.seg_clear || s.seg_len < 64 ||
        s.seg_len > 0x7fffffffLL || s.nseg_alloc < 1)
        return hipErrorInvalidValue;
    // every segment a workgroup can touch has an entry: (nb + n) / seg_len rounded up, plus the slots a
    // chunk's last wave may name beyond its data (always zero, never written: only non-zero maxima are)
    const long long nseg = (s.nb + s.n + s.seg_len - 1) / s.seg_len;
    if (nseg > s.nseg_alloc) return hipErrorInvalidValue;
    StageShape g{};
    g.seg_len = (unsigned)s.seg_len;
    g.seg_magic = (unsigned)(0x100000000ULL / (unsigned long long)s.seg_len + 1ULL);
    const long long ba = (s.n + kStageChunk - 1) / kStageChunk, bb = (s.nb + kStageChunk - 1) / kStageChunk;
    if (ba + bb > 0x7fffffffLL) return hipErrorInvalidValue;
    g.main_blocks = (unsigned)ba;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)(ba + bb)), dim3(256), 0, st, s, g);
    return hipGetLastError();
}

hipError_t launch_ddc_mfma(MfmaKernel kind, int TT, int PK, int W, const MfmaLaunch &a, hipStream_t st) {
    const MfmaShape &sh = a.sh;
    // shapes are checked on the host: a kernel reading past its tables faults the GPU
    if (sh.N < 1 || sh.nout < 1 || sh.M < 1 || sh.MF < 1 || sh.nk8 != (sh.MF + 7) / 8 ||
        sh.ngt != (sh.nout + 31) / 32 || sh.ntg < 1 || W < 1 || sh.ntq != (sh.ntg + W - 1) / W ||
        sh.NT32 != sh.ntg * TT || sh.N > sh.NT32 * 32 || sh.rate < 1 ||
        (long long)(sh.nout - 1 + sh.woff) * sh.M + sh.MF > sh.nx ||
        (long long)sh.woff * sh.M + sh.carry_len < 0 || (sh.woff < 0 && -sh.woff > 32) ||
        !a.x || !a.head || !a.tail || !a.out || !a.bfrag || !a.ptab || !a.dtab || !a.taps ||
        !a.fmod || !a.segmax || sh.seg_k < 1 || sh.F < 1 || sh.F > 8 || sh.MF != sh.M * sh.F)
        return hipErrorInvalidValue;
    // the row tiles between the first and the last read a.x directly, 8 samples at a time
    // (TONES passes its own over-allocated window as x, head and tail alike); a row
    // reads `reach` samples from its start: whole phasor blocks
    const long long reach = (long long)((sh.nk8 + PK / 8 - 1) / (PK / 8)) * PK;
    if (a.x != a.tail && sh.ngt > 2 && (long long)(32 * (sh.ngt - 1) - 1 + sh.woff) * sh.M + reach > sh.nx)
        return hipErrorInvalidValue;
    if (sh.ngt > 1 && (long long)(32 * (sh.ngt - 1) + sh.woff) * sh.M < sh.tail0) return hipErrorInvalidValue;
    if (kind == MfmaKernel::AsmRing16P) {
        // the conversion pass, then the loop that copies its images (a.img: ngt * nhi images of 8 KiB)
        if (TT != 1 || PK != 32 || W != 4 || !a.img) return hipErrorInvalidValue;
        const int nhi = (sh.nk8 + 3) / 4;
        const long long cgrid = (long long)sh.ngt * nhi;
        const int gt8 = (sh.ngt + 7) / 8;
        const long long grid = (long long)gt8 * 8 * sh.ntq;
        if (cgrid < 1 || cgrid > 0x7fffffffLL || grid < 1 || grid > 0x7fffffffLL) return hipErrorInvalidValue;
        hipLaunchKernelGGL(ddc_convert_kernel, dim3((unsigned)cgrid), dim3(256), 0, st, a, const_cast<uint4 *>(a.img), nhi);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(ddc_mfma_ring16p_kernel, dim3((unsigned)grid), dim3(256), 0, st, a);
        return hipGetLastError();
    }
    if (kind == MfmaKernel::AsmRing16W8) {
        if (TT != 1 || PK != 32 || W != 4) return hipErrorInvalidValue;
        const int gt8 = (sh.ngt + 7) / 8;
        const long long grid = (long long)gt8 * 8 * ((sh.ntg + 7) / 8);
        if (grid < 1 || grid > 0x7fffffffLL) return hipErrorInvalidValue;
        hipLaunchKernelGGL(ddc_mfma_ring16w8_kernel, dim3((unsigned)grid), dim3(512), 0, st, a);
        return hipGetLastError();
    }
    if (kind == MfmaKernel::AsmRing16) {
        if (TT != 1 || PK != 32 || W != 4 || sh.rt < 0 || sh.rt > 8) return hipErrorInvalidValue;
        const int rt = sh.rt > 1 ? sh.rt : 1;
        const int gt8 = ((sh.ngt + 7) / 8 + rt - 1) / rt;
        const long long grid = (long long)gt8 * 8 * sh.ntq;
        if (grid < 1 || grid > 0x7fffffffLL) return hipErrorInvalidValue;
        hipLaunchKernelGGL(ddc_mfma_ring16_kernel, dim3((unsigned)grid), dim3(256), 0, st, a);
        return hipGetLastError();
    }
    if (kind == MfmaKernel::AsmRing) {
        if (TT != 1 || PK != 32 || W != 4 || sh.rt < 0 || sh.rt > 8) return hipErrorInvalidValue;
        const int rt = sh.rt > 1 ? sh.rt : 1;
        const int gt8 = ((sh.ngt + 7) / 8 + rt - 1) / rt;   // groups of rt row tiles per XCD
        const long long grid = (long long)gt8 * 8 * sh.ntq;
        if (grid < 1 || grid > 0x7fffffffLL) return hipErrorInvalidValue;
        hipLaunchKernelGGL(ddc_mfma_ring_kernel, dim3((unsigned)grid), dim3(256), 0, st, a);
        return hipGetLastError();
    }
    if (TT == 2 && PK == 32) return launch_tp<2, 32>(W, a, st);
    if (TT == 1 && PK == 32) return launch_tp<1, 32>(W, a, st);
    if (TT == 2 && PK == 16) return launch_tp<2, 16>(W, a, st);
    if (TT == 1 && PK == 16) return launch_tp<1, 16>(W, a, st);
    return hipErrorInvalidValue;
}

const char *ddc_mfma_kernel_name(MfmaKernel kind) {
    return kind == MfmaKernel::AsmRing16P ? "ddc_mfma_ring16p_kernel" : kind == MfmaKernel::AsmRing16W8 ? "ddc_mfma_ring16w8_kernel" : kind == MfmaKernel::AsmRing16 ? "ddc_mfma_ring16_kernel" : kind == MfmaKernel::AsmRing ? "ddc_mfma_ring_kernel" : "ddc_mfma_kernel";
}

}  // namespace gsdr
