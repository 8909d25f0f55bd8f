// demod.cpp -- C ABI of libgsdr.so (include/gsdr.h): handle lifecycle, mode
// dispatch, per-buffer sequencing and carry state of the RX demodulator.
//
// Host-side counterpart of RX_buffer_demodulator
// (ref: cpp/USRP_demodulator.cpp, headers/USRP_demodulator.hpp).  All device
// work is enqueued on one stream per demodulator, like the reference's
// internal_stream (ref: USRP_demodulator.cpp:44), but nothing here blocks
// except the host-pointer entry gsdr_demod_process().
//
// "ref:" citations are relative to /root/reference.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include <unistd.h>

#include "../../include/gsdr.h"
#include "ddc_kernels.h"

extern char **environ;

using gsdr::ChirpShape;
using gsdr::DdcLaunch;
using gsdr::DdcShape;

namespace {

thread_local std::string g_create_error;

constexpr int kMaxF = 8;           // tap phases the DDC kernel is instantiated for
constexpr int kMaxEvents = 8192;   // profiling ring
// Pipelined entries: compute streams that consecutive DIRECT calls go to in turn, and the
// sets of staging buffers / scale slots that keeps the calls in flight apart (pipeline_compute)
constexpr int kPipeStreams = 3;
constexpr int kChirpPartials = 16384;    // float2 slots for the partial sums of split chirp points (chirp_lockin_split_kernel)
constexpr int kStageSets = kPipeStreams + 1;
constexpr int kScaleSlots = kPipeStreams + 2;

inline int env_int(const char *name, int dflt) {
    const char *v = std::getenv(name);
    return (v && *v) ? std::atoi(v) : dflt;
}

}  // namespace

struct gsdr_demod {
    int mode = GSDR_NODSP;
    int device = -1;
    hipStream_t stream = nullptr;
    std::string err;

    int N = 0;              // channels = wave_type.size()
    int ddc_channels = 0;   // tones the DDC kernels run (== N except NOISE: fft_tones)
    long long L = 0;        // buffer_len
    long long decim = 0;
    long long capacity = 0; // max samples process() can return
    float fcut = 0.f;

    // host-pointer entry staging
    float2 *d_in = nullptr, *d_out = nullptr;
    // pipelined host-pointer entry (gsdr_demod_submit / _wait)
    struct Slot {
        float2 *d_in = nullptr, *d_out = nullptr;
        hipEvent_t up = nullptr, done = nullptr, down = nullptr;
        hipEvent_t wait_ev = nullptr;      // what gsdr_demod_wait() waits for: down (host) / done (device)
        int n = 0;
    } slot[GSDR_PIPELINE_DEPTH];
    hipStream_t s_up = nullptr, s_down = nullptr;
    bool pipe_ready = false;             // pipeline_init() succeeded
    int pipe_head = 0, pipe_count = 0;   // oldest outstanding slot, number outstanding

    // ---- DDC (DIRECT / TONES) ----
    int F = 0, K = 16, M = 0, Npad = 0, TW = 0, R = 0;
    unsigned nco_rate = 1;
    unsigned long long idx = 0;        // DIRECT_current_index (ref :88, :437-440)
    int target_waves = 4096;           // resident waves the DDC grid is sized for
    int simds = 1024;                  // SIMDs of the device (4 per CU)
    int nch_max = 1;                   // chunks of the DIRECT launch (fixed nblk)
    int tails_nch = 0;                 // chunk count the tails buffer was sized for (0: not yet)
    int nch_force = 0;                 // GSDR_DDC_NCH: experiment override
    double waves_ratio = 1.3;          // grid waves / resident waves (autotuned in create)
    unsigned lds_bytes = 0;            // GSDR_DDC_LDS: dummy LDS per workgroup (occupancy cap)
    int prefetch = 1;                  // GSDR_DDC_PREFETCH: L2 prefetch of the IQ stream
    std::vector<float> window;         // taps (DIRECT) / PFB window / VNA profile, real part
    float *d_taps_t = nullptr;
    float *d_taps_p = nullptr;         // zero-padded [nsub*K+2][FP] copy for ddc_flat_kernel
    bool pipe = false;                 // ddc_flat_kernel (F <= 4) instead of ddc_kernel
    int pad = 0;                       // samples the flat kernel reads past a block (nsub*K - M)
    float2 *d_stage = nullptr;         // padded copy of the input when pad > 0 (DIRECT only)
    float2 *d_btab = nullptr;
    double2 *d_wk = nullptr, *d_wrem = nullptr;
    unsigned *d_fmod = nullptr;
    float2 *d_tails = nullptr;
    float2 *d_carry[2] = {nullptr, nullptr};
    int parity = 0;
    // ---- DDC on the matrix cores (ddc_mfma.hip) ----
    bool mfma = false;
    bool few = false;                  // DIRECT: a handful of tones at a long decimation run ddc_few_kernel (a wave per chunk and tone)
    int mf_TT = 1, mf_PK = 32, mf_W = 4;   // tone tiles per wave, phasor block, waves per workgroup
    gsdr::MfmaKernel mf_kind = gsdr::MfmaKernel::AsmRing;
    gsdr::MfmaShape mf{};              // fields that do not change between calls
    int last_rt = 0;                   // row tiles per workgroup of the last launch (describe())
    bool w8_auto = true;               // GSDR_MFMA_W8: 8-wave workgroups for in-order launches of one round
    // pre-converted operands (ddc_convert_kernel + ddc_mfma_ring16p_kernel) for launches of many
    // rounds: one image set per staging set (the main kernels of the calls in flight read theirs)
    bool prec = false;                 // image sets allocated: the path may be chosen
    int prec_mode = -1;                // GSDR_MFMA_PREC
    uint4 *d_img[kStageSets] = {};
    uint4 *d_bfrag = nullptr;
    float2 *d_ptab = nullptr, *d_dtab = nullptr;
    float *d_mtaps = nullptr;
    unsigned *d_mfmod = nullptr;
    // kScaleSlots tables of segment maxima (absmax_kernel), one per call in turn; seg_k blocks of M samples per segment
    unsigned *d_segmax = nullptr;
    int seg_k = 1, nseg_alloc = 0;
    // [carry | first rows' samples | zeros] and [last rows' samples | zeros], see absmax_kernel.
    // kStageSets of each, used in turn: the staging pass of call j writes set j (and the carry
    // part of head j+1) while the main kernels of calls j-1 .. j-kPipeStreams+1 may still read theirs.
    float2 *d_head[kStageSets] = {};
    float2 *d_tail[kStageSets] = {};
    // pipelined entries (gsdr_demod_submit*): consecutive DIRECT calls go to two compute
    // streams in turn, so that their kernels overlap (see pipeline_compute)
    hipStream_t s_main[kPipeStreams] = {};
    int pipe_streams = kPipeStreams;   // how many of them are used (GSDR_PIPE_STREAMS)
    hipEvent_t ev_abs[4] = {nullptr, nullptr, nullptr, nullptr};    // staging pass of call j done
    // Streams that carry work of this handle nobody has been ordered behind yet.  The carry, the
    // scale slots, the raw windows and the head/tail copies pass from one call to the next ON THE
    // DEVICE: a call that runs on another stream than its predecessors first joins them (an event
    // recorded on the old stream at that moment covers everything enqueued there before).
    // A stream of the CALLER's stays in this list only until the next call on the handle (which joins it and
    // forgets it): include/gsdr.h asks the caller to keep a stream alive that long.  (An event of the handle's own
    // recorded behind every call would lift that condition, and was tried: an event record costs 3 - 4 us of stream
    // time, which doubled the step period of the chirp path -- 5.1 -> 9.2 us -- and showed in every in-order figure.)
    // If recording on a remembered stream returns an error, the entry is dropped and the call goes on.  (A stream
    // destroyed before that is beyond help: HIP faults inside hipEventRecord, tests/test_gpu_parity.py.)
    std::vector<hipStream_t> dirty;
    hipEvent_t ev_join = nullptr;
    bool pipe_overlap = false;         // set around the compute of an overlapped call
    bool pipe_overlap_allowed = true;  // GSDR_PIPE_OVERLAP, read when the pipeline is created
    unsigned long long pipe_seq = 0;   // overlapped calls so far
    unsigned long long call_no = 0;    // absmax slot rotation
    // ---- TONES ----
    std::vector<int> bins;
    int nfft = 0, batching = 0;
    gsdr_buffer_helper bh{};
    // raw_input (ref :143): kStageSets windows used in turn.  The staging of call j copies the
    // unconsumed end of window j-1 to the front of window j and appends the new buffer, so the
    // kernels of call j-1 may still read their window (the reference moves it in place, :504-509).
    float2 *d_win[kStageSets] = {};
    unsigned long long win_seq = 0;    // TONES/NOISE calls so far
    long long prev_spare_begin = 0, prev_spare_samples = 0;
    // ---- NOISE through the FFT stage (fft_kernels.hip) ----
    bool noise_fft = false;
    gsdr::FftPlan fft{};
    float2 *d_fft_a = nullptr, *d_fft_b = nullptr;   // frames / scratch, batching * max(nfft, m) each
    float2 *d_fft_c = nullptr;                       // TONES through these stages: the spectra the bins are picked from
    float *d_fft_win = nullptr;                      // the PFB window on the device
    // ---- the parameters this handle was created with (gsdr_demod_prepare's rehearsal builds a twin) ----
    gsdr_param_c pc{};
    std::vector<int> pc_wave_type, pc_freq, pc_chirp_f, pc_swipe_s;
    std::vector<float> pc_chirp_t;
    // ---- TONES / NOISE, a frame per workgroup: filter + in-LDS transform + bin selection (fft_kernels.hip) ----
    bool pfb_lds = false;
    bool pfb_blue = false;                           // ... through Bluestein's identity (h->fft holds chirp, transform, twiddles)
    bool pfb_cu = false;                             // ... by the run-per-compute-unit kernel (pfb_cu_kernel)
    float2 *d_pfb_tw = nullptr;                      // w_nfft^k
    int *d_pfb_sel = nullptr;                        // TONES: bin of every output column
    float2 *d_pfb_carry[kStageSets] = {};            // the samples a call leaves over (at most F*nfft)
    // ---- CHIRP ----
    ChirpShape cs{};
    int ppt = 0;
    gsdr_vna_helper vh{};
    float *d_profile = nullptr;
    float2 *d_chirp_part = nullptr;    // partial sums of chirp_lockin_split_kernel (kChirpPartials float2)
    float2 *d_ccarry[2] = {nullptr, nullptr};
    int cparity = 0;
    int carry_len = 0;                 // spare_size (ref :54,:369)
    unsigned long long last_index = 0; // ref :217,:355

    // ---- profiling ----
    bool prof = false;
    int prof_every = 1;                // time every n-th launch (an event pair costs ~3 us of stream time)
    unsigned long long prof_seen = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    size_t ev_used = 0;
    const char *kernel_name = "none";
};

namespace {

#define HIPCHK(h, expr)                                                                   \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            (h)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                 \
            return -1;                                                                    \
        }                                                                                 \
    } while (0)

int fail_create(gsdr_demod *h, const std::string &msg) {
    g_create_error = msg;
    if (h) gsdr_demod_close(h);
    return -1;
}

template <typename T>
hipError_t dev_alloc(T **p, size_t count) {
    *p = nullptr;
    if (count == 0) count = 1;
    return hipMalloc(reinterpret_cast<void **>(p), count * sizeof(T));
}

template <typename T>
hipError_t upload(T **dst, const std::vector<T> &src) {
    hipError_t e = dev_alloc(dst, src.size());
    if (e != hipSuccess) return e;
    if (src.empty()) return hipSuccess;
    return hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice);
}

// exp(-2*pi*i * ph/rate) for an exact integer phase, in double.
inline void phasor(unsigned long long ph, unsigned rate, double &re, double &im) {
    const double a = 2.0 * M_PI * ((double)ph / (double)rate);
    re = std::cos(a);
    im = -std::sin(a);
}

// Builds the per-tone NCO tables of the DDC kernel (see ddc_kernels.hip):
//   fmod[n] = f_n mod rate, btab[lo][n] = w_n^lo, wk[n] = w_n^K, wrem[n] = w_n^R.
int build_nco_tables(gsdr_demod *h, const std::vector<long long> &tone, unsigned rate) {
    const int Npad = h->Npad, K = h->K;
    std::vector<unsigned> fmod(Npad, 0u);
    std::vector<float2> btab((size_t)K * Npad);
    std::vector<double2> wk(Npad), wrem(Npad);
    for (int n = 0; n < Npad; ++n) {
        unsigned long long fm = 0;
        if (n < h->ddc_channels) {
            long long r = tone[n] % (long long)rate;
            if (r < 0) r += rate;
            fm = (unsigned long long)r;
        }
        fmod[n] = (unsigned)fm;
        for (int lo = 0; lo < K; ++lo) {
            double re, im;
            phasor((fm * (unsigned long long)lo) % rate, rate, re, im);
            btab[(size_t)lo * Npad + n] = make_float2((float)re, (float)im);
        }
        double re, im;
        phasor((fm * (unsigned long long)K) % rate, rate, re, im);
        wk[n] = make_double2(re, im);
        phasor((fm * (unsigned long long)h->R) % rate, rate, re, im);
        wrem[n] = make_double2(re, im);
    }
    HIPCHK(h, upload(&h->d_fmod, fmod));
    HIPCHK(h, upload(&h->d_btab, btab));
    HIPCHK(h, upload(&h->d_wk, wk));
    HIPCHK(h, upload(&h->d_wrem, wrem));
    return 0;
}

// taps_t[m*F + j] = h[j*M + m]: the F tap phases of input sample m, contiguous.
int upload_taps_transposed(gsdr_demod *h) {
    std::vector<float> t((size_t)h->M * h->F);
    for (int j = 0; j < h->F; ++j)
        for (int m = 0; m < h->M; ++m) t[(size_t)m * h->F + j] = h->window[(size_t)j * h->M + m];
    HIPCHK(h, upload(&h->d_taps_t, t));
    if (h->pipe) {
        const int FP = h->F == 3 ? 4 : h->F;
        std::vector<float> p((size_t)(h->M + h->pad + 2) * FP, 0.f);
        for (int j = 0; j < h->F; ++j)
            for (int m = 0; m < h->M; ++m) p[(size_t)m * FP + j] = h->window[(size_t)j * h->M + m];
        HIPCHK(h, upload(&h->d_taps_p, p));
    }
    return 0;
}

// Chunk count closest to `want` (within +-20 %) that splits nblk blocks most
// evenly: the launch ends with its longest chunk.
long long balanced_near(long long want, int nblk, long long cap) {
    if (want < 1) want = 1;
    if (want > cap) want = cap;
    long long lo = want - want / 5, hi = want + want / 5;
    if (lo < 1) lo = 1;
    if (hi > cap) hi = cap;
    long long best_nch = want;
    double best = -1.0;
    for (long long nch = lo; nch <= hi; ++nch) {
        const long long longest = (nblk + nch - 1) / nch;
        const double balance = ((double)nblk / (double)nch) / (double)longest;
        const double score = balance - 0.05 * std::fabs((double)(nch - want)) / (double)want;
        if (score > best) {
            best = score;
            best_nch = nch;
        }
    }
    return best_nch;
}

// Number of chunks the blocks of one launch are cut into (one wave per chunk x
// 64 tones).  `h->waves_ratio` = grid waves / resident waves; 1.3 by default
// (measured on C3: a grid 1.3x what is resident -- the dispatcher hands the
// surplus workgroups to whichever CU frees up first -- beats an exactly
// resident one by 9 %), replaced by the autotuned value when create() could
// time the candidates (profiles/r01_chunk_sweeps.log shows why a fixed rule is
// not enough: C2 is best at 0.8, C3 at 1.3-1.7).
int pick_chunks(const gsdr_demod *h, int nblk) {
    if (nblk <= 0) return 1;
    const int TW = h->TW > 0 ? h->TW : 1;
    const long long cap = (h->F > 1) ? nblk / (h->F - 1) : nblk;  // every chunk >= F-1 blocks
    if (cap < 1) return 1;
    long long nch = h->nch_force > 0
                        ? (h->nch_force < cap ? h->nch_force : cap)
                        : balanced_near((long long)(h->waves_ratio * h->target_waves) / TW, nblk, cap);
    // never more chunks than the tails buffer has slots for (0 = not allocated yet)
    if (h->tails_nch > 0 && nch > h->tails_nch) nch = h->tails_nch;
    return (int)(nch < 1 ? 1 : nch);
}

int setup_ddc_common(gsdr_demod *h, int F, int M, unsigned rate,
                     const std::vector<long long> &tone, int max_nblk, bool allow_flat = true) {
    h->F = F;
    h->M = M;
    h->nco_rate = rate;
    if (h->ddc_channels <= 0) h->ddc_channels = h->N;
    h->Npad = ((h->ddc_channels + 63) / 64) * 64;
    h->TW = h->Npad / 64;
    // ddc_flat_kernel (packed math, pipelined scalar loads) covers F <= 4 and is the engine of
    // GSDR_DDC_MFMA=0, where no matrix-core loop runs on the GPU at all.  With the matrix cores enabled,
    // a shape they do not take (windows shorter than a buffer row, ...) goes to the generic ddc_kernel:
    // it is compiled without packed FP32, which is not safe beside the matrix-core loop of another
    // handle on the same GPU (rule 3).  GSDR_DDC_PIPE=0 forces the generic kernel.
    h->pipe = allow_flat && env_int("GSDR_DDC_PIPE", 1) != 0 && F <= 4 && env_int("GSDR_DDC_MFMA", 1) == 0;
    if (h->pipe) {
        // sub-block length: whole sub-blocks per block, cheapest total
        // (padded samples + ~3.3 sample-equivalents of fold work per sub-block).  40 since round 3: half the
        // folds and phasor steps of 20 at 124 instead of 82 registers (4 waves per SIMD instead of 5):
        // C3 460 -> 440 us in order (profiles/r03_flat_k.log)
        int forced = env_int("GSDR_DDC_K", 0);
        if (forced != 12 && forced != 16 && forced != 20 && forced != 40) forced = 0;
        double best = 1e300;
        for (int k : {40, 20, 16, 12}) {
            if (forced && forced != k) continue;
            const long long nsub = (M + k - 1) / k;
            const double cost = (double)nsub * (k + 3.3);
            if (cost < best) {
                best = cost;
                h->K = k;
            }
        }
        const int nsub = (M + h->K - 1) / h->K;
        h->pad = nsub * h->K - M;
        h->R = M - (nsub - 1) * h->K;   // length of the last sub-block in real samples
    } else {
        h->K = env_int("GSDR_DDC_K", 16) == 32 ? 32 : 16;
        h->R = M % h->K;
    }
    int cus = 256;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess)
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    // resident waves per SIMD of the kernel actually used (VGPR-limited)
    const int wps = env_int("GSDR_DDC_WAVES_PER_SIMD", h->pipe ? 6 : 4);
    h->simds = cus * 4;
    h->target_waves = h->simds * (wps > 0 ? wps : 4);
    h->nch_force = env_int("GSDR_DDC_NCH", 0);
    h->lds_bytes = (unsigned)env_int("GSDR_DDC_LDS", 0);
    h->prefetch = env_int("GSDR_DDC_PREFETCH", 1);
    // the tails buffer must hold the largest chunk count any launch may pick:
    // largest autotune ratio (1.7) x the +20 % window of balanced_near(), or the
    // forced count, but never more than one chunk per F-1 blocks
    {
        const long long cap = (F > 1) ? max_nblk / (F - 1) : max_nblk;
        long long worst = (long long)(1.7 * 1.2 * h->target_waves) / h->TW + 2;
        if (h->nch_force > worst) worst = h->nch_force;
        if (worst > cap) worst = cap;
        h->tails_nch = (int)(worst < 1 ? 1 : worst);
    }
    h->waves_ratio = 1.3;
    h->nch_max = pick_chunks(h, max_nblk);
    if (build_nco_tables(h, tone, rate)) return -1;
    if (upload_taps_transposed(h)) return -1;
    const size_t tail_elems = (size_t)(h->tails_nch + 1) * (size_t)(F > 1 ? F - 1 : 1) * h->Npad;
    HIPCHK(h, dev_alloc(&h->d_tails, tail_elems));
    HIPCHK(h, hipMemset(h->d_tails, 0, tail_elems * sizeof(float2)));
    return 0;
}

// fields of DdcShape the kernels derive nothing from on the device
void finish_shape(DdcShape &sh) {
    sh.cbase = sh.nblk / (sh.nch > 0 ? sh.nch : 1);
    sh.crem = sh.nblk % (sh.nch > 0 ? sh.nch : 1);
    sh.rate_magic = 0xffffffffffffffffULL / sh.rate;
    sh.inv_rate = 1.0 / (double)sh.rate;
    sh.m_mod_rate = (unsigned)sh.M % sh.rate;
}

// Tables and fixed shape of ddc_mfma_kernel.  `direct`: rows reach F-1 blocks
// back into the previous buffer (raw-sample carry); otherwise (TONES/NOISE) row o
// starts at block o of the raw window.
int setup_mfma(gsdr_demod *h, bool direct, const std::vector<long long> &tone) {
    const int F = h->F, M = h->M;
    const unsigned rate = h->nco_rate;
    h->mf_TT = env_int("GSDR_MFMA_TT", 1) == 2 ? 2 : 1;
    h->mf_PK = env_int("GSDR_MFMA_PK", 32) == 16 ? 16 : 32;
    h->mf_W = env_int("GSDR_MFMA_W", 4);
    if (h->mf_W != 2 && h->mf_W != 4) h->mf_W = 4;
    if (h->mf_W > h->mf_PK / 8) h->mf_W = h->mf_PK / 8;   // every wave converts whole k-steps
    // the assembly main loops exist for the default shape only; GSDR_MFMA_ASM: 4 = LDS operand ring
    // on v_mfma_f32_16x16x32_f16 (default since round 2: the same cycles per FLOP for less energy,
    // +7 % on C3 under the power cap), 5 = that loop for workgroups of eight waves, 2 = the ring on
    // v_mfma_f32_32x32x16_f16 (round 1's production kernel), 0 = the compiler-scheduled kernel (A/B runs, tests)
    const int asm_kind = env_int("GSDR_MFMA_ASM", 4);
    const bool asm_shape = h->mf_TT == 1 && h->mf_PK == 32 && h->mf_W == 4;
    h->mf_kind = gsdr::MfmaKernel::Cxx;
    if (asm_kind == 2 && asm_shape) h->mf_kind = gsdr::MfmaKernel::AsmRing;
    if (asm_kind == 4 && asm_shape) h->mf_kind = gsdr::MfmaKernel::AsmRing16;      // tools/gen_ddc_mfma_ring16.py
    if (asm_kind == 5 && asm_shape) h->mf_kind = gsdr::MfmaKernel::AsmRing16W8;    // tools/gen_ddc_mfma_ring16w8.py
    gsdr::MfmaPlan pl{};
    pl.TT = h->mf_TT;
    pl.PK = h->mf_PK;
    pl.M = M;
    pl.MF = M * F;
    pl.nk8 = (pl.MF + 7) / 8;
    pl.rate = rate;
    pl.x16 = h->mf_kind == gsdr::MfmaKernel::AsmRing16 || h->mf_kind == gsdr::MfmaKernel::AsmRing16W8;
    const int nt32 = (h->ddc_channels + 31) / 32;
    pl.ntg = (nt32 + pl.TT - 1) / pl.TT;
    std::vector<unsigned> fmod_in(h->ddc_channels);
    for (int n = 0; n < h->ddc_channels; ++n) {
        long long r = tone[n] % (long long)rate;
        if (r < 0) r += rate;
        fmod_in[n] = (unsigned)r;
    }
    std::vector<uint4> bfrag;
    std::vector<float2> ptab, dtab;
    std::vector<float> taps;
    std::vector<unsigned> fmod;
    float unscale = 1.f;
    gsdr::mfma_build_tables(pl, fmod_in, h->window.data(), bfrag, ptab, dtab, taps, fmod, unscale);
    HIPCHK(h, upload(&h->d_bfrag, bfrag));
    HIPCHK(h, upload(&h->d_ptab, ptab));
    HIPCHK(h, upload(&h->d_dtab, dtab));
    HIPCHK(h, upload(&h->d_mtaps, taps));
    HIPCHK(h, upload(&h->d_mfmod, fmod));
    {
        // maxima per segment of the call's logical stream [carry | buffer] (TONES: the raw window, allocated
        // 2 * nfft * batching long): a segment is one block of M samples, several when blocks are short
        h->seg_k = M >= 64 ? 1 : (64 + M - 1) / M;
        const long long t_max = direct ? (long long)(F - 1) * M + h->L : 2LL * h->nfft * h->batching;
        const long long seg_len = (long long)h->seg_k * M;
        h->nseg_alloc = (int)((t_max + seg_len - 1) / seg_len) + 2;
        HIPCHK(h, dev_alloc(&h->d_segmax, (size_t)kScaleSlots * h->nseg_alloc));
        HIPCHK(h, hipMemset(h->d_segmax, 0, (size_t)kScaleSlots * h->nseg_alloc * sizeof(unsigned)));
    }
    gsdr::MfmaShape &sh = h->mf;
    sh.N = h->ddc_channels;
    sh.NT32 = pl.ntg * pl.TT;
    sh.ntg = pl.ntg;
    sh.ntq = (pl.ntg + h->mf_W - 1) / h->mf_W;
    sh.M = M;
    sh.MF = pl.MF;
    sh.F = F;
    sh.nk8 = pl.nk8;
    sh.woff = direct ? -(F - 1) : 0;
    sh.carry_len = direct ? (F - 1) * M : 0;
    sh.rate = rate;
    sh.rate_magic = 0xffffffffffffffffULL / rate;
    sh.inv_rate = 1.0 / (double)rate;
    sh.m_mod_rate = (unsigned)M % rate;
    sh.unscale = unscale;
#ifdef GSDR_TIMING_BUILD
    sh.timing_mode = env_int("GSDR_MFMA_TIMING", 0);   // ablation builds only (scratch/), never shipped
#else
    sh.timing_mode = 0;
#endif
    sh.rt = env_int("GSDR_MFMA_RT", 0);   // 0: chosen per launch in enqueue_mfma
    h->w8_auto = env_int("GSDR_MFMA_W8", 1) != 0;
    if (sh.rt < 0 || sh.rt > 2) sh.rt = 0;
    if (direct) {
        // row tile 0 and the last row tile read from copies with the carry in front
        // and zeros behind (sizes: ddc_mfma_kernel's reach, 8*nk8 samples per row)
        const size_t reach = (size_t)((pl.nk8 + pl.PK / 8 - 1) / (pl.PK / 8)) * pl.PK;   // whole phasor blocks
        const size_t head_n = (size_t)sh.carry_len + 32u * (size_t)M + reach + 8;
        const size_t tail_n = (size_t)(32 + F) * (size_t)M + reach + 8;
        for (int i = 0; i < kStageSets; ++i) {
            HIPCHK(h, dev_alloc(&h->d_head[i], head_n));
            HIPCHK(h, hipMemset(h->d_head[i], 0, head_n * sizeof(float2)));
            HIPCHK(h, dev_alloc(&h->d_tail[i], tail_n));
            HIPCHK(h, hipMemset(h->d_tail[i], 0, tail_n * sizeof(float2)));
        }
    }
    // Pre-converted operands (ddc_convert_kernel + ddc_mfma_ring16p_kernel).
    // GSDR_MFMA_PREC: 1 = always (tests), 0 = never, default = per launch in enqueue_mfma:
    //   in-order entries      launches of four rounds of workgroups or more (-9 % at 16 k ... 64 k tones;
    //                         below that the pass, which runs in front of the loop there, costs more);
    //   overlapped entries    launches of half a round or more with windows of 32 blocks or more: the
    //                         pass of buffer j+1 runs beside the loop of buffer j (C3 139 -> 130.5 us per
    //                         buffer, TONES 1024/1230 76.8 -> 74.0; C2, 13 blocks: neutral, not used).
    if (h->mf_kind == gsdr::MfmaKernel::AsmRing16) {
        h->prec_mode = env_int("GSDR_MFMA_PREC", -1);
        const long long max_rows = direct ? h->L / M : (long long)h->batching;
        const long long ngt_max = (max_rows + 31) / 32;
        const long long nhi = (pl.nk8 + 3) / 4;
        const long long wgs4 = ((ngt_max + 7) / 8) * 8 * sh.ntq;
        bool want = h->prec_mode == 1 ||
                    (h->prec_mode < 0 && (wgs4 >= 4LL * (h->simds / 2) || (wgs4 >= h->simds / 4 && nhi >= 32)));
        const size_t img_n = (size_t)ngt_max * (size_t)nhi * 512;   // uint4 per image set
        if (want && img_n * sizeof(uint4) > (size_t)8 << 30) want = false;
        for (int i = 0; i < kStageSets && want; ++i) HIPCHK(h, dev_alloc(&h->d_img[i], img_n));
        h->prec = want;
    }
    h->mfma = true;
    h->kernel_name = gsdr::ddc_mfma_kernel_name(h->mf_kind);
    return 0;
}

// absmax pass over `in` + ddc_mfma_kernel over `nout` rows.  DIRECT: raw == nullptr,
// the rows read `in` (and its head/tail copies).  TONES: the pass also appends
// `in` to the raw window at raw_new0 and the rows read raw[0 .. nx).
int enqueue_mfma(gsdr_demod *h, const float2 *in, float2 *raw, long long raw_new0, long long nx,
                 int nout, unsigned idx_base, float2 *out, hipStream_t st, const float2 *spare_src = nullptr,
                 long long spare_n = 0);

// Orders `st` behind every stream in h->dirty for which keep(s) is false, and forgets those.
template <typename Keep>
int join_streams(gsdr_demod *h, hipStream_t st, Keep keep) {
    size_t w = 0;
    for (size_t i = 0; i < h->dirty.size(); ++i) {
        hipStream_t s = h->dirty[i];
        if (s == st || keep(s)) {
            h->dirty[w++] = s;
            continue;
        }
        if (!h->ev_join) HIPCHK(h, hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
        if (hipEventRecord(h->ev_join, s) != hipSuccess) {
            // nothing that could still be waited for: forget the stream rather than fail every later call
            (void)hipGetLastError();
            continue;
        }
        HIPCHK(h, hipStreamWaitEvent(st, h->ev_join, 0));
    }
    h->dirty.resize(w);
    bool have = false;
    for (hipStream_t s : h->dirty) have |= (s == st);
    if (!have) h->dirty.push_back(st);
    return 0;
}

int record_begin(gsdr_demod *h, hipStream_t st, hipEvent_t *stop) {
    *stop = nullptr;
    if (!h->prof || h->ev_used >= (size_t)kMaxEvents) return 0;
    if (h->prof_seen++ % (unsigned long long)h->prof_every != 0) return 0;
    if (h->ev_used == h->ev_pool.size()) {
        hipEvent_t a, b;
        HIPCHK(h, hipEventCreate(&a));
        HIPCHK(h, hipEventCreate(&b));
        h->ev_pool.emplace_back(a, b);
    }
    HIPCHK(h, hipEventRecord(h->ev_pool[h->ev_used].first, st));
    *stop = h->ev_pool[h->ev_used].second;
    h->ev_used++;
    return 0;
}

// ---------------------------------------------------------------------------
// per-mode enqueue
// ---------------------------------------------------------------------------

// Grid size of the DDC launch, measured instead of guessed: times the real
// launch (ddc kernel + fixup) on scratch buffers for a few grid/resident ratios
// and keeps the fastest.  Runs once in create(); GSDR_DDC_AUTOTUNE=0 keeps 1.3.
int autotune_chunks(gsdr_demod *h, int nblk) {
    h->waves_ratio = 1.3;
    if (!h->pipe || h->nch_force > 0 || env_int("GSDR_DDC_AUTOTUNE", 1) == 0 || nblk < 1) {
        h->nch_max = pick_chunks(h, nblk);
        return 0;
    }
    const size_t n_in = (size_t)nblk * h->M + h->pad + 8;
    const size_t n_out = (size_t)nblk * h->ddc_channels;
    float2 *x = nullptr, *y = nullptr;
    HIPCHK(h, dev_alloc(&x, n_in));
    if (dev_alloc(&y, n_out) != hipSuccess) {
        (void)hipFree(x);
        h->err = "autotune scratch allocation failed";
        return -1;
    }
    (void)hipMemset(x, 0x3c, n_in * sizeof(float2));  // 0.0115f everywhere: finite, non-zero
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const double cand[] = {0.8, 1.0, 1.3, 1.7};
    double best_ms = 1e30, best_ratio = 1.3;
    int rc = 0;
    for (double r : cand) {
        h->waves_ratio = r;
        DdcLaunch a{};
        a.x = x;
        a.taps_t = h->d_taps_t;
        a.taps_p = h->d_taps_p;
        a.btab = h->d_btab;
        a.wk = h->d_wk;
        a.wrem = h->d_wrem;
        a.fmod = h->d_fmod;
        a.out = y;
        a.tails = h->d_tails;
        a.tails_nch = h->tails_nch;
        a.pipe = true;
        a.lds_bytes = h->lds_bytes;
        a.sh.N = h->ddc_channels;
        a.sh.Npad = h->Npad;
        a.sh.TW = h->TW;
        a.sh.rate = h->nco_rate;
        a.sh.M = h->M;
        a.sh.nblk = nblk;
        a.sh.nch = pick_chunks(h, nblk);
        a.sh.xlast = (long long)nblk * h->M + h->pad - 4;
        a.sh.prefetch = h->prefetch;
        finish_shape(a.sh);
        float ms = 0.f;
        for (int it = 0; it < 4 && !rc; ++it) {  // first iteration warms up
            if (it == 1) rc |= hipEventRecord(e0, h->stream) != hipSuccess;
            rc |= gsdr::launch_ddc(h->F, h->K, a, h->stream, nullptr) != hipSuccess;
        }
        rc |= hipEventRecord(e1, h->stream) != hipSuccess;
        rc |= hipEventSynchronize(e1) != hipSuccess;
        if (rc || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) {
            rc = 1;
            break;
        }
        if (ms < best_ms) {
            best_ms = ms;
            best_ratio = r;
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(x);
    (void)hipFree(y);
    if (rc) {
        h->err = "autotune launch failed";
        return -1;
    }
    h->waves_ratio = best_ratio;
    h->nch_max = pick_chunks(h, nblk);
    return 0;
}

int enqueue_mfma(gsdr_demod *h, const float2 *in, float2 *raw, long long raw_new0, long long nx,
                 int nout, unsigned idx_base, float2 *out, hipStream_t st, const float2 *spare_src,
                 long long spare_n) {
    // tables of segment maxima: this call's, and the one the staging pass clears for the next call.
    // kScaleSlots of them, so that the pass of call j (filling table j, clearing table j+1) leaves
    // alone what the main kernels of the calls still in flight read (tables j-1 .. j-kPipeStreams+1).
    const int cur = (int)(h->call_no % kScaleSlots), next = (int)((h->call_no + 1) % kScaleSlots);
    const int hs = (int)(h->call_no % kStageSets), hs_next = (int)((h->call_no + 1) % kStageSets);
    gsdr::MfmaLaunch a{};
    a.sh = h->mf;
    a.sh.nout = nout;
    a.sh.ngt = (nout + 31) / 32;
    a.sh.nx = nx;
    a.sh.idx_base = idx_base;
    a.sh.seg_k = h->seg_k;
    gsdr::StageLaunch sg{};
    sg.x = in;
    sg.n = h->L;
    sg.seg = h->d_segmax + (size_t)cur * h->nseg_alloc;
    sg.seg_clear = h->d_segmax + (size_t)next * h->nseg_alloc;
    sg.nseg_alloc = h->nseg_alloc;
    sg.seg_len = (long long)h->seg_k * a.sh.M;
    if (a.sh.rt == 0) {
        // Two row tiles per workgroup (the second keeps the phasor images: 64 KiB less to load, one
        // preamble less, half as many workgroups).  Measured at decim 100 (13-block windows) for
        // 128 .. 2048 tones, in order and overlapped (scratch/rt_probe.py): it pays in exactly one
        // place, a launch of between one and two rounds of workgroups in the overlapped entries
        // (256 tones, 626 workgroups: 31.3 -> 29.4 us per buffer), where the neighbouring launches
        // fill the compute units that 313 longer workgroups leave free; everywhere else it is
        // neutral or costs up to 15 % (in order, few workgroups).
        const int nblk = (a.sh.nk8 + 3) / 4;
        const long long wgs = (long long)a.sh.ngt * a.sh.ntq;
        a.sh.rt = h->pipe_overlap && nblk <= 32 && wgs >= 512 && wgs < 1024 ? 2 : 1;
    }
    h->last_rt = a.sh.rt;
    if (raw) {
        // TONES: one pass brings the carried samples to the front of this call's raw window,
        // appends the buffer and takes its maximum; every row reads the window itself
        // (allocated twice as long as it gets)
        if (spare_n != raw_new0) {
            h->err = "raw window bookkeeping out of step";
            return -1;
        }
        sg.b = spare_src;
        sg.nb = spare_n;
        sg.b_dst = raw;
        sg.head_cur = raw + raw_new0;
        sg.head_n = h->L;
        sg.tail0 = h->L;
        HIPCHK(h, gsdr::launch_absmax(sg, st));
        if (h->pipe_overlap) HIPCHK(h, hipEventRecord(h->ev_abs[h->pipe_seq % 4], st));
        a.x = a.head = a.tail = raw;
        a.sh.tail0 = 0;
    } else {
        const int cl = a.sh.carry_len;
        const long long t0 = a.sh.ngt > 1 ? (long long)(32 * (a.sh.ngt - 1) + a.sh.woff) * a.sh.M : h->L;
        const int ks = h->mf_PK / 8;
        long long head_n = 32LL * a.sh.M + (long long)((a.sh.nk8 + ks - 1) / ks) * h->mf_PK + 8;
        if (head_n > h->L) head_n = h->L;
        sg.b = h->d_head[hs];       // the carry: the last cl samples of the previous buffer, left there by its pass
        sg.nb = cl;
        sg.head_cur = h->d_head[hs];
        sg.head_n = head_n;
        sg.head_next = h->d_head[hs_next];
        sg.carry_len = cl;
        sg.tail = h->d_tail[hs];
        sg.tail0 = t0;
        HIPCHK(h, gsdr::launch_absmax(sg, st));
        if (h->pipe_overlap) HIPCHK(h, hipEventRecord(h->ev_abs[h->pipe_seq % 4], st));
        a.x = in;
        a.head = h->d_head[hs];
        a.tail = h->d_tail[hs];
        a.sh.tail0 = t0;
    }
    a.taps = h->d_mtaps;
    a.bfrag = h->d_bfrag;
    a.ptab = h->d_ptab;
    a.dtab = h->d_dtab;
    a.fmod = h->d_mfmod;
    a.segmax = sg.seg;
    a.out = out;
    hipEvent_t stop = nullptr;
    if (record_begin(h, st, &stop)) return -1;
    // One launch at a time (in-order entries) that needs the two workgroups per compute unit the
    // 4-wave kernel is resident with: the hardware serves the older of the two first, the younger ends
    // 40 % later and runs the tail alone.  If the launch fits the compute
    // units as 8-wave workgroups (same tables, same loop, the two waves of a SIMD barrier-coupled
    // partners), that kernel ends 4-7 % earlier (C3: 145 against 152-158 us).  The overlapped entries
    // keep the 4-wave kernel: there the next buffer's workgroups fill the slots the older ones free.
    gsdr::MfmaKernel kind = h->mf_kind;
    bool use_prec = false;
    if (kind == gsdr::MfmaKernel::AsmRing16 && h->prec && a.sh.rt <= 1) {
        const long long wgs4 = (long long)((a.sh.ngt + 7) / 8) * 8 * a.sh.ntq;
        const int nhi = (a.sh.nk8 + 3) / 4;
        use_prec = h->prec_mode == 1 || wgs4 >= 4LL * (h->simds / 2) ||
                   (h->pipe_overlap && wgs4 >= h->simds / 4 && nhi >= 32);
    }
    if (use_prec) {
        kind = gsdr::MfmaKernel::AsmRing16P;
        a.img = h->d_img[hs];
    } else if (kind == gsdr::MfmaKernel::AsmRing16 && !h->pipe_overlap && h->w8_auto) {
        const long long wgs4 = (long long)((a.sh.ngt + 7) / 8) * 8 * a.sh.ntq;
        const long long wgs8 = (long long)((a.sh.ngt + 7) / 8) * 8 * ((a.sh.ntg + 7) / 8);
        if (a.sh.rt <= 1 && wgs4 > h->simds / 4 && wgs4 <= h->simds / 2 && wgs8 <= h->simds / 4)
            kind = gsdr::MfmaKernel::AsmRing16W8;
    }
    h->kernel_name = gsdr::ddc_mfma_kernel_name(kind);
    HIPCHK(h, gsdr::launch_ddc_mfma(kind, h->mf_TT, h->mf_PK, h->mf_W, a, st));
    if (stop) HIPCHK(h, hipEventRecord(stop, st));
    h->call_no++;
    return 0;
}

// ref: process_direct, cpp/USRP_demodulator.cpp:400-464
int enqueue_direct(gsdr_demod *h, const float2 *in, float2 *out, hipStream_t st) {
    DdcLaunch a{};
    a.x = in;
    a.taps_t = h->d_taps_t;
    a.taps_p = h->d_taps_p;
    a.pipe = h->pipe && h->decim > 0 && h->L >= 4;
    a.few = h->few;
    a.lds_bytes = h->lds_bytes;
    a.sh.prefetch = h->prefetch;
    a.btab = h->d_btab;
    a.wk = h->d_wk;
    a.wrem = h->d_wrem;
    a.fmod = h->d_fmod;
    a.out = out;
    a.sh.N = h->N;
    a.sh.Npad = h->Npad;
    a.sh.TW = h->TW;
    a.sh.rate = h->nco_rate;
    a.sh.idx0 = h->idx;
    a.sh.g_off = 0;
    long long ret;
    hipEvent_t stop = nullptr;
    if (h->decim > 0 && h->mfma) {
        const unsigned rate = h->nco_rate;
        const unsigned back = (unsigned)(((unsigned long long)(h->F - 1) * h->M) % rate);
        const unsigned idx_base = (unsigned)((h->idx + rate - back) % rate);
        if (enqueue_mfma(h, in, nullptr, 0, h->L, (int)(h->L / h->M), idx_base, out, st)) return -1;
        ret = (long long)h->N * (h->L / h->M);               // :459
    } else if (h->decim > 0) {
        a.tails = h->d_tails;
        a.tails_nch = h->tails_nch;
        a.carry_in = h->d_carry[h->parity];
        a.carry_out = h->d_carry[h->parity ^ 1];
        a.sh.M = h->M;
        a.sh.nblk = (int)(h->L / h->M);
        a.sh.nch = h->nch_max;
        a.sh.xlast = h->L - 4;
        if (a.pipe && h->pad > 0) {
            // the last sub-block of a block reads `pad` samples past it (zero taps);
            // behind the last block that would leave the caller's buffer
            HIPCHK(h, hipMemcpyAsync(h->d_stage, in, (size_t)h->L * sizeof(float2),
                                     hipMemcpyDeviceToDevice, st));
            a.x = h->d_stage;
            a.sh.xlast = h->L + h->pad - 4;
        }
        finish_shape(a.sh);
        if (record_begin(h, st, &stop)) return -1;
        HIPCHK(h, gsdr::launch_ddc(h->F, h->K, a, st, stop));
        h->parity ^= 1;
        ret = (long long)h->N * (h->L / h->M);               // :459
    } else {
        a.sh.M = h->K;
        a.sh.total = h->L;
        a.sh.nblk = (int)((h->L + h->K - 1) / h->K);
        long long nch = h->target_waves / h->TW;
        if (nch < 1) nch = 1;
        if (nch > a.sh.nblk) nch = a.sh.nblk;
        a.sh.nch = (int)nch;
        finish_shape(a.sh);
        if (record_begin(h, st, &stop)) return -1;
        HIPCHK(h, gsdr::launch_mix(h->K, a, st));
        if (stop) HIPCHK(h, hipEventRecord(stop, st));
        ret = (long long)h->N * h->L;                        // :457
    }
    h->idx = (h->idx + (unsigned long long)h->L) % h->nco_rate;  // :437-440
    return (int)ret;
}

// ref: process_pfb_spec (decim == 0), cpp/USRP_demodulator.cpp:568-649: polyphase filter, forward FFT of
// every complete frame, all bins kept.  Frame bookkeeping and the carry of the unconsumed samples
// are those of enqueue_pfb (the raw windows rotate, see there).
int enqueue_noise_fft(gsdr_demod *h, const float2 *in, float2 *out, hipStream_t st) {
    const int cb = h->bh.current_batch;
    float2 *win = h->d_win[h->win_seq % kStageSets];
    if (h->prev_spare_samples > 0)   // :590-596 of the previous call
        HIPCHK(h, hipMemcpyAsync(win, h->d_win[(h->win_seq + kStageSets - 1) % kStageSets] + h->prev_spare_begin,
                                 (size_t)h->prev_spare_samples * sizeof(float2), hipMemcpyDeviceToDevice, st));
    HIPCHK(h, hipMemcpyAsync(win + h->bh.new_0, in, (size_t)h->L * sizeof(float2), hipMemcpyDeviceToDevice, st));  // :573-577
    if (cb > 0) {
        hipEvent_t stop = nullptr;
        if (record_begin(h, st, &stop)) return -1;
        HIPCHK(h, gsdr::launch_pfb_filter(win, h->d_fft_win, h->nfft, h->F, cb, h->d_fft_a, st));   // :580
        if (h->d_fft_c) {                                                                            // TONES: :531-540
            HIPCHK(h, gsdr::fft_forward(h->fft, h->d_fft_a, h->d_fft_c, h->d_fft_b, cb, st));
            HIPCHK(h, gsdr::launch_pfb_select(h->d_fft_c, h->nfft, cb, h->d_pfb_sel, h->ddc_channels, out, st));
        } else {
            HIPCHK(h, gsdr::fft_forward(h->fft, h->d_fft_a, out, h->d_fft_b, cb, st));               // :583
        }
        if (stop) HIPCHK(h, hipEventRecord(stop, st));
    }
    h->prev_spare_begin = h->bh.spare_begin;
    h->prev_spare_samples = h->bh.spare_samples > 0 ? h->bh.spare_samples : 0;
    h->win_seq++;
    const int ret = h->ddc_channels * cb;   // :546 (TONES), copy_size :638 (NOISE: every bin)
    gsdr_buffer_helper_update(&h->bh);      // :552 / :644
    return ret;
}

// ref: process_pfb (:486-565) and process_pfb_spec (:568-649), decim == 0: one launch, a frame per
// workgroup.  The logical raw_input is [what the previous call left over | the new buffer]; nothing is
// staged: the kernel reads both parts in place and writes this call's leftovers (:504-509, :590-596)
// into the other carry buffer.
int enqueue_pfb_lds(gsdr_demod *h, const float2 *in, float2 *out, hipStream_t st) {
    const int cb = h->bh.current_batch;
    const float2 *carry = h->d_pfb_carry[h->win_seq % kStageSets];
    float2 *carry_out = h->d_pfb_carry[(h->win_seq + 1) % kStageSets];
    const int spare_n = h->bh.spare_samples > 0 ? h->bh.spare_samples : 0;
    if (spare_n > h->nfft * h->F) {
        h->err = "PFB carry larger than a window";
        return -1;
    }
    const int *sel = h->mode == GSDR_NOISE ? nullptr : h->d_pfb_sel;
    const long long wlen = (long long)h->bh.new_0 + h->L;
    const gsdr::FftPlan *blue = h->pfb_blue ? &h->fft : nullptr;
    const float2 *tw = h->pfb_blue ? h->fft.d_tw : h->d_pfb_tw;
    hipEvent_t stop = nullptr;
    if (record_begin(h, st, &stop)) return -1;
    HIPCHK(h, gsdr::launch_pfb_lds(carry, h->bh.new_0, in, h->d_fft_win, tw, h->nfft, h->F, cb,
                                   sel, h->ddc_channels, out, carry_out, h->bh.spare_begin, spare_n, wlen, st, blue));
    if (stop) HIPCHK(h, hipEventRecord(stop, st));
    h->win_seq++;
    const int ret = h->ddc_channels * cb;  // :546 (TONES), copy_size :638 (NOISE)
    gsdr_buffer_helper_update(&h->bh);     // :552 / :644
    return ret;
}

// ref: process_pfb (decim == 0 branch), cpp/USRP_demodulator.cpp:486-565
int enqueue_pfb(gsdr_demod *h, const float2 *in, float2 *out, hipStream_t st) {
    if (h->pfb_lds) return enqueue_pfb_lds(h, in, out, st);
    if (h->noise_fft) return enqueue_noise_fft(h, in, out, st);
    // :491-495  new buffer goes after the carried samples
    const int cb = h->bh.current_batch;
    float2 *win = h->d_win[h->win_seq % kStageSets];
    const float2 *spare = h->prev_spare_samples > 0
                              ? h->d_win[(h->win_seq + kStageSets - 1) % kStageSets] + h->prev_spare_begin
                              : nullptr;
    if (!(cb > 0 && h->mfma)) {
        // :504-509 of the previous call (carry the unconsumed samples to the front), then :491-495
        if (spare)
            HIPCHK(h, hipMemcpyAsync(win, spare, (size_t)h->prev_spare_samples * sizeof(float2),
                                     hipMemcpyDeviceToDevice, st));
        HIPCHK(h, hipMemcpyAsync(win + h->bh.new_0, in, (size_t)h->L * sizeof(float2),
                                 hipMemcpyDeviceToDevice, st));
        if (h->pipe_overlap) HIPCHK(h, hipEventRecord(h->ev_abs[h->pipe_seq % 4], st));
    }
    if (cb > 0 && h->mfma) {
        if (enqueue_mfma(h, in, win, h->bh.new_0, (long long)(cb + h->F - 1) * h->M, cb, 0u, out, st, spare,
                         h->prev_spare_samples))
            return -1;
    } else if (cb > 0) {
        DdcLaunch a{};
        a.x = win;
        a.taps_t = h->d_taps_t;
        a.taps_p = h->d_taps_p;
        a.btab = h->d_btab;
        a.wk = h->d_wk;
        a.wrem = h->d_wrem;
        a.fmod = h->d_fmod;
        a.out = out;
        a.tails = h->d_tails;
        a.tails_nch = h->tails_nch;
        a.carry_in = nullptr;   // frames never reach back before raw_input[0]
        a.carry_out = nullptr;
        a.sh.N = h->ddc_channels;
        a.sh.Npad = h->Npad;
        a.sh.TW = h->TW;
        a.sh.rate = h->nco_rate;
        a.sh.idx0 = 0;          // raw_input[0] is always on the frame grid
        a.sh.M = h->M;
        a.sh.nblk = cb + h->F - 1;  // frame r spans blocks r .. r+F-1
        a.pipe = h->pipe;
        a.lds_bytes = h->lds_bytes;
        a.sh.prefetch = h->prefetch;
        a.sh.xlast = (long long)a.sh.nblk * h->M + h->pad - 4;  // a window is allocated twice as long
        a.sh.g_off = h->F - 1;      // DDC output G <-> frame r = G-(F-1)
        const int nch = pick_chunks(h, a.sh.nblk);
        a.sh.nch = nch;
        finish_shape(a.sh);
        hipEvent_t stop = nullptr;
        if (record_begin(h, st, &stop)) return -1;
        HIPCHK(h, gsdr::launch_ddc(h->F, h->K, a, st, stop));
    }
    // :504-509 the unconsumed samples go to the front of the next call's window (see above)
    h->prev_spare_begin = h->bh.spare_begin;
    h->prev_spare_samples = h->bh.spare_samples > 0 ? h->bh.spare_samples : 0;
    h->win_seq++;
    const int ret = h->ddc_channels * cb;  // :546 (TONES), copy_size :638 (NOISE)
    gsdr_buffer_helper_update(&h->bh);   // :552
    return ret;
}

// ref: process_chirp, cpp/USRP_demodulator.cpp:342-397
int enqueue_chirp(gsdr_demod *h, const float2 *in, float2 *out, hipStream_t st) {
    hipEvent_t stop = nullptr;
    int ret;
    if (h->decim <= 0) {
        if (record_begin(h, st, &stop)) return -1;
        HIPCHK(h, gsdr::launch_chirp_demod(in, out, h->L, h->last_index, h->cs, st));
        if (stop) HIPCHK(h, hipEventRecord(stop, st));
        ret = (int)h->L;                 // :390
    } else {
        const int valid = h->vh.valid_size;      // :361
        // chirp index of the first carried sample
        const unsigned long long idx0 =
            (h->last_index + h->cs.period - (unsigned long long)h->carry_len % h->cs.period) %
            h->cs.period;
        if (record_begin(h, st, &stop)) return -1;
        HIPCHK(h, gsdr::launch_chirp_lockin(h->d_ccarry[h->cparity], h->carry_len, in,
                                            h->d_profile, h->ppt, valid, out, idx0, h->cs, st, h->d_chirp_part,
                                            h->d_chirp_part ? kChirpPartials : 0));
        if (stop) HIPCHK(h, hipEventRecord(stop, st));
        // :369-380 the reference keeps the last new0 DEMODULATED samples; we
        // keep the same raw samples and re-demodulate them next call.
        const int new0 = h->vh.new0;
        if (new0 > 0) {
            // they are the tail of the logical stage [carry | in]
            const long long from_in = (long long)new0 <= h->L ? new0 : h->L;
            const long long from_carry = new0 - from_in;
            float2 *dst = h->d_ccarry[h->cparity ^ 1];
            if (from_carry > 0)
                HIPCHK(h, hipMemcpyAsync(dst, h->d_ccarry[h->cparity] + (h->carry_len - from_carry),
                                         (size_t)from_carry * sizeof(float2),
                                         hipMemcpyDeviceToDevice, st));
            HIPCHK(h, hipMemcpyAsync(dst + from_carry, in + (h->L - from_in),
                                     (size_t)from_in * sizeof(float2), hipMemcpyDeviceToDevice,
                                     st));
            h->cparity ^= 1;
        }
        h->carry_len = new0;
        gsdr_vna_helper_update(&h->vh);  // :382
        ret = valid;
    }
    h->last_index = (h->last_index + (unsigned long long)h->L) % h->cs.period;  // :355
    return ret;
}

}  // namespace

extern "C" {

const char *gsdr_last_error(const gsdr_demod *h) {
    return h ? h->err.c_str() : g_create_error.c_str();
}

gsdr_demod *gsdr_demod_create(const gsdr_param_c *p) {
    g_create_error.clear();
    if (!p) {
        g_create_error = "null parameters";
        return nullptr;
    }
    gsdr_demod *h = new gsdr_demod();
    {
        h->pc = *p;
        auto keep = [](auto &dst, const auto *src, int n) {
            dst.assign(src && n > 0 ? src : nullptr, src && n > 0 ? src + n : nullptr);
            return dst.empty() ? nullptr : dst.data();
        };
        h->pc.wave_type = keep(h->pc_wave_type, p->wave_type, p->n_wave_type);
        h->pc.freq = keep(h->pc_freq, p->freq, p->n_freq);
        h->pc.chirp_f = keep(h->pc_chirp_f, p->chirp_f, p->n_chirp_f);
        h->pc.swipe_s = keep(h->pc_swipe_s, p->swipe_s, p->n_swipe_s);
        h->pc.chirp_t = keep(h->pc_chirp_t, p->chirp_t, p->n_chirp_t);
    }

    // ---- mode selection, ref: USRP_demodulator.cpp:15-39 ----
    int last = GSDR_NODSP;
    if (p->n_wave_type > 0 && p->wave_type) last = p->wave_type[0];
    bool mixed = false;
    int chirps = 0;
    for (int i = 0; i < p->n_wave_type; ++i) {
        if (p->wave_type[i] != last) mixed = true;
        if (p->wave_type[i] == GSDR_CHIRP) chirps++;
    }
    if (chirps > 1) {
        fail_create(h, "Multiple chirp RX buffer demodulation has been requested. This feature is not implemented yet.");
        return nullptr;
    }
    if (mixed) {
        fail_create(h, "Mixed RX buffer demodulation has been requested. This feature is not implemented yet.");
        return nullptr;
    }
    h->mode = last;
    h->N = p->n_wave_type;
    h->L = p->buffer_len;
    h->decim = p->decim;
    if (h->L <= 0) {
        fail_create(h, "buffer_len must be positive");
        return nullptr;
    }

    {
        h->device = p->device_index;
        if (h->device >= 0 && hipSetDevice(h->device) != hipSuccess) {
            fail_create(h, "hipSetDevice failed (no such GPU?)");
            return nullptr;
        }
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        // ref: :41-44 low priority stream for tone modes, :186-189 high for chirp
        hipError_t e = hipStreamCreateWithPriority(&h->stream, hipStreamNonBlocking,
                                                   last == GSDR_CHIRP ? hi : lo);
        if (e != hipSuccess) {
            h->stream = nullptr;
            fail_create(h, std::string("cannot create a HIP stream: ") + hipGetErrorString(e));
            return nullptr;
        }
    }

    auto need = [&](bool ok, const char *msg) {
        if (!ok) fail_create(h, msg);
        return ok;
    };

    int rc = 0;
    switch (last) {
        case GSDR_DIRECT: {  // ref: :59-119
            if (!need(p->rate > 0, "rate must be positive")) return nullptr;
            if (!need(p->n_freq >= h->N && p->freq, "DIRECT needs one frequency per wave_type entry")) return nullptr;
            std::vector<long long> tone(p->freq, p->freq + h->N);
            if (h->decim > 0) {
                if (!need(h->L % h->decim == 0, "buffer_len must be a multiple of decim (ref: fir.cu:20)")) return nullptr;
                if (!need(p->pf_average >= 1 && p->pf_average <= kMaxF, "pf_average must be in [1,8] for DIRECT with decimation")) return nullptr;
                if (!need(h->decim <= 0x7fffffffLL / p->pf_average, "decim*pf_average overflows")) return nullptr;
                const int F = (int)p->pf_average, M = (int)h->decim;
                h->window.resize((size_t)M * F);
                // ref: :99 taps, cut-off 0.75/(2*decim) narrowed to float
                gsdr_make_sinc_window(M * F, (float)(0.75 / (M * 2)), h->window.data());
                rc = setup_ddc_common(h, F, M, (unsigned)p->rate, tone, (int)(h->L / M));
                if (!rc && h->pipe && h->pad > 0) {
                    const size_t n = (size_t)h->L + h->pad;
                    if (dev_alloc(&h->d_stage, n) != hipSuccess ||
                        hipMemset(h->d_stage, 0, n * sizeof(float2)) != hipSuccess) {
                        h->err = "staging allocation failed";
                        rc = -1;
                    }
                }
                h->kernel_name = h->pipe ? gsdr::ddc_flat_kernel_name() : gsdr::ddc_kernel_name();
                if (!rc) {
                    for (int i = 0; i < 2 && !rc; ++i) {
                        const size_t n = (size_t)(F > 1 ? F - 1 : 1) * h->Npad;
                        if (dev_alloc(&h->d_carry[i], n) != hipSuccess ||
                            hipMemset(h->d_carry[i], 0, n * sizeof(float2)) != hipSuccess) {
                            h->err = "carry allocation failed";
                            rc = -1;
                        }
                    }
                }
                // (rows read whole 32-sample phasor blocks: the padding behind a window must stay
                //  within the next block, or middle rows would read past the buffer)
                // A handful of tones at a long decimation: every engine below walks a block's samples in sequence per
                // tone lane / matrix column, and a launch is as long as one workgroup's walk (72 us per 1 M-sample
                // buffer for 1 ... 256 tones at decim 1000).  ddc_few_kernel splits the block over the lanes of a wave
                // per (chunk, tone): 16 tones at decim 1000 in 15 us (profiles/r03_shape_sweep.log).  GSDR_DDC_FEW=0: off.
                h->few = !rc && env_int("GSDR_DDC_MFMA", 1) != 0 && env_int("GSDR_DDC_FEW", 1) != 0 && !h->pipe &&
                         M >= 512 && h->N <= 32 && h->TW == 1;
                if (h->few) h->kernel_name = gsdr::ddc_few_kernel_name();
                if (!rc && !h->few && env_int("GSDR_DDC_MFMA", 1) != 0 && h->L / M >= F - 1 && h->L >= 4 && F <= 33 &&
                    (M * F + 31) / 32 * 32 - M * F <= M)
                    rc = setup_mfma(h, /*direct=*/true, tone);
                if (!rc && !h->mfma) rc = autotune_chunks(h, (int)(h->L / M));
                h->capacity = (long long)h->N * (h->L / M);
            } else {
                // undecimated: only the NCO tables are needed
                h->F = 1;
                h->M = 1;
                h->window.assign(1, 1.f);
                h->kernel_name = gsdr::mix_kernel_name(h->N);
                rc = setup_ddc_common(h, 1, 1, (unsigned)p->rate, tone, 1, /*allow_flat=*/false);
                h->capacity = (long long)h->N * h->L;
            }
            break;
        }
        case GSDR_TONES:    // ref: :121-175, :702-768
        case GSDR_NOISE: {  // ref: :264-313 (full spectrum: every FFT bin is a channel)
            const bool noise = (last == GSDR_NOISE);
            if (!need(p->rate > 0, "rate must be positive")) return nullptr;
            if (!need(p->fft_tones >= 1, "fft_tones must be >= 1")) return nullptr;
            if (!need(p->pf_average >= 1 && p->pf_average <= kMaxF, "pf_average must be in [1,8] for TONES/NOISE")) return nullptr;
            if (!need(noise || (p->n_freq >= h->N && p->freq), "TONES needs one frequency per wave_type entry")) return nullptr;
            if (!need(h->decim <= 0,
                      "TONES/NOISE with decim > 0 is not supported: the reference path is broken "
                      "(ref: kernels.cu:718-719,747,779, USRP_demodulator.cpp:172,516)")) return nullptr;
            if (!need((long long)p->fft_tones * p->pf_average <= 0x7fffffffLL, "fft_tones*pf_average overflows")) return nullptr;
            // NOISE: polyphase filter + batched FFT of every frame (fft_kernels.hip), any fft_tones.
            // GSDR_NOISE_FFT=0 evaluates every bin as a DDC tone instead (round 1's path: O(fft_tones)
            // per sample, kept for A/B runs and refused above 16384 bins)
            const bool noise_fft = noise && env_int("GSDR_NOISE_FFT", 1) != 0;
            if (!need(!noise || noise_fft || p->fft_tones <= 16384,
                      "NOISE without the FFT stage (GSDR_NOISE_FFT=0) supports fft_tones <= 16384")) return nullptr;
            h->nfft = p->fft_tones;
            const int F = (int)p->pf_average;
            h->fcut = (float)(1. / (2 * h->nfft));                     // :131, :274
            h->window.resize((size_t)h->nfft * F);
            gsdr_make_sinc_window(h->nfft * F, h->fcut, h->window.data());  // :134, :277
            h->batching = gsdr_pfb_batching(h->L, h->nfft, F);         // :706
            const int n_ch = noise ? h->nfft : h->N;                   // channels of the DDC launch
            h->bins.resize(n_ch);
            if (noise) {
                for (int u = 0; u < n_ch; ++u) h->bins[u] = u;         // process_pfb_spec keeps every bin
            } else {
                gsdr_pfb_tone_bins(p->rate, h->nfft, p->freq, h->N, h->bins.data());  // :722-733
            }
            std::vector<long long> tone(n_ch);
            for (int u = 0; u < n_ch; ++u) tone[u] = h->bins[u] < 0 ? 0 : h->bins[u];
            // buffer_helper(n_tones, buffer_len, average, n_eff_tones): :159 / :301
            gsdr_buffer_helper_init(&h->bh, h->nfft, (int)h->L, F, n_ch);
            h->ddc_channels = n_ch;
            // A frame per workgroup -- polyphase filter, transform inside the LDS, bin selection: the
            // reference's own algorithm (:486-565, :568-649) at one read of the window and one write of
            // the selected bins per buffer.  Frames of up to 8192 points without a prime factor above 127;
            // GSDR_PFB_LDS=0, GSDR_TONES_FFT=0 (TONES only) or such a length leave TONES to the DDC
            // kernels (every selected bin as a tone) and NOISE to the global-memory FFT stages.
            int radices16[16];
            const bool direct_ok = gsdr::pfb_lds_plan(h->nfft, radices16) >= 0;
            // a prime factor above 127 (or GSDR_PFB_BLUESTEIN=1: any length, for tests): Bluestein's identity inside
            // the workgroup, when a frame at the padded length m = 2^ceil(log2(2 nfft - 1)) fits the LDS
            long long blue_m = 1;
            while (blue_m < 2LL * h->nfft - 1) blue_m <<= 1;
            const bool blue_ok = (!direct_ok || env_int("GSDR_PFB_BLUESTEIN", 0) != 0) && env_int("GSDR_PFB_BLUESTEIN", 1) != 0 &&
                                 blue_m <= gsdr::kPfbLdsMaxN && gsdr::pfb_cu_fits(h->nfft, F, (int)blue_m);
            const bool lds_path = env_int("GSDR_PFB_LDS", 1) != 0 && (direct_ok || blue_ok) &&
                                  (noise ? noise_fft : env_int("GSDR_TONES_FFT", 1) != 0);
            if (lds_path) {
                h->pfb_lds = true;
                h->pfb_blue = blue_ok;
                h->F = F;
                h->M = h->nfft;
                h->pfb_cu = gsdr::pfb_cu_takes(h->nfft, F, blue_ok ? (int)blue_m : h->nfft, blue_ok, (int)(h->L / h->nfft));
                h->kernel_name = h->pfb_cu ? gsdr::pfb_cu_kernel_name() : gsdr::pfb_lds_kernel_name();
                std::vector<float2> tw((size_t)h->nfft);
                for (int k = 0; k < h->nfft; ++k) {
                    const double a = -2.0 * M_PI * (double)k / (double)h->nfft;
                    tw[(size_t)k] = make_float2((float)std::cos(a), (float)std::sin(a));
                }
                std::vector<int> sel(tone.begin(), tone.end());
                bool ok = upload(&h->d_pfb_tw, tw) == hipSuccess && upload(&h->d_fft_win, h->window) == hipSuccess &&
                          (noise || upload(&h->d_pfb_sel, sel) == hipSuccess);
                // Bluestein: the chirp, its transform and the twiddles of length m (fft_plan_build makes exactly these)
                if (ok && blue_ok) ok = gsdr::fft_plan_build(h->fft, h->nfft) == 0 && h->fft.m == (int)blue_m;
                const size_t ncarry = (size_t)h->nfft * F + 8;
                for (int i = 0; i < kStageSets && ok; ++i)
                    ok = dev_alloc(&h->d_pfb_carry[i], ncarry) == hipSuccess &&
                         hipMemset(h->d_pfb_carry[i], 0, ncarry * sizeof(float2)) == hipSuccess;
                if (!need(ok, "PFB allocation failed")) return nullptr;
                h->capacity = (long long)n_ch * h->batching;               // :147 / :288
                break;
            }
            // frames the LDS kernel does not take: the polyphase filter and one launch per radix stage through
            // memory (§4.5) -- NOISE keeps every bin, TONES picks its bins out of a scratch spectrum.  For
            // TONES this replaces one DDC per bin when the frame is long (above 8192 points a frame is a
            // window of 32 768+ samples: the DDC rows become thousand-block loops on a handful of workgroups)
            const bool tones_global = !noise && env_int("GSDR_TONES_FFT", 1) != 0;
            if (noise_fft || tones_global) {
                h->noise_fft = true;
                h->F = F;
                h->M = h->nfft;
                h->kernel_name = gsdr::fft_kernel_name();
                if (tones_global) {
                    std::vector<int> sel(tone.begin(), tone.end());
                    if (!need(upload(&h->d_pfb_sel, sel) == hipSuccess &&
                                  dev_alloc(&h->d_fft_c, (size_t)h->nfft * (size_t)h->batching) == hipSuccess,
                              "TONES allocation failed")) return nullptr;
                }
                if (!need(gsdr::fft_plan_build(h->fft, h->nfft) == 0, "cannot plan an FFT of fft_tones points")) return nullptr;
                const size_t len = (size_t)(h->fft.m > h->nfft ? h->fft.m : h->nfft) * (size_t)h->batching;
                const size_t nwin = (size_t)h->nfft * h->batching * 2;
                bool ok = dev_alloc(&h->d_fft_a, len) == hipSuccess && dev_alloc(&h->d_fft_b, len) == hipSuccess &&
                          upload(&h->d_fft_win, h->window) == hipSuccess;
                for (int i = 0; i < kStageSets && ok; ++i)
                    ok = dev_alloc(&h->d_win[i], nwin) == hipSuccess &&
                         hipMemset(h->d_win[i], 0, nwin * sizeof(float2)) == hipSuccess;
                if (!need(ok, "NOISE allocation failed")) return nullptr;
                h->capacity = (long long)n_ch * h->batching;               // :288
                break;
            }
            rc = setup_ddc_common(h, F, h->nfft, (unsigned)h->nfft, tone,
                                  (int)(h->L / h->nfft) + F + 6);
            h->kernel_name = h->pipe ? gsdr::ddc_flat_kernel_name() : gsdr::ddc_kernel_name();
            if (!rc) {
                // raw_input (:143) plus an equally long half of padding behind it, kStageSets times
                const size_t n = (size_t)h->nfft * h->batching * 2;
                for (int i = 0; i < kStageSets && !rc; ++i)
                    if (dev_alloc(&h->d_win[i], n) != hipSuccess ||
                        hipMemset(h->d_win[i], 0, n * sizeof(float2)) != hipSuccess) {
                        h->err = "raw_input allocation failed";
                        rc = -1;
                    }
            }
            // every carried sample of the raw window must come from the previous buffer
            // (absmax covers this buffer and the one before)
            // and the padding behind the last window must stay inside the raw buffer's spare half
            if (!rc && env_int("GSDR_DDC_MFMA", 1) != 0 && (long long)h->nfft * (F + 1) <= h->L &&
                (long long)h->nfft * h->batching >= 40)
                rc = setup_mfma(h, /*direct=*/false, tone);
            if (!rc && !h->mfma) rc = autotune_chunks(h, (int)(h->L / h->nfft) + F - 1);
            h->capacity = (long long)n_ch * h->batching;               // :147 / :288
            break;
        }
        case GSDR_CHIRP: {  // ref: :177-262
            if (!need(p->rate > 0, "rate must be positive")) return nullptr;
            if (!need(p->n_freq >= 1 && p->n_chirp_f >= 1 && p->n_swipe_s >= 1 && p->n_chirp_t >= 1 &&
                          p->freq && p->chirp_f && p->swipe_s && p->chirp_t,
                      "CHIRP needs freq[0], chirp_f[0], swipe_s[0] and chirp_t[0]")) return nullptr;
            gsdr_chirp_param cp;
            gsdr_chirp_derive(p->rate, p->freq[0], p->chirp_f[0], p->swipe_s[0], p->chirp_t[0], &cp);
            if (!need(cp.num_steps >= 1 && cp.length >= 1 &&
                          cp.num_steps <= 0x7fffffffffffffffULL / cp.length,
                      "chirp period overflows")) return nullptr;
            h->cs.num_steps = cp.num_steps;
            h->cs.length = cp.length;
            h->cs.period = cp.num_steps * cp.length;
            h->cs.chirpness = cp.chirpness;
            h->cs.f0 = cp.f0;
            if (h->decim > 0) {
                const unsigned long long ppt = cp.length * (unsigned long long)h->decim;  // :231
                if (!need(ppt >= 1 && ppt <= (unsigned long long)h->L,
                          "chirp lock-in needs length*decim <= buffer_len")) return nullptr;
                h->ppt = (int)ppt;
                gsdr_vna_helper_init(&h->vh, h->ppt, (int)h->L);       // :235
                h->window.resize(h->ppt);
                gsdr_make_flat_window(h->ppt, h->ppt / 10, h->window.data());  // :246
                h->kernel_name = gsdr::chirp_lockin_kernel_name();
                if (upload(&h->d_profile, h->window) != hipSuccess ||
                    dev_alloc(&h->d_ccarry[0], (size_t)h->ppt) != hipSuccess ||
                    dev_alloc(&h->d_ccarry[1], (size_t)h->ppt) != hipSuccess ||
                    dev_alloc(&h->d_chirp_part, (size_t)kChirpPartials) != hipSuccess) {
                    h->err = "chirp allocation failed";
                    rc = -1;
                }
                h->capacity = h->L / h->ppt + 1;
            } else {
                h->kernel_name = gsdr::chirp_demod_kernel_name();
                h->capacity = h->L;
            }
            break;
        }
        case GSDR_NODSP:  // ref: :315-321
            h->capacity = h->L;
            h->kernel_name = "memcpy";
            break;
        default:  // ref: :322-325
            fail_create(h, "Void demodulation operation has not been implemented yet!");
            return nullptr;
    }
    if (rc) {
        fail_create(h, h->err.empty() ? "device setup failed" : h->err);
        return nullptr;
    }
    // hipMemset returns before the fill has run (it is asynchronous on the null stream), and the streams
    // of this library do not wait for the null stream: without this the zeroing of a carry buffer could
    // land after the first call had written the carry (seen with several handles created and used from
    // several threads, scratch/concurrent_handles.py: the second buffer of a TONES handle came out wrong)
    if (hipStreamSynchronize(nullptr) != hipSuccess) {
        fail_create(h, "device setup did not complete");
        return nullptr;
    }
    return h;
}

int gsdr_demod_process_device(gsdr_demod *h, const gsdr_c64 *in_dev, gsdr_c64 *out_dev,
                              void *hip_stream) {
    if (!h) return -1;
    if (!in_dev || !out_dev) {
        h->err = "null buffer";
        return -1;
    }
    if (h->device >= 0) HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)hip_stream;  // NULL is HIP's null stream, as everywhere in HIP
    // An in-order call is a join point: it runs behind everything this handle has in flight on
    // other streams (an earlier call on another stream, overlapped calls still running).  Free on
    // the usual path (same stream as the call before).  Overlapped calls order themselves among
    // their compute streams (pipeline_compute) and only join the in-order streams.
    if (!h->pipe_overlap) {
        if (join_streams(h, st, [](hipStream_t) { return false; })) return -1;
    }
    const float2 *in = reinterpret_cast<const float2 *>(in_dev);
    float2 *out = reinterpret_cast<float2 *>(out_dev);
    int n;
    switch (h->mode) {
        case GSDR_DIRECT: n = enqueue_direct(h, in, out, st); break;
        case GSDR_TONES:
        case GSDR_NOISE: n = enqueue_pfb(h, in, out, st); break;
        case GSDR_CHIRP: n = enqueue_chirp(h, in, out, st); break;
        case GSDR_NODSP:  // ref: process_nodsp :335-339
            HIPCHK(h, hipMemcpyAsync(out, in, (size_t)h->L * sizeof(float2),
                                     hipMemcpyDeviceToDevice, st));
            return (int)h->L;     // no state passes from call to call
        default: h->err = "unsupported mode"; return -1;
    }
    return n;
}

int gsdr_demod_process(gsdr_demod *h, const gsdr_c64 *in_host, gsdr_c64 *out_host) {
    if (!h) return -1;
    if (!in_host || !out_host) {
        h->err = "null buffer";
        return -1;
    }
    if (h->mode == GSDR_NODSP) {  // ref: :335-339, a host memcpy
        std::memcpy(out_host, in_host, (size_t)h->L * sizeof(gsdr_c64));
        return (int)h->L;
    }
    if (h->device >= 0) HIPCHK(h, hipSetDevice(h->device));
    if (!h->d_in) {
        HIPCHK(h, dev_alloc(&h->d_in, (size_t)h->L));
        HIPCHK(h, dev_alloc(&h->d_out, (size_t)h->capacity));
    }
    HIPCHK(h, hipMemcpyAsync(h->d_in, in_host, (size_t)h->L * sizeof(float2),
                             hipMemcpyHostToDevice, h->stream));
    const int ret = gsdr_demod_process_device(h, reinterpret_cast<gsdr_c64 *>(h->d_in),
                                              reinterpret_cast<gsdr_c64 *>(h->d_out), h->stream);
    if (ret < 0) return ret;
    if (ret > 0)
        HIPCHK(h, hipMemcpyAsync(out_host, h->d_out, (size_t)ret * sizeof(float2),
                                 hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));  // ref: :393,:462,:555
    return ret;
}

static void pipeline_teardown(gsdr_demod *h) {
    for (auto &sl : h->slot) {
        if (sl.up) (void)hipEventDestroy(sl.up);
        if (sl.done) (void)hipEventDestroy(sl.done);
        if (sl.down) (void)hipEventDestroy(sl.down);
        sl.up = sl.done = sl.down = sl.wait_ev = nullptr;
    }
    for (int i = 0; i < kPipeStreams; ++i) {
        if (h->s_main[i]) (void)hipStreamDestroy(h->s_main[i]);
        h->s_main[i] = nullptr;
    }
    for (int i = 0; i < 4; ++i) {
        if (h->ev_abs[i]) (void)hipEventDestroy(h->ev_abs[i]);
        h->ev_abs[i] = nullptr;
    }
    if (h->s_up) (void)hipStreamDestroy(h->s_up);
    if (h->s_down) (void)hipStreamDestroy(h->s_down);
    h->s_up = h->s_down = nullptr;
}

static int pipeline_init_parts(gsdr_demod *h);

// streams and events of the pipelined entries, created on first use; all or nothing: a partial
// failure leaves no half-built pipeline behind for the next call to trip over
static int pipeline_init(gsdr_demod *h) {
    if (h->pipe_ready) return 0;
    if (pipeline_init_parts(h)) {
        const std::string keep = h->err;
        pipeline_teardown(h);
        h->err = keep;
        return -1;
    }
    h->pipe_ready = true;
    return 0;
}

static int pipeline_init_parts(gsdr_demod *h) {
    HIPCHK(h, hipStreamCreateWithFlags(&h->s_up, hipStreamNonBlocking));
    HIPCHK(h, hipStreamCreateWithFlags(&h->s_down, hipStreamNonBlocking));
    for (auto &sl : h->slot) {
        HIPCHK(h, hipEventCreateWithFlags(&sl.up, hipEventDisableTiming));
        HIPCHK(h, hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
        HIPCHK(h, hipEventCreateWithFlags(&sl.down, hipEventDisableTiming));
    }
    // The compute streams must sit on different hardware queues to overlap.  HIP deals the
    // streams of one priority class out to few queues (4 by default) that every other stream
    // of that class in the process shares too (torch alone creates dozens), and which streams
    // end up together is luck (measured: 53 vs 33 us per C2 buffer from run to run, and again
    // with the second handle of a process).  A stream created with a compute-unit mask gets a
    // queue of its own: ask for one with every unit enabled.  GSDR_PIPE_QUEUES=0 (or a runtime
    // that refuses) falls back to streams of the low-priority class, which is what the
    // reference gives its demodulator stream (cpp/USRP_demodulator.cpp:43-44).
    int prio_least = 0, prio_greatest = 0;
    HIPCHK(h, hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    hipDeviceProp_t prop;
    int dev = 0;
    HIPCHK(h, hipGetDevice(&dev));
    HIPCHK(h, hipGetDeviceProperties(&prop, dev));
    std::vector<uint32_t> all_units((size_t)(prop.multiProcessorCount + 31) / 32, 0xffffffffu);
    if (prop.multiProcessorCount % 32) all_units.back() = (1u << (prop.multiProcessorCount % 32)) - 1u;
    const bool own_queues = env_int("GSDR_PIPE_QUEUES", 1) != 0;
    for (int i = 0; i < kPipeStreams; ++i) {
        if (own_queues &&
            hipExtStreamCreateWithCUMask(&h->s_main[i], (uint32_t)all_units.size(), all_units.data()) == hipSuccess)
            continue;
        (void)hipGetLastError();
        HIPCHK(h, hipStreamCreateWithPriority(&h->s_main[i], hipStreamNonBlocking, prio_least));
    }
    h->pipe_streams = env_int("GSDR_PIPE_STREAMS", kPipeStreams);
    h->pipe_overlap_allowed = env_int("GSDR_PIPE_OVERLAP", 1) != 0;
    if (h->pipe_streams < 1 || h->pipe_streams > kPipeStreams) h->pipe_streams = kPipeStreams;
    for (int i = 0; i < 4; ++i) HIPCHK(h, hipEventCreateWithFlags(&h->ev_abs[i], hipEventDisableTiming));
    return 0;
}

// The kernels of one pipelined buffer; records sl.done behind them.  `up`: event the input
// becomes ready with (nullptr: it is ready).
//
// DIRECT on the staged matrix-core kernel: call j runs on compute stream j % S (S =
// GSDR_PIPE_STREAMS, kPipeStreams = 3 by default), so the kernels of consecutive buffers
// overlap and the next buffer fills the compute units that the last workgroups of this one
// leave idle.  What call j needs from its neighbours:
//   - its staging pass follows the pass of call j-1 (which wrote the carry in front of this
//     call's head copy and cleared this call's scale slot): one event, long complete when
//     it is waited for;
//   - everything it overwrites was last read by main kernels that are finished: call j-S ran
//     on the same stream, calls j-S-1 .. sit in front of pass j-1 on their streams.  The main
//     kernels of calls j-1 .. j-S+1 may still run: they read head/tail sets j-1 .. j-S+1 and
//     slots j-1 .. j-S, the pass of call j writes head/tail set j, the carry part of head j+1,
//     slot j and clears slot j+1 -- disjoint modulo kStageSets = S+1 and kScaleSlots = S+2.
// Every other mode keeps the one compute stream.  GSDR_PIPE_OVERLAP=0 does so for DIRECT too.
static int pipeline_compute(gsdr_demod *h, gsdr_demod::Slot &sl, hipEvent_t up, const float2 *in, float2 *out) {
    const bool ddc = (h->mode == GSDR_DIRECT && h->decim > 0) ||
                     h->mode == GSDR_TONES || h->mode == GSDR_NOISE;
    // TONES / NOISE inside the LDS keep the one compute stream: a launch is 10 - 15 us, the events that tie the calls
    // of rotating streams together (carry, completion) cost more than their overlap gives -- per 1 M-sample buffer
    // 16.3 against 12.0 us in order at 1024 points, 17.0 against 10.2 at 256, 25.7 against 15.4 at 1230
    // (profiles/r03_pfb_api_ab.log)
    const bool overlap = h->pipe_overlap_allowed && h->mfma && !h->pfb_lds && ddc;
    hipStream_t cs = overlap ? h->s_main[h->pipe_seq % (unsigned)h->pipe_streams] : h->stream;
    if (up) HIPCHK(h, hipStreamWaitEvent(cs, up, 0));
    if (overlap) {
        // behind in-order calls made on other streams since (their carry, slot and window writes);
        // the compute streams of the pipeline are ordered by the protocol above
        if (join_streams(h, cs, [h](hipStream_t s) {
                for (int i = 0; i < kPipeStreams; ++i)
                    if (s == h->s_main[i]) return true;
                return false;
            }))
            return -1;
    }
    if (overlap && h->pipe_seq > 0) HIPCHK(h, hipStreamWaitEvent(cs, h->ev_abs[(h->pipe_seq - 1) % 4], 0));
    h->pipe_overlap = overlap;
    const int n = gsdr_demod_process_device(h, reinterpret_cast<const gsdr_c64 *>(in),
                                            reinterpret_cast<gsdr_c64 *>(out), cs);
    h->pipe_overlap = false;
    if (overlap) h->pipe_seq++;
    if (n < 0) return -1;
    HIPCHK(h, hipEventRecord(sl.done, cs));
    return n;
}

int gsdr_demod_prepare(gsdr_demod *h, int what) {
    if (!h) return -1;
    if (h->device >= 0) HIPCHK(h, hipSetDevice(h->device));
    if (h->mode == GSDR_NODSP) return 0;
    if ((what & GSDR_PREPARE_HOST) && !h->d_in) {
        HIPCHK(h, dev_alloc(&h->d_in, (size_t)h->L));
        HIPCHK(h, dev_alloc(&h->d_out, (size_t)h->capacity));
    }
    if (what & (GSDR_PREPARE_PIPELINE | GSDR_PREPARE_PIPELINE_HOST)) {
        if (pipeline_init(h)) return -1;
    }
    if (what & GSDR_PREPARE_PIPELINE_HOST) {
        for (auto &sl : h->slot)
            if (!sl.d_in) {
                HIPCHK(h, dev_alloc(&sl.d_in, (size_t)h->L));
                HIPCHK(h, dev_alloc(&sl.d_out, (size_t)h->capacity));
            }
    }
    // first use of a stream (its hardware queue), of the copy engines in both directions and of
    // the code object costs milliseconds: pay them here, not on the first packets
    HIPCHK(h, gsdr::launch_warm(h->stream));
    if (h->pipe_ready) {
        for (int i = 0; i < kPipeStreams; ++i) HIPCHK(h, gsdr::launch_warm(h->s_main[i]));
        float2 *pin = nullptr;
        if (h->slot[0].d_in && hipHostMalloc((void **)&pin, 4096) == hipSuccess) {
            std::memset(pin, 0, 4096);
            (void)hipMemcpyAsync(h->slot[0].d_in, pin, 4096, hipMemcpyHostToDevice, h->s_up);
            (void)hipStreamSynchronize(h->s_up);
            (void)hipMemcpyAsync(pin, h->slot[0].d_in, 4096, hipMemcpyDeviceToHost, h->s_down);
            (void)hipStreamSynchronize(h->s_down);
            (void)hipHostFree(pin);
        }
    }
    HIPCHK(h, hipDeviceSynchronize());
    if (what & GSDR_PREPARE_REHEARSE) {
        // a twin with the same parameters takes the process-wide first-use costs (see include/gsdr.h)
        gsdr_demod *twin = gsdr_demod_create(&h->pc);
        gsdr_c64 *pin_in = nullptr, *pin_out = nullptr;
        bool ok = twin != nullptr;
        ok = ok && hipHostMalloc((void **)&pin_in, (size_t)h->L * sizeof(gsdr_c64)) == hipSuccess;
        ok = ok && hipHostMalloc((void **)&pin_out, (size_t)(h->capacity > 0 ? h->capacity : 1) * sizeof(gsdr_c64)) == hipSuccess;
        if (ok) {
            std::memset(pin_in, 0, (size_t)h->L * sizeof(gsdr_c64));
            ok = gsdr_demod_prepare(twin, what & ~GSDR_PREPARE_REHEARSE) == 0;
            if (ok && (what & GSDR_PREPARE_HOST))
                for (int k = 0; k < 2 && ok; ++k) ok = gsdr_demod_process(twin, pin_in, pin_out) >= 0;
            if (ok && (what & GSDR_PREPARE_PIPELINE_HOST)) {
                int pending = 0;
                for (int k = 0; k < 2 * GSDR_PIPELINE_DEPTH + 2 && ok; ++k) {
                    if (pending == GSDR_PIPELINE_DEPTH) {
                        ok = gsdr_demod_wait(twin) >= 0;
                        --pending;
                    }
                    ok = ok && gsdr_demod_submit(twin, pin_in, pin_out) == 0;
                    ++pending;
                }
                while (ok && pending-- > 0) ok = gsdr_demod_wait(twin) >= 0;
            }
        }
        if (!ok) h->err = std::string("rehearsal failed: ") + (twin ? twin->err : g_create_error);
        if (pin_in) (void)hipHostFree(pin_in);
        if (pin_out) (void)hipHostFree(pin_out);
        if (twin) gsdr_demod_close(twin);
        if (!ok) return -1;
    }
    return 0;
}

int gsdr_demod_submit(gsdr_demod *h, const gsdr_c64 *in_host, gsdr_c64 *out_host) {
    if (!h) return -1;
    if (!in_host || !out_host) {
        h->err = "null buffer";
        return -1;
    }
    if (h->pipe_count >= GSDR_PIPELINE_DEPTH) {
        h->err = "pipeline full: call gsdr_demod_wait() first";
        return -1;
    }
    if (h->device >= 0) HIPCHK(h, hipSetDevice(h->device));
    if (pipeline_init(h)) return -1;
    auto &sl = h->slot[(h->pipe_head + h->pipe_count) % GSDR_PIPELINE_DEPTH];
    if (!sl.d_in) {
        HIPCHK(h, dev_alloc(&sl.d_in, (size_t)h->L));
        HIPCHK(h, dev_alloc(&sl.d_out, (size_t)h->capacity));
    }
    // the slot is free: its previous download was waited for in gsdr_demod_wait()
    HIPCHK(h, hipMemcpyAsync(sl.d_in, in_host, (size_t)h->L * sizeof(float2), hipMemcpyHostToDevice, h->s_up));
    HIPCHK(h, hipEventRecord(sl.up, h->s_up));
    const int n = pipeline_compute(h, sl, sl.up, sl.d_in, sl.d_out);
    if (n < 0) return -1;
    HIPCHK(h, hipStreamWaitEvent(h->s_down, sl.done, 0));
    if (n > 0)
        HIPCHK(h, hipMemcpyAsync(out_host, sl.d_out, (size_t)n * sizeof(float2), hipMemcpyDeviceToHost, h->s_down));
    HIPCHK(h, hipEventRecord(sl.down, h->s_down));
    sl.wait_ev = sl.down;
    sl.n = n;
    h->pipe_count++;
    return 0;
}

int gsdr_demod_submit_device(gsdr_demod *h, const gsdr_c64 *in_dev, gsdr_c64 *out_dev) {
    if (!h) return -1;
    if (!in_dev || !out_dev) {
        h->err = "null buffer";
        return -1;
    }
    if (h->pipe_count >= GSDR_PIPELINE_DEPTH) {
        h->err = "pipeline full: call gsdr_demod_wait() first";
        return -1;
    }
    if (h->device >= 0) HIPCHK(h, hipSetDevice(h->device));
    if (pipeline_init(h)) return -1;
    auto &sl = h->slot[(h->pipe_head + h->pipe_count) % GSDR_PIPELINE_DEPTH];
    const int n = pipeline_compute(h, sl, nullptr, reinterpret_cast<const float2 *>(in_dev),
                                   reinterpret_cast<float2 *>(out_dev));
    if (n < 0) return -1;
    sl.wait_ev = sl.done;
    sl.n = n;
    h->pipe_count++;
    return 0;
}

int gsdr_demod_wait(gsdr_demod *h) {
    if (!h) return -1;
    if (h->pipe_count == 0) return -2;
    if (h->device >= 0) HIPCHK(h, hipSetDevice(h->device));
    auto &sl = h->slot[h->pipe_head];
    const hipError_t e = hipEventSynchronize(sl.wait_ev);
    // the slot leaves the queue whatever happened: a failed wait must not leave the pipeline "full"
    h->pipe_head = (h->pipe_head + 1) % GSDR_PIPELINE_DEPTH;
    h->pipe_count--;
    if (e != hipSuccess) {
        h->err = std::string("hipEventSynchronize: ") + hipGetErrorString(e);
        return -1;
    }
    return sl.n;
}

void gsdr_demod_close(gsdr_demod *h) {
    if (!h) return;
    if (h->device >= 0) (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (int i = 0; i < kPipeStreams; ++i)
        if (h->s_main[i]) (void)hipStreamSynchronize(h->s_main[i]);
    if (h->s_up) (void)hipStreamSynchronize(h->s_up);
    if (h->s_down) (void)hipStreamSynchronize(h->s_down);
    for (auto &sl : h->slot) {
        if (sl.d_in) (void)hipFree(sl.d_in);
        if (sl.d_out) (void)hipFree(sl.d_out);
    }
    pipeline_teardown(h);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    for (auto &e : h->ev_pool) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    if (h->d_chirp_part) (void)hipFree(h->d_chirp_part);
    void *ptrs[] = {h->d_in,      h->d_out,     h->d_taps_t,   h->d_taps_p,   h->d_stage,    h->d_btab,     h->d_wk,
                    h->d_wrem,    h->d_fmod,    h->d_tails,    h->d_carry[0], h->d_carry[1],
                    h->d_profile, h->d_ccarry[0], h->d_ccarry[1],
                    h->d_bfrag,   h->d_ptab,    h->d_dtab,     h->d_mtaps,    h->d_mfmod,
                    h->d_segmax};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    for (int i = 0; i < kStageSets; ++i) {
        if (h->d_img[i]) (void)hipFree(h->d_img[i]);
        if (h->d_win[i]) (void)hipFree(h->d_win[i]);
        if (h->d_head[i]) (void)hipFree(h->d_head[i]);
        if (h->d_tail[i]) (void)hipFree(h->d_tail[i]);
    }
    gsdr::fft_plan_free(h->fft);
    if (h->d_fft_a) (void)hipFree(h->d_fft_a);
    if (h->d_fft_b) (void)hipFree(h->d_fft_b);
    if (h->d_fft_c) (void)hipFree(h->d_fft_c);
    if (h->d_fft_win) (void)hipFree(h->d_fft_win);
    if (h->d_pfb_tw) (void)hipFree(h->d_pfb_tw);
    if (h->d_pfb_sel) (void)hipFree(h->d_pfb_sel);
    for (int i = 0; i < kStageSets; ++i)
        if (h->d_pfb_carry[i]) (void)hipFree(h->d_pfb_carry[i]);
    if (h->stream) (void)hipStreamDestroy(h->stream);  // ref: 03_implement.md:58-63
    delete h;
}

int gsdr_pfb_lds_stages(int fft_tones, int *radices) {
    int tmp[16];
    const int n = gsdr::pfb_lds_plan(fft_tones, tmp);
    if (radices)
        for (int i = 0; i < n && i < 16; ++i) radices[i] = tmp[i];
    return n;
}

int gsdr_demod_mode(const gsdr_demod *h) { return h ? h->mode : -1; }
int gsdr_demod_channels(const gsdr_demod *h) { return h ? h->N : 0; }
long long gsdr_demod_out_capacity(const gsdr_demod *h) { return h ? h->capacity : 0; }
float gsdr_demod_fcut(const gsdr_demod *h) { return h ? h->fcut : 0.f; }

int gsdr_demod_get_window(const gsdr_demod *h, float *w, int cap) {
    if (!h) return 0;
    const int n = (int)h->window.size();
    if (w)
        for (int i = 0; i < n && i < cap; ++i) w[i] = h->window[i];
    return n;
}

int gsdr_demod_get_bins(const gsdr_demod *h, int *bins, int cap) {
    if (!h) return 0;
    const int n = (int)h->bins.size();
    if (bins)
        for (int i = 0; i < n && i < cap; ++i) bins[i] = h->bins[i];
    return n;
}

void gsdr_demod_profile_enable(gsdr_demod *h, int enable) {
    if (!h) return;
    h->prof = enable != 0;
    h->prof_every = enable > 1 ? enable : 1;
    h->prof_seen = 0;
    h->ev_used = 0;
}

int gsdr_demod_profile_read(gsdr_demod *h, double *total_ms) {
    if (total_ms) *total_ms = 0.0;
    if (!h || h->ev_used == 0) return 0;
    if (h->device >= 0) (void)hipSetDevice(h->device);
    double sum = 0.0;
    int n = 0;
    for (size_t i = 0; i < h->ev_used; ++i) {
        if (hipEventSynchronize(h->ev_pool[i].second) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, h->ev_pool[i].first, h->ev_pool[i].second) == hipSuccess) {
            sum += ms;
            n++;
        }
    }
    if (total_ms) *total_ms = sum;
    return n;
}

const char *gsdr_demod_kernel_name(const gsdr_demod *h) { return h ? h->kernel_name : "none"; }

void gsdr_reload_env(void) { gsdr::fft_env_reload(); }

const char *gsdr_build_info(void) {
#ifdef GSDR_TIMING_BUILD
    return "abi 1; arch gfx950; timing_build 1";
#else
    return "abi 1; arch gfx950; timing_build 0";
#endif
}

int gsdr_demod_describe(const gsdr_demod *h, char *buf, int cap) {
    if (!h || !buf || cap < 1) return 0;
    static const char *modes[] = {"TONES", "CHIRP", "NOISE", "RAMP", "NODSP", "SWONLY", "DIRECT"};
    std::string s = "{\"mode\": \"";
    s += (h->mode >= 0 && h->mode <= 6) ? modes[h->mode] : "?";
    s += "\", \"kernel\": \"";
    s += h->kernel_name;
    s += "\", \"family\": \"";
    s += h->pfb_lds ? (h->pfb_blue ? "polyphase filter + Bluestein (two fp32 Stockham FFTs) inside the LDS + bin selection, one launch"
                                   : "polyphase filter + fp32 Stockham FFT inside the LDS + bin selection, one launch") :
         h->noise_fft ? "fp32 Stockham FFT behind the polyphase filter" : h->mfma ? "f16 MFMA, hi/lo split" : (h->mode == GSDR_CHIRP ? "fp32 VALU, integer phase" : (h->pipe ? "packed fp32 VALU" : "fp32 VALU"));
    s += "\", \"channels\": " + std::to_string(h->ddc_channels > 0 ? h->ddc_channels : h->N);
    s += ", \"row_tiles_per_workgroup\": " + std::to_string(h->mfma ? h->last_rt : 0);
    s += ", \"pipeline_streams\": " + std::to_string(h->pipe_ready ? h->pipe_streams : env_int("GSDR_PIPE_STREAMS", kPipeStreams));
    s += ", \"timing_build\": ";
#ifdef GSDR_TIMING_BUILD
    s += "1";
#else
    s += "0";
#endif
    s += ", \"env\": {";
    bool first = true;
    for (char **e = environ; e && *e; ++e) {
        if (std::strncmp(*e, "GSDR_", 5) != 0) continue;
        const char *eq = std::strchr(*e, '=');
        if (!eq) continue;
        std::string k(*e, eq - *e), v(eq + 1);
        for (auto &c : v)
            if (c == '"' || c == '\\' || (unsigned char)c < 0x20) c = '?';
        s += (first ? "\"" : ", \"") + k + "\": \"" + v + "\"";
        first = false;
    }
    s += "}}";
    const int n = (int)s.size() < cap - 1 ? (int)s.size() : cap - 1;
    std::memcpy(buf, s.data(), (size_t)n);
    buf[n] = 0;
    return n;
}

// ---- synthetic sources -----------------------------------------------------
int gsdr_source_tones(gsdr_c64 *out_dev, long long n, long long start, int rate, const int *freq,
                      const float *ampl, const float *phase, int n_tones, float sigma,
                      unsigned long long seed, void *hip_stream) {
    if (!out_dev || n < 0 || rate <= 0 || n_tones < 0) return -1;
    std::vector<unsigned> fm(n_tones > 0 ? n_tones : 1, 0u);
    for (int k = 0; k < n_tones; ++k) {
        long long r = (long long)freq[k] % rate;
        if (r < 0) r += rate;
        fm[k] = (unsigned)r;
    }
    unsigned *d_f = nullptr;
    float *d_a = nullptr, *d_p = nullptr;
    const size_t cnt = fm.size();
    hipError_t e = hipMalloc((void **)&d_f, cnt * sizeof(unsigned));
    if (e == hipSuccess) e = hipMalloc((void **)&d_a, cnt * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&d_p, cnt * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(d_f, fm.data(), cnt * sizeof(unsigned), hipMemcpyHostToDevice);
    if (e == hipSuccess && n_tones > 0) e = hipMemcpy(d_a, ampl, n_tones * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess && n_tones > 0) e = hipMemcpy(d_p, phase, n_tones * sizeof(float), hipMemcpyHostToDevice);
    hipStream_t st = (hipStream_t)hip_stream;
    long long start_mod = start % rate;
    if (start_mod < 0) start_mod += rate;
    if (e == hipSuccess)
        e = gsdr::launch_source_tones(reinterpret_cast<float2 *>(out_dev), n, start_mod, (unsigned)rate,
                                      d_f, d_a, d_p, n_tones, sigma, seed, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (d_f) (void)hipFree(d_f);
    if (d_a) (void)hipFree(d_a);
    if (d_p) (void)hipFree(d_p);
    if (e != hipSuccess) {
        g_create_error = std::string("gsdr_source_tones: ") + hipGetErrorString(e);
        return -1;
    }
    return 0;
}

// ---- TX tone comb at scale (row f3) ------------------------------------------
struct gsdr_txgen {
    int device = -1;
    unsigned rate = 1;
    int n_tones = 0;
    unsigned *d_fmod = nullptr;
    float2 *d_q0 = nullptr, *d_btab = nullptr, *d_ctab = nullptr;
    // the TX_buffer_generator state (gsdr_txgen_create)
    int mode = -1;                     // GSDR_TONES / GSDR_CHIRP, -1: a bare tone comb (gsdr_txgen_tones_create)
    long long buffer_len = 0;
    unsigned long long period = 1, last = 0;
    gsdr_chirp_param cp{};
    float scale = 1.f;
    float2 *d_stage = nullptr;         // get() to host memory goes through here
    size_t stage_n = 0;                // samples d_stage holds
    // TONES through gsdr_txgen_get_ptr: one period + one buffer of the comb in host memory, made once
    // (the reference's base_buffer, cpp/USRP_buffer_generator.cpp:77-95)
    float2 *h_period = nullptr;
    bool h_period_pinned = false;
};

gsdr_txgen *gsdr_txgen_tones_create(int rate, const int *freq, const float *ampl, const float *phase, int n_tones,
                                    int device_index) {
    g_create_error.clear();
    if (rate <= 0 || n_tones < 0 || (n_tones > 0 && (!freq || !ampl))) {
        g_create_error = "gsdr_txgen_tones_create: bad arguments";
        return nullptr;
    }
    if (device_index >= 0 && hipSetDevice(device_index) != hipSuccess) {
        g_create_error = "gsdr_txgen_tones_create: hipSetDevice failed (no such GPU?)";
        return nullptr;
    }
    gsdr_txgen *g = new gsdr_txgen();
    g->device = device_index;
    g->rate = (unsigned)rate;
    g->n_tones = n_tones;
    const size_t N = (size_t)(n_tones > 0 ? n_tones : 1);
    std::vector<unsigned> fm(N, 0u);
    std::vector<float2> q0(N, make_float2(0.f, 0.f)), bt(N * 64), ct(N * 16);
    for (int k = 0; k < n_tones; ++k) {
        long long r = (long long)freq[k] % rate;
        if (r < 0) r += rate;
        fm[(size_t)k] = (unsigned)r;
        const double ph0 = phase ? (double)phase[k] : 0.0;
        q0[(size_t)k] = make_float2((float)((double)ampl[k] * std::cos(ph0)), (float)((double)ampl[k] * std::sin(ph0)));
        // w^m for the exact integer phase (f m) mod rate, TX sign: e^(+2 pi i ...)
        auto w = [&](unsigned long long m) {
            double re, im;
            phasor(((unsigned long long)r * m) % (unsigned long long)rate, (unsigned)rate, re, im);   // e^(-...)
            return make_float2((float)re, (float)-im);
        };
        for (int lo = 0; lo < 64; ++lo) bt[(size_t)k * 64 + lo] = w((unsigned long long)lo);
        for (int j = 0; j < 16; ++j) ct[(size_t)k * 16 + j] = w(64ULL * (unsigned long long)j);
    }
    const bool ok = upload(&g->d_fmod, fm) == hipSuccess && upload(&g->d_q0, q0) == hipSuccess &&
                    upload(&g->d_btab, bt) == hipSuccess && upload(&g->d_ctab, ct) == hipSuccess &&
                    hipStreamSynchronize(nullptr) == hipSuccess;
    if (!ok) {
        g_create_error = "gsdr_txgen_tones_create: device allocation failed";
        gsdr_txgen_close(g);
        return nullptr;
    }
    return g;
}

int gsdr_txgen_tones_fill(gsdr_txgen *g, gsdr_c64 *out_dev, long long n, long long start, void *hip_stream) {
    if (!g || !out_dev || n < 0) {
        g_create_error = "gsdr_txgen_tones_fill: bad arguments";
        return -1;
    }
    if (g->device >= 0 && hipSetDevice(g->device) != hipSuccess) {
        g_create_error = "gsdr_txgen_tones_fill: hipSetDevice failed";
        return -1;
    }
    long long sm = start % (long long)g->rate;
    if (sm < 0) sm += g->rate;
    const hipError_t e = gsdr::launch_tones_synth(reinterpret_cast<float2 *>(out_dev), n, (unsigned long long)sm, g->rate,
                                                  g->d_fmod, g->d_q0, g->d_btab, g->d_ctab, g->n_tones, (hipStream_t)hip_stream);
    if (e != hipSuccess) {
        g_create_error = std::string("gsdr_txgen_tones_fill: ") + hipGetErrorString(e);
        return -1;
    }
    return 0;
}

void gsdr_txgen_close(gsdr_txgen *g) {
    if (!g) return;
    if (g->device >= 0) (void)hipSetDevice(g->device);
    (void)hipDeviceSynchronize();
    for (void *p : {(void *)g->d_fmod, (void *)g->d_q0, (void *)g->d_btab, (void *)g->d_ctab, (void *)g->d_stage})
        if (p) (void)hipFree(p);
    if (g->h_period) {
        if (g->h_period_pinned) (void)hipHostFree(g->h_period);
        else std::free(g->h_period);
    }
    delete g;
}

// ref: TX_buffer_generator::TX_buffer_generator, cpp/USRP_buffer_generator.cpp:10-160
gsdr_txgen *gsdr_txgen_create(const gsdr_param_c *p, const float *ampl, int n_ampl) {
    g_create_error.clear();
    auto fail = [](const char *msg) {
        g_create_error = msg;
        return (gsdr_txgen *)nullptr;
    };
    if (!p) return fail("null parameters");
    if (p->buffer_len < 1) return fail("buffer_len must be positive");
    if (p->rate < 1) return fail("rate must be positive");
    if (p->n_wave_type < 1 || !p->wave_type) return fail("TX buffer generation needs at least one wave_type");
    const int last = p->wave_type[0];
    int chirps = 0;
    bool mixed = false;
    for (int i = 0; i < p->n_wave_type; ++i) {
        chirps += p->wave_type[i] == GSDR_CHIRP;
        mixed |= p->wave_type[i] != last;
    }
    if (chirps > 1)      // :26-29
        return fail("Multiple chirp TX buffer generation has been requested. This feature is not implemented yet.");
    if (mixed)           // :31-34
        return fail("Mixed TX buffer generation has been requested. This feature is not implemented yet.");
    if (last == GSDR_NODSP || last == GSDR_SWONLY) return fail("NODSP CASE NOT IMPLEMENTED.");   // :41-44
    if (last == GSDR_RAMP || last == GSDR_DIRECT) return fail("RAMP CASE NOT IMPLEMENTED.");      // :46-49
    gsdr_txgen *g = nullptr;
    // NOISE: the reference's `case NOISE:` (:52-58) has no break and falls through into TONES, which overwrites its
    // get/close pointers: a TX NOISE request generates the tone comb of freq[] / ampl[] there, and so it does here
    if (last == GSDR_TONES || last == GSDR_NOISE) {
        const int n = p->n_wave_type;
        if (p->n_freq < n || !p->freq || n_ampl < n || !ampl) return fail("TONES needs freq[] and ampl[] for every wave_type entry");
        std::vector<int> tf((size_t)n);
        std::vector<float> ta((size_t)n);
        const int nt = gsdr_tx_tone_bins(p->rate, p->freq, ampl, n, tf.data(), ta.data());
        g = gsdr_txgen_tones_create(p->rate, tf.data(), ta.data(), nullptr, nt > 0 ? nt : 0, p->device_index);
        if (!g) return nullptr;
        // TONES_buffer_len: rate, or the multiple of it that holds one buffer (:60-75)
        g->period = (unsigned long long)p->rate * (unsigned long long)((p->buffer_len + p->rate - 1) / p->rate);
    } else if (last == GSDR_CHIRP) {
        if (p->n_freq < 1 || p->n_chirp_f < 1 || p->n_swipe_s < 1 || p->n_chirp_t < 1 || !p->freq || !p->chirp_f ||
            !p->swipe_s || !p->chirp_t)
            return fail("CHIRP needs freq[0], chirp_f[0], swipe_s[0] and chirp_t[0]");
        if (p->device_index >= 0 && hipSetDevice(p->device_index) != hipSuccess)
            return fail("hipSetDevice failed (no such GPU?)");
        g = new gsdr_txgen();
        g->device = p->device_index;
        g->rate = (unsigned)p->rate;
        // the TX side's own derivation: a step shorter than one sample also resets num_steps, and the slope
        // follows the reset value (:107-129)
        gsdr_chirp_derive_tx(p->rate, p->freq[0], p->chirp_f[0], p->swipe_s[0], p->chirp_t[0], &g->cp);
        if (g->cp.num_steps < 1 || g->cp.length < 1 || g->cp.num_steps > 0x7fffffffffffffffULL / g->cp.length) {
            delete g;
            return fail("chirp period overflows");
        }
        g->period = g->cp.num_steps * g->cp.length;
        g->scale = n_ampl > 0 && ampl ? ampl[0] : 1.f;
    } else {
        return fail("Void TX generation operation has not been implemented yet!");
    }
    g->mode = last == GSDR_NOISE ? GSDR_TONES : last;
    g->buffer_len = p->buffer_len;
    g->last = 0;
    return g;
}

long long gsdr_txgen_buffer_len(const gsdr_txgen *g) { return g ? g->buffer_len : 0; }

// ref: get_from_tones :226-229, get_from_chirp :208-221
int gsdr_txgen_get_device(gsdr_txgen *g, gsdr_c64 *out_dev, void *hip_stream) {
    if (!g || !out_dev || g->mode < 0) {
        g_create_error = "gsdr_txgen_get: bad arguments";
        return -1;
    }
    int rc;
    if (g->mode == GSDR_TONES) {
        rc = gsdr_txgen_tones_fill(g, out_dev, g->buffer_len, (long long)(g->last % g->rate), hip_stream);
    } else {
        if (g->device >= 0 && hipSetDevice(g->device) != hipSuccess) {
            g_create_error = "gsdr_txgen_get: hipSetDevice failed";
            return -1;
        }
        rc = gsdr_source_chirp(out_dev, g->buffer_len, g->last, &g->cp, g->scale, hip_stream);
    }
    if (rc == 0) g->last = (g->last + (unsigned long long)g->buffer_len) % g->period;
    return rc;
}

int gsdr_txgen_get(gsdr_txgen *g, gsdr_c64 *out_host) {
    if (!g || !out_host || g->mode < 0) {
        g_create_error = "gsdr_txgen_get: bad arguments";
        return -1;
    }
    if (g->device >= 0 && hipSetDevice(g->device) != hipSuccess) {
        g_create_error = "gsdr_txgen_get: hipSetDevice failed";
        return -1;
    }
    if (g->d_stage && g->stage_n < (size_t)g->buffer_len) {
        (void)hipFree(g->d_stage);
        g->d_stage = nullptr;
    }
    if (!g->d_stage) {
        if (dev_alloc(&g->d_stage, (size_t)g->buffer_len) != hipSuccess) {
            g_create_error = "gsdr_txgen_get: device allocation failed";
            return -1;
        }
        g->stage_n = (size_t)g->buffer_len;
    }
    if (gsdr_txgen_get_device(g, reinterpret_cast<gsdr_c64 *>(g->d_stage), nullptr) != 0) return -1;
    const hipError_t e = hipMemcpy(out_host, g->d_stage, (size_t)g->buffer_len * sizeof(float2), hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
        g_create_error = std::string("gsdr_txgen_get: ") + hipGetErrorString(e);
        return -1;
    }
    return 0;
}

// ref: the TONES branch of the constructor (:77-95): base_buffer = one period (TONES_buffer_len samples) plus
// buffer_len more (a copy of its beginning), in host memory.  Made once, in pieces through a device buffer.
int gsdr_txgen_prepare_host(gsdr_txgen *g) {
    if (!g || g->mode != GSDR_TONES) {
        g_create_error = "gsdr_txgen_prepare_host: a TONES generator is needed";
        return -1;
    }
    if (g->h_period) return 0;
    if (g->device >= 0 && hipSetDevice(g->device) != hipSuccess) {
        g_create_error = "gsdr_txgen_prepare_host: hipSetDevice failed";
        return -1;
    }
    const unsigned long long total = g->period + (unsigned long long)g->buffer_len;
    float2 *hp = nullptr;
    bool pinned = hipHostMalloc((void **)&hp, (size_t)total * sizeof(float2)) == hipSuccess;
    if (!pinned) {
        (void)hipGetLastError();
        hp = (float2 *)std::malloc((size_t)total * sizeof(float2));
    }
    if (!hp) {
        g_create_error = "gsdr_txgen_prepare_host: cannot allocate the period buffer in host memory";
        return -1;
    }
    const size_t piece = (size_t)(total < (8u << 20) ? total : (8u << 20));
    if (g->d_stage && g->stage_n < piece) {
        (void)hipFree(g->d_stage);
        g->d_stage = nullptr;
    }
    bool ok = true;
    if (!g->d_stage) {
        ok = dev_alloc(&g->d_stage, piece) == hipSuccess;
        g->stage_n = ok ? piece : 0;
    }
    for (unsigned long long off = 0; ok && off < total; off += piece) {
        const long long n = (long long)(total - off < piece ? total - off : piece);
        ok = gsdr_txgen_tones_fill(g, reinterpret_cast<gsdr_c64 *>(g->d_stage), n, (long long)(off % g->rate), nullptr) == 0 &&
             hipMemcpy(hp + off, g->d_stage, (size_t)n * sizeof(float2), hipMemcpyDeviceToHost) == hipSuccess;
    }
    if (!ok) {
        if (pinned) (void)hipHostFree(hp);
        else std::free(hp);
        if (g_create_error.empty()) g_create_error = "gsdr_txgen_prepare_host: generating the period failed";
        return -1;
    }
    g->h_period = hp;
    g->h_period_pinned = pinned;
    return 0;
}

// ref: get_from_tones (:226-229): *target = base_buffer + TONES_last_sample -- the caller's pointer is REPLACED by one
// into the generator's own period buffer (tx_single_link hands in an unallocated pointer for TONES,
// cpp/USRP_server_link_threads.cpp:568-584, and never frees what it gets back).
const gsdr_c64 *gsdr_txgen_get_ptr(gsdr_txgen *g) {
    if (!g || g->mode != GSDR_TONES) {
        g_create_error = "gsdr_txgen_get_ptr: a TONES generator is needed";
        return nullptr;
    }
    if (!g->h_period && gsdr_txgen_prepare_host(g) != 0) return nullptr;
    const gsdr_c64 *p = reinterpret_cast<const gsdr_c64 *>(g->h_period + g->last);
    g->last = (g->last + (unsigned long long)g->buffer_len) % g->period;
    return p;
}

int gsdr_txgen_mode(const gsdr_txgen *g) { return g ? g->mode : -1; }

int gsdr_source_chirp(gsdr_c64 *out_dev, long long n, unsigned long long last_index,
                      const gsdr_chirp_param *cp, float scale, void *hip_stream) {
    if (!out_dev || !cp || n < 0 || cp->num_steps < 1 || cp->length < 1) return -1;
    ChirpShape cs{};
    cs.num_steps = cp->num_steps;
    cs.length = cp->length;
    cs.period = cp->num_steps * cp->length;
    cs.chirpness = cp->chirpness;
    cs.f0 = cp->f0;
    hipError_t e = gsdr::launch_source_chirp(reinterpret_cast<float2 *>(out_dev), n,
                                             last_index % cs.period, cs, scale,
                                             (hipStream_t)hip_stream);
    if (e != hipSuccess) {
        g_create_error = std::string("gsdr_source_chirp: ") + hipGetErrorString(e);
        return -1;
    }
    return 0;
}

}  // extern "C"
