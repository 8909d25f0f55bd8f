// USRP_demodulator.hpp -- header-only C++ drop-in for the reference's
// RX_buffer_demodulator on top of the C ABI in gsdr.h.
//
// A GPU_SDR server tree that replaces its headers/USRP_demodulator.hpp by this
// file (and drops cpp/USRP_demodulator.cpp, cpp/kernels.cu, cpp/fir.cu from the
// build, linking -lgsdr instead of -lcufft -lcublas) keeps compiling its
// callers unchanged: TXRX::set constructs `new RX_buffer_demodulator(param*)`
// (ref: cpp/USRP_server_link_threads.cpp:121,136), rx_single_link calls
// `demodulator->process(&in,&out)` and reads
// `demodulator->parameters->wave_type.size()` (ref: :657,:666), TXRX::stop
// calls `close()` (ref: :475,:485).
//
// The types below mirror headers/USRP_server_settings.hpp:114,130-167,216-224
// field for field; when this header is used INSIDE the reference tree, define
// GSDR_USE_REFERENCE_SETTINGS before including it so that the tree's own
// USRP_server_settings.hpp provides them instead.
#pragma once
#ifndef USRP_DEMODULATOR_INCLUDED
#define USRP_DEMODULATOR_INCLUDED

#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gsdr.h"

#ifdef GSDR_USE_REFERENCE_SETTINGS
#include "USRP_server_settings.hpp"
#else

// float2 as the CUDA/HIP runtime headers define it (8-byte POD, x = re, y = im)
#if !defined(__HIPCC__) && !defined(__CUDACC__) && !defined(HIP_INCLUDE_HIP_AMD_DETAIL_HIP_VECTOR_TYPES_H) && \
    !defined(__VECTOR_TYPES_H__)
struct float2 {
    float x, y;
};
#endif

// ref: headers/USRP_server_settings.hpp:114
enum w_type { TONES, CHIRP, NOISE, RAMP, NODSP, SWONLY, DIRECT };
// ref: headers/USRP_server_settings.hpp:123
enum ant_mode { TX, RX, OFF };

// ref: headers/USRP_server_settings.hpp:130-167
struct param {
    ant_mode mode = OFF;
    int rate, gain, bw;
    size_t tone;
    size_t samples;
    double delay;
    float burst_on;
    float burst_off;
    size_t buffer_len;
    bool tuning_mode;
    std::vector<int> freq;
    std::vector<w_type> wave_type;
    std::vector<float> ampl;
    size_t decim;
    std::vector<float> chirp_t;
    std::vector<int> chirp_f;
    std::vector<int> swipe_s;
    size_t data_mem_mult;
    int fft_tones;
    size_t pf_average;
    //! does a TX measurement with these parameters need buffers allocated per packet?
    //! (ref: param::dynamic_buffer, cpp/USRP_server_settings.cpp:98-102: everything but TONES)
    bool dynamic_buffer() {
        bool dynamic = false;
        for (size_t i = 0; i < wave_type.size(); i++)
            if (wave_type[i] != TONES) dynamic = true;
        return dynamic;
    }
};

// ref: headers/USRP_server_settings.hpp:216-224
struct RX_wrapper {
    float2* buffer;
    int usrp_number;
    char front_end_code;
    int packet_number;
    int length;
    int errors;
    int channels;
};
#endif  // GSDR_USE_REFERENCE_SETTINGS

static_assert(sizeof(float2) == sizeof(gsdr_c64), "float2 must be two packed floats");

//! Same public surface as ref: headers/USRP_demodulator.hpp:13-33.
class RX_buffer_demodulator {
   public:
    //! stores the signal processing parameters (borrowed, like the reference)
    param* parameters;

    //! PFB cut-off frequency of the window (ref: USRP_demodulator.hpp:21)
    float fcut;

    //! GPU used by demodulators created afterwards; the reference takes it from
    //! server_settings::GPU_device_index via cudaSetDevice at start-up
    //! (ref: cpp/USRP_hardware_manager.cpp:68). -1 keeps the current device.
    static int& device_index() {
        static int idx = -1;
        return idx;
    }

    RX_buffer_demodulator(param* init_parameters, bool init_diagnostic = false)
        : parameters(init_parameters), fcut(0.f), handle_(nullptr), diagnostic_(init_diagnostic) {
        std::vector<int> wt(parameters->wave_type.begin(), parameters->wave_type.end());
        gsdr_param_c pc;
        pc.rate = parameters->rate;
        pc.decim = (long long)parameters->decim;
        pc.fft_tones = parameters->fft_tones;
        pc.pf_average = (long long)parameters->pf_average;
        pc.buffer_len = (long long)parameters->buffer_len;
        pc.wave_type = wt.data();
        pc.n_wave_type = (int)wt.size();
        pc.freq = parameters->freq.data();
        pc.n_freq = (int)parameters->freq.size();
        pc.chirp_t = parameters->chirp_t.data();
        pc.n_chirp_t = (int)parameters->chirp_t.size();
        pc.chirp_f = parameters->chirp_f.data();
        pc.n_chirp_f = (int)parameters->chirp_f.size();
        pc.swipe_s = parameters->swipe_s.data();
        pc.n_swipe_s = (int)parameters->swipe_s.size();
        pc.device_index = device_index();
        handle_ = gsdr_demod_create(&pc);
        if (!handle_) {
            // the reference print_error()s and exit(-1)s on unsupported requests
            // (ref: cpp/USRP_demodulator.cpp:31-39,322-325)
            std::fprintf(stderr, "ERROR: %s\n", gsdr_last_error(nullptr));
            std::exit(-1);
        }
        fcut = gsdr_demod_fcut(handle_);
        // like the reference's constructor (ref: cpp/USRP_demodulator.cpp:59-119): every device
        // buffer and stream exists before the first packet arrives
        if (gsdr_demod_prepare(handle_, GSDR_PREPARE_HOST | GSDR_PREPARE_PIPELINE | GSDR_PREPARE_PIPELINE_HOST | GSDR_PREPARE_REHEARSE) != 0)
            std::fprintf(stderr, "WARNING: demodulator: %s\n", gsdr_last_error(handle_));
        if (diagnostic_) std::fprintf(stderr, "WARNING: Demodulator diagnostic enabled.\n");
    }

    //! ref: USRP_demodulator.hpp:27-30 -- both are host pointers; returns the
    //! valid length (complex samples) written to *out.
    int process(float2** __restrict__ in, float2** __restrict__ out) {
        const int n = gsdr_demod_process(handle_, reinterpret_cast<const gsdr_c64*>(*in),
                                         reinterpret_cast<gsdr_c64*>(*out));
        if (n < 0) {
            // no error channel exists in the reference API: log and stream nothing
            std::fprintf(stderr, "ERROR: demodulator: %s\n", gsdr_last_error(handle_));
            return 0;
        }
        return n;
    }

    //! device-resident variant (synthetic in-HBM source); not in the reference
    int process_device(const float2* in_dev, float2* out_dev, void* hip_stream) {
        return gsdr_demod_process_device(handle_, reinterpret_cast<const gsdr_c64*>(in_dev),
                                         reinterpret_cast<gsdr_c64*>(out_dev), hip_stream);
    }

    //! pipelined variant of process() (extension, see gsdr_demod_submit in gsdr.h):
    //! submit up to GSDR_PIPELINE_DEPTH pinned buffers, then wait() for the oldest
    bool submit(float2** in, float2** out) {
        return gsdr_demod_submit(handle_, reinterpret_cast<const gsdr_c64*>(*in),
                                 reinterpret_cast<gsdr_c64*>(*out)) == 0;
    }
    //! the same for device-resident buffers (gsdr_demod_submit_device)
    bool submit_device(const float2* in_dev, float2* out_dev) {
        return gsdr_demod_submit_device(handle_, reinterpret_cast<const gsdr_c64*>(in_dev),
                                        reinterpret_cast<gsdr_c64*>(out_dev)) == 0;
    }
    int wait() {
        const int n = gsdr_demod_wait(handle_);
        if (n == -1) std::fprintf(stderr, "ERROR: demodulator: %s\n", gsdr_last_error(handle_));
        return n < 0 ? 0 : n;
    }

    //! ref: USRP_demodulator.hpp:33
    void close() {
        if (handle_) gsdr_demod_close(handle_);
        handle_ = nullptr;
    }

    //! room *out must have (complex samples); the reference sizes its pool as
    //! buffer_len * max(data_mem_mult, 1) (ref: cpp/USRP_server_link_threads.cpp:143-150)
    long long out_capacity() const { return gsdr_demod_out_capacity(handle_); }

   private:
    gsdr_demod* handle_;
    bool diagnostic_;
};

inline long long gsdr_demod_out_capacity_of(const RX_buffer_demodulator* d) { return d->out_capacity(); }

#endif
