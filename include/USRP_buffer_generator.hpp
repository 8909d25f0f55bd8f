// USRP_buffer_generator.hpp -- drop-in for the reference's TX_buffer_generator
// (ref: headers/USRP_buffer_generator.hpp:47-66, cpp/USRP_buffer_generator.cpp), header-only over the C ABI of
// libgsdr.so (include/gsdr.h: gsdr_txgen_*).  Same public surface: buffer_len, parameters, the constructor
// from param*, get(float2**) to a host buffer, close().  prefill_queue() is not mirrored: it takes the server's
// queue and allocator types, and the reference documents it as not working ("it doesn't update the index",
// headers/USRP_buffer_generator.hpp:62-64).  Where the reference exits on unsupported requests
// (cpp/USRP_buffer_generator.cpp:26-49) this prints the same message and exits as well.
//
// The `param`, `w_type`, `float2` types come from USRP_demodulator.hpp (or from the reference's own settings
// header with -DGSDR_USE_REFERENCE_SETTINGS).
#pragma once
#ifndef GSDR_USRP_BUFFER_GEN_INCLUDED
#define GSDR_USRP_BUFFER_GEN_INCLUDED

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "USRP_demodulator.hpp"
#include "gsdr.h"

class TX_buffer_generator {
   public:
    //! length of the buffer segment retrieved with get() (ref: USRP_buffer_generator.hpp:50)
    int buffer_len;
    //! the parameters used to generate the signal (borrowed, like the reference)
    param* parameters;

    TX_buffer_generator(param* init_parameters) : buffer_len((int)init_parameters->buffer_len), parameters(init_parameters), handle_(nullptr) {
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
        pc.device_index = RX_buffer_demodulator::device_index();
        handle_ = gsdr_txgen_create(&pc, parameters->ampl.data(), (int)parameters->ampl.size());
        if (!handle_) {
            std::fprintf(stderr, "ERROR: %s\n", gsdr_last_error(nullptr));
            std::exit(-1);
        }
        tones_ = gsdr_txgen_mode(handle_) == GSDR_TONES;
        // TONES: the whole period exists in host memory when the constructor returns, like the reference's
        // base_buffer (ref: cpp/USRP_buffer_generator.cpp:77-95)
        if (tones_ && gsdr_txgen_prepare_host(handle_) != 0) {
            std::fprintf(stderr, "ERROR: %s\n", gsdr_last_error(nullptr));
            std::exit(-1);
        }
    }

    //! The next buffer_len samples (ref: cpp/USRP_buffer_generator.cpp:163-169).  As in the reference the two
    //! kinds of generator treat `in` differently:
    //!   TONES  *in is REPLACED by a pointer into the generator's own period buffer (get_from_tones, :226-229);
    //!          whatever *in was is neither read nor written -- tx_single_link passes an unallocated pointer
    //!          here (param::dynamic_buffer() is false for TONES, ref: cpp/USRP_server_link_threads.cpp:568-584);
    //!   CHIRP  *in must point to buffer_len samples of host memory, which are filled (get_from_chirp, :208-221).
    void get(float2** in) {
        if (tones_) {
            const gsdr_c64* p = gsdr_txgen_get_ptr(handle_);
            if (!p) {
                std::fprintf(stderr, "ERROR: %s\n", gsdr_last_error(nullptr));
                std::exit(-1);
            }
            *in = reinterpret_cast<float2*>(const_cast<gsdr_c64*>(p));
            return;
        }
        if (gsdr_txgen_get(handle_, reinterpret_cast<gsdr_c64*>(*in)) != 0) {
            std::fprintf(stderr, "ERROR: %s\n", gsdr_last_error(nullptr));
            std::exit(-1);
        }
    }

    //! (ref: cpp/USRP_buffer_generator.cpp:172-178)
    void close() {
        if (handle_) gsdr_txgen_close(handle_);
        handle_ = nullptr;
    }

   private:
    gsdr_txgen* handle_;
    bool tones_ = false;
};

#endif
