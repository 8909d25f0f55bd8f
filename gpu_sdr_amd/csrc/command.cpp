// command.cpp -- the pyUSRP command surface, host-only (no HIP): JSON command ->
// usrp_param, the server-side sanity pass, ack/nack messages and the two wire
// headers.  Pure functions, exported through the C ABI (include/gsdr.h) and
// usable on a machine without a GPU.
//
// ref: string2param cpp/USRP_JSON_interpreter.cpp:19-257, chk_param :268-439,
// server_ack/nack :441-457, Async_server::format_header
// cpp/USRP_server_network.cpp:497-501, Sync_server::format_net_buffer :164-191,
// string_to_w_type / ant_mode_from_string cpp/USRP_server_settings.cpp.
// (citations relative to /root/reference)
#include <cctype>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/gsdr.h"

namespace {

// ---- a small JSON reader (objects, arrays, strings, numbers, true/false/null) ----
struct JValue {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    double num = 0;
    bool b = false;
    std::string str;
    std::vector<JValue> arr;
    std::vector<std::pair<std::string, JValue>> obj;
    const JValue *get(const std::string &k) const {
        for (auto &kv : obj)
            if (kv.first == k) return &kv.second;
        return nullptr;
    }
};

struct JParser {
    const char *p, *end;
    std::string err;
    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
    bool fail(const char *m) { if (err.empty()) err = m; return false; }
    bool str(std::string &out) {
        if (p >= end || *p != '"') return fail("expected string");
        ++p;
        while (p < end && *p != '"') {
            if (*p == '\\' && p + 1 < end) {
                ++p;
                switch (*p) {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'u': if (end - p >= 5) { out += '?'; p += 4; } break;
                    default: out += *p;
                }
                ++p;
            } else {
                out += *p++;
            }
        }
        if (p >= end) return fail("unterminated string");
        ++p;
        return true;
    }
    bool value(JValue &v, int depth = 0) {
        if (depth > 32) return fail("nesting too deep");
        ws();
        if (p >= end) return fail("unexpected end");
        if (*p == '{') {
            v.kind = JValue::Obj;
            ++p; ws();
            if (p < end && *p == '}') { ++p; return true; }
            for (;;) {
                ws();
                std::string k;
                if (!str(k)) return false;
                ws();
                if (p >= end || *p != ':') return fail("expected ':'");
                ++p;
                JValue c;
                if (!value(c, depth + 1)) return false;
                v.obj.emplace_back(std::move(k), std::move(c));
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == '}') { ++p; return true; }
                return fail("expected ',' or '}'");
            }
        }
        if (*p == '[') {
            v.kind = JValue::Arr;
            ++p; ws();
            if (p < end && *p == ']') { ++p; return true; }
            for (;;) {
                JValue c;
                if (!value(c, depth + 1)) return false;
                v.arr.push_back(std::move(c));
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == ']') { ++p; return true; }
                return fail("expected ',' or ']'");
            }
        }
        if (*p == '"') { v.kind = JValue::Str; return str(v.str); }
        if (end - p >= 4 && !std::strncmp(p, "true", 4)) { v.kind = JValue::Bool; v.b = true; p += 4; return true; }
        if (end - p >= 5 && !std::strncmp(p, "false", 5)) { v.kind = JValue::Bool; v.b = false; p += 5; return true; }
        if (end - p >= 4 && !std::strncmp(p, "null", 4)) { v.kind = JValue::Null; p += 4; return true; }
        // a JSON number: plain decimal text only (strtod alone would also take nan, inf and hex)
        const char *t = p;
        while (t < end && t - p < 64 && (std::isdigit((unsigned char)*t) || *t == '-' || *t == '+' || *t == '.' || *t == 'e' || *t == 'E')) ++t;
        if (t == p) return fail("unexpected character");
        char *q = nullptr;
        std::string tmp(p, (size_t)(t - p));
        v.num = std::strtod(tmp.c_str(), &q);
        if (q != tmp.c_str() + tmp.size() || !std::isfinite(v.num)) return fail("malformed number");
        v.kind = JValue::Num;
        p = t;
        return true;
    }
};

// boost::property_tree keeps every leaf as text and converts on get<T>(): a
// number may arrive as a JSON number or as a numeric string.
bool as_double(const JValue *v, double &out) {
    if (!v) return false;
    if (v->kind == JValue::Num) { out = v->num; return true; }
    if (v->kind == JValue::Bool) { out = v->b ? 1 : 0; return true; }
    if (v->kind == JValue::Str) {
        // numeric text: plain decimal only, finite (a stream extraction refuses nan/inf/hex too)
        if (v->str.empty() || v->str.size() > 64) return false;
        for (char ch : v->str)
            if (!(std::isdigit((unsigned char)ch) || ch == '-' || ch == '+' || ch == '.' || ch == 'e' || ch == 'E')) return false;
        char *q = nullptr;
        out = std::strtod(v->str.c_str(), &q);
        return q != v->str.c_str() && *q == 0 && std::isfinite(out);
    }
    return false;
}
bool as_integer(const JValue *v, long long &out) {  // get<int>/get<size_t>: text must be an integer
    double d;
    if (!as_double(v, d) || d != std::floor(d) || std::fabs(d) > 9.0e18) return false;
    out = (long long)d;
    return true;
}
// The reference reads most scalars with get<double> and lets C++ narrow them (undefined for
// values the target cannot hold).  Here a value outside the target's range is a type error:
// int fields within (INT_MIN, INT_MAX], size_t fields within [0, 2^53].
bool to_int(double d, int &out) {
    if (!(d > (double)INT_MIN && d <= (double)INT_MAX)) return false;
    out = (int)d;
    return true;
}
bool to_size(double d, unsigned long long &out) {
    if (!(d >= 0.0 && d <= 9007199254740992.0)) return false;
    out = (unsigned long long)d;
    return true;
}

struct Ant {                       // struct param, ref: headers/USRP_server_settings.hpp:130-167
    int mode = 2;                  // ant_mode { TX, RX, OFF }
    int rate = 0, gain = 0, bw = 0;
    unsigned long long tone = 0, samples = 0, buffer_len = 0, decim = 0, data_mem_mult = 0, pf_average = 0;
    double delay = 0;
    float burst_on = 0, burst_off = 0;
    int tuning_mode = 0, fft_tones = 0;
    std::vector<int> freq, wave_type, chirp_f, swipe_s;
    std::vector<float> ampl, chirp_t;
};

}  // namespace

struct gsdr_command {
    int usrp_number = 0;
    Ant ant[4];                    // A_TXRX, B_TXRX, A_RX2, B_RX2
};

namespace {

thread_local std::string g_cmd_error;
const char *kAnt[4] = {"A_TXRX", "B_TXRX", "A_RX2", "B_RX2"};

int w_type_from_string(const std::string &s) {  // ref: cpp/USRP_server_settings.cpp:38-54
    if (s == "CHIRP") return GSDR_CHIRP;
    if (s == "NOISE") return GSDR_NOISE;
    if (s == "TONES") return GSDR_TONES;
    if (s == "SWONLY") return GSDR_SWONLY;
    if (s == "DIRECT") return GSDR_DIRECT;
    return GSDR_NODSP;             // "NODSP", "RAMP" and anything unknown
}

int ant_mode_from_string(const std::string &s) {  // ref: cpp/USRP_server_settings.cpp (ant_mode_from_string)
    if (s == "TX") return 0;
    if (s == "RX") return 1;
    return 2;                      // "OFF" and anything else
}

bool type_error(const char *key) {
    g_cmd_error = std::string("could not parse the JSON file correctly: be sure that the data type used "
                              "for descriptor \"") + key + "\" match the specifications!";
    return false;
}

// ref: string2param -- every key of every antenna object is mandatory
bool fill(const JValue &root, gsdr_command &c) {
    long long dev;
    if (root.kind != JValue::Obj || !as_integer(root.get("device"), dev)) {
        g_cmd_error = "missing device ID or wrong JSON string";
        return false;
    }
    c.usrp_number = (int)dev;
    const JValue *a[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = root.get(kAnt[i]);
        if (!a[i] || a[i]->kind != JValue::Obj) return type_error("mode");
    }
#define EACH(stmt) for (int i = 0; i < 4; ++i) { Ant &A = c.ant[i]; const JValue &J = *a[i]; (void)A; (void)J; stmt }
    // the reference reads key by key across the four antennas, in this order
    EACH({ const JValue *v = J.get("mode"); if (!v || v->kind != JValue::Str) return type_error("mode");
           A.mode = ant_mode_from_string(v->str); })
    double d; long long n;
    EACH({ if (!as_double(J.get("rf"), d) || !to_size(d, A.tone)) return type_error("rf"); })
    EACH({ if (!as_integer(J.get("tuning_mode"), n) || n < 0 || n > INT_MAX) return type_error("tuning_mode"); A.tuning_mode = (int)n; })
    EACH({ if (!as_double(J.get("rate"), d) || !to_int(d, A.rate)) return type_error("rate"); })
    EACH({ if (!as_double(J.get("decim"), d) || !to_size(d, A.decim)) return type_error("decim"); })
    EACH({ if (!as_double(J.get("fft_tones"), d) || !to_int(d, A.fft_tones)) return type_error("fft_tones"); })
    EACH({ if (!as_double(J.get("pf_average"), d) || !to_size(d, A.pf_average)) return type_error("pf_average"); })
    EACH({ if (!as_integer(J.get("samples"), n) || n < 0) return type_error("samples"); A.samples = (unsigned long long)n; })
    EACH({ if (!as_double(J.get("buffer_len"), d) || !to_size(d, A.buffer_len)) return type_error("buffer_len"); })
    EACH({ if (!as_double(J.get("burst_off"), d)) return type_error("burst_off"); A.burst_off = (float)d; })
    EACH({ if (!as_double(J.get("burst_on"), d)) return type_error("burst_on"); A.burst_on = (float)d; })
    EACH({ if (!as_double(J.get("bw"), d) || !to_int(d, A.bw)) return type_error("bw"); })
    EACH({ if (!as_double(J.get("delay"), d)) return type_error("delay"); A.delay = d; })
    EACH({ if (!as_double(J.get("gain"), d) || !to_int(d, A.gain)) return type_error("gain"); })
#define INT_LIST(key, field)                                                                     \
    EACH({ const JValue *v = J.get(key); if (!v || v->kind != JValue::Arr) return type_error(key); \
           for (auto &e : v->arr) { if (!as_integer(&e, n) || n <= (long long)INT_MIN || n > (long long)INT_MAX) return type_error(key);  \
                                    A.field.push_back((int)n); } })
#define FLT_LIST(key, field)                                                                     \
    EACH({ const JValue *v = J.get(key); if (!v || v->kind != JValue::Arr) return type_error(key); \
           for (auto &e : v->arr) { if (!as_double(&e, d)) return type_error(key); A.field.push_back((float)d); } })
    INT_LIST("freq", freq)
    FLT_LIST("ampl", ampl)
    EACH({ const JValue *v = J.get("wave_type"); if (!v || v->kind != JValue::Arr) return type_error("wave_type");
           for (auto &e : v->arr) {  // as_vector<std::string>: numbers come through as text and map to NODSP
               A.wave_type.push_back(e.kind == JValue::Str ? w_type_from_string(e.str) : (int)GSDR_NODSP); } })
    FLT_LIST("chirp_t", chirp_t)
    INT_LIST("chirp_f", chirp_f)
    INT_LIST("swipe_s", swipe_s)
    EACH({ if (!as_double(J.get("data_mem_mult"), d) || !to_size(d, A.data_mem_mult)) return type_error("data_mem_mult"); })
#undef EACH
#undef INT_LIST
#undef FLT_LIST
    return true;
}

// ref: chk_param, cpp/USRP_JSON_interpreter.cpp:268-439 (same for the four antennas)
bool check(gsdr_command &c) {
    for (int i = 0; i < 4; ++i) {
        Ant &A = c.ant[i];
        if (A.mode == 2) continue;
        bool pfb = false;
        for (int w : A.wave_type) pfb |= (w == GSDR_TONES || w == GSDR_NOISE);
        if (pfb) {
            if (A.pf_average <= 0) A.pf_average = 1;
            if (A.fft_tones <= 0) A.fft_tones = 2;
        }
        if (A.mode == 0) {
            // a TX tone comb reads ampl[k] and freq[k] for every wave_type entry (ref: tone_gen,
            // cpp/kernels.cu:589-684, buffer_generator.cpp:60-157): short lists would be read past
            size_t tones = 0;
            for (int w : A.wave_type) tones += (w == GSDR_TONES);
            if (tones > 0 && (A.ampl.size() < A.wave_type.size() || A.freq.size() < A.wave_type.size())) {
                g_cmd_error = std::string("Number of amplitude/frequency descriptors does not match the number of "
                                          "signal mode descriptor in parameter '") + kAnt[i] + "'";
                return false;
            }
        }
        if (A.buffer_len == 0) A.buffer_len = 1000000;                         // DEFAULT_BUFFER_LEN
        if (A.buffer_len > 6000000 || A.buffer_len < 50000) A.buffer_len = 1000000;  // MAX/MIN_USEFULL_BUFFER
        for (size_t k = 0; k < A.wave_type.size(); ++k) {
            const int w = A.wave_type[k];
            if (w == GSDR_CHIRP || w == GSDR_TONES) {
                if (k >= A.freq.size()) {
                    g_cmd_error = std::string("Number of frequency descriptor does not match the number of "
                                              "signal mode descriptor in parameter '") + kAnt[i] + "'";
                    return false;
                }
                if (std::abs(A.freq[k]) > A.rate) {
                    g_cmd_error = "frequency descriptor " + std::to_string(k) + " in '" + kAnt[i] +
                                  "' parameter is out of Nyquist range: " + std::to_string(A.freq[k]) + ">" +
                                  std::to_string(A.rate);
                    return false;
                }
            }
            if (w == GSDR_CHIRP) {
                if (k >= A.chirp_f.size()) {
                    g_cmd_error = std::string("Number of frequency descriptor does not match the number of "
                                              "signal mode descriptor in parameter '") + kAnt[i] + "'";
                    return false;
                }
                if (std::abs(A.chirp_f[k]) > A.rate) {
                    g_cmd_error = "second frequency descriptor " + std::to_string(k) + " in '" + kAnt[i] +
                                  "' parameter is out of Nyquist range: " + std::to_string(A.chirp_f[k]) + ">" +
                                  std::to_string(A.rate);
                    return false;
                }
            }
        }
    }
    return true;
}

}  // namespace

extern "C" {

const char *gsdr_command_error(void) { return g_cmd_error.c_str(); }

gsdr_command *gsdr_command_parse(const char *json, int len) {
    g_cmd_error.clear();
    if (!json || len < 0) {
        g_cmd_error = "missing device ID or wrong JSON string";
        return nullptr;
    }
    JParser jp{json, json + len, {}};
    JValue root;
    if (!jp.value(root)) {
        g_cmd_error = "missing device ID or wrong JSON string (" + jp.err + ")";
        return nullptr;
    }
    std::unique_ptr<gsdr_command> c(new gsdr_command());
    if (!fill(root, *c)) return nullptr;
    if (!check(*c)) return nullptr;
    return c.release();
}

void gsdr_command_free(gsdr_command *c) { delete c; }

int gsdr_command_device(const gsdr_command *c) { return c ? c->usrp_number : -1; }

int gsdr_command_antenna(const gsdr_command *c, int antenna, gsdr_param_c *p, gsdr_antenna_info *info) {
    if (!c || antenna < 0 || antenna > 3) return -1;
    const Ant &A = c->ant[antenna];
    if (p) {
        std::memset(p, 0, sizeof(*p));
        p->rate = A.rate;
        p->decim = (long long)A.decim;
        p->fft_tones = A.fft_tones;
        p->pf_average = (long long)A.pf_average;
        p->buffer_len = (long long)A.buffer_len;
        p->wave_type = A.wave_type.data();
        p->n_wave_type = (int)A.wave_type.size();
        p->freq = A.freq.data();
        p->n_freq = (int)A.freq.size();
        p->chirp_t = A.chirp_t.data();
        p->n_chirp_t = (int)A.chirp_t.size();
        p->chirp_f = A.chirp_f.data();
        p->n_chirp_f = (int)A.chirp_f.size();
        p->swipe_s = A.swipe_s.data();
        p->n_swipe_s = (int)A.swipe_s.size();
        p->device_index = -1;
    }
    if (info) {
        info->mode = A.mode;
        info->rf = (double)A.tone;
        info->gain = A.gain;
        info->bw = A.bw;
        info->tuning_mode = A.tuning_mode;
        info->samples = (long long)A.samples;
        info->delay = A.delay;
        info->burst_on = A.burst_on;
        info->burst_off = A.burst_off;
        info->data_mem_mult = (long long)A.data_mem_mult;
        info->ampl = A.ampl.data();
        info->n_ampl = (int)A.ampl.size();
    }
    return 0;
}

// ref: server_ack / server_nack (:441-457): boost::property_tree::write_json layout
int gsdr_server_reply(int is_ack, const char *payload, char *buf, int cap) {
    std::string esc;
    for (const char *q = payload ? payload : ""; *q; ++q) {
        switch (*q) {
            case '"': esc += "\\\""; break;
            case '\\': esc += "\\\\"; break;
            case '/': esc += "\\/"; break;   // boost escapes the solidus
            case '\n': esc += "\\n"; break;
            case '\t': esc += "\\t"; break;
            case '\r': esc += "\\r"; break;
            default: esc += *q;
        }
    }
    const std::string s = std::string("{\n    \"type\": \"") + (is_ack ? "ack" : "nack") +
                          "\",\n    \"payload\": \"" + esc + "\"\n}\n";
    if (buf && cap > 0) {
        const size_t n = s.size() < (size_t)(cap - 1) ? s.size() : (size_t)(cap - 1);
        std::memcpy(buf, s.data(), n);
        buf[n] = 0;
    }
    return (int)s.size();
}

// ref: Async_server::format_header, cpp/USRP_server_network.cpp:497-501
void gsdr_format_async_header(int message_len, unsigned char out[8]) {
    const int head[2] = {0, message_len};
    std::memcpy(out, head, 8);
}

// ref: Sync_server::format_net_buffer, cpp/USRP_server_network.cpp:164-191 (header part)
void gsdr_format_rx_header(const gsdr_rx_header *h, unsigned char out[21]) {
    int off = 0;
    std::memcpy(out + off, &h->usrp_number, 4); off += 4;
    std::memcpy(out + off, &h->front_end_code, 1); off += 1;
    std::memcpy(out + off, &h->packet_number, 4); off += 4;
    std::memcpy(out + off, &h->length, 4); off += 4;
    std::memcpy(out + off, &h->errors, 4); off += 4;
    std::memcpy(out + off, &h->channels, 4);
}

}  // extern "C"
