// LADSPA plugin "speech_separator" on the MI355X path: same C ABI as the reference's
// speech-ladspa-onnx.cpp (descriptor :293-337, ladspa_descriptor :339-342; identical in
// speech-ladspa-torch.cpp:224-273), with the ONNX Runtime session + FFTW replaced by
// libbsrnn_hip's device-resident streaming step (bsrnn_stream_step_host).
//
// Kept from the reference: UniqueID/label/name/maker/ports/hints; run() re-blocks any
// sampleCount into 1024-sample chunks, reading input then writing output per sample so
// in-place hosts work (:152-169); channel 0 feeds both model rows (:183-188); the mono result
// goes to both outputs (:258-260); wet/dry control (:215-226); 1024-sample output delay; LSTM
// state carried across chunks (:264).
// Changed on purpose: the model file is the flat weight file named by $BSRNN_WEIGHTS (default
// ./model-always.bsrnnw) instead of a hard-coded ONNX path (:73); instantiate() returns NULL
// on any failure and nothing throws across the C ABI (the reference logs and may dereference
// null, :117-119); run() allocates, compiles and captures nothing: instantiate() has done all of it (bsrnn_stream_create runs two
// throw-away steps: every kernel loaded, both parity graphs captured and instantiated, carry zeroed again) - as the reference's
// constructor builds its session, FFT plans and state (:55-120) and its run() only uses them (:152-169).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/bsrnn_hip.h"
#include "ladspa_min.h"

namespace {

constexpr int kChunk = 1024;
constexpr int kChannels = 2;

struct Separator {
    LADSPA_Data* control = nullptr;
    LADSPA_Data* in[2] = {nullptr, nullptr};
    LADSPA_Data* out[2] = {nullptr, nullptr};
    bsrnn_ctx* ctx = nullptr;
    bsrnn_stream* stream = nullptr;
    int pos = 0;
    float buf_in[kChannels][kChunk] = {};
    float buf_out[kChannels][kChunk] = {};
    float chunk[kChannels * kChunk] = {};
    float result[kChannels * kChunk] = {};

    ~Separator()
    {
        if (stream) bsrnn_stream_destroy(stream);
        if (ctx) bsrnn_destroy(ctx);
    }

    bool init()
    {
        const char* path = getenv("BSRNN_WEIGHTS");
        if (!path || !*path) path = "model-always.bsrnnw";
        const char* dev = getenv("BSRNN_DEVICE");
        // the band table travels in the weight file header (weights.py): magic, n_bands, widths
        FILE* f = fopen(path, "rb");
        if (!f) { fprintf(stderr, "speech_separator: cannot open weight file %s\n", path); return false; }
        char magic[8];
        uint32_t nb = 0;
        std::vector<int32_t> widths;
        bool ok = fread(magic, 1, 8, f) == 8 && !memcmp(magic, "BSRNNW01", 8) && fread(&nb, 4, 1, f) == 1 && nb > 0 && nb <= 256;
        if (ok) {
            widths.resize(nb);
            ok = fread(widths.data(), 4, nb, f) == nb;
        }
        fclose(f);
        if (!ok) { fprintf(stderr, "speech_separator: %s is not a BSRNNW01 weight file\n", path); return false; }
        if (bsrnn_create(dev ? atoi(dev) : 0, widths.data(), (int32_t)nb, &ctx) ||
            bsrnn_load_weights_file(ctx, path) || bsrnn_stream_create(ctx, kChannels, &stream)) {
            fprintf(stderr, "speech_separator: %s\n", bsrnn_last_error());
            return false;
        }
        return true;
    }

    void new_chunk()
    {
        // channel 0 drives both model rows (speech-ladspa-onnx.cpp:183-188)
        memcpy(chunk, buf_in[0], sizeof(float) * kChunk);
        memcpy(chunk + kChunk, buf_in[0], sizeof(float) * kChunk);
        const float mix = control ? *control : 1.0f;
        if (bsrnn_stream_step_host(stream, chunk, result, mix) != 0) {
            fprintf(stderr, "speech_separator: %s\n", bsrnn_last_error());
            memset(result, 0, sizeof result);
        }
        for (int c = 0; c < kChannels; ++c) memcpy(buf_out[c], result, sizeof(float) * kChunk);   // mono to both (:258-260)
        pos = 0;
    }

    void run(unsigned long n)
    {
        if (!in[0] || !in[1] || !out[0] || !out[1]) return;
        unsigned long p = 0;
        while (n > 0) {
            const unsigned long k = (unsigned long)(kChunk - pos) < n ? (unsigned long)(kChunk - pos) : n;
            for (unsigned long i = 0; i < k; ++i) {          // read before write: in-place safe
                buf_in[0][pos] = in[0][p + i];
                buf_in[1][pos] = in[1][p + i];
                out[0][p + i] = buf_out[0][pos];
                out[1][p + i] = buf_out[1][pos];
                ++pos;
            }
            n -= k;
            p += k;
            if (pos == kChunk) new_chunk();
        }
    }
};

LADSPA_Handle instantiate(const LADSPA_Descriptor*, unsigned long /*sampleRate*/)
{
    Separator* s = new (std::nothrow) Separator();
    if (!s) return nullptr;
    if (!s->init()) { delete s; return nullptr; }
    return s;
}
void connect_port(LADSPA_Handle h, unsigned long port, LADSPA_Data* data)
{
    Separator* s = static_cast<Separator*>(h);
    switch (port) {
    case 0: s->control = data; break;
    case 1: s->in[0] = data; break;
    case 2: s->in[1] = data; break;
    case 3: s->out[0] = data; break;
    case 4: s->out[1] = data; break;
    default: break;
    }
}
void run(LADSPA_Handle h, unsigned long n) { static_cast<Separator*>(h)->run(n); }
void cleanup(LADSPA_Handle h) { delete static_cast<Separator*>(h); }

const LADSPA_PortDescriptor kPorts[5] = {
    LADSPA_PORT_INPUT | LADSPA_PORT_CONTROL, LADSPA_PORT_INPUT | LADSPA_PORT_AUDIO, LADSPA_PORT_INPUT | LADSPA_PORT_AUDIO,
    LADSPA_PORT_OUTPUT | LADSPA_PORT_AUDIO, LADSPA_PORT_OUTPUT | LADSPA_PORT_AUDIO};
const char* const kPortNames[5] = {"Control", "Input (Left)", "Input (Right)", "Output (Left)", "Output (Right)"};
const LADSPA_PortRangeHint kHints[5] = {
    {LADSPA_HINT_DEFAULT_1 | LADSPA_HINT_BOUNDED_BELOW, 0.0f, 0.0f}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};

const LADSPA_Descriptor kDescriptor = {
    0xf433b044UL, "speech_separator", 0, "Speech Separator", "Pierre-Hugues Husson @ Freebox", "None",
    5, kPorts, kPortNames, kHints, nullptr,
    instantiate, connect_port, nullptr /*activate*/, run, nullptr /*run_adding*/, nullptr /*set_run_adding_gain*/,
    nullptr /*deactivate*/, cleanup};

}  // namespace

extern "C" const LADSPA_Descriptor* ladspa_descriptor(unsigned long idx) { return idx == 0 ? &kDescriptor : nullptr; }
