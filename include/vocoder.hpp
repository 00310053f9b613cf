// vocoder.hpp -- C++ host-side mirror of the reference's `Vocoder` module
// (vocoder/src/Vocoder.hs:12-42) over the C ABI in vocoder_hip.h.  Header only.
//
// The reference is compiled Haskell; where its toolchain is absent this is the host language
// the path is used from.  Names, argument meaning and error behaviour follow the reference: a
// failing call throws std::runtime_error where the reference calls `error` (Vocoder.hs:132).
#pragma once

#include <complex>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "vocoder_hip.h"

namespace vocoder {

using Frame = std::vector<double>;     // Vocoder.hs:65
using Window = Frame;                  // Vocoder.hs:68
using Phase = std::vector<double>;     // Vocoder.hs:57
using Moduli = std::vector<double>;    // Vocoder.hs:54
using PhaseInc = std::vector<double>;  // Vocoder.hs:62
using HopSize = int;                   // Vocoder.hs:71
using Length = int;                    // Vocoder.hs:74
using STFTFrame = std::pair<Moduli, PhaseInc>;          // Vocoder.hs:77
using FFTOutput = std::vector<std::complex<double>>;    // Vocoder.hs:80; {re, im} pairs, the library's layout
struct STFTFrames {                    // a run of STFTFrame = (Moduli, PhaseInc), Vocoder.hs:77, F x K each
    std::vector<double> mag, ph_inc;
};

inline void check(int32_t st, const char *what, const voc_ctx *ctx = nullptr)
{
    if (st == VOC_OK) return;
    std::string msg = std::string(what) + ": " + voc_strerror(st);
    if (ctx && st == VOC_ERR_HIP) msg += std::string(" (") + voc_last_hip_error(ctx) + ")";
    throw std::runtime_error(msg);
}

// VocoderParams (Vocoder.hs:89-99) / vocoderParams len hop window (Vocoder.hs:114-115)
class VocoderParams {
public:
    VocoderParams(int length, int hop, const Window &window, int device = 0)
    {
        check(voc_ctx_create(length, hop, window.data(), (int32_t)window.size(), device, VOC_PRECISION_F64, &ctx_),
              "vocoderParams");
    }
    ~VocoderParams() { voc_ctx_destroy(ctx_); }
    VocoderParams(const VocoderParams &) = delete;
    VocoderParams &operator=(const VocoderParams &) = delete;

    int vocFrameLength() const { return voc_frame_length(ctx_); }             // Vocoder.hs:106-107
    int vocInputFrameLength() const { return voc_input_frame_length(ctx_); }  // Vocoder.hs:110-111
    int vocFreqFrameLength() const { return voc_freq_frame_length(ctx_); }    // Vocoder.hs:102-103
    int vocHopSize() const { return voc_hop_size(ctx_); }                     // Vocoder.hs:95
    double volumeCoeff() const { return voc_volume_coeff(ctx_); }             // Vocoder.hs:209-210
    Phase zeroPhase() const { return Phase((size_t)vocFreqFrameLength(), 0.0); }  // Vocoder.hs:204-205
    Window vocWindow() const                                                   // Vocoder.hs:97
    {
        Window w((size_t)vocInputFrameLength());
        check(voc_window(ctx_, w.data()), "vocWindow");
        return w;
    }
    voc_ctx *handle() const { return ctx_; }

private:
    voc_ctx *ctx_ = nullptr;
};

// doFFT par frame (Vocoder.hs:143-145): window, zero padding, rewind, real-to-complex transform
inline FFTOutput doFFT(const VocoderParams &par, const Frame &frame)
{
    if (frame.size() != (size_t)par.vocInputFrameLength()) throw std::runtime_error("doFFT: size mismatch");
    FFTOutput out((size_t)par.vocFreqFrameLength());
    check(voc_do_fft(par.handle(), frame.data(), reinterpret_cast<double *>(out.data()), 1), "doFFT", par.handle());
    return out;
}

// doIFFT par spectrum (Vocoder.hs:194-196): complex-to-real transform, rewind, cutCenter, window
inline Frame doIFFT(const VocoderParams &par, const FFTOutput &spec)
{
    if (spec.size() != (size_t)par.vocFreqFrameLength()) throw std::runtime_error("doIFFT: size mismatch");
    Frame out((size_t)par.vocInputFrameLength());
    check(voc_do_ifft(par.handle(), reinterpret_cast<const double *>(spec.data()), out.data(), 1), "doIFFT", par.handle());
    return out;
}

// analysisStep h eN prev_ph vec -> (ph, (mag, ph_inc))  (Vocoder.hs:158-163): no VocoderParams, as in the reference
inline std::pair<Phase, STFTFrame> analysisStep(HopSize h, Length eN, const Phase &prev_ph, const FFTOutput &vec)
{
    if (prev_ph.size() != vec.size()) throw std::runtime_error("analysisStep: size mismatch");
    const size_t k = vec.size();
    Phase ph(k);
    STFTFrame fr{Moduli(k), PhaseInc(k)};
    check(voc_analysis_step_hn(h, eN, (int64_t)k, prev_ph.data(), reinterpret_cast<const double *>(vec.data()), ph.data(),
                               fr.first.data(), fr.second.data(), 1),
          "analysisStep");
    return {std::move(ph), std::move(fr)};
}

// synthesisStep hop ph (mag, ph_inc) -> (new_ph, spectrum)  (Vocoder.hs:186-190)
inline std::pair<Phase, FFTOutput> synthesisStep(HopSize hop, const Phase &ph, const STFTFrame &fr)
{
    const size_t k = ph.size();
    if (fr.first.size() != k || fr.second.size() != k) throw std::runtime_error("synthesisStep: size mismatch");
    Phase out(k);
    FFTOutput spec(k);
    check(voc_synthesis_step_h(hop, (int64_t)k, ph.data(), fr.first.data(), fr.second.data(), out.data(),
                               reinterpret_cast<double *>(spec.data()), 1),
          "synthesisStep");
    return {std::move(out), std::move(spec)};
}

// frameFromComplex / frameToComplex (Vocoder.hs:217-218, 213-214)
inline STFTFrame frameFromComplex(const FFTOutput &vec)
{
    STFTFrame fr{Moduli(vec.size()), Phase(vec.size())};
    check(voc_frame_from_complex(reinterpret_cast<const double *>(vec.data()), fr.first.data(), fr.second.data(),
                                 (int64_t)vec.size()),
          "frameFromComplex");
    return fr;
}
inline FFTOutput frameToComplex(const STFTFrame &fr)
{
    if (fr.first.size() != fr.second.size()) throw std::runtime_error("frameToComplex: size mismatch");
    FFTOutput out(fr.first.size());
    check(voc_frame_to_complex(fr.first.data(), fr.second.data(), reinterpret_cast<double *>(out.data()), (int64_t)out.size()),
          "frameToComplex");
    return out;
}

// addFrames f1 f2 (Vocoder.hs:220-222)
inline STFTFrame addFrames(const STFTFrame &f1, const STFTFrame &f2)
{
    const size_t k = f1.first.size();
    if (f1.second.size() != k || f2.first.size() != k || f2.second.size() != k) throw std::runtime_error("addFrames: size mismatch");
    STFTFrame out{Moduli(k), PhaseInc(k)};
    check(voc_add_frames_n(f1.first.data(), f1.second.data(), f2.first.data(), f2.second.data(), out.first.data(),
                           out.second.data(), (int64_t)k),
          "addFrames");
    return out;
}

// analysisStage par ph frames -> (ph', stft)  (Vocoder.hs:149-150); frames is F x W, row major
inline std::pair<Phase, STFTFrames> analysisStage(const VocoderParams &par, Phase ph, const std::vector<double> &frames)
{
    const size_t w = (size_t)par.vocInputFrameLength(), k = (size_t)par.vocFreqFrameLength();
    if (frames.size() % w) throw std::runtime_error("analysisStage: size mismatch");
    const size_t f = frames.size() / w;
    STFTFrames out{std::vector<double>(f * k), std::vector<double>(f * k)};
    check(voc_analysis_stage(par.handle(), ph.data(), frames.data(), out.mag.data(), out.ph_inc.data(), 1, (int64_t)f),
          "analysisStage", par.handle());
    return {std::move(ph), std::move(out)};
}

// synthesisStage par ph stft -> (ph', frames)  (Vocoder.hs:177-178)
inline std::pair<Phase, std::vector<double>> synthesisStage(const VocoderParams &par, Phase ph, const STFTFrames &stft)
{
    const size_t w = (size_t)par.vocInputFrameLength(), k = (size_t)par.vocFreqFrameLength();
    const size_t f = stft.mag.size() / k;
    std::vector<double> frames(f * w);
    check(voc_synthesis_stage(par.handle(), ph.data(), stft.mag.data(), stft.ph_inc.data(), frames.data(), 1, (int64_t)f),
          "synthesisStage", par.handle());
    return {std::move(ph), std::move(frames)};
}

// analysisBlock / synthesisBlock (Vocoder.hs:153-154,181-182): the one-frame case
inline std::pair<Phase, STFTFrames> analysisBlock(const VocoderParams &par, Phase ph, const Frame &frame)
{
    return analysisStage(par, std::move(ph), frame);
}
inline std::pair<Phase, Frame> synthesisBlock(const VocoderParams &par, Phase ph, const STFTFrames &one)
{
    return synthesisStage(par, std::move(ph), one);
}

struct Processed {
    std::vector<double> y, leftover;  // C x out_len, C x keep (planar)
    std::vector<double> ana_phase, syn_phase;
};

// One `processWith q phs idFilter` step of Vocoder.Conduit (Conduit.hs:46-53) for C planar channels:
// x holds (previous leftover ++ new samples) per channel, `length` samples each.
inline Processed process(const VocoderParams &par, const std::vector<double> &x, int64_t channels, int64_t chunk,
                         std::vector<double> ana_phase, std::vector<double> syn_phase)
{
    const int64_t len = channels ? (int64_t)x.size() / channels : 0;
    const int64_t f = voc_num_frames(par.handle(), len), n = voc_ola_length(par.handle(), f, chunk),
                  keep = voc_leftover_length(par.handle(), len);
    Processed r{std::vector<double>((size_t)(channels * n)), std::vector<double>((size_t)(channels * keep)),
                std::move(ana_phase), std::move(syn_phase)};
    check(voc_roundtrip(par.handle(), x.data(), len, len, channels, chunk, r.y.data(), n, r.ana_phase.data(),
                        r.syn_phase.data(), r.leftover.data(), keep),
          "process", par.handle());
    return r;
}


// Vocoder.Filter (vocoder/src/Vocoder/Filter.hs): a chain in composeFilters order.
class Filter {
public:
    Filter(const VocoderParams &par, double freq_step) : par_(par)
    {
        check(voc_filter_create(par.handle(), freq_step, &f_), "filter", par.handle());
    }
    ~Filter() { voc_filter_destroy(f_); }
    Filter(const Filter &) = delete;
    Filter &operator=(const Filter &) = delete;

    Filter &append(int32_t kind, const std::vector<double> &params)
    {
        check(voc_filter_append(f_, kind, params.data(), (int32_t)params.size()), "filter", par_.handle());
        return *this;
    }
    Filter &amplify(double k) { return append(VOC_FILTER_AMPLIFY, {k}); }                              // Filter.hs:80-81
    Filter &lowpassBrickwall(double t) { return append(VOC_FILTER_LOWPASS_BRICKWALL, {t}); }           // Filter.hs:84-85
    Filter &highpassBrickwall(double t) { return append(VOC_FILTER_HIGHPASS_BRICKWALL, {t}); }         // Filter.hs:88-89
    Filter &bandpassBrickwall(double t, double u) { return append(VOC_FILTER_BANDPASS_BRICKWALL, {t, u}); }
    Filter &bandstopBrickwall(double t, double u) { return append(VOC_FILTER_BANDSTOP_BRICKWALL, {t, u}); }
    Filter &lowpassButterworth(double n, double t) { return append(VOC_FILTER_LOWPASS_BUTTERWORTH, {n, t}); }
    Filter &highpassButterworth(double n, double t) { return append(VOC_FILTER_HIGHPASS_BUTTERWORTH, {n, t}); }
    Filter &bandpassButterworth(double n, double t, double u) { return append(VOC_FILTER_BANDPASS_BUTTERWORTH, {n, t, u}); }
    Filter &bandstopButterworth(double n, double t, double u) { return append(VOC_FILTER_BANDSTOP_BUTTERWORTH, {n, t, u}); }
    Filter &pitchShiftInterpolate(double n) { return append(VOC_FILTER_PITCH_SHIFT_INTERPOLATE, {n}); }  // Filter.hs:128-129
    Filter &convolutionFilter(const std::vector<double> &ker) { return append(VOC_FILTER_CONVOLUTION, ker); }
    Filter &envelopeFilter(int ksize) { return append(VOC_FILTER_ENVELOPE, {(double)ksize}); }        // Filter.hs:151-152
    Filter &randomPhaseFilter(uint64_t seed) { return append(VOC_FILTER_RANDOM_PHASE, {(double)seed}); }

    // runs the chain on F frames of one channel, in place
    void apply(STFTFrames &stft) const
    {
        const size_t k = (size_t)par_.vocFreqFrameLength();
        check(voc_filter_apply(f_, stft.mag.data(), stft.ph_inc.data(), (int64_t)(stft.mag.size() / k), 1), "filter",
              par_.handle());
    }
    voc_filter *handle() const { return f_; }

private:
    const VocoderParams &par_;
    voc_filter *f_ = nullptr;
};

// Vocoder.Conduit.process fed chunk by chunk (Conduit.hs:46-53): push returns what sumFramesE has
// published so far, flush the rest.
class Stream {
public:
    Stream(const VocoderParams &par, int64_t channels, int64_t chunk, const Filter *filter = nullptr, bool dunai = false)
        : par_(par), channels_(channels)
    {
        check(voc_stream_create_mode(par.handle(), channels, chunk, filter ? filter->handle() : nullptr,
                                     dunai ? VOC_STREAM_DUNAI : VOC_STREAM_CONDUIT, &s_),
              "stream", par.handle());
    }
    ~Stream() { voc_stream_destroy(s_); }
    Stream(const Stream &) = delete;
    Stream &operator=(const Stream &) = delete;

    // playSpeed num/den between the stream's chain and `after` (Conduit/Filter.hs:122-136); before the first push
    void playSpeed(int64_t num, int64_t den, const Filter *after = nullptr)
    {
        check(voc_stream_set_play_speed(s_, num, den, after ? after->handle() : nullptr), "playSpeed", par_.handle());
    }

    // x: channels x n planar; returns channels x m planar
    std::vector<double> push(const std::vector<double> &x)
    {
        const int64_t n = channels_ ? (int64_t)x.size() / channels_ : 0;
        const int64_t m = voc_stream_push_length(s_, n);
        std::vector<double> y((size_t)(channels_ * (m > 0 ? m : 0)));
        int64_t got = 0;
        check(voc_stream_push(s_, x.data(), n, n, y.data(), m, m, &got), "stream push", par_.handle());
        return y;
    }
    std::vector<double> flush()
    {
        const int64_t m = voc_stream_flush_length(s_);
        std::vector<double> y((size_t)(channels_ * (m > 0 ? m : 0)));
        int64_t got = 0;
        check(voc_stream_flush(s_, y.data(), m, m, &got), "stream flush", par_.handle());
        return y;
    }

private:
    const VocoderParams &par_;
    int64_t channels_;
    voc_stream *s_ = nullptr;
};

}  // namespace vocoder
