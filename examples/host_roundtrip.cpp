// host_roundtrip.cpp -- the C++ host mirror (include/vocoder.hpp) in use: analysis + synthesis
// of a 440 Hz tone, stage by stage and fused, and the identity check of the round trip.
//   g++ -std=c++17 -Iinclude examples/host_roundtrip.cpp -Lvocoder_amd -lvocoder_hip \
//       -Wl,-rpath,$PWD/vocoder_amd -Wl,-rpath,/opt/rocm/lib -o examples/host_roundtrip
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <string>
#include <vector>

#include "vocoder.hpp"

// With a directory as its argument the program also leaves what it computed there -- the input, the fused
// output, the stage-by-stage output, the filtered output, the window -- as raw doubles, for a checker that
// holds them against a CPU restatement of the reference (tests/test_gpu_cpp_host.py).
static void dump(const char *dir, const char *name, const std::vector<double> &v)
{
    if (!dir) return;
    const std::string path = std::string(dir) + "/" + name + ".f64";
    if (FILE *f = std::fopen(path.c_str(), "wb")) {
        std::fwrite(v.data(), sizeof(double), v.size(), f);
        std::fclose(f);
    }
}

int main(int argc, char **argv)
{
    const char *out_dir = argc > 1 ? argv[1] : nullptr;
    const int n = 2048, h = 512;
    vocoder::Window win(n);
    for (int k = 0; k < n; ++k) win[k] = 0.5 * (1 - std::cos(2 * M_PI * k / n));  // periodic Hann: exact overlap-add
    vocoder::VocoderParams par(n, h, win);
    const int64_t len = 48000;
    std::vector<double> x(len);
    for (int64_t i = 0; i < len; ++i) x[i] = 0.5 * std::sin(2 * M_PI * 440.0 * i / 48000.0);

    auto r = vocoder::process(par, x, 1, h, par.zeroPhase(), par.zeroPhase());
    double err = 0;
    for (int64_t i = n; i < len - 2 * n; ++i) err = std::fmax(err, std::fabs(r.y[i] - x[i]));
    std::printf("fused round trip: %zu samples out, %zu left over, max |y - x| = %.3g\n", r.y.size(), r.leftover.size(), err);

    // the same through the stage functions, as vocoder-dunai drives them (Dunai.hs:30,34)
    const int64_t f = voc_num_frames(par.handle(), len);
    std::vector<double> frames((size_t)f * n);
    vocoder::check(voc_frames_of(par.handle(), x.data(), len, len, 1, frames.data()), "framesOf");
    auto [aph, stft] = vocoder::analysisStage(par, par.zeroPhase(), frames);
    for (auto &m : stft.mag) m *= par.volumeCoeff();  // volumeFix (Conduit.hs:34-35)
    auto [sph, out] = vocoder::synthesisStage(par, par.zeroPhase(), stft);
    std::vector<double> y((size_t)voc_ola_length(par.handle(), f, h));
    vocoder::check(voc_sum_frames(par.handle(), out.data(), f, 1, h, y.data(), (int64_t)y.size()), "sumFrames");
    double diff = 0;
    for (size_t i = 0; i < y.size(); ++i) diff = std::fmax(diff, std::fabs(y[i] - r.y[i]));
    std::printf("stage by stage vs fused: max difference %.3g\n", diff);

    // chunk by chunk through a stream, as a conduit would feed it (Conduit.hs:46-53): same bits as the fused call
    vocoder::Stream st(par, 1, h);
    std::vector<double> ys;
    for (int64_t pos = 0; pos < len; pos += 7001) {
        const int64_t m = std::min<int64_t>(7001, len - pos);
        auto part = st.push(std::vector<double>(x.begin() + pos, x.begin() + pos + m));
        ys.insert(ys.end(), part.begin(), part.end());
    }
    auto rest = st.flush();
    ys.insert(ys.end(), rest.begin(), rest.end());
    const bool same = ys == r.y;
    std::printf("streamed in pieces of 7001 samples: %s\n", same ? "identical to the fused call" : "DIFFERENT");

    // a filter chain on the STFT (vocoder-file --lowpassBrickwall 1000 --amplify 0.5): the 440 Hz tone passes, halved
    vocoder::Filter flt(par, 48000.0 / n);
    flt.lowpassBrickwall(1000.0).amplify(0.5);
    flt.apply(stft);
    auto [sph2, out2] = vocoder::synthesisStage(par, par.zeroPhase(), stft);
    std::vector<double> y2(y.size());
    vocoder::check(voc_sum_frames(par.handle(), out2.data(), f, 1, h, y2.data(), (int64_t)y2.size()), "sumFrames");
    double ferr = 0;
    for (int64_t i = n; i < len - 2 * n; ++i) ferr = std::fmax(ferr, std::fabs(y2[i] - 0.5 * x[i]));
    std::printf("lowpass 1000 Hz + amplify 0.5: max |y - x/2| = %.3g\n", ferr);

    // the reference's frame-level functions, with its own argument lists (Vocoder.hs:143-222): one frame by hand
    const vocoder::Frame fr0(x.begin(), x.begin() + n), fr1(x.begin() + h, x.begin() + h + n);
    auto [ph0, st0] = vocoder::analysisStep(h, n, par.zeroPhase(), vocoder::doFFT(par, fr0));
    auto [ph1, st1] = vocoder::analysisStep(h, n, ph0, vocoder::doFFT(par, fr1));
    double step_diff = 0;  // against the stage call above (before its volumeFix and filter: recompute)
    auto [aph2, stft2] = vocoder::analysisStage(par, par.zeroPhase(), std::vector<double>(frames.begin(), frames.begin() + 2 * n));
    double top = 0;
    for (double m : st1.first) top = std::fmax(top, m);
    for (size_t k = 0; k < st1.first.size(); ++k) {
        step_diff = std::fmax(step_diff, std::fabs(st1.first[k] - stft2.mag[st1.first.size() + k]));
        // (the phase of a bin in the leakage floor depends on the last bits of the transform, and the two paths
        //  factor their transforms differently)
        if (st1.first[k] > 1e-3 * top)
            step_diff = std::fmax(step_diff, std::fabs(st1.second[k] - stft2.ph_inc[st1.first.size() + k]));
    }
    auto [nph, spec1] = vocoder::synthesisStep(h, par.zeroPhase(), st0);
    const vocoder::Frame back = vocoder::doIFFT(par, spec1);
    const auto polar = vocoder::frameFromComplex(vocoder::frameToComplex(st0));
    double polar_diff = 0;
    for (size_t k = 0; k < polar.first.size(); ++k) polar_diff = std::fmax(polar_diff, std::fabs(polar.first[k] - st0.first[k]));
    const auto doubled = vocoder::addFrames(st0, st0);
    double add_diff = 0;
    for (size_t k = 0; k < doubled.first.size(); ++k) add_diff = std::fmax(add_diff, std::fabs(doubled.first[k] - 2 * st0.first[k]));
    std::printf("step functions vs stage call: %.3g; polar round trip: %.3g; addFrames f f = 2 |f|: %.3g; frame of %zu samples back\n",
                step_diff, polar_diff, add_diff, back.size());

    // playSpeed 2 inside a stream: twice the frames, twice the samples
    vocoder::Stream fast(par, 1, h);
    fast.playSpeed(2, 1);
    auto y2x = fast.push(x);
    auto y2r = fast.flush();
    std::printf("playSpeed 2 in a stream: %zu samples out for %lld in\n", y2x.size() + y2r.size(), (long long)len);
    const bool speed_ok = y2x.size() + y2r.size() > (size_t)(2 * (len - 2 * n));
    dump(out_dir, "window", std::vector<double>(win.begin(), win.end()));
    dump(out_dir, "x", x);
    dump(out_dir, "y_fused", r.y);
    dump(out_dir, "y_stages", y);
    dump(out_dir, "y_lowpass1000_amplify0.5", y2);
    y2x.insert(y2x.end(), y2r.begin(), y2r.end());
    dump(out_dir, "y_playspeed2", y2x);
    return (err < 1e-9 && diff < 1e-12 && same && ferr < 1e-3 && step_diff < 1e-12 && polar_diff < 1e-12 && add_diff < 1e-12 &&
            speed_ok) ? 0 : 1;
}
