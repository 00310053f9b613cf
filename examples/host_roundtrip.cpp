// host_roundtrip.cpp -- the C++ host mirror (include/vocoder.hpp) in use: analysis + synthesis
// of a 440 Hz tone, stage by stage and fused, and the identity check of the round trip.
//   g++ -std=c++17 -Iinclude examples/host_roundtrip.cpp -Lvocoder_amd -lvocoder_hip \
//       -Wl,-rpath,$PWD/vocoder_amd -Wl,-rpath,/opt/rocm/lib -o examples/host_roundtrip
#include <algorithm>
#include <cmath>
#include <cstdio>

#include "vocoder.hpp"

int main()
{
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
    return (err < 1e-9 && diff < 1e-12 && same && ferr < 1e-3) ? 0 : 1;
}
