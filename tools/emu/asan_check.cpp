// asan_check.cpp -- DEVELOPMENT HARNESS: the C-ABI library compiled for the host (hip_emu.h) into one
// executable under AddressSanitizer / UBSan, driven through a set of scenarios that cover every
// kernel family (radix kernels, fast kernels, chirp-z in LDS and through device memory, filters, streaming with
// playSpeed, the context-free functions, the single-precision forms).
//   make -C tools/emu asan && tools/emu/asan_check
#include "../../vocoder_amd/csrc/vocoder_hip.cpp"

#include <cstdio>
#include <random>

static std::vector<double> hann(int n)
{
    std::vector<double> w(n);
    for (int i = 0; i < n; ++i) w[i] = 0.5 * (1 - std::cos(2 * M_PI * i / n));
    return w;
}

#define OK(call)                                                                  \
    do {                                                                          \
        int32_t st_ = (call);                                                     \
        if (st_ != 0) { std::printf("FAILED %s -> %d\n", #call, (int)st_); return 1; } \
    } while (0)

static int scenario(int n, int w, int h, int channels, int frames, int chunk)
{
    std::mt19937_64 rng(n * 31 + w);
    std::normal_distribution<double> g;
    voc_ctx *ctx = nullptr;
    const auto win = hann(w);
    OK(voc_ctx_create(n, h, win.data(), w, 0, VOC_PRECISION_F64, &ctx));
    const int64_t len = w + (int64_t)h * (frames - 1) + 3, k = voc_freq_frame_length(ctx);
    std::vector<double> x((size_t)channels * len);
    for (auto &v : x) v = g(rng);
    const int64_t f = voc_num_frames(ctx, len), olen = voc_ola_length(ctx, f, chunk), keep = voc_leftover_length(ctx, len);
    std::vector<double> y((size_t)channels * olen), left((size_t)channels * (keep ? keep : 1)), a((size_t)channels * k, 0.0),
        s((size_t)channels * k, 0.0);
    OK(voc_roundtrip(ctx, x.data(), len, len, channels, chunk, y.data(), olen, a.data(), s.data(), left.data(), keep));

    // stage by stage + filters
    std::vector<double> fr((size_t)channels * f * w), mag((size_t)channels * f * k), inc(mag.size()), ph((size_t)channels * k, 0.0);
    OK(voc_frames_of(ctx, x.data(), len, len, channels, fr.data()));
    OK(voc_analysis_stage(ctx, ph.data(), fr.data(), mag.data(), inc.data(), channels, f));
    voc_filter *flt = nullptr;
    OK(voc_filter_create(ctx, 48000.0 / n, &flt));
    const double q1[] = {1.5}, q2[] = {2.0, 3000.0}, q3[] = {5.0}, q4[] = {11.0};
    OK(voc_filter_append(flt, VOC_FILTER_PITCH_SHIFT_INTERPOLATE, q1, 1));
    OK(voc_filter_append(flt, VOC_FILTER_LOWPASS_BUTTERWORTH, q2, 2));
    OK(voc_filter_append(flt, VOC_FILTER_ENVELOPE, q3, 1));
    OK(voc_filter_append(flt, VOC_FILTER_RANDOM_PHASE, q4, 1));
    OK(voc_filter_apply(flt, mag.data(), inc.data(), f, channels));
    std::fill(ph.begin(), ph.end(), 0.0);
    OK(voc_synthesis_stage(ctx, ph.data(), mag.data(), inc.data(), fr.data(), channels, f));
    OK(voc_sum_frames(ctx, fr.data(), f, channels, chunk, y.data(), olen));

    // streaming in uneven pieces, with the filter, then flush
    voc_stream *st = nullptr;
    OK(voc_stream_create(ctx, channels, chunk, flt, &st));
    std::vector<double> piece, out;
    for (int64_t pos = 0; pos < len;) {
        const int64_t m = std::min<int64_t>(len - pos, 1 + (int64_t)(rng() % (3 * h + 1)));
        piece.assign((size_t)channels * m, 0.0);
        for (int c = 0; c < channels; ++c) std::copy_n(&x[(size_t)c * len + pos], m, &piece[(size_t)c * m]);
        const int64_t cap = voc_stream_push_length(st, m);
        out.assign((size_t)channels * (cap > 0 ? cap : 1), 0.0);
        int64_t got = 0;
        OK(voc_stream_push(st, piece.data(), m, m, out.data(), cap, cap, &got));
        pos += m;
    }
    const int64_t cap = voc_stream_flush_length(st);
    out.assign((size_t)channels * (cap > 0 ? cap : 1), 0.0);
    int64_t got = 0;
    OK(voc_stream_flush(st, out.data(), cap, cap, &got));
    OK(voc_stream_destroy(st));

    // playSpeed 3/2 between two chains, pushed in pieces
    voc_filter *after = nullptr;
    OK(voc_filter_create(ctx, 48000.0 / n, &after));
    const double q5[] = {0.5};
    OK(voc_filter_append(after, VOC_FILTER_AMPLIFY, q5, 1));
    OK(voc_stream_create(ctx, channels, chunk, flt, &st));
    OK(voc_stream_set_play_speed(st, 3, 2, after));
    for (int64_t pos = 0; pos < len;) {
        const int64_t m = std::min<int64_t>(len - pos, 1 + (int64_t)(rng() % (3 * h + 1)));
        piece.assign((size_t)channels * m, 0.0);
        for (int c = 0; c < channels; ++c) std::copy_n(&x[(size_t)c * len + pos], m, &piece[(size_t)c * m]);
        const int64_t cap2 = voc_stream_push_length(st, m);
        out.assign((size_t)channels * (cap2 > 0 ? cap2 : 1), 0.0);
        OK(voc_stream_push(st, piece.data(), m, m, out.data(), cap2, cap2, &got));
        pos += m;
    }
    const int64_t cap3 = voc_stream_flush_length(st);
    out.assign((size_t)channels * (cap3 > 0 ? cap3 : 1), 0.0);
    OK(voc_stream_flush(st, out.data(), cap3, cap3, &got));
    OK(voc_stream_destroy(st));
    OK(voc_filter_destroy(after));

    // the FFT layer and the functions with the reference's own argument lists
    std::vector<double> spec((size_t)2 * k), frame0(fr.begin(), fr.begin() + w), m1((size_t)k), p1((size_t)k), i1((size_t)k),
        zero((size_t)k, 0.0), spec2((size_t)2 * k), back((size_t)w);
    OK(voc_do_fft(ctx, frame0.data(), spec.data(), 1));
    OK(voc_analysis_step_hn(h, n, k, zero.data(), spec.data(), p1.data(), m1.data(), i1.data(), 1));
    OK(voc_synthesis_step_h(h, k, zero.data(), m1.data(), i1.data(), p1.data(), spec2.data(), 1));
    OK(voc_do_ifft(ctx, spec2.data(), back.data(), 1));
    OK(voc_frame_from_complex(spec.data(), m1.data(), p1.data(), k));
    OK(voc_frame_to_complex(m1.data(), p1.data(), spec2.data(), k));
    OK(voc_add_frames_n(m1.data(), i1.data(), m1.data(), i1.data(), m1.data(), i1.data(), k));
    OK(voc_filter_destroy(flt));
    OK(voc_ctx_destroy(ctx));
    std::printf("ok  N=%d W=%d H=%d C=%d F=%lld\n", n, w, h, channels, (long long)f);
    return 0;
}

static int scenario_f32(int n, int h, int channels, int frames)
{
    voc_ctx *ctx = nullptr;
    const auto win = hann(n);
    OK(voc_ctx_create(n, h, win.data(), n, 0, VOC_PRECISION_F32, &ctx));
    const int64_t len = n + (int64_t)h * (frames - 1) + 1, k = voc_freq_frame_length(ctx);
    std::vector<float> x((size_t)channels * len, 0.25f);
    const int64_t f = voc_num_frames(ctx, len), olen = voc_ola_length(ctx, f, h), keep = voc_leftover_length(ctx, len);
    std::vector<float> y((size_t)channels * olen), left((size_t)channels * (keep ? keep : 1)), a((size_t)channels * k, 0.f),
        s((size_t)channels * k, 0.f);
    OK(voc_roundtrip_f32(ctx, x.data(), len, len, channels, h, y.data(), olen, a.data(), s.data(), left.data(), keep));
    // stage level, a filter chain on the float STFT, and a stream
    std::vector<float> fr((size_t)channels * f * n, 0.125f), mag((size_t)channels * f * k), inc(mag.size()), ph((size_t)channels * k, 0.f);
    OK(voc_analysis_stage_f32(ctx, ph.data(), fr.data(), mag.data(), inc.data(), channels, f));
    voc_filter *flt = nullptr;
    OK(voc_filter_create(ctx, 48000.0 / n, &flt));
    const double q1[] = {1.5}, q2[] = {7.0};
    OK(voc_filter_append(flt, VOC_FILTER_PITCH_SHIFT_INTERPOLATE, q1, 1));
    OK(voc_filter_append(flt, VOC_FILTER_RANDOM_PHASE, q2, 1));
    OK(voc_filter_apply_f32(flt, mag.data(), inc.data(), f, channels));
    std::fill(ph.begin(), ph.end(), 0.f);
    OK(voc_synthesis_stage_f32(ctx, ph.data(), mag.data(), inc.data(), fr.data(), channels, f));
    voc_stream *st = nullptr;
    OK(voc_stream_create(ctx, channels, h, flt, &st));
    OK(voc_stream_set_play_speed(st, 2, 1, nullptr));
    int64_t got = 0;
    for (int64_t pos = 0; pos < len; pos += 3 * h + 1) {
        const int64_t m = std::min<int64_t>(len - pos, 3 * h + 1);
        std::vector<float> piece((size_t)channels * m, 0.25f);
        const int64_t cap = voc_stream_push_length(st, m);
        std::vector<float> out((size_t)channels * (cap > 0 ? cap : 1));
        OK(voc_stream_push_f32(st, piece.data(), m, m, out.data(), cap, cap, &got));
    }
    const int64_t cap = voc_stream_flush_length(st);
    std::vector<float> out((size_t)channels * (cap > 0 ? cap : 1));
    OK(voc_stream_flush_f32(st, out.data(), cap, cap, &got));
    OK(voc_stream_get_phases_f32(st, a.data(), s.data()));
    OK(voc_stream_destroy(st));
    OK(voc_filter_destroy(flt));
    // FFT-compatible layer, step level and addFrames on float data
    {
        std::vector<float> fr1((size_t)n, 0.5f), spec((size_t)2 * k), back((size_t)n), p0((size_t)k, 0.f), p1((size_t)k),
            m1((size_t)k), i1((size_t)k), spec2((size_t)2 * k);
        fr1[3] = -1.0f;
        OK(voc_r2c_f32(ctx, fr1.data(), spec.data(), 1));
        OK(voc_c2r_f32(ctx, spec.data(), back.data(), 1));
        OK(voc_do_fft_f32(ctx, fr1.data(), spec.data(), 1));
        OK(voc_analysis_step_f32(ctx, p0.data(), spec.data(), p1.data(), m1.data(), i1.data(), 1));
        OK(voc_synthesis_step_f32(ctx, p0.data(), m1.data(), i1.data(), p1.data(), spec2.data(), 1));
        OK(voc_do_ifft_f32(ctx, spec2.data(), back.data(), 1));
        OK(voc_add_frames_f32(ctx, m1.data(), i1.data(), m1.data(), i1.data(), m1.data(), i1.data(), 1));
    }
    OK(voc_ctx_destroy(ctx));
    std::printf("ok  f32 N=%d\n", n);
    return 0;
}

// Entry points that take "device" pointers (plain host memory in this build): frame maps with
// entries outside the array, 16-bit PCM, addFrames, the split analysis / synthesis pair.
static int scenario_dev(int n, int h, int channels, int frames)
{
    voc_ctx *ctx = nullptr;
    const auto win = hann(n);
    OK(voc_ctx_create(n, h, win.data(), n, 0, VOC_PRECISION_F64, &ctx));
    const int64_t k = voc_freq_frame_length(ctx), ldk = (k + 1) & ~(int64_t)1;
    const int64_t len = n + (int64_t)h * (frames - 1);
    std::vector<int16_t> pcm((size_t)channels * len);
    for (size_t i = 0; i < pcm.size(); ++i) pcm[i] = (int16_t)((i * 2654435761u) >> 16);
    std::vector<double> x((size_t)channels * len);
    OK(voc_pcm16_to_planar_dev(ctx, pcm.data(), len, channels, x.data(), len, nullptr));
    std::vector<double> mag((size_t)channels * frames * ldk), inc(mag.size()), ph((size_t)channels * k, 0.0);
    OK(voc_analysis_signal_dev(ctx, x.data(), len, frames, channels, ph.data(), mag.data(), inc.data(), ldk, 1, nullptr));
    OK(voc_add_frames_dev(ctx, mag.data(), inc.data(), mag.data(), inc.data(), mag.data(), inc.data(), (int64_t)mag.size(),
                          nullptr));
    const int64_t fo = 2 * frames + 3;                       // more output frames than input rows
    std::vector<int64_t> rows((size_t)fo);
    for (int64_t j = 0; j < fo; ++j) rows[(size_t)j] = j / 2 - 1;   // -1 at the front, past the end at the back
    const int64_t ylen = (fo - 1) * h + n;
    std::vector<double> y((size_t)channels * ylen), sph((size_t)channels * k, 0.0);
    OK(voc_synthesis_mapped_dev(ctx, mag.data(), frames, rows.data(), inc.data(), frames, rows.data(), ldk, fo, channels,
                                sph.data(), y.data(), ylen, ylen, nullptr));
    std::vector<int16_t> out((size_t)channels * ylen);
    OK(voc_planar_to_pcm16_dev(ctx, y.data(), ylen, ylen, channels, 0, out.data(), nullptr));
    OK(voc_planar_to_pcm16_dev(ctx, y.data(), ylen, ylen, channels, 1, out.data(), nullptr));
    OK(voc_ctx_destroy(ctx));
    std::printf("ok  dev N=%d C=%d\n", n, channels);
    return 0;
}

int main()
{
    int bad = 0;
    bad |= scenario(64, 64, 16, 2, 9, 16);
    bad |= scenario(64, 40, 24, 1, 7, 50);      // padded frame, hop not a divisor
    bad |= scenario(1024, 1024, 256, 2, 11, 256);  // fast kernels, register overlap-add
    bad |= scenario(1024, 1000, 250, 1, 6, 100);   // fast kernels, scalar loads
    bad |= scenario(100, 81, 20, 2, 8, 20);        // chirp-z
    bad |= scenario(7, 7, 2, 1, 12, 3);
    bad |= scenario_dev(64, 16, 2, 9);
    bad |= scenario_dev(1024, 256, 3, 7);          // specialised kernels through the frame maps
    bad |= scenario_dev(2048, 512, 2, 6);          // registers for the accumulated phases, rows loaded at use
    bad |= scenario(2048, 2048, 512, 1, 6, 512);
    bad |= scenario_dev(100, 20, 1, 6);            // chirp-z
    bad |= scenario(4100, 3000, 1000, 1, 3, 1000); // chirp-z through device memory (two interleaved 8192-point transforms)
    bad |= scenario_f32(64, 16, 2, 9);
    bad |= scenario_f32(1024, 256, 1, 7);
    bad |= scenario(4096, 4096, 1024, 1, 6, 1024);   // sixteen points per thread (analysis), sliding sample window
    bad |= scenario(4096, 4000, 1000, 1, 4, 1000);   // the same kernels with scalar loads and a padded frame
    bad |= scenario_f32(4096, 1024, 2, 6);           // sixteen points per thread on float, two bins at a time
    bad |= scenario(16400, 16400, 4100, 1, 2, 4100); // chirp-z over eight interleaved transforms; rows of 8201 bins: the filter build for long rows
    std::printf(bad ? "FAILURES\n" : "all scenarios ran\n");
    return bad;
}
