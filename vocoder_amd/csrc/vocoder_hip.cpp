// vocoder_hip.cpp -- C-ABI implementation (include/vocoder_hip.h) over the gfx950 kernels.
// Built by hipcc (-x hip --offload-arch=gfx950 -ffp-contract=off) into libvocoder_hip.so.
#ifndef VOC_EMU
#include <hip/hip_runtime.h>
#endif

#include <cmath>
#include <complex>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <optional>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/vocoder_hip.h"
#include "voc_kernels.h"
#include "voc_kernels_fast.h"
#include "voc_kernels_fast16.h"
#include "voc_filters.h"
#include "voc_kernels_bluestein.h"
#include "voc_kernels_bluestein_big.h"

#define VOC_EXPORT extern "C" __attribute__((visibility("default")))

// Scratch of voc_roundtrip_static_dev: device memory that exists from the moment the code object is
// loaded, one copy per GPU, held until the process ends.  In host code the array's name is only a
// handle for hipGetSymbolAddress; the device address is looked up once per device (voc_ctx_create)
// and kept in the context.
constexpr size_t kStaticScratchBytes = (size_t)VOC_STATIC_SCRATCH_BYTES;
#ifndef VOC_EMU
static __device__ __align__(256) unsigned char voc_static_scratch[kStaticScratchBytes];
#else
alignas(256) static unsigned char voc_static_scratch[kStaticScratchBytes];
#endif

// Grow-only staging of the host-pointer entry points: one device buffer that holds the device images of
// a call's host arrays, and one pinned host buffer through which small arrays travel (a copy from
// pinned memory is a plain DMA; from pageable memory the runtime stages it and blocks).  Held by the
// context (and one more for the entry points that take no context), so that a caller who crosses the
// boundary once per frame -- Vocoder.Conduit.analysis / synthesis do (Conduit.hs:39,43) -- pays no
// allocation per call.
struct Staging {
    char *dev = nullptr;
    size_t dev_bytes = 0;
    char *pin = nullptr;
    size_t pin_bytes = 0;
    void release()
    {
        if (dev) (void)hipFree(dev);
        if (pin) (void)hipHostFree(pin);
        dev = pin = nullptr;
        dev_bytes = pin_bytes = 0;
    }
};

struct voc_ctx {
    int N = 0, W = 0, H = 0, K = 0, M = 0, log2m = 0, device = 0;
    std::vector<double> window;
    double vol = 0.0;
    double *d_window = nullptr;
    double2 *d_twP = nullptr, *d_twS = nullptr, *d_cs64 = nullptr;
    voc::Params params{};
    int precision = VOC_PRECISION_F64;
    // frame lengths that are not powers of two: chirp-z transform over an L-point FFT (M and log2m
    // then describe L, not N/2)
    bool bluestein = false;
    double2 *d_chirp = nullptr, *d_bfft = nullptr;
    voc::BlueTab blue{};
    // ... and above 4096 the chirp-z transform runs through device memory (M = 8192: the transform of each of
    // its L1 interleaved sequences); big_ws holds the frames in flight, grow-only
    bool blue_big = false;
    double2 *d_wl = nullptr;
    voc::BlueBigTab big{};
    void *big_ws = nullptr;
    size_t big_ws_bytes = 0;
    // single-precision contexts: float copies of the tables (the double ones stay for the accessors)
    float *d_window_f = nullptr;
    float2 *d_twP_f = nullptr, *d_twS_f = nullptr;
    voc::ParamsT<float> params_f{};
    void *ws = nullptr;
    size_t ws_bytes = 0;
    Staging staging;  // host-pointer entry points
    // The library-owned workspace is shared by every call on the context, and _dev calls return
    // after enqueueing: ws_event marks the end of the last enqueued use, and a call on another HIP
    // stream waits for it before it touches the workspace.
    hipEvent_t ws_event = nullptr;
    hipStream_t ws_stream = nullptr;
    bool ws_busy = false;
    void *static_scratch = nullptr;  // this device's copy of voc_static_scratch (allocation-free entry point)
    int ana_tile_override = 0, syn_tile_override = 0, syn_slots_override = 0;
    int cus = 256;                       // compute units and LDS bytes per compute unit of the context's device
    long long lds_per_cu = 160 * 1024;   // (read from the device's properties when the context is created)
    bool force_generic = false;  // VOC_FORCE_GENERIC=1: never take the specialised kernels
    // sixteen instead of eight points per thread at N = 4096 / 8192 (voc_kernels_fast16.h): analysis by default
    // (VOC_FAST16_ANA=0 turns it off), synthesis on request (VOC_FAST16_SYN=1: measured slower, kept for comparison)
    bool fast16_ana = true;
    int fast16_syn = -1;  // -1: only where there is no eight-point kernel (N = 16384)
    // the same on single-precision contexts (VOC_FAST16_F32_ANA, VOC_FAST16_F32_SYN)
    bool fast16_ana_f32 = VOC_FAST16_F32_ANA_DEFAULT != 0;
    int fast16_syn_f32 = VOC_FAST16_F32_SYN_DEFAULT;
#ifndef VOC_EMU
    bool profiling = false;
    struct Timed { int kind; hipEvent_t start, stop; };
    std::vector<Timed> timed;  // one pair of events per kernel launch while profiling
#endif
    std::mutex mu;
    std::string hip_err;
};

namespace {

struct HipFail {
    hipError_t e;
    const char *what;
};

struct StatusFail {  // a plain status code raised below the C boundary
    int32_t status;
};

#define HIPCHK(call)                          \
    do {                                      \
        hipError_t e_ = (call);               \
        if (e_ != hipSuccess) throw HipFail{e_, #call}; \
    } while (0)

// Runs body() and maps failures to status codes; nothing propagates across the C boundary.
template <class Body>
int32_t guarded(voc_ctx *ctx, Body body)
{
    try {
        return body();
    } catch (const HipFail &f) {
        if (ctx) ctx->hip_err = std::string(f.what) + ": " + hipGetErrorString(f.e);
        return f.e == hipErrorOutOfMemory ? VOC_ERR_NOMEM : VOC_ERR_HIP;
    } catch (const StatusFail &f) {
        return f.status;
    } catch (const std::bad_alloc &) {
        return VOC_ERR_NOMEM;
    } catch (...) {
        return VOC_ERR_INVALID;
    }
}

// The device images of the host arrays of one host-pointer call, carved out of a Staging.  Usage: add()
// every array (inputs first, then arrays that go both ways, outputs last), commit() once (grows the
// staging if needed), upload() the inputs, uploads_done(), launch, download() the outputs, finish() (one
// stream synchronisation, then the pinned images are copied to the caller).
//
// Arrays up to kPinPerArray travel through the pinned buffer, whose layout the front of the device
// buffer mirrors: neighbouring arrays that move in the same direction go as ONE copy, so a one-frame
// stage call costs one copy in, its kernels, one copy out and one synchronisation.  Larger arrays are
// copied directly between the caller's memory and the device.
class HostCall {
public:
    HostCall(Staging &s, hipStream_t st) : s_(s), st_(st) {}
    HostCall(const HostCall &) = delete;
    HostCall &operator=(const HostCall &) = delete;
    int add(size_t bytes)
    {
        Buf b{};
        b.bytes = bytes;
        const size_t padded = (bytes + 255) / 256 * 256;
        b.pinned = bytes <= kPinPerArray && pin_need_ + padded <= kPinTotal;
        if (b.pinned) {
            b.off = pin_need_;
            pin_need_ += padded;
        } else {
            b.off = direct_need_;
            direct_need_ += padded + 256;
        }
        bufs_.push_back(b);
        return (int)bufs_.size() - 1;
    }
    void commit()
    {
        const size_t dev_need = pin_need_ + direct_need_ + 256;
        if (dev_need > s_.dev_bytes) {
            HIPCHK(hipDeviceSynchronize());
            if (s_.dev) (void)hipFree(s_.dev);
            s_.dev = nullptr;
            s_.dev_bytes = 0;
            HIPCHK(hipMalloc((void **)&s_.dev, dev_need));
            s_.dev_bytes = dev_need;
        }
        if (pin_need_ > s_.pin_bytes) {
            HIPCHK(hipDeviceSynchronize());
            if (s_.pin) (void)hipHostFree(s_.pin);
            s_.pin = nullptr;
            s_.pin_bytes = 0;
            HIPCHK(hipHostMalloc((void **)&s_.pin, pin_need_, 0));
            s_.pin_bytes = pin_need_;
        }
        for (Buf &b : bufs_)
            if (!b.pinned) b.off += pin_need_;  // the directly copied arrays follow the mirrored part
    }
    template <class T = double> T *dev(int i) const { return reinterpret_cast<T *>(s_.dev + bufs_[(size_t)i].off); }
    void upload(int i, const void *host)
    {
        Buf &b = bufs_[(size_t)i];
        if (b.bytes == 0) return;
        if (b.pinned) {
            std::memcpy(s_.pin + b.off, host, b.bytes);
            b.moving = true;
            pending_uploads_ = true;
        } else {
            HIPCHK(hipMemcpyAsync(s_.dev + b.off, host, b.bytes, hipMemcpyHostToDevice, st_));
        }
    }
    // rows of `width` bytes, `host_pitch` bytes apart in the caller's array, packed on the device
    void upload_rows(int i, const void *host, size_t host_pitch, size_t width, size_t rows)
    {
        if (width == 0 || rows == 0) return;
        if (bufs_[(size_t)i].pinned) {
            for (size_t r = 0; r < rows; ++r)
                std::memcpy(s_.pin + bufs_[(size_t)i].off + r * width, (const char *)host + r * host_pitch, width);
            bufs_[(size_t)i].moving = true;
            pending_uploads_ = true;
        } else {
            HIPCHK(hipMemcpy2DAsync(s_.dev + bufs_[(size_t)i].off, width, host, host_pitch, width, rows, hipMemcpyHostToDevice, st_));
        }
    }
    void uploads_done()  // before the first launch
    {
        move_runs(true);
        pending_uploads_ = false;
    }
    void download(int i, void *host)
    {
        if (pending_uploads_) throw StatusFail{VOC_ERR_INVALID};
        Buf &b = bufs_[(size_t)i];
        if (b.bytes == 0) return;
        if (b.pinned) {
            b.moving = true;
            b.deliver_to = host;
            b.deliver_pitch = b.deliver_width = b.bytes;
        } else {
            HIPCHK(hipMemcpyAsync(host, s_.dev + b.off, b.bytes, hipMemcpyDeviceToHost, st_));
        }
    }
    void download_rows(int i, void *host, size_t host_pitch, size_t width, size_t rows)
    {
        if (pending_uploads_) throw StatusFail{VOC_ERR_INVALID};
        if (width == 0 || rows == 0) return;
        Buf &b = bufs_[(size_t)i];
        if (b.pinned) {
            b.moving = true;
            b.deliver_to = host;
            b.deliver_pitch = host_pitch;
            b.deliver_width = width;
        } else {
            HIPCHK(hipMemcpy2DAsync(host, host_pitch, s_.dev + b.off, width, width, rows, hipMemcpyDeviceToHost, st_));
        }
    }
    void finish()
    {
        if (pending_uploads_) throw StatusFail{VOC_ERR_INVALID};
        move_runs(false);
        HIPCHK(hipStreamSynchronize(st_));
        for (Buf &b : bufs_)
            if (b.deliver_to) {
                const char *src = s_.pin + b.off;
                for (size_t done = 0, r = 0; done + b.deliver_width <= b.bytes; done += b.deliver_width, ++r)
                    std::memcpy((char *)b.deliver_to + r * b.deliver_pitch, src + done, b.deliver_width);
                b.deliver_to = nullptr;
            }
    }

private:
    static constexpr size_t kPinPerArray = (size_t)4 << 20, kPinTotal = (size_t)32 << 20;
    struct Buf {
        size_t bytes, off;  // off: into the device buffer, and into the pinned one for pinned arrays
        bool pinned, moving;
        void *deliver_to;
        size_t deliver_pitch, deliver_width;
    };
    // one copy per run of neighbouring pinned arrays marked `moving`
    void move_runs(bool to_device)
    {
        size_t lo = 0, hi = 0;
        auto flush = [&]() {
            if (hi > lo) {
                if (to_device)
                    HIPCHK(hipMemcpyAsync(s_.dev + lo, s_.pin + lo, hi - lo, hipMemcpyHostToDevice, st_));
                else
                    HIPCHK(hipMemcpyAsync(s_.pin + lo, s_.dev + lo, hi - lo, hipMemcpyDeviceToHost, st_));
            }
            lo = hi = 0;
        };
        for (Buf &b : bufs_) {
            if (!b.pinned) continue;
            if (b.moving) {
                if (hi == lo) lo = b.off;
                hi = b.off + b.bytes;
                b.moving = false;
            } else if (b.bytes > 0) {
                flush();
            }
        }
        flush();
    }
    Staging &s_;
    hipStream_t st_;
    std::vector<Buf> bufs_;
    size_t pin_need_ = 0, direct_need_ = 0;
    bool pending_uploads_ = false;
};

enum { KIND_ANALYSIS = 0, KIND_CARRY = 1, KIND_SYNTHESIS = 2, KIND_OTHER = 3, KIND_COUNT = 4 };

// While profiling is on, brackets one kernel launch with HIP events on the launch stream.
struct LaunchTimer {
#ifndef VOC_EMU
    voc_ctx *c;
    hipStream_t st;
    hipEvent_t stop = nullptr;
    LaunchTimer(voc_ctx *c_, int kind, hipStream_t st_) : c(c_), st(st_)
    {
        if (!c->profiling) return;
        hipEvent_t start;
        HIPCHK(hipEventCreate(&start));
        HIPCHK(hipEventCreate(&stop));
        HIPCHK(hipEventRecord(start, st));
        c->timed.push_back({kind, start, stop});
    }
    ~LaunchTimer() { if (stop) (void)hipEventRecord(stop, st); }
#else
    LaunchTimer(voc_ctx *, int, hipStream_t) {}
#endif
};

inline long long floordiv(long long a, long long b)
{
    long long q = a / b;
    if ((a % b != 0) && ((a < 0) != (b < 0))) --q;
    return q;
}

inline int env_int(const char *name)
{
    const char *s = std::getenv(name);
    return s ? std::atoi(s) : 0;
}

template <class R = double>
inline size_t lds_bytes(const voc_ctx *c, bool ola)
{
    size_t b = (size_t)(c->M + (c->M >> 3) + 2) * 2 * sizeof(R);
    if (ola) b += (size_t)c->W * sizeof(R);
    return b;
}

template <class R> const voc::ParamsT<R> &params_of(const voc_ctx *c);
template <> const voc::ParamsT<double> &params_of<double>(const voc_ctx *c) { return c->params; }
template <> const voc::ParamsT<float> &params_of<float>(const voc_ctx *c) { return c->params_f; }

#ifdef VOC_DEV_ONLY_LOG2M  // development builds for timing experiments: one frame length only, compiled in seconds
#define VOC_FOR_EACH_LOG2M(X) X(VOC_DEV_ONLY_LOG2M)
#define VOC_FOR_EACH_BLUE_LOG2L(X) X(VOC_DEV_ONLY_LOG2M)
#else
#define VOC_FOR_EACH_LOG2M(X) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13)
#define VOC_FOR_EACH_BLUE_LOG2L(X) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13)
#endif

inline unsigned ew_grid(long long total);

// ---- chirp-z transform through device memory (voc_kernels_bluestein_big.h) ----------------------------

// frames per pass: the transform buffer takes 16 L bytes per frame (0.25 - 0.5 MiB)
inline long long big_blue_batch(voc_ctx *c, long long frames, size_t per_frame_extra)
{
    const size_t per_frame = (size_t)c->big.L1 * voc::kBigP * sizeof(double2) + per_frame_extra;
    long long nb = (long long)(((size_t)256 << 20) / per_frame);
    nb = nb < 1 ? 1 : nb;
    nb = nb < frames ? nb : frames;
    const size_t need = (size_t)nb * per_frame;
    if (need > c->big_ws_bytes) {
        HIPCHK(hipDeviceSynchronize());
        if (c->big_ws) (void)hipFree(c->big_ws);
        c->big_ws = nullptr;
        c->big_ws_bytes = 0;
        HIPCHK(hipMalloc(&c->big_ws, need));
        c->big_ws_bytes = need;
    }
    return nb;
}

template <int SRC, int DST>
void big_blue_transform(voc_ctx *c, const voc::BigSrc &s, const voc::BigDst &d, long long n, double2 *u, hipStream_t st)
{
    const unsigned grid = (unsigned)(n * c->big.L1);
    hipLaunchKernelGGL((voc::blue_big_first_kernel<SRC>), dim3(grid), dim3(voc::Cfg<voc::kBigLog2P>::T), lds_bytes<double>(c, false),
                       st, c->params, c->big, s, u);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(voc::blue_big_middle_kernel, dim3(ew_grid(n * voc::kBigP)), dim3(256), 0, st, c->big, u, n);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL((voc::blue_big_last_kernel<DST>), dim3(grid), dim3(voc::Cfg<voc::kBigLog2P>::T), lds_bytes<double>(c, false),
                       st, c->params, c->big, (const double2 *)u, d);
    HIPCHK(hipGetLastError());
}

// The contract of analysis_blue_kernel (tiles are an internal matter there; here the frames of a channel
// are walked in order from phase_in).
template <bool RAW, bool SPEC>
void big_blue_analysis(voc_ctx *c, const voc::AnaArgs &a, long long channels, hipStream_t st)
{
    const long long total = channels * a.F;
    if (total <= 0) return;
    const size_t spec_bytes = SPEC ? 0 : (size_t)c->K * sizeof(double2);
    const long long nb = big_blue_batch(c, total, spec_bytes);
    double2 *u = static_cast<double2 *>(c->big_ws);
    double *spec_tmp = reinterpret_cast<double *>(u + (size_t)nb * c->big.L1 * voc::kBigP);
    if (!SPEC && a.phase_out != a.phase_in)
        HIPCHK(hipMemcpyAsync(a.phase_out, a.phase_in, sizeof(double) * (size_t)(channels * c->K), hipMemcpyDeviceToDevice, st));
    for (long long b0 = 0; b0 < total; b0 += nb) {
        const long long n = total - b0 < nb ? total - b0 : nb;
        voc::BigSrc s{a.x, a.chan_stride, a.frame_stride, nullptr, a.F, b0, 0};
        voc::BigDst d{};
        d.first = b0;
        d.F = a.F;
        if (SPEC) {
            d.spec = a.spec;
            d.rpc = a.rpc;
            d.row_base = 0;
        } else {
            d.spec = spec_tmp;
            d.rpc = a.F;
            d.row_base = b0;
        }
        big_blue_transform<RAW ? voc::BIG_SRC_RAW : voc::BIG_SRC_FRAME, voc::BIG_DST_SPEC>(c, s, d, n, u, st);
        if (!SPEC) {
            hipLaunchKernelGGL(voc::spec_rows_to_polar_kernel, dim3(ew_grid(channels * c->K)), dim3(256), 0, st, c->params,
                               c->big.n_d, (const double *)spec_tmp, b0, a.phase_out, a.mag, a.inc, a.ldk, a.rpc, a.F, b0,
                               b0 + n, a.vol, channels);
            HIPCHK(hipGetLastError());
        }
    }
}

template <int IN_MODE, int OUT_MODE>
void big_blue_synthesis(voc_ctx *c, const voc::SynArgs &a, long long channels, hipStream_t st)
{
    const long long total = channels * a.F;
    if (total <= 0) return;
    constexpr bool POLAR = IN_MODE == voc::SYN_IN_POLAR;
    const size_t spec_bytes = POLAR ? (size_t)c->K * sizeof(double2) : 0;
    const long long nb = big_blue_batch(c, total, spec_bytes);
    double2 *u = static_cast<double2 *>(c->big_ws);
    double *spec_tmp = reinterpret_cast<double *>(u + (size_t)nb * c->big.L1 * voc::kBigP);
    if (POLAR && a.phase_out != a.phase_in)
        HIPCHK(hipMemcpyAsync(a.phase_out, a.phase_in, sizeof(double) * (size_t)(channels * c->K), hipMemcpyDeviceToDevice, st));
    for (long long b0 = 0; b0 < total; b0 += nb) {
        const long long n = total - b0 < nb ? total - b0 : nb;
        voc::BigSrc s{nullptr, 0, 0, nullptr, a.F, b0, 0};
        if (POLAR) {
            hipLaunchKernelGGL(voc::polar_rows_to_spec_kernel, dim3(ew_grid(channels * c->K)), dim3(256), 0, st, c->params, a,
                               a.phase_out, spec_tmp, b0, b0, b0 + n, channels);
            HIPCHK(hipGetLastError());
            s.spec = spec_tmp;
            s.spec_base = b0;
        } else {
            s.spec = a.spec;
            s.spec_base = 0;
        }
        voc::BigDst d{};
        d.out = a.out;
        d.first = b0;
        d.F = a.F;
        d.rpc = a.F;
        d.row_base = 0;
        big_blue_transform<voc::BIG_SRC_SPEC, OUT_MODE == voc::SYN_OUT_RAW ? voc::BIG_DST_RAW : voc::BIG_DST_FRAME>(c, s, d, n, u, st);
    }
}

template <bool RAW, bool SPEC, class R = double>
void launch_analysis(voc_ctx *c, const voc::AnaArgsT<R> &a, long long channels, hipStream_t st)
{
    const long long grid = channels * a.tiles;
    if (grid <= 0) return;
    if (grid > 0x7fffffffLL) throw HipFail{hipErrorInvalidValue, "analysis grid too large"};
    LaunchTimer timer(c, SPEC ? KIND_OTHER : KIND_ANALYSIS, st);
    if (c->bluestein) {
        if constexpr (std::is_same<R, double>::value) {
            if (c->blue_big) {
                big_blue_analysis<RAW, SPEC>(c, a, channels, st);
                return;
            }
            switch (c->log2m) {
#define X(L)                                                                                              \
    case L:                                                                                               \
        hipLaunchKernelGGL((voc::analysis_blue_kernel<L, RAW, SPEC>), dim3((unsigned)grid), dim3(voc::Cfg<L>::T), \
                           lds_bytes<double>(c, false), st, c->params, a, c->blue);                       \
        break;
                VOC_FOR_EACH_BLUE_LOG2L(X)
#undef X
            }
            HIPCHK(hipGetLastError());
            return;
        } else {
            throw StatusFail{VOC_ERR_UNSUPPORTED};
        }
    }
    switch (c->log2m) {
#define X(L)                                                                                              \
    case L:                                                                                               \
        hipLaunchKernelGGL((voc::analysis_kernel<L, RAW, SPEC, R>), dim3((unsigned)grid), dim3(voc::Cfg<L>::T), \
                           lds_bytes<R>(c, false), st, params_of<R>(c), a);                               \
        break;
        VOC_FOR_EACH_LOG2M(X)
#undef X
    }
    HIPCHK(hipGetLastError());
}

template <int IN_MODE, int OUT_MODE, class R = double>
void launch_synthesis(voc_ctx *c, const voc::SynArgsT<R> &a, long long channels, hipStream_t st)
{
    const long long grid = a.unsplit + (channels - a.unsplit) * a.tiles;
    if (grid <= 0) return;
    if (grid > 0x7fffffffLL) throw HipFail{hipErrorInvalidValue, "synthesis grid too large"};
    LaunchTimer timer(c, IN_MODE == voc::SYN_IN_SPEC ? KIND_OTHER : KIND_SYNTHESIS, st);
    if (c->bluestein) {
        if constexpr (std::is_same<R, double>::value && OUT_MODE != voc::SYN_OUT_OLA) {
            if (c->blue_big) {
                big_blue_synthesis<IN_MODE, OUT_MODE>(c, a, channels, st);
                return;
            }
            switch (c->log2m) {
#define X(L)                                                                                              \
    case L:                                                                                               \
        hipLaunchKernelGGL((voc::synthesis_blue_kernel<L, IN_MODE, OUT_MODE>), dim3((unsigned)grid),      \
                           dim3(voc::Cfg<L>::T), lds_bytes<double>(c, false), st, c->params, a, c->blue); \
        break;
                VOC_FOR_EACH_BLUE_LOG2L(X)
#undef X
            }
            HIPCHK(hipGetLastError());
            return;
        } else {
            throw StatusFail{VOC_ERR_UNSUPPORTED};  // overlap-add of these lengths goes through frames
        }
    }
    switch (c->log2m) {
#define X(L)                                                                                                    \
    case L:                                                                                                     \
        hipLaunchKernelGGL((voc::synthesis_kernel<L, IN_MODE, OUT_MODE, R>), dim3((unsigned)grid), dim3(voc::Cfg<L>::T), \
                           lds_bytes<R>(c, OUT_MODE == voc::SYN_OUT_OLA), st, params_of<R>(c), a);                 \
        break;
        VOC_FOR_EACH_LOG2M(X)
#undef X
    }
    HIPCHK(hipGetLastError());
}

#ifdef VOC_DEV_ONLY_LOG2M
#define VOC_FOR_EACH_FAST_LOG2M(X) X(VOC_DEV_ONLY_LOG2M)
#else
#define VOC_FOR_EACH_FAST_LOG2M(X) X(9) X(10) X(11) X(12)
#endif

inline bool fast_sizes(const voc_ctx *c) { return !c->force_generic && !c->bluestein && c->log2m >= 9 && c->log2m <= 12; }

// The kernels with sixteen points per thread (voc_kernels_fast16.h) serve double-precision contexts with
// N = 4096 or 8192.  Whether a context's analysis takes them depends on the context alone -- never on the
// alignment, the strides or the shape of one call's arrays -- so that every entry point of a context
// (stages, signals, streams pushed in any chunking) produces the same bits.
inline bool fast16_sizes(const voc_ctx *c)
{
    return !c->force_generic && !c->bluestein && c->log2m >= 11 && c->log2m <= 13;
}

inline bool fast16_ana_on(const voc_ctx *c) { return c->precision == VOC_PRECISION_F32 ? c->fast16_ana_f32 : c->fast16_ana; }
inline int fast16_syn_on(const voc_ctx *c) { return c->precision == VOC_PRECISION_F32 ? c->fast16_syn_f32 : c->fast16_syn; }

inline bool fast16_analysis(const voc_ctx *c) { return fast16_ana_on(c) && fast16_sizes(c); }

// The sixteen-point synthesis kernel (overlap-add in registers) needs W == N and a hop of 1, 2, 4 or 8 times
// N/16 samples.  Returns that multiple (the pair slots its overlap-add emits per frame) or 0.
inline int fast16_hop_slots(const voc_ctx *c, bool synthesis)
{
    if (!fast16_sizes(c) || c->W != c->N) return 0;
    if (synthesis ? !(fast16_syn_on(c) == 1 || (fast16_syn_on(c) < 0 && c->log2m == 13)) : !fast16_ana_on(c)) return 0;
    const int t2 = c->M / 8;  // 2 T samples
    if (c->H % t2 != 0) return 0;
    const int s = c->H / t2;
    return (s == 1 || s == 2 || s == 4 || s == 8) ? s : 0;
}

template <class R>
inline bool rows_aligned16(const R *y, long long stride) { return reinterpret_cast<uintptr_t>(y) % (2 * sizeof(R)) == 0 && stride % 2 == 0; }

#ifdef VOC_DEV_ONLY_LOG2M
#define VOC_FOR_EACH_FAST16_LOG2M(X) X(VOC_DEV_ONLY_LOG2M)
#else
#define VOC_FOR_EACH_FAST16_LOG2M(X) X(11) X(12) X(13)
#endif

template <int L, class R>
void launch_analysis_fast16(voc_ctx *c, const voc::AnaArgsT<R> &a, unsigned grid, bool vec2, int slide, hipStream_t st)
{
    const dim3 g(grid), b(voc::Fast16Cfg<L>::T);
    const size_t lds = voc::Fast16Cfg<L>::template lds_bytes<R>();
    const voc::ParamsT<R> &p = params_of<R>(c);
    if (!vec2) hipLaunchKernelGGL((voc::analysis_fast16_kernel<L, false, 0, R>), g, b, lds, st, p, a);
    else if (slide == 1) hipLaunchKernelGGL((voc::analysis_fast16_kernel<L, true, 1, R>), g, b, lds, st, p, a);
    else if (slide == 2) hipLaunchKernelGGL((voc::analysis_fast16_kernel<L, true, 2, R>), g, b, lds, st, p, a);
    else if (slide == 4) hipLaunchKernelGGL((voc::analysis_fast16_kernel<L, true, 4, R>), g, b, lds, st, p, a);
    else if (slide == 8) hipLaunchKernelGGL((voc::analysis_fast16_kernel<L, true, 8, R>), g, b, lds, st, p, a);
    else hipLaunchKernelGGL((voc::analysis_fast16_kernel<L, true, 0, R>), g, b, lds, st, p, a);
}

template <int L, bool MAPPED, class R, bool ALIGNED>
void launch_synthesis_fast16_al(voc_ctx *c, const voc::SynArgsT<R> &a, unsigned grid, int nhp, hipStream_t st)
{
    const dim3 g(grid), b(voc::Fast16Cfg<L>::T);
    const size_t lds = voc::Fast16Cfg<L>::template lds_bytes<R>();
    const voc::ParamsT<R> &p = params_of<R>(c);
    switch (nhp) {
    case 1: hipLaunchKernelGGL((voc::synthesis_fast16_kernel<L, 1, R, MAPPED, ALIGNED>), g, b, lds, st, p, a); break;
    case 2: hipLaunchKernelGGL((voc::synthesis_fast16_kernel<L, 2, R, MAPPED, ALIGNED>), g, b, lds, st, p, a); break;
    case 4: hipLaunchKernelGGL((voc::synthesis_fast16_kernel<L, 4, R, MAPPED, ALIGNED>), g, b, lds, st, p, a); break;
    case 8: hipLaunchKernelGGL((voc::synthesis_fast16_kernel<L, 8, R, MAPPED, ALIGNED>), g, b, lds, st, p, a); break;
    default:  // 16: whole frames out, no overlap-add
        if constexpr (!MAPPED) hipLaunchKernelGGL((voc::synthesis_fast16_kernel<L, 16, R, false, ALIGNED>), g, b, lds, st, p, a);
        break;
    }
}

// (output rows that are not 16-byte aligned take the build that stores the sample pairs as two values each)
template <int L, bool MAPPED, class R>
void launch_synthesis_fast16(voc_ctx *c, const voc::SynArgsT<R> &a, unsigned grid, int nhp, hipStream_t st)
{
    // (whole frames out: rows of W = N samples, aligned with their base)
    const bool aligned = nhp == 16 ? rows_aligned16(a.out, 0) : rows_aligned16(a.out, a.out_stride);
    if (aligned) launch_synthesis_fast16_al<L, MAPPED, R, true>(c, a, grid, nhp, st);
    else launch_synthesis_fast16_al<L, MAPPED, R, false>(c, a, grid, nhp, st);
}

#if defined(VOC_STAMPS) && !defined(VOC_EMU)
// diagnostic builds: prints and clears the table of in-kernel time marks (voc_kernels_fast16.h)
inline void print_stamps(const char *what, int kid)
{
    unsigned long long h[2][voc::kStampSlots + 1];
    if (hipDeviceSynchronize() != hipSuccess) return;
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(voc::voc_stamps), sizeof(h)) != hipSuccess) return;
    const double n = (double)h[kid][voc::kStampSlots];
    if (n > 0) {
        double total = 0;
        for (int i = 0; i < voc::kStampSlots; ++i) total += (double)h[kid][i];
        std::fprintf(stderr, "[stamps] %s: %.0f wave-frames, %.0f clocks per wave-frame:", what, n, total / n);
        for (int i = 0; i < voc::kStampSlots; ++i) std::fprintf(stderr, " %d:%.0f", i, (double)h[kid][i] / n);
        std::fprintf(stderr, "\n");
    }
    std::memset(h, 0, sizeof(h));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(voc::voc_stamps), h, sizeof(h));
}
#else
inline void print_stamps(const char *, int) {}
#endif

// Specialised analysis kernel (polar output) when the frame length allows it.
template <class R>
bool launch_analysis_fast(voc_ctx *c, const voc::AnaArgsT<R> &a, long long channels, hipStream_t st)
{
    if (!fast_sizes(c) && !fast16_analysis(c)) return false;
    const long long grid = channels * a.tiles;
    if (grid <= 0) return true;
    if (grid > 0x7fffffffLL) throw HipFail{hipErrorInvalidValue, "analysis grid too large"};
    const bool vec2 = c->W == c->N && (reinterpret_cast<uintptr_t>(a.x) % (2 * sizeof(R)) == 0) && a.chan_stride % 2 == 0 &&
                      a.frame_stride % 2 == 0;
    if (fast16_analysis(c)) {
        // frames cut from one signal, hop 1, 2, 4 or 8 times N/16: the sample window slides in registers
        const int s16 = (vec2 && a.frame_stride == c->H) ? fast16_hop_slots(c, false) : 0;
        LaunchTimer timer(c, KIND_ANALYSIS, st);
        switch (c->log2m) {
#define X(L) case L: launch_analysis_fast16<L, R>(c, a, (unsigned)grid, vec2, s16, st); break;
            VOC_FOR_EACH_FAST16_LOG2M(X)
#undef X
        }
        HIPCHK(hipGetLastError());
        print_stamps("analysis16", 0);
        return true;
    }
    if (!fast_sizes(c)) return false;
    // frames cut from one signal, hop a multiple of 2T: the sample window slides in registers
    const int t2 = 2 * (c->M / 8);
    int slide = 0;
    if (vec2 && a.frame_stride == c->H && c->H % t2 == 0) {
        const int sl = c->H / t2;
        if (sl == 1 || sl == 2 || sl == 4) slide = sl;
    }
    LaunchTimer timer(c, KIND_ANALYSIS, st);
    const dim3 g((unsigned)grid);
    switch (c->log2m) {
#define X(L)                                                                                                        \
    case L: {                                                                                                       \
        const dim3 b(voc::FastCfg<L>::T);                                                                           \
        const size_t lds = voc::FastCfg<L>::template lds_bytes<R>();                                                \
        if (!vec2) hipLaunchKernelGGL((voc::analysis_fast_kernel<L, false, 0, R>), g, b, lds, st, params_of<R>(c), a);       \
        else if (slide == 1) hipLaunchKernelGGL((voc::analysis_fast_kernel<L, true, 1, R>), g, b, lds, st, params_of<R>(c), a); \
        else if (slide == 2) hipLaunchKernelGGL((voc::analysis_fast_kernel<L, true, 2, R>), g, b, lds, st, params_of<R>(c), a); \
        else if (slide == 4) hipLaunchKernelGGL((voc::analysis_fast_kernel<L, true, 4, R>), g, b, lds, st, params_of<R>(c), a); \
        else hipLaunchKernelGGL((voc::analysis_fast_kernel<L, true, 0, R>), g, b, lds, st, params_of<R>(c), a);              \
    } break;
        VOC_FOR_EACH_FAST_LOG2M(X)
#undef X
    }
    HIPCHK(hipGetLastError());
    print_stamps("analysis8", 0);
    return true;
}

// The register-resident overlap-add needs W == N (eight sample pairs per thread) and a hop that is a
// multiple of twice the thread count.  Returns H / 2T or 0.  It depends on the context alone: whatever the
// alignment of a call's output rows, the same kernel arithmetic runs (16-byte or 8-byte stores), so that every
// entry point of a context gives the same bits.
template <class R>
inline int fast_ola_nhp(const voc_ctx *c, const voc::SynArgsT<R> &)
{
    if (!fast_sizes(c) || c->W != c->N) return 0;
    const int t2 = 2 * (c->M / 8);
    if (c->H % t2 != 0) return 0;
    const int nhp = c->H / t2;
    return (nhp == 1 || nhp == 2 || nhp == 4 || nhp == 8) ? nhp : 0;
}

template <int L, bool OLA, class R, bool MAPPED, bool ALIGNED>
void launch_synthesis_fast_al(voc_ctx *c, const voc::SynArgsT<R> &a, unsigned grid, int nhp, hipStream_t st)
{
    const dim3 g(grid), b(voc::FastCfg<L>::T);
    const size_t lds = voc::syn_lds_bytes<L, OLA, R>();
    switch (nhp) {
    case 1: hipLaunchKernelGGL((voc::synthesis_fast_kernel<L, OLA, 1, R, MAPPED, ALIGNED>), g, b, lds, st, params_of<R>(c), a); break;
    case 2: hipLaunchKernelGGL((voc::synthesis_fast_kernel<L, OLA, 2, R, MAPPED, ALIGNED>), g, b, lds, st, params_of<R>(c), a); break;
    case 4: hipLaunchKernelGGL((voc::synthesis_fast_kernel<L, OLA, 4, R, MAPPED, ALIGNED>), g, b, lds, st, params_of<R>(c), a); break;
    default: hipLaunchKernelGGL((voc::synthesis_fast_kernel<L, OLA, 8, R, MAPPED, ALIGNED>), g, b, lds, st, params_of<R>(c), a); break;
    }
}

// the build that filters the rows as it reads them (SynArgsT::pitch_n / gain): double precision, two waves per SIMD
template <int L, bool ALIGNED, int FILT>
void launch_synthesis_filtered_al(voc_ctx *c, const voc::SynArgsT<double> &a, unsigned grid, int nhp, hipStream_t st)
{
    if constexpr (voc::syn_fat<L, true, double>()) {
        const dim3 g(grid), b(voc::FastCfg<L>::T);
        const size_t lds = voc::syn_lds_bytes<L, true, double>();
        switch (nhp) {
        case 1: hipLaunchKernelGGL((voc::synthesis_fast_kernel<L, true, 1, double, false, ALIGNED, FILT>), g, b, lds, st, c->params, a); break;
        case 2: hipLaunchKernelGGL((voc::synthesis_fast_kernel<L, true, 2, double, false, ALIGNED, FILT>), g, b, lds, st, c->params, a); break;
        case 4: hipLaunchKernelGGL((voc::synthesis_fast_kernel<L, true, 4, double, false, ALIGNED, FILT>), g, b, lds, st, c->params, a); break;
        default: hipLaunchKernelGGL((voc::synthesis_fast_kernel<L, true, 8, double, false, ALIGNED, FILT>), g, b, lds, st, c->params, a); break;
        }
    } else {
        throw StatusFail{VOC_ERR_UNSUPPORTED};
    }
}

template <int L, bool OLA, class R, bool MAPPED = false>
void launch_synthesis_fast_nh(voc_ctx *c, const voc::SynArgsT<R> &a, unsigned grid, int nhp, hipStream_t st)
{
    if (!OLA || rows_aligned16(a.out, a.out_stride)) launch_synthesis_fast_al<L, OLA, R, MAPPED, true>(c, a, grid, nhp, st);
    else launch_synthesis_fast_al<L, OLA, R, MAPPED, false>(c, a, grid, nhp, st);
}

// Specialised synthesis kernel (polar input): frames out, or overlap-add out when fast_ola_nhp() != 0.
// Whether a context's synthesis can filter its rows as it reads them (voc_synthesis_filtered_dev): the two-wave
// double-precision overlap-add build of the eight-point kernel.
inline bool synthesis_can_filter(const voc_ctx *c)
{
    voc::SynArgsT<double> none{};
    return c->precision == VOC_PRECISION_F64 && VOC_SYN_FAT != 0 && fast_sizes(c) && c->log2m >= 10 && fast_ola_nhp(c, none) > 0 &&
           fast16_hop_slots(c, true) == 0;
}

template <class R>
bool launch_synthesis_fast(voc_ctx *c, const voc::SynArgsT<R> &a, long long channels, bool ola, hipStream_t st)
{
    // Whether a context's synthesis takes the specialised kernels depends on the context alone, and is the same
    // for frames out (synthesisStage) and overlap-add out (the fused call, streams): a context whose overlap-add
    // form has no specialised build (W < N, a hop that is no multiple of the pair slots) takes the generic kernels
    // for both, so that the stage-level chain and the fused call give the same bits.
    const int s16 = fast16_hop_slots(c, true);
    const int nh_ctx = fast_ola_nhp(c, a);
    if (s16 == 0 && nh_ctx == 0) return false;
    const int nh = ola ? nh_ctx : 8;
    const bool mapped = a.mag_rows || a.inc_rows;
    if (mapped && !ola) return false;  // (frames out through row maps: the generic kernel reads through maps too)
    const long long grid = a.unsplit + (channels - a.unsplit) * a.tiles;
    if (grid <= 0) return true;
    if (grid > 0x7fffffffLL) throw HipFail{hipErrorInvalidValue, "synthesis grid too large"};
    LaunchTimer timer(c, KIND_SYNTHESIS, st);
    if (s16 > 0) {
        switch (c->log2m) {
#define X(L)                                                                                                     \
    case L:                                                                                                      \
        if (mapped) launch_synthesis_fast16<L, true, R>(c, a, (unsigned)grid, s16, st);                          \
        else launch_synthesis_fast16<L, false, R>(c, a, (unsigned)grid, ola ? s16 : 16, st);                     \
        break;
            VOC_FOR_EACH_FAST16_LOG2M(X)
#undef X
        }
        HIPCHK(hipGetLastError());
        print_stamps("synthesis16", 1);
        return true;
    }
    if (!fast_sizes(c)) return false;
    if constexpr (std::is_same<R, double>::value) {
        if (a.pitch_n != 0.0 || a.gain) {  // (the caller has checked synthesis_can_filter)
            if (!ola || mapped || !synthesis_can_filter(c)) throw StatusFail{VOC_ERR_UNSUPPORTED};
            const bool al = rows_aligned16(a.out, a.out_stride);
            // a whole-number factor (or none) reads one entry per bin: the lighter build
            const bool one = a.pitch_n == 0.0 || (a.pitch_n >= 1.0 && a.pitch_n <= 64.0 && a.pitch_n == std::floor(a.pitch_n));
            switch (c->log2m) {
#define X(L)                                                                                          \
    case L:                                                                                           \
        if (one && al) launch_synthesis_filtered_al<L, true, 1>(c, a, (unsigned)grid, nh, st);        \
        else if (one) launch_synthesis_filtered_al<L, false, 1>(c, a, (unsigned)grid, nh, st);        \
        else if (al) launch_synthesis_filtered_al<L, true, 2>(c, a, (unsigned)grid, nh, st);          \
        else launch_synthesis_filtered_al<L, false, 2>(c, a, (unsigned)grid, nh, st);                 \
        break;
                VOC_FOR_EACH_FAST_LOG2M(X)
#undef X
            }
            HIPCHK(hipGetLastError());
            return true;
        }
    }
    switch (c->log2m) {
#define X(L)                                                                          \
    case L:                                                                           \
        if (mapped) launch_synthesis_fast_nh<L, true, R, true>(c, a, (unsigned)grid, nh, st);                     \
        else if (ola) launch_synthesis_fast_nh<L, true, R>(c, a, (unsigned)grid, nh, st);     \
        else hipLaunchKernelGGL((voc::synthesis_fast_kernel<L, false, 8, R>), dim3((unsigned)grid), \
                                dim3(voc::FastCfg<L>::T), voc::FastCfg<L>::template lds_bytes<R>(), st, params_of<R>(c), a); \
        break;
        VOC_FOR_EACH_FAST_LOG2M(X)
#undef X
    }
    HIPCHK(hipGetLastError());
    print_stamps("synthesis8", 1);
    return true;
}

inline unsigned ew_grid(long long total)
{
    long long g = (total + 255) / 256;
    if (g < 1) g = 1;
    if (g > 65536) g = 65536;
    return (unsigned)g;
}

// Frames per analysis tile.  A tile recomputes the frame before its first one (to recover that
// frame's raw phase), so long tiles are cheap; short ones keep small problems spread over the chip.
int analysis_tile_len(const voc_ctx *c, long long channels, long long F)
{
    if (c->ana_tile_override > 0) return c->ana_tile_override;
    long long tl = channels * F / 4096;
    if (tl < 8) tl = 8;
    if (tl > 64) tl = 64;
    if (tl > F) tl = F;
    return (int)(tl < 1 ? 1 : tl);
}

// Synthesis tiles per channel.  The phase chain is sequential along frames, so a channel is cut
// only when there are too few channels to fill the chip; a cut costs one pass of the carry
// kernel over ph_inc plus `halo` recomputed frames per tile in overlap-add mode.
void synthesis_tiling(const voc_ctx *c, long long channels, long long F, int halo, int *tile_len, int *tiles)
{
    long long tl;
    if (c->syn_tile_override > 0) {
        tl = c->syn_tile_override;
    } else {
        long long want = (3072 + channels - 1) / channels;  // tiles per channel to reach ~3072 workgroups (4 rounds of 3 per CU)
        long long min_tile = 8LL * (halo + 1);
        if (min_tile < 16) min_tile = 16;
        long long max_tiles = F / min_tile;
        if (max_tiles < 1) max_tiles = 1;
        if (want > max_tiles) want = max_tiles;
        if (want < 1) want = 1;
        tl = (F + want - 1) / want;
    }
    if (tl <= halo) tl = halo + 1;
    if (tl > F) tl = F;
    if (tl < 1) tl = 1;
    *tile_len = (int)tl;
    *tiles = (int)((F + tl - 1) / tl);
}

void ensure_ws(voc_ctx *c, size_t bytes)
{
    if (bytes <= c->ws_bytes) return;
    if (c->ws) {
        HIPCHK(hipDeviceSynchronize());
        (void)hipFree(c->ws);
        c->ws = nullptr;
        c->ws_bytes = 0;
    }
    HIPCHK(hipMalloc(&c->ws, bytes));
    c->ws_bytes = bytes;
}

// Brackets one enqueued use of the context's workspace on stream st (see voc_ctx::ws_event).
struct WsUse {
    voc_ctx *c;
    hipStream_t st;
    WsUse(voc_ctx *c_, hipStream_t st_) : c(c_), st(st_)
    {
        if (c->ws_busy && c->ws_stream != st) HIPCHK(hipStreamWaitEvent(st, c->ws_event, 0));
    }
    ~WsUse()
    {
        if (hipEventRecord(c->ws_event, st) == hipSuccess) {
            c->ws_busy = true;
            c->ws_stream = st;
        }
    }
    WsUse(const WsUse &) = delete;
    WsUse &operator=(const WsUse &) = delete;
};

inline long long num_frames(const voc_ctx *c, long long len) { return len < c->W ? 0 : (len - c->W) / c->H + 1; }

inline long long ola_length(const voc_ctx *c, long long F, long long chunk)
{
    if (F <= 0) return 0;
    const long long last = (F - 1) * c->H + c->W - 1;
    return (last / chunk + 1) * chunk;
}

inline long long leftover_length(const voc_ctx *c, long long len)
{
    long long drop = floordiv(len - c->W, c->H) * c->H + c->H;
    if (drop < 0) drop = 0;
    if (drop > len) drop = len;
    return len - drop;
}

// ---- device-pointer bodies (ctx->mu held, device set) -------------------------------------------

template <class R = double>
void analysis_stage_dev(voc_ctx *c, R *phase_io, const R *frames, R *mag, R *inc, long long C,
                        long long F, hipStream_t st)
{
    if (C <= 0 || F <= 0) return;
    WsUse use(c, st);
    // the kernel reads the initial phases while another tile may already write the final ones
    ensure_ws(c, (size_t)C * c->K * sizeof(R));
    R *pin = static_cast<R *>(c->ws);
    HIPCHK(hipMemcpyAsync(pin, phase_io, (size_t)C * c->K * sizeof(R), hipMemcpyDeviceToDevice, st));
    voc::AnaArgsT<R> a{};
    a.x = frames;
    a.chan_stride = F * c->W;
    a.frame_stride = c->W;
    a.phase_in = pin;
    a.phase_out = phase_io;
    a.mag = mag;
    a.inc = inc;
    a.ldk = c->K;
    a.F = F;
    a.rpc = F;
    a.tile_len = analysis_tile_len(c, C, F);
    a.tiles = (int)((F + a.tile_len - 1) / a.tile_len);
    a.vol = (R)1;
    if (!launch_analysis_fast<R>(c, a, C, st)) launch_analysis<false, false, R>(c, a, C, st);
}

template <class R = double>
void synthesis_stage_dev(voc_ctx *c, R *phase_io, const R *mag, const R *inc, R *frames,
                         long long C, long long F, hipStream_t st)
{
    if (C <= 0 || F <= 0) return;
    WsUse use(c, st);
    int tile_len, tiles;
    synthesis_tiling(c, C, F, 0, &tile_len, &tiles);
    const size_t pin_bytes = (size_t)C * c->K * sizeof(R);
    const size_t carry_bytes = tiles > 1 ? (size_t)C * tiles * c->K * sizeof(R) : 0;
    ensure_ws(c, pin_bytes + carry_bytes);
    R *pin = static_cast<R *>(c->ws);
    R *carry = pin + (size_t)C * c->K;
    HIPCHK(hipMemcpyAsync(pin, phase_io, pin_bytes, hipMemcpyDeviceToDevice, st));
    if (tiles > 1) {
        const long long chains = C * c->K;
        LaunchTimer timer(c, KIND_CARRY, st);
        hipLaunchKernelGGL(voc::phase_carry_kernel<R>, dim3((unsigned)((chains + 255) / 256)), dim3(256), 0, st,
                           params_of<R>(c), inc, (long long)c->K, F, (const R *)pin, carry, tile_len, tiles, 0, C,
                           (const long long *)nullptr, 0LL);
        HIPCHK(hipGetLastError());
    }
    voc::SynArgsT<R> a{};
    a.mag = mag;
    a.inc = inc;
    a.ldk = c->K;
    a.phase_in = pin;
    a.carry = carry;
    a.phase_out = phase_io;
    a.out = frames;
    a.F = F;
    a.tile_len = tile_len;
    a.tiles = tiles;
    a.halo = 0;
    if (!launch_synthesis_fast<R>(c, a, C, false, st)) launch_synthesis<voc::SYN_IN_POLAR, voc::SYN_OUT_FRAMES, R>(c, a, C, st);
}

// genFramesOfE + analysisStage (+ volumeFix) on signal rows: frame j of a channel is read in place
// at sample j*H (Frames.hs:34-35).
// rpc: rows per channel of mag / inc (0: F); with more, mag / inc point at the row frame 0 goes to
template <class R = double>
void analysis_signal_dev(voc_ctx *c, const R *x, long long x_stride, long long F, long long C, R *phase_io,
                         R *mag, R *inc, long long ldk, bool volume_fix, hipStream_t st, long long rpc = 0)
{
    if (C <= 0 || F <= 0) return;
    WsUse use(c, st);
    ensure_ws(c, (size_t)C * c->K * sizeof(R));
    R *pin = static_cast<R *>(c->ws);
    HIPCHK(hipMemcpyAsync(pin, phase_io, (size_t)C * c->K * sizeof(R), hipMemcpyDeviceToDevice, st));
    voc::AnaArgsT<R> a{};
    a.x = x;
    a.chan_stride = x_stride;
    a.frame_stride = c->H;
    a.phase_in = pin;
    a.phase_out = phase_io;
    a.mag = mag;
    a.inc = inc;
    a.ldk = ldk;
    a.F = F;
    a.rpc = rpc ? rpc : F;
    a.tile_len = analysis_tile_len(c, C, F);
    a.tiles = (int)((F + a.tile_len - 1) / a.tile_len);
    a.vol = volume_fix ? (R)c->vol : (R)1;
    if (!launch_analysis_fast<R>(c, a, C, st)) launch_analysis<false, false, R>(c, a, C, st);
}

constexpr size_t kLdsLimit = 160 * 1024;

// synthesisStage + sumFramesE: y rows receive y_len samples each (zeros past the last frame).
long long synthesis_unsplit(const voc_ctx *c, long long channels, long long F, int halo, bool fused, int *tile_len, int *tiles);

template <class R = double>
void synthesis_signal_dev(voc_ctx *c, const R *mag, const R *inc, long long ldk, long long F, long long C,
                          R *phase_io, R *y, long long y_stride, long long y_len, hipStream_t st,
                          const long long *mag_rows = nullptr, long long F_mag = 0, const long long *inc_rows = nullptr,
                          long long F_inc = 0, double pitch_n = 0.0, const double *gain = nullptr)
{
    if (C <= 0 || F <= 0) return;
    WsUse use(c, st);
    const bool filtered = pitch_n != 0.0 || gain != nullptr;  // rows filtered as they are read (synthesis_can_filter contexts)
    double pitch_rcp = 0.0;
    if (pitch_n != 0.0) {
        int e = 0;
        if (std::frexp(pitch_n, &e) == 0.5) pitch_rcp = 1.0 / pitch_n;  // a power of two: the product is the quotient
    }
    const bool fused = !c->bluestein && (lds_bytes<R>(c, true) <= kLdsLimit ||
                                         fast16_hop_slots(c, true) > 0);
    const int halo = fused ? (c->W + c->H - 1) / c->H - 1 : 0;
    int tile_len, tiles;
    synthesis_tiling(c, C, F, halo, &tile_len, &tiles);
    const long long unsplit = synthesis_unsplit(c, C, F, halo, fused, &tile_len, &tiles);
    const size_t nph = (size_t)C * c->K;
    const size_t ncarry = tiles > 1 ? (size_t)C * tiles * c->K : 0;
    const size_t nframes = fused ? 0 : (size_t)C * F * c->W;
    ensure_ws(c, (nph + ncarry + nframes) * sizeof(R));
    R *pin = static_cast<R *>(c->ws);
    R *carry = pin + nph;
    R *frames = carry + ncarry;
    HIPCHK(hipMemcpyAsync(pin, phase_io, nph * sizeof(R), hipMemcpyDeviceToDevice, st));
    if (tiles > 1 && C > unsplit) {
        const long long chains = (C - unsplit) * c->K;
        LaunchTimer timer(c, KIND_CARRY, st);
        if constexpr (std::is_same<R, double>::value) {
            if (pitch_n != 0.0) {  // the chain runs on the increments the filter would have written
                hipLaunchKernelGGL(voc::phase_carry_pitch_kernel, dim3((unsigned)((chains + 255) / 256)), dim3(256), 0, st, c->params,
                                   inc + (size_t)unsplit * F * ldk, ldk, F, (const double *)(pin + unsplit * c->K), carry, tile_len,
                                   tiles, halo, C - unsplit, pitch_n, pitch_rcp);
                HIPCHK(hipGetLastError());
            }
        }
        if (pitch_n == 0.0) {
            hipLaunchKernelGGL(voc::phase_carry_kernel<R>, dim3((unsigned)((chains + 255) / 256)), dim3(256), 0, st, params_of<R>(c),
                               inc + (size_t)unsplit * (F_inc ? F_inc : F) * ldk, ldk, F, (const R *)(pin + unsplit * c->K), carry,
                               tile_len, tiles, halo, C - unsplit, inc_rows, F_inc);
            HIPCHK(hipGetLastError());
        }
    }
    voc::SynArgsT<R> s{};
    s.mag = mag;
    s.inc = inc;
    s.ldk = ldk;
    s.phase_in = pin;
    s.carry = carry;
    s.phase_out = phase_io;
    s.out = y;
    s.out_stride = y_stride;
    s.out_len = y_len;
    s.F = F;
    s.tile_len = tile_len;
    s.tiles = tiles;
    s.halo = halo;
    s.unsplit = unsplit;
    s.mag_rows = mag_rows;
    s.inc_rows = inc_rows;
    s.F_mag = F_mag;
    s.F_inc = F_inc;
    s.pitch_n = pitch_n;
    s.pitch_rcp = pitch_rcp;
    s.gain = gain;
    if (filtered && !(fused && synthesis_can_filter(c))) throw StatusFail{VOC_ERR_UNSUPPORTED};
    if (fused) {
        if (!launch_synthesis_fast<R>(c, s, C, true, st)) launch_synthesis<voc::SYN_IN_POLAR, voc::SYN_OUT_OLA, R>(c, s, C, st);
    } else {
        s.out = frames;
        if (!launch_synthesis_fast<R>(c, s, C, false, st)) launch_synthesis<voc::SYN_IN_POLAR, voc::SYN_OUT_FRAMES, R>(c, s, C, st);
        const long long total = C * y_len;
        hipLaunchKernelGGL(voc::sum_frames_kernel<R>, dim3(ew_grid(total)), dim3(256), 0, st, (const R *)frames, F, c->W, c->H, y,
                           y_stride, y_len, total);
        HIPCHK(hipGetLastError());
    }
}

template <class R = double>
void phase_advance_dev(voc_ctx *c, const R *inc, long long ldk, long long F, long long C, R *phase_io,
                       hipStream_t st, long long rpc = 0)
{
    if (C <= 0 || F <= 0) return;
    const long long chains = C * c->K;
    LaunchTimer timer(c, KIND_CARRY, st);
    hipLaunchKernelGGL(voc::phase_advance_kernel<R>, dim3((unsigned)((chains + 255) / 256)), dim3(256), 0, st, params_of<R>(c), inc,
                       ldk, F, phase_io, C, rpc ? rpc : F);
    HIPCHK(hipGetLastError());
}

// Workgroups of the specialised synthesis kernel the chip holds at once (0: the generic kernel runs).
long long synthesis_slots(const voc_ctx *c)
{
    if (!fast_sizes(c) && fast16_hop_slots(c, true) == 0) return 0;
    if (c->syn_slots_override > 0) return c->syn_slots_override;  // VOC_SYN_SLOTS: lets small tests reach this path
    if (fast16_hop_slots(c, true) > 0) {  // sixteen points per thread: one frame of LDS and two (float: three) waves per SIMD
        const bool f32_16 = c->precision == VOC_PRECISION_F32;
        const long long t16 = c->M / 16, by_lds16 = c->lds_per_cu / ((long long)c->M * (f32_16 ? 8 : 16)),
                        by_regs16 = (f32_16 && c->log2m < 13 ? 4 * VOC_F32_SYN16_WAVES : 8) * 64 / t16;
        const long long per_cu16 = by_lds16 < by_regs16 ? by_lds16 : by_regs16;
        return c->cus * (per_cu16 < 1 ? 1 : per_cu16);
    }
    const bool f32 = c->precision == VOC_PRECISION_F32;
    const long long t = c->M / 8;
    const bool fat = VOC_SYN_FAT != 0 && c->log2m >= 10 && !f32;  // voc::syn_fat: two waves per SIMD instead of three
    const bool pingpong = fat && c->log2m >= 11 && voc::syn_pingpong<11, true, double>();  // voc::syn_lds_bytes
    const long long lds = pingpong ? (long long)(2 * c->M + 512) * 16 : (long long)(c->M * 2 + ((c->K + 1) & ~1)) * (f32 ? 4 : 8);
    long long by_lds = c->lds_per_cu / lds, by_regs = (f32 ? 16 : (fat ? 8 : 12)) * 64 / t;
    long long per_cu = by_lds < by_regs ? by_lds : by_regs;
    if (per_cu < 1) per_cu = 1;
    return c->cus * per_cu;
}

// With more channels than the chip holds workgroups, whole rounds of channels need no cut at all:
// they run as one tile each, first, and only the channels of the last, partial round are cut into
// tiles (and need the carry pass).  Returns the number of uncut leading channels and, for the
// others, the tiling.
long long synthesis_unsplit(const voc_ctx *c, long long channels, long long F, int halo, bool fused, int *tile_len, int *tiles)
{
    const long long slots = synthesis_slots(c);
    if (c->syn_tile_override > 0 || slots == 0 || !fused) return 0;
    if (channels < slots) {
        // one nearly full round: cutting would buy little and cost the carry pass
        if (channels * 20 < slots * 17) return 0;
        *tile_len = (int)F;
        *tiles = 1;
        return channels;
    }
    const long long unsplit = channels / slots * slots, rest = channels - unsplit;
    if (rest == 0) {
        *tile_len = (int)F;
        *tiles = 1;
        return channels;
    }
    long long min_tile = 8LL * (halo + 1);
    if (min_tile < 16) min_tile = 16;
    long long t = slots / rest, max_tiles = F / min_tile;
    if (t > max_tiles) t = max_tiles;
    if (t <= 1) {
        *tile_len = (int)F;
        *tiles = 1;
        return channels;
    }
    long long tl = (F + t - 1) / t;
    if (tl <= halo) tl = halo + 1;
    *tile_len = (int)tl;
    *tiles = (int)((F + tl - 1) / tl);
    return unsplit;
}

struct RtPlan {
    long long F, out_len, ldk;
    int ana_tile, ana_tiles, syn_tile, syn_tiles, halo;
    long long syn_unsplit;     // leading channels that run as one synthesis tile each
    bool fused_ola;            // overlap-add ring fits in LDS beside the FFT buffer
    size_t per_channel_bytes;  // STFT + carry + two phase snapshots (+ frames when not fused)
};

// ola16: the overlap-add of this call can run in the registers of the sixteen-point synthesis kernel (a frame
// length whose ring does not fit in LDS is then fused all the same)
RtPlan roundtrip_plan(const voc_ctx *c, long long len, long long group, long long chunk, bool ola16 = false)
{
    RtPlan r{};
    r.F = num_frames(c, len);
    r.out_len = ola_length(c, r.F, chunk > 0 ? chunk : c->H);
    r.ldk = (c->K + 1) & ~1LL;
    const bool f32 = c->precision == VOC_PRECISION_F32;
    r.fused_ola = !c->bluestein && ((f32 ? lds_bytes<float>(c, true) : lds_bytes<double>(c, true)) <= kLdsLimit || ola16);
    r.halo = r.fused_ola ? (c->W + c->H - 1) / c->H - 1 : 0;
    if (r.F > 0) {
        r.ana_tile = analysis_tile_len(c, group, r.F);
        r.ana_tiles = (int)((r.F + r.ana_tile - 1) / r.ana_tile);
        synthesis_tiling(c, group, r.F, r.halo, &r.syn_tile, &r.syn_tiles);
        r.syn_unsplit = synthesis_unsplit(c, group, r.F, r.halo, r.fused_ola, &r.syn_tile, &r.syn_tiles);
    }
    r.per_channel_bytes = (size_t)(2 * r.F * r.ldk + (long long)(r.syn_tiles > 1 ? r.syn_tiles : 0) * c->K + 2LL * c->K +
                                   (r.fused_ola ? 0 : r.F * c->W)) *
                          (f32 ? sizeof(float) : sizeof(double));
    return r;
}

template <class R>
void roundtrip_dev(voc_ctx *c, const R *x, long long x_stride, long long len, long long C, long long chunk,
                   R *y, long long y_stride, R *ana_phase, R *syn_phase, R *leftover,
                   long long leftover_stride, void *ws, long long ws_bytes, hipStream_t st)
{
    if (C <= 0) return;
    const long long F = num_frames(c, len);
    const long long keep = leftover_length(c, len);
    if (leftover && keep > 0)
        HIPCHK(hipMemcpy2DAsync(leftover, (size_t)leftover_stride * sizeof(R), x + (len - keep),
                                (size_t)x_stride * sizeof(R), (size_t)keep * sizeof(R), (size_t)C,
                                hipMemcpyDeviceToDevice, st));
    if (F <= 0) return;

    const bool ola16 = fast16_hop_slots(c, true) > 0;
    // channels per pass: as many as the workspace holds
    long long group = C;
    std::optional<WsUse> own;  // the library's own workspace is shared between calls; a caller's is the caller's to order
    // (the chirp-z transform of long frames runs through one more buffer of the context, big_ws, whoever brings
    //  the workspace: calls on two streams are ordered for it all the same)
    if (!ws || c->blue_big) own.emplace(c, st);
    if (!ws) {
        size_t cap = (size_t)160 << 30;
        const int cap_gb = env_int("VOC_WS_CAP_GB");
        if (cap_gb > 0) cap = (size_t)cap_gb << 30;
        while (group > 1 && roundtrip_plan(c, len, group, chunk, ola16).per_channel_bytes * (size_t)group > cap)
            group = (group + 1) / 2;
        ensure_ws(c, roundtrip_plan(c, len, group, chunk, ola16).per_channel_bytes * (size_t)group);
        ws = c->ws;
    } else {
        while (group > 1 && roundtrip_plan(c, len, group, chunk, ola16).per_channel_bytes * (size_t)group > (size_t)ws_bytes)
            group = (group + 1) / 2;
        if (roundtrip_plan(c, len, group, chunk, ola16).per_channel_bytes * (size_t)group > (size_t)ws_bytes)
            throw StatusFail{VOC_ERR_SIZE};  // workspace smaller than one channel needs
    }

    for (long long c0 = 0; c0 < C; c0 += group) {
        const long long G = (C - c0 < group) ? C - c0 : group;
        const RtPlan r = roundtrip_plan(c, len, group, chunk, ola16);
        R *mag = static_cast<R *>(ws);
        R *inc = mag + (size_t)G * F * r.ldk;
        R *pinA = inc + (size_t)G * F * r.ldk;
        R *pinS = pinA + (size_t)G * c->K;
        R *carry = pinS + (size_t)G * c->K;
        R *frames = carry + (size_t)(r.syn_tiles > 1 ? r.syn_tiles : 0) * G * c->K;  // only when !fused_ola
        R *aph = ana_phase + c0 * c->K, *sph = syn_phase + c0 * c->K;
        HIPCHK(hipMemcpyAsync(pinA, aph, (size_t)G * c->K * sizeof(R), hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemcpyAsync(pinS, sph, (size_t)G * c->K * sizeof(R), hipMemcpyDeviceToDevice, st));

        voc::AnaArgsT<R> a{};
        a.x = x + c0 * x_stride;
        a.chan_stride = x_stride;
        a.frame_stride = c->H;  // framing fused: frame j starts at sample j*H (Frames.hs:34-35)
        a.phase_in = pinA;
        a.phase_out = aph;
        a.mag = mag;
        a.inc = inc;
        a.ldk = r.ldk;
        a.F = F;
        a.rpc = F;
        a.tile_len = r.ana_tile;
        a.tiles = r.ana_tiles;
        a.vol = (R)c->vol;  // volumeFix folded into the magnitude store (Conduit.hs:34-35)
        bool done = launch_analysis_fast<R>(c, a, G, st);
        if (!done) launch_analysis<false, false, R>(c, a, G, st);

        // (a group smaller than the planned one keeps the plan's tiling; only its uncut part shrinks)
        const long long unsplit = r.syn_unsplit < G ? r.syn_unsplit : (r.syn_tiles > 1 ? 0 : G);
        if (r.syn_tiles > 1 && G > unsplit) {
            const long long chains = (G - unsplit) * c->K;
            LaunchTimer timer(c, KIND_CARRY, st);
            hipLaunchKernelGGL(voc::phase_carry_kernel<R>, dim3((unsigned)((chains + 255) / 256)), dim3(256), 0, st,
                               params_of<R>(c), (const R *)(inc + (size_t)unsplit * F * r.ldk), r.ldk, F,
                               (const R *)(pinS + unsplit * c->K), carry, r.syn_tile, r.syn_tiles, r.halo, G - unsplit,
                           (const long long *)nullptr, 0LL);
            HIPCHK(hipGetLastError());
        }
        voc::SynArgsT<R> s{};
        s.mag = mag;
        s.inc = inc;
        s.ldk = r.ldk;
        s.phase_in = pinS;
        s.carry = carry;
        s.phase_out = sph;
        s.out = y + c0 * y_stride;
        s.out_stride = y_stride;
        s.out_len = r.out_len;
        s.F = F;
        s.tile_len = r.syn_tile;
        s.tiles = r.syn_tiles;
        s.unsplit = unsplit;
        s.halo = r.halo;
        if (r.fused_ola) {
            done = launch_synthesis_fast<R>(c, s, G, true, st);
            if (!done) launch_synthesis<voc::SYN_IN_POLAR, voc::SYN_OUT_OLA, R>(c, s, G, st);
        } else {
            // frame too long for an LDS-resident overlap-add ring: frames go through HBM once
            s.out = frames;
            done = launch_synthesis_fast<R>(c, s, G, false, st);
            if (!done) launch_synthesis<voc::SYN_IN_POLAR, voc::SYN_OUT_FRAMES, R>(c, s, G, st);
            const long long total = G * r.out_len;
            hipLaunchKernelGGL(voc::sum_frames_kernel<R>, dim3(ew_grid(total)), dim3(256), 0, st, (const R *)frames, F, c->W, c->H,
                               y + c0 * y_stride, (long long)y_stride, r.out_len, total);
            HIPCHK(hipGetLastError());
        }
    }
}

struct Locked {  // serialises calls on a context and pins its device for the calling thread
    std::lock_guard<std::mutex> g;
    // every entry point works in one precision; a context of the other one is refused
    explicit Locked(voc_ctx *c, int precision = VOC_PRECISION_F64) : g(c->mu)
    {
        if (c->precision != precision) throw StatusFail{VOC_ERR_UNSUPPORTED};
        HIPCHK(hipSetDevice(c->device));
    }
};

}  // namespace

// ================================================================================ C ABI

VOC_EXPORT const char *voc_strerror(int32_t s)
{
    switch (s) {
    case VOC_OK: return "ok";
    case VOC_ERR_INVALID: return "invalid argument";
    case VOC_ERR_UNSUPPORTED: return "unsupported frame length or precision (frame length: 2 to 32768; single precision: powers of two from 8 to 16384)";
    case VOC_ERR_HIP: return "HIP runtime call failed";
    case VOC_ERR_NOMEM: return "out of memory";
    case VOC_ERR_SIZE: return "size mismatch";
    case VOC_ERR_NO_DEVICE: return "no usable GPU";
    case VOC_ERR_SCRATCH: return "one channel of this input needs more scratch than the library's static buffer holds";
    default: return "unknown status";
    }
}

VOC_EXPORT const char *voc_last_hip_error(const voc_ctx *ctx) { return ctx ? ctx->hip_err.c_str() : ""; }

constexpr int kMaxFrameLength = 32768;  // chirp-z transform of 8 * 8192 points (voc_kernels_bluestein_big.h)

VOC_EXPORT int32_t voc_ctx_create(int32_t N, int32_t H, const double *window, int32_t W, int32_t device,
                                  int32_t precision, voc_ctx **out)
{
    if (!out) return VOC_ERR_INVALID;
    *out = nullptr;
    if (!window || W < 1 || H < 1 || N < 1) return VOC_ERR_INVALID;
    if (W > N) return VOC_ERR_INVALID;  // addZeroPadding: input longer than target length (Vocoder.hs:132)
    if (precision != VOC_PRECISION_F64 && precision != VOC_PRECISION_F32) return VOC_ERR_UNSUPPORTED;
    const bool pow2 = N >= 8 && N <= 16384 && (N & (N - 1)) == 0;
    // every other length, and the powers of two above 16384, go through the chirp-z transform
    const bool chirp = !pow2 && N >= 2 && N <= kMaxFrameLength && precision == VOC_PRECISION_F64;
    const bool chirp_big = chirp && N > 4096;  // its transform no longer fits in LDS
    if (!pow2 && !chirp) return VOC_ERR_UNSUPPORTED;
    voc_ctx *c = new (std::nothrow) voc_ctx;
    if (!c) return VOC_ERR_NOMEM;
    int32_t rc = guarded(c, [&]() -> int32_t {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return VOC_ERR_NO_DEVICE;
        if (device < 0 || device >= ndev) return VOC_ERR_INVALID;
        HIPCHK(hipSetDevice(device));
        HIPCHK(hipEventCreateWithFlags(&c->ws_event, hipEventDisableTiming));
#ifndef VOC_EMU
        {   // how many workgroups the device holds at once follows from these (synthesis_slots)
            hipDeviceProp_t prop;
            HIPCHK(hipGetDeviceProperties(&prop, device));
            if (prop.multiProcessorCount > 0) c->cus = prop.multiProcessorCount;
            if (prop.maxSharedMemoryPerMultiProcessor > 0) c->lds_per_cu = (long long)prop.maxSharedMemoryPerMultiProcessor;
        }
#endif
        // every GPU holds its own copy of the static scratch array at its own address: look it up with
        // this context's device current and use only that pointer from here on
        HIPCHK(hipGetSymbolAddress(&c->static_scratch, HIP_SYMBOL(voc_static_scratch)));
        c->N = N; c->W = W; c->H = H; c->K = N / 2 + 1; c->M = N / 2; c->device = device;
        c->bluestein = chirp;
        c->blue_big = chirp_big;
        int chirp_len = 0;  // L = 2^m >= 2N - 1 complex points of the convolution
        if (chirp) {
            chirp_len = 8;
            while (chirp_len < 2 * N - 1) chirp_len *= 2;
            // the LDS-resident FFT underneath: all of it, or each of L / 8192 interleaved sequences
            c->M = chirp_big ? voc::kBigP : chirp_len;
        }
        c->precision = precision;
        c->log2m = 0;
        while ((1 << c->log2m) < c->M) ++c->log2m;
        c->window.assign(window, window + W);
        // volumeCoeff = H / sum (map (**2) window), left fold from 0 (Vocoder.hs:209-210)
        double s = 0.0;
        for (int i = 0; i < W; ++i) s = s + std::pow(window[i], 2.0);
        c->vol = (double)H / s;
        c->ana_tile_override = env_int("VOC_ANA_TILE_LEN");
        c->syn_tile_override = env_int("VOC_SYN_TILE_LEN");
        c->syn_slots_override = env_int("VOC_SYN_SLOTS");
        c->force_generic = env_int("VOC_FORCE_GENERIC") != 0;
        c->fast16_ana = !(std::getenv("VOC_FAST16_ANA") && env_int("VOC_FAST16_ANA") == 0);
        c->fast16_syn = std::getenv("VOC_FAST16_SYN") ? (env_int("VOC_FAST16_SYN") != 0 ? 1 : 0) : -1;
        if (std::getenv("VOC_FAST16_F32_ANA")) c->fast16_ana_f32 = env_int("VOC_FAST16_F32_ANA") != 0;
        if (std::getenv("VOC_FAST16_F32_SYN")) c->fast16_syn_f32 = env_int("VOC_FAST16_F32_SYN") != 0 ? 1 : 0;

        // per-pass Stockham twiddles, laid out as voc::fft_passes reads them: the pass at stride NS
        // (radix r = min(8, M/NS)) occupies entries NS-1 .. NS-1 + (r-1)*NS, [j-1][k] with k fastest,
        // value exp(-2 pi i j k / (NS r))
        std::vector<double2> twP((size_t)c->M), twS((size_t)c->M / 2 + 1);
        const long double two_pi = 2.0L * 3.141592653589793238462643383279502884L;
        for (int ns = 1; ns < c->M;) {
            const int rad = (c->M / ns) >= 8 ? 8 : (c->M / ns);
            if (ns > 1)
                for (int j = 1; j < rad; ++j)
                    for (int k = 0; k < ns; ++k) {
                        const long double ang = two_pi * (long double)((long long)j * k) / (long double)((long long)ns * rad);
                        twP[(size_t)(ns - 1) + (size_t)(j - 1) * ns + k] = make_double2((double)cosl(ang), (double)-sinl(ang));
                    }
            ns *= rad;
        }
        for (int k = 0; k <= c->M / 2; ++k) {
            const long double ang = two_pi * (long double)k / (long double)N;
            twS[k] = make_double2((double)cosl(ang), (double)-sinl(ang));
        }
        twS[c->M / 2] = make_double2(0.0, -1.0);  // exp(-i pi/2) exactly
        // the window, then the window times 1/N: with N a power of two that product is exact, and the
        // overlap-add of the specialised synthesis kernel multiplies by it instead of scaling every sample
        std::vector<double> win2(window, window + W);
        win2.resize(2 * (size_t)W);
        for (int i = 0; i < W; ++i) win2[(size_t)W + i] = window[i] * (1.0 / (double)N);
        HIPCHK(hipMalloc((void **)&c->d_window, sizeof(double) * win2.size()));
        HIPCHK(hipMalloc((void **)&c->d_twP, sizeof(double2) * twP.size()));
        HIPCHK(hipMalloc((void **)&c->d_twS, sizeof(double2) * twS.size()));
        HIPCHK(hipMemcpy(c->d_window, win2.data(), sizeof(double) * win2.size(), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(c->d_twP, twP.data(), sizeof(double2) * twP.size(), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(c->d_twS, twS.data(), sizeof(double2) * twS.size(), hipMemcpyHostToDevice));
        {   // the circle in 64 steps (fast_sincos_tab); the multiples of a quarter turn are exact
            std::vector<double2> cs64(64);
            for (int j = 0; j < 64; ++j) {
                const long double ang = two_pi * (long double)j / 64.0L;
                cs64[j] = make_double2(j % 32 == 16 ? 0.0 : (double)cosl(ang), j % 32 == 0 ? 0.0 : (double)sinl(ang));
            }
            HIPCHK(hipMalloc((void **)&c->d_cs64, sizeof(double2) * cs64.size()));
            HIPCHK(hipMemcpy(c->d_cs64, cs64.data(), sizeof(double2) * cs64.size(), hipMemcpyHostToDevice));
        }

        voc::Params &p = c->params;
        p.N = N; p.W = W; p.H = H; p.K = c->K; p.M = c->M;
        const int diff = N - W;
        p.hd = diff - diff / 2;
        p.off = diff / 2;
        p.hop_pow2 = (H & (H - 1)) == 0;
        p.hop_d = (double)H;
        p.inv_hop = 1.0 / (double)H;
        p.inv_n = 1.0 / (double)N;
        p.omega1 = (p.hop_pow2 && !chirp) ? (voc::kTwoPi * (double)H) / (double)N : 0.0;  // exact scalings of 2 pi
        p.window = c->d_window;
        p.window_n = c->d_window + W;
        p.twP = c->d_twP;
        p.twS = c->d_twS;
        p.cs64 = c->d_cs64;
        if (chirp) {
            const int L = chirp_len;
            const long double pi_l = 3.141592653589793238462643383279502884L;
            std::vector<double2> w((size_t)N);
            std::vector<std::complex<long double>> bl((size_t)L, std::complex<long double>(0.0L, 0.0L));
            for (int n = 0; n < N; ++n) {
                const long long r = ((long long)n * n) % (2LL * N);  // n^2 mod 2N keeps the angle small
                const long double ang = pi_l * (long double)r / (long double)N;
                const std::complex<long double> v(cosl(ang), sinl(ang));
                w[n] = make_double2((double)v.real(), (double)v.imag());
                bl[n] = v;
                if (n > 0) bl[L - n] = v;
            }
            // FFT of b in long double (iterative radix 2)
            for (int i = 1, j = 0; i < L; ++i) {
                int bit = L >> 1;
                for (; j & bit; bit >>= 1) j ^= bit;
                j ^= bit;
                if (i < j) std::swap(bl[i], bl[j]);
            }
            for (int len = 2; len <= L; len <<= 1) {
                for (int i = 0; i < L; i += len)
                    for (int k = 0; k < len / 2; ++k) {
                        const long double ang = -2.0L * pi_l * (long double)k / (long double)len;
                        const std::complex<long double> t = std::complex<long double>(cosl(ang), sinl(ang)) * bl[i + k + len / 2];
                        const std::complex<long double> u = bl[i + k];
                        bl[i + k] = u + t;
                        bl[i + k + len / 2] = u - t;
                    }
            }
            std::vector<double2> bf((size_t)L);
            for (int m = 0; m < L; ++m) bf[m] = make_double2((double)bl[m].real(), (double)bl[m].imag());
            HIPCHK(hipMalloc((void **)&c->d_chirp, sizeof(double2) * w.size()));
            HIPCHK(hipMalloc((void **)&c->d_bfft, sizeof(double2) * bf.size()));
            HIPCHK(hipMemcpy(c->d_chirp, w.data(), sizeof(double2) * w.size(), hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(c->d_bfft, bf.data(), sizeof(double2) * bf.size(), hipMemcpyHostToDevice));
            c->blue.chirp = c->d_chirp;
            c->blue.bfft = c->d_bfft;
            c->blue.inv_l = 1.0 / (double)L;
            c->blue.n_d = (double)N;
            if (chirp_big) {
                std::vector<double2> wl((size_t)L);
                for (int m = 0; m < L; ++m) {
                    const long double ang = two_pi * (long double)m / (long double)L;
                    wl[m] = make_double2((double)cosl(ang), (double)-sinl(ang));
                }
                HIPCHK(hipMalloc((void **)&c->d_wl, sizeof(double2) * wl.size()));
                HIPCHK(hipMemcpy(c->d_wl, wl.data(), sizeof(double2) * wl.size(), hipMemcpyHostToDevice));
                c->big.chirp = c->d_chirp;
                c->big.bfft = c->d_bfft;
                c->big.wl = c->d_wl;
                c->big.inv_l = c->blue.inv_l;
                c->big.n_d = c->blue.n_d;
                c->big.L1 = L / voc::kBigP;
            }
        }
        if (precision == VOC_PRECISION_F32) {
            // the same tables rounded once to float (twiddles from the long double values, not from the doubles)
            std::vector<float> wf(window, window + W);
            wf.resize(2 * (size_t)W);
            for (int i = 0; i < W; ++i) wf[(size_t)W + i] = wf[i] * (1.0f / (float)N);
            std::vector<float2> tpf(twP.size()), tsf(twS.size());
            for (int ns = 1; ns < c->M;) {
                const int rad = (c->M / ns) >= 8 ? 8 : (c->M / ns);
                if (ns > 1)
                    for (int j = 1; j < rad; ++j)
                        for (int k = 0; k < ns; ++k) {
                            const long double ang = two_pi * (long double)((long long)j * k) / (long double)((long long)ns * rad);
                            tpf[(size_t)(ns - 1) + (size_t)(j - 1) * ns + k] = make_float2((float)cosl(ang), (float)-sinl(ang));
                        }
                ns *= rad;
            }
            for (int k = 0; k <= c->M / 2; ++k) {
                const long double ang = two_pi * (long double)k / (long double)N;
                tsf[k] = make_float2((float)cosl(ang), (float)-sinl(ang));
            }
            tsf[c->M / 2] = make_float2(0.0f, -1.0f);
            HIPCHK(hipMalloc((void **)&c->d_window_f, sizeof(float) * wf.size()));
            HIPCHK(hipMalloc((void **)&c->d_twP_f, sizeof(float2) * tpf.size()));
            HIPCHK(hipMalloc((void **)&c->d_twS_f, sizeof(float2) * tsf.size()));
            HIPCHK(hipMemcpy(c->d_window_f, wf.data(), sizeof(float) * wf.size(), hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(c->d_twP_f, tpf.data(), sizeof(float2) * tpf.size(), hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(c->d_twS_f, tsf.data(), sizeof(float2) * tsf.size(), hipMemcpyHostToDevice));
            voc::ParamsT<float> &q = c->params_f;
            q.N = N; q.W = W; q.H = H; q.K = c->K; q.M = c->M;
            q.hd = p.hd; q.off = p.off; q.hop_pow2 = p.hop_pow2; q.omega1 = p.omega1;
            q.om1_hi = (float)p.omega1;
            q.om1_lo = (float)(p.omega1 - (double)q.om1_hi);
            q.turn_mask = (p.omega1 != 0.0 && N >= H) ? N / H - 1 : 0;   // H <= N here: both powers of two
            q.hop_d = (float)H;
            q.inv_hop = 1.0f / (float)H;
            q.inv_n = 1.0f / (float)N;
            q.window = c->d_window_f;
            q.window_n = c->d_window_f + W;
            q.twP = c->d_twP_f;
            q.twS = c->d_twS_f;
        }
        return VOC_OK;
    });
    if (rc != VOC_OK) {
        voc_ctx_destroy(c);
        return rc;
    }
    *out = c;
    return VOC_OK;
}

VOC_EXPORT int32_t voc_ctx_destroy(voc_ctx *c)
{
    if (!c) return VOC_OK;
    (void)hipSetDevice(c->device);
    if (c->ws) { (void)hipDeviceSynchronize(); (void)hipFree(c->ws); }
    if (c->ws_event) (void)hipEventDestroy(c->ws_event);
    c->staging.release();
    if (c->d_window) (void)hipFree(c->d_window);
    if (c->d_twP) (void)hipFree(c->d_twP);
    if (c->d_twS) (void)hipFree(c->d_twS);
    if (c->d_cs64) (void)hipFree(c->d_cs64);
    if (c->d_chirp) (void)hipFree(c->d_chirp);
    if (c->d_bfft) (void)hipFree(c->d_bfft);
    if (c->d_wl) (void)hipFree(c->d_wl);
    if (c->big_ws) (void)hipFree(c->big_ws);
    if (c->d_window_f) (void)hipFree(c->d_window_f);
    if (c->d_twP_f) (void)hipFree(c->d_twP_f);
    if (c->d_twS_f) (void)hipFree(c->d_twS_f);
    delete c;
    return VOC_OK;
}

VOC_EXPORT int32_t voc_frame_length(const voc_ctx *c) { return c ? c->N : VOC_ERR_INVALID; }
VOC_EXPORT int32_t voc_input_frame_length(const voc_ctx *c) { return c ? c->W : VOC_ERR_INVALID; }
VOC_EXPORT int32_t voc_freq_frame_length(const voc_ctx *c) { return c ? c->K : VOC_ERR_INVALID; }
VOC_EXPORT int32_t voc_hop_size(const voc_ctx *c) { return c ? c->H : VOC_ERR_INVALID; }
VOC_EXPORT double voc_volume_coeff(const voc_ctx *c) { return c ? c->vol : 0.0; }
VOC_EXPORT int32_t voc_window(const voc_ctx *c, double *out)
{
    if (!c || !out) return VOC_ERR_INVALID;
    std::memcpy(out, c->window.data(), sizeof(double) * (size_t)c->W);
    return VOC_OK;
}

VOC_EXPORT int64_t voc_num_frames(const voc_ctx *c, int64_t len) { return c ? num_frames(c, len) : VOC_ERR_INVALID; }
VOC_EXPORT int64_t voc_leftover_length(const voc_ctx *c, int64_t len) { return c ? leftover_length(c, len) : VOC_ERR_INVALID; }
VOC_EXPORT int64_t voc_ola_length(const voc_ctx *c, int64_t F, int64_t chunk)
{
    if (!c || chunk < 1) return VOC_ERR_INVALID;
    return ola_length(c, F, chunk);
}

// ---- FFT-compatible layer ---------------------------------------------------------------------------

namespace {
template <class R> constexpr int precision_of() { return std::is_same<R, float>::value ? VOC_PRECISION_F32 : VOC_PRECISION_F64; }

template <bool RAW, class R = double>
int32_t spec_forward(voc_ctx *c, const R *in, R *out, int64_t B)
{
    if (!c || B < 0 || (B > 0 && (!in || !out))) return VOC_ERR_INVALID;
    if (B == 0) return VOC_OK;
    return guarded(c, [&]() -> int32_t {
        Locked lk(c, precision_of<R>());
        const size_t in_len = RAW ? (size_t)c->N : (size_t)c->W;
        HostCall h(c->staging, nullptr);
        const int din = h.add((size_t)B * in_len * sizeof(R)), dout = h.add((size_t)B * c->K * 2 * sizeof(R));
        h.commit();
        h.upload(din, in);
        h.uploads_done();
        voc::AnaArgsT<R> a{};
        a.x = h.dev<R>(din);
        a.chan_stride = 0;
        a.frame_stride = (long long)in_len;
        a.spec = h.dev<R>(dout);
        a.F = B;
        a.rpc = B;
        a.tile_len = 4;
        a.tiles = (int)((B + 3) / 4);
        a.vol = (R)1;
        launch_analysis<RAW, true, R>(c, a, 1, nullptr);
        h.download(dout, out);
        h.finish();
        return VOC_OK;
    });
}

template <int OUT_MODE, class R = double>
int32_t spec_backward(voc_ctx *c, const R *in, R *out, int64_t B)
{
    if (!c || B < 0 || (B > 0 && (!in || !out))) return VOC_ERR_INVALID;
    if (B == 0) return VOC_OK;
    return guarded(c, [&]() -> int32_t {
        Locked lk(c, precision_of<R>());
        const size_t out_len = OUT_MODE == voc::SYN_OUT_RAW ? (size_t)c->N : (size_t)c->W;
        HostCall h(c->staging, nullptr);
        const int din = h.add((size_t)B * c->K * 2 * sizeof(R)), dout = h.add((size_t)B * out_len * sizeof(R));
        h.commit();
        h.upload(din, in);
        h.uploads_done();
        voc::SynArgsT<R> a{};
        a.spec = h.dev<R>(din);
        a.out = h.dev<R>(dout);
        a.F = B;
        a.tile_len = 4;
        a.tiles = (int)((B + 3) / 4);
        launch_synthesis<voc::SYN_IN_SPEC, OUT_MODE, R>(c, a, 1, nullptr);
        h.download(dout, out);
        h.finish();
        return VOC_OK;
    });
}

// analysisStep / synthesisStep / addFrames on B frames of a context, host arrays
template <class R>
int32_t analysis_step_host(voc_ctx *c, const R *prev_ph, const R *spec, R *ph_out, R *mag, R *ph_inc, int64_t B)
{
    if (!c || B < 0 || (B > 0 && (!prev_ph || !spec || !ph_out || !mag || !ph_inc))) return VOC_ERR_INVALID;
    if (B == 0) return VOC_OK;
    return guarded(c, [&]() -> int32_t {
        Locked lk(c, precision_of<R>());
        const size_t n = (size_t)B * c->K, nb = n * sizeof(R);
        HostCall h(c->staging, nullptr);
        const int dprev = h.add(nb), dspec = h.add(2 * nb), dph = h.add(nb), dmag = h.add(nb), dinc = h.add(nb);
        h.commit();
        h.upload(dprev, prev_ph);
        h.upload(dspec, spec);
        h.uploads_done();
        hipLaunchKernelGGL(voc::analysis_step_kernel<R>, dim3(ew_grid((long long)n)), dim3(256), 0, nullptr, params_of<R>(c),
                           h.dev<R>(dprev), h.dev<R>(dspec), h.dev<R>(dph), h.dev<R>(dmag), h.dev<R>(dinc), (long long)n);
        HIPCHK(hipGetLastError());
        h.download(dph, ph_out);
        h.download(dmag, mag);
        h.download(dinc, ph_inc);
        h.finish();
        return VOC_OK;
    });
}

template <class R>
int32_t synthesis_step_host(voc_ctx *c, const R *ph, const R *mag, const R *ph_inc, R *ph_out, R *spec, int64_t B)
{
    if (!c || B < 0 || (B > 0 && (!ph || !mag || !ph_inc || !ph_out || !spec))) return VOC_ERR_INVALID;
    if (B == 0) return VOC_OK;
    return guarded(c, [&]() -> int32_t {
        Locked lk(c, precision_of<R>());
        const size_t n = (size_t)B * c->K, nb = n * sizeof(R);
        HostCall h(c->staging, nullptr);
        const int dph = h.add(nb), dmag = h.add(nb), dinc = h.add(nb), dout = h.add(nb), dspec = h.add(2 * nb);
        h.commit();
        h.upload(dph, ph);
        h.upload(dmag, mag);
        h.upload(dinc, ph_inc);
        h.uploads_done();
        hipLaunchKernelGGL(voc::synthesis_step_kernel<R>, dim3(ew_grid((long long)n)), dim3(256), 0, nullptr, params_of<R>(c),
                           h.dev<R>(dph), h.dev<R>(dmag), h.dev<R>(dinc), h.dev<R>(dout), h.dev<R>(dspec), (long long)n);
        HIPCHK(hipGetLastError());
        h.download(dout, ph_out);
        h.download(dspec, spec);
        h.finish();
        return VOC_OK;
    });
}

template <class R>
int32_t add_frames_host(voc_ctx *c, const R *mag1, const R *inc1, const R *mag2, const R *inc2, R *mag_out, R *inc_out, int64_t B)
{
    if (!c || B < 0 || (B > 0 && (!mag1 || !inc1 || !mag2 || !inc2 || !mag_out || !inc_out))) return VOC_ERR_INVALID;
    if (B == 0) return VOC_OK;
    return guarded(c, [&]() -> int32_t {
        Locked lk(c, precision_of<R>());
        const size_t n = (size_t)B * c->K, nb = n * sizeof(R);
        HostCall h(c->staging, nullptr);
        const int a = h.add(nb), b = h.add(nb), d = h.add(nb), e = h.add(nb), mo = h.add(nb), io = h.add(nb);
        h.commit();
        h.upload(a, mag1);
        h.upload(b, inc1);
        h.upload(d, mag2);
        h.upload(e, inc2);
        h.uploads_done();
        hipLaunchKernelGGL(voc::add_frames_kernel<R>, dim3(ew_grid((long long)n)), dim3(256), 0, nullptr, h.dev<R>(a), h.dev<R>(b),
                           h.dev<R>(d), h.dev<R>(e), h.dev<R>(mo), h.dev<R>(io), (long long)n);
        HIPCHK(hipGetLastError());
        h.download(mo, mag_out);
        h.download(io, inc_out);
        h.finish();
        return VOC_OK;
    });
}
}  // namespace

VOC_EXPORT int32_t voc_r2c(voc_ctx *c, const double *in, double *out, int64_t B) { return spec_forward<true>(c, in, out, B); }
VOC_EXPORT int32_t voc_do_fft(voc_ctx *c, const double *fr, double *spec, int64_t B) { return spec_forward<false>(c, fr, spec, B); }
VOC_EXPORT int32_t voc_c2r(voc_ctx *c, const double *in, double *out, int64_t B) { return spec_backward<voc::SYN_OUT_RAW>(c, in, out, B); }
VOC_EXPORT int32_t voc_do_ifft(voc_ctx *c, const double *spec, double *fr, int64_t B) { return spec_backward<voc::SYN_OUT_FRAMES>(c, spec, fr, B); }
VOC_EXPORT int32_t voc_r2c_f32(voc_ctx *c, const float *in, float *out, int64_t B) { return spec_forward<true, float>(c, in, out, B); }
VOC_EXPORT int32_t voc_do_fft_f32(voc_ctx *c, const float *fr, float *spec, int64_t B) { return spec_forward<false, float>(c, fr, spec, B); }
VOC_EXPORT int32_t voc_c2r_f32(voc_ctx *c, const float *in, float *out, int64_t B) { return spec_backward<voc::SYN_OUT_RAW, float>(c, in, out, B); }
VOC_EXPORT int32_t voc_do_ifft_f32(voc_ctx *c, const float *spec, float *fr, int64_t B) { return spec_backward<voc::SYN_OUT_FRAMES, float>(c, spec, fr, B); }

// ---- step level ---------------------------------------------------------------------------------------

VOC_EXPORT int32_t voc_analysis_step(voc_ctx *c, const double *prev_ph, const double *spec, double *ph_out, double *mag,
                                     double *ph_inc, int64_t B)
{
    return analysis_step_host<double>(c, prev_ph, spec, ph_out, mag, ph_inc, B);
}
VOC_EXPORT int32_t voc_analysis_step_f32(voc_ctx *c, const float *prev_ph, const float *spec, float *ph_out, float *mag,
                                         float *ph_inc, int64_t B)
{
    return analysis_step_host<float>(c, prev_ph, spec, ph_out, mag, ph_inc, B);
}

VOC_EXPORT int32_t voc_synthesis_step(voc_ctx *c, const double *ph, const double *mag, const double *ph_inc,
                                      double *ph_out, double *spec, int64_t B)
{
    return synthesis_step_host<double>(c, ph, mag, ph_inc, ph_out, spec, B);
}
VOC_EXPORT int32_t voc_synthesis_step_f32(voc_ctx *c, const float *ph, const float *mag, const float *ph_inc,
                                          float *ph_out, float *spec, int64_t B)
{
    return synthesis_step_host<float>(c, ph, mag, ph_inc, ph_out, spec, B);
}

VOC_EXPORT int32_t voc_add_frames(voc_ctx *c, const double *mag1, const double *inc1, const double *mag2,
                                  const double *inc2, double *mag_out, double *inc_out, int64_t B)
{
    return add_frames_host<double>(c, mag1, inc1, mag2, inc2, mag_out, inc_out, B);
}
VOC_EXPORT int32_t voc_add_frames_f32(voc_ctx *c, const float *mag1, const float *inc1, const float *mag2,
                                      const float *inc2, float *mag_out, float *inc_out, int64_t B)
{
    return add_frames_host<float>(c, mag1, inc1, mag2, inc2, mag_out, inc_out, B);
}

// ---- the reference's functions that take no VocoderParams ----------------------------------------------
//
// analysisStep, synthesisStep, frameFromComplex, frameToComplex and addFrames (Vocoder.hs:158,186,213,217,
// 221) are functions of their vectors (and of hop / length where given) alone.  These forms take no
// context: they run on the calling thread's current HIP device, share one staging buffer, and
// serialise among themselves.

namespace {
std::mutex g_free_mu;
Staging g_free_staging;  // lives until the process ends
int g_free_device = -1;  // the device g_free_staging's buffers belong to

// Serialises the context-free entry points and keeps their staging on the calling thread's current
// device (a thread that has switched devices gets a fresh staging there: device memory of one GPU is
// not assumed to be reachable from another).
struct FreeCall {
    std::lock_guard<std::mutex> g;
    FreeCall() : g(g_free_mu)
    {
        int dev = 0;
        HIPCHK(hipGetDevice(&dev));
        if (dev != g_free_device) {
            if (g_free_device >= 0 && (g_free_staging.dev || g_free_staging.pin)) {
                HIPCHK(hipSetDevice(g_free_device));
                (void)hipDeviceSynchronize();
                g_free_staging.release();
                HIPCHK(hipSetDevice(dev));
            }
            g_free_device = dev;
        }
    }
};

voc::Params step_params(int hop, int n, long long k)
{
    voc::Params p{};
    p.N = n;
    p.H = hop;
    p.K = (int)k;
    p.hop_pow2 = (hop & (hop - 1)) == 0;
    p.hop_d = (double)hop;
    p.inv_hop = 1.0 / (double)hop;
    p.inv_n = 1.0 / (double)n;
    p.omega1 = (p.hop_pow2 && (n & (n - 1)) == 0) ? (voc::kTwoPi * (double)hop) / (double)n : 0.0;
    return p;
}
}  // namespace

VOC_EXPORT int32_t voc_analysis_step_hn(int32_t hop, int32_t n, int64_t K, const double *prev_ph, const double *spec,
                                        double *ph_out, double *mag, double *ph_inc, int64_t B)
{
    if (hop < 1 || n < 1 || K < 0 || K > INT32_MAX || B < 0) return VOC_ERR_INVALID;
    if (B * K > 0 && (!prev_ph || !spec || !ph_out || !mag || !ph_inc)) return VOC_ERR_INVALID;
    if (B * K == 0) return VOC_OK;
    return guarded(nullptr, [&]() -> int32_t {
        FreeCall fc;
        const size_t cnt = (size_t)B * (size_t)K, nb = cnt * sizeof(double);
        HostCall h(g_free_staging, nullptr);
        const int dprev = h.add(nb), dspec = h.add(2 * nb), dph = h.add(nb), dmag = h.add(nb), dinc = h.add(nb);
        h.commit();
        h.upload(dprev, prev_ph);
        h.upload(dspec, spec);
        h.uploads_done();
        hipLaunchKernelGGL(voc::analysis_step_kernel<double>, dim3(ew_grid((long long)cnt)), dim3(256), 0, nullptr,
                           step_params(hop, n, K), h.dev(dprev), h.dev(dspec), h.dev(dph), h.dev(dmag), h.dev(dinc),
                           (long long)cnt);
        HIPCHK(hipGetLastError());
        h.download(dph, ph_out);
        h.download(dmag, mag);
        h.download(dinc, ph_inc);
        h.finish();
        return VOC_OK;
    });
}

VOC_EXPORT int32_t voc_synthesis_step_h(int32_t hop, int64_t K, const double *ph, const double *mag, const double *ph_inc,
                                        double *ph_out, double *spec, int64_t B)
{
    if (hop < 1 || K < 0 || K > INT32_MAX || B < 0) return VOC_ERR_INVALID;
    if (B * K > 0 && (!ph || !mag || !ph_inc || !ph_out || !spec)) return VOC_ERR_INVALID;
    if (B * K == 0) return VOC_OK;
    return guarded(nullptr, [&]() -> int32_t {
        FreeCall fc;
        const size_t cnt = (size_t)B * (size_t)K, nb = cnt * sizeof(double);
        HostCall h(g_free_staging, nullptr);
        const int dph = h.add(nb), dmag = h.add(nb), dinc = h.add(nb), dout = h.add(nb), dspec = h.add(2 * nb);
        h.commit();
        h.upload(dph, ph);
        h.upload(dmag, mag);
        h.upload(dinc, ph_inc);
        h.uploads_done();
        hipLaunchKernelGGL(voc::synthesis_step_kernel<double>, dim3(ew_grid((long long)cnt)), dim3(256), 0, nullptr,
                           step_params(hop, 2, K), h.dev(dph), h.dev(dmag), h.dev(dinc), h.dev(dout), h.dev(dspec),
                           (long long)cnt);
        HIPCHK(hipGetLastError());
        h.download(dout, ph_out);
        h.download(dspec, spec);
        h.finish();
        return VOC_OK;
    });
}

VOC_EXPORT int32_t voc_frame_from_complex(const double *spec, double *mag, double *ph, int64_t count)
{
    if (count < 0 || (count > 0 && (!spec || !mag || !ph))) return VOC_ERR_INVALID;
    if (count == 0) return VOC_OK;
    return guarded(nullptr, [&]() -> int32_t {
        FreeCall fc;
        const size_t nb = (size_t)count * sizeof(double);
        HostCall h(g_free_staging, nullptr);
        const int dspec = h.add(2 * nb), dmag = h.add(nb), dph = h.add(nb);
        h.commit();
        h.upload(dspec, spec);
        h.uploads_done();
        hipLaunchKernelGGL(voc::frame_from_complex_kernel, dim3(ew_grid((long long)count)), dim3(256), 0, nullptr,
                           h.dev(dspec), h.dev(dmag), h.dev(dph), (long long)count);
        HIPCHK(hipGetLastError());
        h.download(dmag, mag);
        h.download(dph, ph);
        h.finish();
        return VOC_OK;
    });
}

VOC_EXPORT int32_t voc_frame_to_complex(const double *mag, const double *ph, double *spec, int64_t count)
{
    if (count < 0 || (count > 0 && (!spec || !mag || !ph))) return VOC_ERR_INVALID;
    if (count == 0) return VOC_OK;
    return guarded(nullptr, [&]() -> int32_t {
        FreeCall fc;
        const size_t nb = (size_t)count * sizeof(double);
        HostCall h(g_free_staging, nullptr);
        const int dmag = h.add(nb), dph = h.add(nb), dspec = h.add(2 * nb);
        h.commit();
        h.upload(dmag, mag);
        h.upload(dph, ph);
        h.uploads_done();
        hipLaunchKernelGGL(voc::frame_to_complex_kernel, dim3(ew_grid((long long)count)), dim3(256), 0, nullptr,
                           h.dev(dmag), h.dev(dph), h.dev(dspec), (long long)count);
        HIPCHK(hipGetLastError());
        h.download(dspec, spec);
        h.finish();
        return VOC_OK;
    });
}

VOC_EXPORT int32_t voc_add_frames_n(const double *mag1, const double *inc1, const double *mag2, const double *inc2,
                                    double *mag_out, double *inc_out, int64_t count)
{
    if (count < 0 || (count > 0 && (!mag1 || !inc1 || !mag2 || !inc2 || !mag_out || !inc_out))) return VOC_ERR_INVALID;
    if (count == 0) return VOC_OK;
    return guarded(nullptr, [&]() -> int32_t {
        FreeCall fc;
        const size_t nb = (size_t)count * sizeof(double);
        HostCall h(g_free_staging, nullptr);
        const int a = h.add(nb), b = h.add(nb), d = h.add(nb), e = h.add(nb), mo = h.add(nb), io = h.add(nb);
        h.commit();
        h.upload(a, mag1);
        h.upload(b, inc1);
        h.upload(d, mag2);
        h.upload(e, inc2);
        h.uploads_done();
        hipLaunchKernelGGL(voc::add_frames_kernel<double>, dim3(ew_grid((long long)count)), dim3(256), 0, nullptr, h.dev(a),
                           h.dev(b), h.dev(d), h.dev(e), h.dev(mo), h.dev(io), (long long)count);
        HIPCHK(hipGetLastError());
        h.download(mo, mag_out);
        h.download(io, inc_out);
        h.finish();
        return VOC_OK;
    });
}

VOC_EXPORT int32_t voc_add_frames_dev(voc_ctx *c, const double *mag1, const double *inc1, const double *mag2,
                                      const double *inc2, double *mag_out, double *inc_out, int64_t count, void *stream)
{
    if (!c || count < 0 || (count > 0 && (!mag1 || !inc1 || !mag2 || !inc2 || !mag_out || !inc_out))) return VOC_ERR_INVALID;
    if (count == 0) return VOC_OK;
    return guarded(c, [&]() -> int32_t {
        Locked lk(c);
        hipLaunchKernelGGL(voc::add_frames_kernel<double>, dim3(ew_grid((long long)count)), dim3(256), 0, (hipStream_t)stream, mag1,
                           inc1, mag2, inc2, mag_out, inc_out, (long long)count);
        HIPCHK(hipGetLastError());
        return VOC_OK;
    });
}

VOC_EXPORT int32_t voc_binary_frames_dev(voc_ctx *c, int32_t op, const double *mag1, const double *inc1, const double *mag2,
                                         const double *inc2, double *mag_out, double *inc_out, int64_t count, void *stream)
{
    if (op == VOC_BINARY_ADD) return voc_add_frames_dev(c, mag1, inc1, mag2, inc2, mag_out, inc_out, count, stream);
    if (op != VOC_BINARY_MULTIPLY && op != VOC_BINARY_DIVIDE) return VOC_ERR_INVALID;
    if (!c || count < 0 || (count > 0 && (!mag1 || !inc1 || !mag2 || !inc2 || !mag_out || !inc_out))) return VOC_ERR_INVALID;
    if (count == 0) return VOC_OK;
    return guarded(c, [&]() -> int32_t {
        Locked lk(c);
        if (op == VOC_BINARY_DIVIDE)
            hipLaunchKernelGGL(voc::binary_frames_kernel<true>, dim3(ew_grid((long long)count)), dim3(256), 0,
                               (hipStream_t)stream, mag1, inc1, mag2, inc2, mag_out, inc_out, (long long)count);
        else
            hipLaunchKernelGGL(voc::binary_frames_kernel<false>, dim3(ew_grid((long long)count)), dim3(256), 0,
                               (hipStream_t)stream, mag1, inc1, mag2, inc2, mag_out, inc_out, (long long)count);
        HIPCHK(hipGetLastError());
        return VOC_OK;
    });
}

// ---- stage level ------------------------------------------------------------------------------------------

VOC_EXPORT int32_t voc_analysis_stage_dev(voc_ctx *c, double *phase_io, const double *frames, double *mag,
                                          double *ph_inc, int64_t C, int64_t F, void *stream)
{
    if (!c || C < 0 || F < 0) return VOC_ERR_INVALID;
    if (C > 0 && F > 0 && (!phase_io || !frames || !mag || !ph_inc)) return VOC_ERR_INVALID;
    return guarded(c, [&]() -> int32_t {
        Locked lk(c);
        analysis_stage_dev(c, phase_io, frames, mag, ph_inc, C, F, (hipStream_t)stream);
        return VOC_OK;
    });
}

VOC_EXPORT int32_t voc_synthesis_stage_dev(voc_ctx *c, double *phase_io, const double *mag, const double *ph_inc,
                                           double *frames, int64_t C, int64_t F, void *stream)
{
    if (!c || C < 0 || F < 0) return VOC_ERR_INVALID;
    if (C > 0 && F > 0 && (!phase_io || !frames || !mag || !ph_inc)) return VOC_ERR_INVALID;
    return guarded(c, [&]() -> int32_t {
        Locked lk(c);
        synthesis_stage_dev(c, phase_io, mag, ph_inc, frames, C, F, (hipStream_t)stream);
        return VOC_OK;
    });
}

VOC_EXPORT int32_t voc_analysis_stage(voc_ctx *c, double *phase_io, const double *frames, double *mag, double *ph_inc,
                                      int64_t C, int64_t F)
{
    if (!c || C < 0 || F < 0) return VOC_ERR_INVALID;
    if (C == 0 || F == 0) return VOC_OK;  // mapAccumL over no frames returns the phase unchanged
    if (!phase_io || !frames || !mag || !ph_inc) return VOC_ERR_INVALID;
    return guarded(c, [&]() -> int32_t {
        Locked lk(c);
        const size_t nph = (size_t)C * c->K * sizeof(double), nfr = (size_t)C * F * c->W * sizeof(double),
                     nst = (size_t)C * F * c->K * sizeof(double);
        HostCall h(c->staging, nullptr);
        const int dfr = h.add(nfr), dph = h.add(nph), dmag = h.add(nst), dinc = h.add(nst);
        h.commit();
        h.upload(dfr, frames);
        h.upload(dph, phase_io);
        h.uploads_done();
        analysis_stage_dev(c, h.dev(dph), h.dev(dfr), h.dev(dmag), h.dev(dinc), C, F, nullptr);
        h.download(dph, phase_io);
        h.download(dmag, mag);
        h.download(dinc, ph_inc);
        h.finish();
        return VOC_OK;
    });
}

VOC_EXPORT int32_t voc_synthesis_stage(voc_ctx *c, double *phase_io, const double *mag, const double *ph_inc,
                                       double *frames, int64_t C, int64_t F)
{
    if (!c || C < 0 || F < 0) return VOC_ERR_INVALID;
    if (C == 0 || F == 0) return VOC_OK;
    if (!phase_io || !frames || !mag || !ph_inc) return VOC_ERR_INVALID;
    return guarded(c, [&]() -> int32_t {
        Locked lk(c);
        const size_t nph = (size_t)C * c->K * sizeof(double), nfr = (size_t)C * F * c->W * sizeof(double),
                     nst = (size_t)C * F * c->K * sizeof(double);
        HostCall h(c->staging, nullptr);
        const int dmag = h.add(nst), dinc = h.add(nst), dph = h.add(nph), dfr = h.add(nfr);
        h.commit();
        h.upload(dph, phase_io);
        h.upload(dmag, mag);
        h.upload(dinc, ph_inc);
        h.uploads_done();
        synthesis_stage_dev(c, h.dev(dph), h.dev(dmag), h.dev(dinc), h.dev(dfr), C, F, nullptr);
        h.download(dph, phase_io);
        h.download(dfr, frames);
        h.finish();
        return VOC_OK;
    });
}

// ---- framing and overlap-add --------------------------------------------------------------------------------

VOC_EXPORT int32_t voc_frames_of(voc_ctx *c, const double *x, int64_t x_stride, int64_t len, int64_t C, double *frames)
{
    if (!c || C < 0 || len < 0 || x_stride < len) return VOC_ERR_INVALID;
    const long long F = c ? num_frames(c, len) : 0;
    if (C == 0 || F == 0) return VOC_OK;
    if (!x || !frames) return VOC_ERR_INVALID;
    return guarded(c, [&]() -> int32_t {
        Locked lk(c);
        const size_t nx = (size_t)C * x_stride * sizeof(double), nfr = (size_t)C * F * c->W * sizeof(double);
        HostCall h(c->staging, nullptr);
        const int dx = h.add(nx), dfr = h.add(nfr);
        h.commit();
        h.upload(dx, x);
        h.uploads_done();
        const long long total = C * F * c->W;
        hipLaunchKernelGGL(voc::frames_of_kernel, dim3(ew_grid(total)), dim3(256), 0, nullptr, h.dev(dx),
                           (long long)x_stride, h.dev(dfr), F, c->W, c->H, total);
        HIPCHK(hipGetLastError());
        h.download(dfr, frames);
        h.finish();
        return VOC_OK;
    });
}

VOC_EXPORT int32_t voc_sum_frames(voc_ctx *c, const double *frames, int64_t F, int64_t C, int64_t chunk, double *y,
                                  int64_t y_stride)
{
    if (!c || C < 0 || F < 0 || chunk < 1) return VOC_ERR_INVALID;
    const long long out_len = ola_length(c, F, chunk);
    if (C == 0 || F == 0) return VOC_OK;
    if (!frames || !y || y_stride < out_len) return VOC_ERR_INVALID;
    return guarded(c, [&]() -> int32_t {
        Locked lk(c);
        const size_t nfr = (size_t)C * F * c->W * sizeof(double), ny = (size_t)C * y_stride * sizeof(double);
        HostCall h(c->staging, nullptr);
        const int dfr = h.add(nfr), dy = h.add(ny);
        h.commit();
        h.upload(dfr, frames);
        h.upload(dy, y);  // keeps the caller's bytes between rows
        h.uploads_done();
        const long long total = C * out_len;
        hipLaunchKernelGGL(voc::sum_frames_kernel<double>, dim3(ew_grid(total)), dim3(256), 0, nullptr, h.dev(dfr), (long long)F,
                           c->W, c->H, h.dev(dy), (long long)y_stride, out_len, total);
        HIPCHK(hipGetLastError());
        h.download(dy, y);
        h.finish();
        return VOC_OK;
    });
}

// ---- fused round trip -----------------------------------------------------------------------------------------

VOC_EXPORT int64_t voc_roundtrip_workspace_bytes(const voc_ctx *c, int64_t len, int64_t C)
{
    if (!c || len < 0 || C < 0) return VOC_ERR_INVALID;
    if (C == 0) return 0;
    return (int64_t)(roundtrip_plan(c, len, C, c->H).per_channel_bytes * (size_t)C);
}

VOC_EXPORT int32_t voc_roundtrip_dev(voc_ctx *c, const double *x, int64_t x_stride, int64_t len, int64_t C,
                                     int64_t chunk, double *y, int64_t y_stride, double *ana_phase_io,
                                     double *syn_phase_io, double *leftover, int64_t leftover_stride, void *workspace,
                                     int64_t workspace_bytes, void *stream)
{
    if (!c || C < 0 || len < 0 || chunk < 1 || x_stride < len) return VOC_ERR_INVALID;
    if (C == 0) return VOC_OK;
    const long long F = num_frames(c, len);
    if (!x || !ana_phase_io || !syn_phase_io) return VOC_ERR_INVALID;
    if (F > 0 && (!y || y_stride < ola_length(c, F, chunk))) return VOC_ERR_SIZE;
    if (leftover && leftover_stride < leftover_length(c, len)) return VOC_ERR_SIZE;
    if (workspace && workspace_bytes < 0) return VOC_ERR_INVALID;
    return guarded(c, [&]() -> int32_t {
        Locked lk(c);
        roundtrip_dev(c, x, x_stride, len, C, chunk, y, y_stride, ana_phase_io, syn_phase_io, leftover,
                      leftover_stride, workspace, workspace_bytes, (hipStream_t)stream);
        return VOC_OK;
    });
}

// voc_roundtrip_dev on the library's static scratch array: no hipMalloc / hipFree, no allocating
// library routine, no per-launch event.  Every call on one GPU shares that GPU's array.
VOC_EXPORT int32_t voc_roundtrip_static_dev(voc_ctx *c, const double *x, int64_t x_stride, int64_t len, int64_t C,
                                            int64_t chunk, double *y, int64_t y_stride, double *ana_phase_io,
                                            double *syn_phase_io, double *leftover, int64_t leftover_stride, void *stream)
{
    if (!c || C < 0 || len < 0 || chunk < 1 || x_stride < len) return VOC_ERR_INVALID;
    if (C == 0) return VOC_OK;
    const long long F = num_frames(c, len);
    if (!x || !ana_phase_io || !syn_phase_io) return VOC_ERR_INVALID;
    if (F > 0 && (!y || y_stride < ola_length(c, F, chunk))) return VOC_ERR_SIZE;
    if (leftover && leftover_stride < leftover_length(c, len)) return VOC_ERR_SIZE;
    if (!c->static_scratch) return VOC_ERR_HIP;
    if (c->blue_big) return VOC_ERR_UNSUPPORTED;  // those transforms go through a buffer the library allocates
    // the channels run in groups that fit the array; one channel alone has to fit
    if (roundtrip_plan(c, len, 1, chunk).per_channel_bytes > kStaticScratchBytes) return VOC_ERR_SCRATCH;
    return guarded(c, [&]() -> int32_t {
        Locked lk(c);
#ifndef VOC_EMU
        struct NoEvents {  // launch timing creates events: off for the duration of this call
            voc_ctx *c;
            bool was;
            explicit NoEvents(voc_ctx *c_) : c(c_), was(c_->profiling) { c->profiling = false; }
            ~NoEvents() { c->profiling = was; }
        } quiet(c);
#endif
        // nothing in the scratch layout (STFT rows, phase snapshots, tile carries) is read before this
        // call has written it, so the previous call's contents need no clearing
        roundtrip_dev(c, x, x_stride, len, C, chunk, y, y_stride, ana_phase_io, syn_phase_io, leftover,
                      leftover_stride, c->static_scratch, (long long)kStaticScratchBytes, (hipStream_t)stream);
        return VOC_OK;
    });
}

VOC_EXPORT int32_t voc_analysis_signal_dev(voc_ctx *c, const double *x, int64_t x_stride, int64_t F, int64_t C,
                                           double *phase_io, double *mag, double *ph_inc, int64_t ldk,
                                           int32_t volume_fix, void *stream)
{
    if (!c || C < 0 || F < 0) return VOC_ERR_INVALID;
    if (C == 0 || F == 0) return VOC_OK;
    if (!x || !phase_io || !mag || !ph_inc) return VOC_ERR_INVALID;
    if (ldk < c->K || x_stride < (F - 1) * (int64_t)c->H + c->W) return VOC_ERR_SIZE;
    return guarded(c, [&]() -> int32_t {
        Locked lk(c);
        analysis_signal_dev(c, x, x_stride, F, C, phase_io, mag, ph_inc, ldk, volume_fix != 0, (hipStream_t)stream);
        return VOC_OK;
    });
}

VOC_EXPORT int32_t voc_synthesis_signal_dev(voc_ctx *c, const double *mag, const double *ph_inc, int64_t ldk, int64_t F,
                                            int64_t C, double *phase_io, double *y, int64_t y_stride, int64_t y_len,
                                            void *stream)
{
    if (!c || C < 0 || F < 0) return VOC_ERR_INVALID;
    if (C == 0 || F == 0) return VOC_OK;
    if (!mag || !ph_inc || !phase_io || !y) return VOC_ERR_INVALID;
    if (ldk < c->K || y_len < (F - 1) * (int64_t)c->H + c->W || y_stride < y_len) return VOC_ERR_SIZE;
    return guarded(c, [&]() -> int32_t {
        Locked lk(c);
        synthesis_signal_dev(c, mag, ph_inc, ldk, F, C, phase_io, y, y_stride, y_len, (hipStream_t)stream);
        return VOC_OK;
    });
}

VOC_EXPORT int32_t voc_synthesis_mapped_dev(voc_ctx *c, const double *mag, int64_t mag_frames, const int64_t *mag_rows,
                                            const double *ph_inc, int64_t inc_frames, const int64_t *inc_rows, int64_t ldk,
                                            int64_t F, int64_t C, double *phase_io, double *y, int64_t y_stride,
                                            int64_t y_len, void *stream)
{
    if (!c || C < 0 || F < 0) return VOC_ERR_INVALID;
    if (C == 0 || F == 0) return VOC_OK;
    if (!mag || !ph_inc || !phase_io || !y) return VOC_ERR_INVALID;
    if (ldk < c->K || y_len < (F - 1) * (int64_t)c->H + c->W || y_stride < y_len) return VOC_ERR_SIZE;
    // an array without a map holds exactly the frames that are synthesised; a mapped one at least one row
    if ((mag_rows ? mag_frames < 1 : mag_frames != F) || (inc_rows ? inc_frames < 1 : inc_frames != F)) return VOC_ERR_SIZE;
    static_assert(sizeof(long long) == sizeof(int64_t), "row maps are passed through as long long");
    return guarded(c, [&]() -> int32_t {
        Locked lk(c);
        synthesis_signal_dev(c, mag, ph_inc, ldk, F, C, phase_io, y, y_stride, y_len, (hipStream_t)stream,
                             reinterpret_cast<const long long *>(mag_rows), mag_frames,
                             reinterpret_cast<const long long *>(inc_rows), inc_frames);
        return VOC_OK;
    });
}

VOC_EXPORT int32_t voc_phase_advance_dev(voc_ctx *c, const double *ph_inc, int64_t ldk, int64_t F, int64_t C,
                                         double *phase_io, void *stream)
{
    if (!c || C < 0 || F < 0) return VOC_ERR_INVALID;
    if (C == 0 || F == 0) return VOC_OK;
    if (!ph_inc || !phase_io) return VOC_ERR_INVALID;
    if (ldk < c->K) return VOC_ERR_SIZE;
    return guarded(c, [&]() -> int32_t {
        Locked lk(c);
        phase_advance_dev(c, ph_inc, ldk, F, C, phase_io, (hipStream_t)stream);
        return VOC_OK;
    });
}

VOC_EXPORT int32_t voc_roundtrip(voc_ctx *c, const double *x, int64_t x_stride, int64_t len, int64_t C, int64_t chunk,
                                 double *y, int64_t y_stride, double *ana_phase_io, double *syn_phase_io,
                                 double *leftover, int64_t leftover_stride)
{
    if (!c || C < 0 || len < 0 || chunk < 1 || x_stride < len) return VOC_ERR_INVALID;
    if (C == 0) return VOC_OK;
    const long long F = num_frames(c, len), out_len = ola_length(c, F, chunk), keep = leftover_length(c, len);
    if (!x || !ana_phase_io || !syn_phase_io) return VOC_ERR_INVALID;
    if (F > 0 && (!y || y_stride < out_len)) return VOC_ERR_SIZE;
    if (leftover && leftover_stride < keep) return VOC_ERR_SIZE;
    return guarded(c, [&]() -> int32_t {
        Locked lk(c);
        const size_t nx = (size_t)C * len * sizeof(double), ny = (size_t)C * out_len * sizeof(double),
                     nph = (size_t)C * c->K * sizeof(double), nl = (size_t)C * keep * sizeof(double);
        HostCall h(c->staging, nullptr);
        const int dx = h.add(nx), da = h.add(nph), ds = h.add(nph), dy = h.add(ny), dl = h.add(nl);
        h.commit();
        h.upload_rows(dx, x, (size_t)x_stride * sizeof(double), (size_t)len * sizeof(double), (size_t)C);
        h.upload(da, ana_phase_io);
        h.upload(ds, syn_phase_io);
        h.uploads_done();
        roundtrip_dev(c, h.dev(dx), len, len, C, chunk, h.dev(dy), out_len, h.dev(da), h.dev(ds),
                      leftover ? h.dev(dl) : nullptr, keep, nullptr, 0, nullptr);
        h.download_rows(dy, y, (size_t)y_stride * sizeof(double), (size_t)out_len * sizeof(double), (size_t)C);
        if (leftover) h.download_rows(dl, leftover, (size_t)leftover_stride * sizeof(double), (size_t)keep * sizeof(double), (size_t)C);
        h.download(da, ana_phase_io);
        h.download(ds, syn_phase_io);
        h.finish();
        return VOC_OK;
    });
}

// ---- fused path, single precision ---------------------------------------------------------------------

VOC_EXPORT int32_t voc_roundtrip_f32_dev(voc_ctx *c, const float *x, int64_t x_stride, int64_t len, int64_t C,
                                         int64_t chunk, float *y, int64_t y_stride, float *ana_phase_io,
                                         float *syn_phase_io, float *leftover, int64_t leftover_stride,
                                         void *workspace, int64_t workspace_bytes, void *stream)
{
    if (!c || C < 0 || len < 0 || chunk < 1 || x_stride < len) return VOC_ERR_INVALID;
    if (C == 0) return VOC_OK;
    const long long F = num_frames(c, len);
    if (!x || !ana_phase_io || !syn_phase_io) return VOC_ERR_INVALID;
    if (F > 0 && (!y || y_stride < ola_length(c, F, chunk))) return VOC_ERR_SIZE;
    if (leftover && leftover_stride < leftover_length(c, len)) return VOC_ERR_SIZE;
    if (workspace && workspace_bytes < 0) return VOC_ERR_INVALID;
    return guarded(c, [&]() -> int32_t {
        Locked lk(c, VOC_PRECISION_F32);
        roundtrip_dev<float>(c, x, x_stride, len, C, chunk, y, y_stride, ana_phase_io, syn_phase_io, leftover,
                             leftover_stride, workspace, workspace_bytes, (hipStream_t)stream);
        return VOC_OK;
    });
}

VOC_EXPORT int32_t voc_roundtrip_f32(voc_ctx *c, const float *x, int64_t x_stride, int64_t len, int64_t C, int64_t chunk,
                                     float *y, int64_t y_stride, float *ana_phase_io, float *syn_phase_io,
                                     float *leftover, int64_t leftover_stride)
{
    if (!c || C < 0 || len < 0 || chunk < 1 || x_stride < len) return VOC_ERR_INVALID;
    if (C == 0) return VOC_OK;
    const long long F = num_frames(c, len), out_len = ola_length(c, F, chunk), keep = leftover_length(c, len);
    if (!x || !ana_phase_io || !syn_phase_io) return VOC_ERR_INVALID;
    if (F > 0 && (!y || y_stride < out_len)) return VOC_ERR_SIZE;
    if (leftover && leftover_stride < keep) return VOC_ERR_SIZE;
    return guarded(c, [&]() -> int32_t {
        Locked lk(c, VOC_PRECISION_F32);
        const size_t e = sizeof(float);
        const size_t nx = (size_t)C * len * e, ny = (size_t)C * out_len * e, nph = (size_t)C * c->K * e,
                     nl = (size_t)C * keep * e;
        HostCall h(c->staging, nullptr);
        const int dx = h.add(nx), da = h.add(nph), ds = h.add(nph), dy = h.add(ny), dl = h.add(nl);
        h.commit();
        h.upload_rows(dx, x, (size_t)x_stride * e, (size_t)len * e, (size_t)C);
        h.upload(da, ana_phase_io);
        h.upload(ds, syn_phase_io);
        h.uploads_done();
        roundtrip_dev<float>(c, h.dev<float>(dx), len, len, C, chunk, h.dev<float>(dy), out_len, h.dev<float>(da),
                             h.dev<float>(ds), leftover ? h.dev<float>(dl) : nullptr, keep, nullptr, 0, nullptr);
        h.download_rows(dy, y, (size_t)y_stride * e, (size_t)out_len * e, (size_t)C);
        if (leftover) h.download_rows(dl, leftover, (size_t)leftover_stride * e, (size_t)keep * e, (size_t)C);
        h.download(da, ana_phase_io);
        h.download(ds, syn_phase_io);
        h.finish();
        return VOC_OK;
    });
}

// ---- measurement hook -----------------------------------------------------------------------------------------

// ---- single-precision forms of the stage-level and signal-level entry points ---------------------------
// (the same kernel templates on float; contexts made with VOC_PRECISION_F32)

VOC_EXPORT int32_t voc_analysis_stage_f32_dev(voc_ctx *c, float *phase_io, const float *frames, float *mag, float *ph_inc,
                                              int64_t C, int64_t F, void *stream)
{
    if (!c || C < 0 || F < 0) return VOC_ERR_INVALID;
    if (C > 0 && F > 0 && (!phase_io || !frames || !mag || !ph_inc)) return VOC_ERR_INVALID;
    return guarded(c, [&]() -> int32_t {
        Locked lk(c, VOC_PRECISION_F32);
        analysis_stage_dev<float>(c, phase_io, frames, mag, ph_inc, C, F, (hipStream_t)stream);
        return VOC_OK;
    });
}

VOC_EXPORT int32_t voc_synthesis_stage_f32_dev(voc_ctx *c, float *phase_io, const float *mag, const float *ph_inc,
                                               float *frames, int64_t C, int64_t F, void *stream)
{
    if (!c || C < 0 || F < 0) return VOC_ERR_INVALID;
    if (C > 0 && F > 0 && (!phase_io || !frames || !mag || !ph_inc)) return VOC_ERR_INVALID;
    return guarded(c, [&]() -> int32_t {
        Locked lk(c, VOC_PRECISION_F32);
        synthesis_stage_dev<float>(c, phase_io, mag, ph_inc, frames, C, F, (hipStream_t)stream);
        return VOC_OK;
    });
}

VOC_EXPORT int32_t voc_analysis_stage_f32(voc_ctx *c, float *phase_io, const float *frames, float *mag, float *ph_inc,
                                          int64_t C, int64_t F)
{
    if (!c || C < 0 || F < 0) return VOC_ERR_INVALID;
    if (C == 0 || F == 0) return VOC_OK;
    if (!phase_io || !frames || !mag || !ph_inc) return VOC_ERR_INVALID;
    return guarded(c, [&]() -> int32_t {
        Locked lk(c, VOC_PRECISION_F32);
        const size_t nph = (size_t)C * c->K * sizeof(float), nfr = (size_t)C * F * c->W * sizeof(float),
                     nst = (size_t)C * F * c->K * sizeof(float);
        HostCall h(c->staging, nullptr);
        const int dfr = h.add(nfr), dph = h.add(nph), dmag = h.add(nst), dinc = h.add(nst);
        h.commit();
        h.upload(dfr, frames);
        h.upload(dph, phase_io);
        h.uploads_done();
        analysis_stage_dev<float>(c, h.dev<float>(dph), h.dev<float>(dfr), h.dev<float>(dmag), h.dev<float>(dinc), C, F, nullptr);
        h.download(dph, phase_io);
        h.download(dmag, mag);
        h.download(dinc, ph_inc);
        h.finish();
        return VOC_OK;
    });
}

VOC_EXPORT int32_t voc_synthesis_stage_f32(voc_ctx *c, float *phase_io, const float *mag, const float *ph_inc, float *frames,
                                           int64_t C, int64_t F)
{
    if (!c || C < 0 || F < 0) return VOC_ERR_INVALID;
    if (C == 0 || F == 0) return VOC_OK;
    if (!phase_io || !frames || !mag || !ph_inc) return VOC_ERR_INVALID;
    return guarded(c, [&]() -> int32_t {
        Locked lk(c, VOC_PRECISION_F32);
        const size_t nph = (size_t)C * c->K * sizeof(float), nfr = (size_t)C * F * c->W * sizeof(float),
                     nst = (size_t)C * F * c->K * sizeof(float);
        HostCall h(c->staging, nullptr);
        const int dmag = h.add(nst), dinc = h.add(nst), dph = h.add(nph), dfr = h.add(nfr);
        h.commit();
        h.upload(dmag, mag);
        h.upload(dinc, ph_inc);
        h.upload(dph, phase_io);
        h.uploads_done();
        synthesis_stage_dev<float>(c, h.dev<float>(dph), h.dev<float>(dmag), h.dev<float>(dinc), h.dev<float>(dfr), C, F, nullptr);
        h.download(dph, phase_io);
        h.download(dfr, frames);
        h.finish();
        return VOC_OK;
    });
}

VOC_EXPORT int32_t voc_analysis_signal_f32_dev(voc_ctx *c, const float *x, int64_t x_stride, int64_t F, int64_t C,
                                               float *phase_io, float *mag, float *ph_inc, int64_t ldk, int32_t volume_fix,
                                               void *stream)
{
    if (!c || C < 0 || F < 0) return VOC_ERR_INVALID;
    if (C == 0 || F == 0) return VOC_OK;
    if (!x || !phase_io || !mag || !ph_inc) return VOC_ERR_INVALID;
    if (ldk < c->K || x_stride < (F - 1) * (int64_t)c->H + c->W) return VOC_ERR_SIZE;
    return guarded(c, [&]() -> int32_t {
        Locked lk(c, VOC_PRECISION_F32);
        analysis_signal_dev<float>(c, x, x_stride, F, C, phase_io, mag, ph_inc, ldk, volume_fix != 0, (hipStream_t)stream);
        return VOC_OK;
    });
}

VOC_EXPORT int32_t voc_synthesis_signal_f32_dev(voc_ctx *c, const float *mag, const float *ph_inc, int64_t ldk, int64_t F,
                                                int64_t C, float *phase_io, float *y, int64_t y_stride, int64_t y_len,
                                                void *stream)
{
    if (!c || C < 0 || F < 0) return VOC_ERR_INVALID;
    if (C == 0 || F == 0) return VOC_OK;
    if (!mag || !ph_inc || !phase_io || !y) return VOC_ERR_INVALID;
    if (ldk < c->K || y_len < (F - 1) * (int64_t)c->H + c->W || y_stride < y_len) return VOC_ERR_SIZE;
    return guarded(c, [&]() -> int32_t {
        Locked lk(c, VOC_PRECISION_F32);
        synthesis_signal_dev<float>(c, mag, ph_inc, ldk, F, C, phase_io, y, y_stride, y_len, (hipStream_t)stream);
        return VOC_OK;
    });
}

VOC_EXPORT int32_t voc_synthesis_mapped_f32_dev(voc_ctx *c, const float *mag, int64_t mag_frames, const int64_t *mag_rows,
                                                const float *ph_inc, int64_t inc_frames, const int64_t *inc_rows, int64_t ldk,
                                                int64_t F, int64_t C, float *phase_io, float *y, int64_t y_stride,
                                                int64_t y_len, void *stream)
{
    if (!c || C < 0 || F < 0) return VOC_ERR_INVALID;
    if (C == 0 || F == 0) return VOC_OK;
    if (!mag || !ph_inc || !phase_io || !y) return VOC_ERR_INVALID;
    if (ldk < c->K || y_len < (F - 1) * (int64_t)c->H + c->W || y_stride < y_len) return VOC_ERR_SIZE;
    if ((mag_rows ? mag_frames < 1 : mag_frames != F) || (inc_rows ? inc_frames < 1 : inc_frames != F)) return VOC_ERR_SIZE;
    return guarded(c, [&]() -> int32_t {
        Locked lk(c, VOC_PRECISION_F32);
        synthesis_signal_dev<float>(c, mag, ph_inc, ldk, F, C, phase_io, y, y_stride, y_len, (hipStream_t)stream,
                                    reinterpret_cast<const long long *>(mag_rows), mag_frames,
                                    reinterpret_cast<const long long *>(inc_rows), inc_frames);
        return VOC_OK;
    });
}

VOC_EXPORT int32_t voc_profile_enable(voc_ctx *c, int32_t on)
{
    if (!c) return VOC_ERR_INVALID;
#ifndef VOC_EMU
    std::lock_guard<std::mutex> g(c->mu);
    c->profiling = on != 0;
#else
    (void)on;
#endif
    return VOC_OK;
}

VOC_EXPORT int32_t voc_profile_read(voc_ctx *c, double *ms, int64_t *launches)
{
    if (!c || !ms || !launches) return VOC_ERR_INVALID;
    for (int k = 0; k < KIND_COUNT; ++k) { ms[k] = 0.0; launches[k] = 0; }
#ifndef VOC_EMU
    return guarded(c, [&]() -> int32_t {
        Locked lk(c, c->precision);
        for (auto &t : c->timed) {
            float e = 0.f;
            HIPCHK(hipEventSynchronize(t.stop));
            HIPCHK(hipEventElapsedTime(&e, t.start, t.stop));
            ms[t.kind] += e;
            launches[t.kind] += 1;
            (void)hipEventDestroy(t.start);
            (void)hipEventDestroy(t.stop);
        }
        c->timed.clear();
        return VOC_OK;
    });
#else
    return VOC_OK;
#endif
}

#include "vocoder_filters.inc"
#include "vocoder_stream.inc"
