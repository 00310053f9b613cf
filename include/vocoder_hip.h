/*
 * vocoder_hip.h -- C ABI of the MI355X (gfx950) phase-vocoder engine.
 *
 * This is the drop-in boundary for the STFT round-trip path of tilk/vocoder.
 * In the reference that boundary is the `vector-fftw` binding
 * (Numeric.FFT.Vector.Invertible / .Plan, imported at vocoder/src/Vocoder.hs:49-50);
 * here it is widened to the whole per-frame / per-stage arithmetic of
 * vocoder/src/Vocoder.hs so that a batch of frames and channels crosses the
 * FFI once.  Every entry point cites the reference definition it replaces
 * (paths relative to the reference checkout).
 *
 * Conventions
 *   - plain C: pointers, sizes, int32_t status.  0 = VOC_OK, negative = error;
 *     voc_strerror() maps a status to text.  Nothing throws across the boundary.
 *   - all sample / spectrum data is IEEE double, laid out planar
 *     [channel][frame][bin or sample], bin / sample fastest.  A complex value is
 *     {re, im} (16 bytes), the layout of Haskell's `Storable (Complex Double)`.
 *   - N = FFT frame length (vocFrameLength), W = window / input frame length
 *     (vocInputFrameLength, W <= N), H = hop (vocHopSize), K = N/2+1
 *     (vocFreqFrameLength).  This build supports every N in [2, 32768]: powers of
 *     two from 8 to 16384 directly, every other length as a chirp-z transform (double precision
 *     only; above 4096 through device memory, an order of magnitude slower).
 *   - the caller owns every buffer; the library owns only the opaque voc_ctx.
 *   - entry points without a `_dev` suffix take HOST pointers and copy through
 *     device memory they allocate; `_dev` entry points take DEVICE pointers and
 *     a HIP stream (hipStream_t passed as void*, NULL = the default stream) and
 *     return after enqueueing.
 *   - a context may be used from any thread; calls on one context serialise on
 *     an internal mutex (the reference leaves this open: Vocoder.hs:98).  `_dev` calls on one
 *     context may name different HIP streams: the library orders its own workspace between
 *     them with an event (a call waits, on its stream, for the previous call's use of the
 *     workspace to finish); ordering the caller's buffers between streams is the caller's
 *     business, as with any stream-ordered API.  A voc_stream is one sequential state: its
 *     pushes must be ordered by the caller (one HIP stream, or synchronised).
 */
#ifndef VOCODER_HIP_H
#define VOCODER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct voc_ctx voc_ctx;

enum {
    VOC_OK = 0,
    VOC_ERR_INVALID = -1,     /* NULL pointer, negative count, W > N (Vocoder.hs:132), H < 1 */
    VOC_ERR_UNSUPPORTED = -2, /* frame length out of range (see above), unknown precision */
    VOC_ERR_HIP = -3,         /* a HIP runtime call failed; see voc_last_hip_error() */
    VOC_ERR_NOMEM = -4,       /* device or host allocation failed */
    VOC_ERR_SIZE = -5,        /* buffer / workspace too small, size mismatch (vector-fftw `execute`) */
    VOC_ERR_NO_DEVICE = -6,   /* no usable GPU */
    VOC_ERR_SCRATCH = -7      /* voc_roundtrip_static_dev: one channel needs more than VOC_STATIC_SCRATCH_BYTES */
};

/* VOC_PRECISION_F64: the reference's arithmetic; every entry point without `_f32` in its name.
 * VOC_PRECISION_F32: a reduced-precision variant (every entry point with `_f32` in its name: the fused
 * round trip, the stage and signal levels, filter chains, streams): float samples, STFT and phases, the synthesis phase kept wrapped into [-pi, pi) (an
 * unwrapped float phase loses all resolution within a few thousand frames).  Its parity bar
 * is its own (tests/test_f32_cpu.py, tests/test_gpu_f32.py): relative L2 <= 2e-4 on output
 * samples against the double-precision oracle.  A context serves one precision; entry points
 * of the other one return VOC_ERR_UNSUPPORTED. */
enum { VOC_PRECISION_F64 = 0, VOC_PRECISION_F32 = 1 };

const char *voc_strerror(int32_t status);
/* Text of the last failing HIP call made through this context (empty if none). */
const char *voc_last_hip_error(const voc_ctx *ctx);

/* ---- lifecycle ------------------------------------------------------------------
 * voc_ctx_create   replaces  vocoderParams len hop window           Vocoder.hs:114-115
 *                  (and the two `plan dftR2C / dftC2R len` it makes: Vocoder.hs:115)
 * accessors        replace   vocFrameLength / vocInputFrameLength /
 *                            vocFreqFrameLength / vocHopSize / vocWindow   Vocoder.hs:95-111
 * voc_volume_coeff replaces  volumeCoeff                             Vocoder.hs:209-210
 */
int32_t voc_ctx_create(int32_t frame_length, int32_t hop, const double *window, int32_t window_length,
                       int32_t device, int32_t precision, voc_ctx **out);
int32_t voc_ctx_destroy(voc_ctx *ctx);
int32_t voc_frame_length(const voc_ctx *ctx);       /* N */
int32_t voc_input_frame_length(const voc_ctx *ctx); /* W */
int32_t voc_freq_frame_length(const voc_ctx *ctx);  /* K */
int32_t voc_hop_size(const voc_ctx *ctx);           /* H */
int32_t voc_window(const voc_ctx *ctx, double *out /* [W] */);
double voc_volume_coeff(const voc_ctx *ctx);

/* ---- FFT-compatible layer ---------------------------------------------------------
 * voc_r2c / voc_c2r replace `FFT.execute` on the two plans          Vocoder.hs:145,196
 *   r2c: B x N reals -> B x K complex, unnormalised, DC / Nyquist imaginary parts +0
 *   c2r: B x K complex -> B x N reals, multiplied by 1/N, DC / Nyquist imaginary parts ignored
 * voc_do_fft  replaces doFFT  (window, symmetric zero pad, rewind, R2C)   Vocoder.hs:143-145
 * voc_do_ifft replaces doIFFT (C2R, rewind, cutCenter, window)            Vocoder.hs:194-196
 */
int32_t voc_r2c(voc_ctx *ctx, const double *in, double *out, int64_t batch);
int32_t voc_c2r(voc_ctx *ctx, const double *in, double *out, int64_t batch);
int32_t voc_do_fft(voc_ctx *ctx, const double *frames /* [B][W] */, double *spec /* [B][K] complex */, int64_t batch);
int32_t voc_do_ifft(voc_ctx *ctx, const double *spec /* [B][K] complex */, double *frames /* [B][W] */, int64_t batch);

/* ---- step level ---------------------------------------------------------------------
 * voc_analysis_step  replaces analysisStep h eN prev_ph vec          Vocoder.hs:158-163
 *                    (frameFromComplex :217-218, calcPhaseInc :169-173, wrap :166-167)
 * voc_synthesis_step replaces synthesisStep hop ph (mag, ph_inc)      Vocoder.hs:186-190
 *                    (frameToComplex :213-214)
 * voc_add_frames     replaces addFrames                               Vocoder.hs:220-222
 * B independent frames per call; prev_ph / ph are [B][K].  ph_out may alias prev_ph / ph.
 */
int32_t voc_analysis_step(voc_ctx *ctx, const double *prev_ph, const double *spec, double *ph_out, double *mag,
                          double *ph_inc, int64_t batch);
int32_t voc_synthesis_step(voc_ctx *ctx, const double *ph, const double *mag, const double *ph_inc, double *ph_out,
                           double *spec, int64_t batch);
int32_t voc_add_frames(voc_ctx *ctx, const double *mag1, const double *inc1, const double *mag2, const double *inc2,
                       double *mag_out, double *inc_out, int64_t batch);
/* addFrames on device arrays: element-wise over `count` doubles of six arrays with one common
 * layout (rows with any pitch included); an output may be one of the inputs.  With two filter
 * chains run on copies of an STFT this is addFilters (Filter.hs:58-59). */
int32_t voc_add_frames_dev(voc_ctx *ctx, const double *mag1, const double *inc1, const double *mag2, const double *inc2,
                           double *mag_out, double *inc_out, int64_t count, void *stream);

/* The same functions with the reference's own argument lists: analysisStep and synthesisStep take the
 * hop size (and frame length) as plain numbers, frameFromComplex / frameToComplex / addFrames take only
 * their vectors, none of them a VocoderParams.  These forms need no context: they run on the calling
 * thread's current HIP device, keep one staging buffer for the process, and serialise among themselves.
 *   voc_analysis_step_hn    analysisStep h eN prev_ph vec          Vocoder.hs:158-163   (K = length vec)
 *   voc_synthesis_step_h    synthesisStep hop ph (mag, ph_inc)     Vocoder.hs:186-190
 *   voc_frame_from_complex  frameFromComplex                       Vocoder.hs:217-218   spec: count complex
 *   voc_frame_to_complex    frameToComplex                         Vocoder.hs:213-214
 *   voc_add_frames_n        addFrames                              Vocoder.hs:220-222
 */
int32_t voc_analysis_step_hn(int32_t hop, int32_t frame_length, int64_t k, const double *prev_ph, const double *spec,
                             double *ph_out, double *mag, double *ph_inc, int64_t batch);
int32_t voc_synthesis_step_h(int32_t hop, int64_t k, const double *ph, const double *mag, const double *ph_inc,
                             double *ph_out, double *spec, int64_t batch);
int32_t voc_frame_from_complex(const double *spec, double *mag, double *ph, int64_t count);
int32_t voc_frame_to_complex(const double *mag, const double *ph, double *spec, int64_t count);
int32_t voc_add_frames_n(const double *mag1, const double *inc1, const double *mag2, const double *inc2,
                         double *mag_out, double *inc_out, int64_t count);

/* Element-wise combination of two STFT streams, as the `--multiply`, `--divide` and `--add` commands of
 * vocoder-jack combine them (vocoder-dunai/example/VocoderJack.hs:107-110,127-135):
 *   VOC_BINARY_MULTIPLY  mkBinary (*) (+): mag1 * mag2, ph_inc1 + ph_inc2
 *   VOC_BINARY_DIVIDE    mkBinary (/) (-): mag1 / mag2, ph_inc1 - ph_inc2
 *   VOC_BINARY_ADD       addFrames (the same as voc_add_frames_dev)
 * over `count` doubles of six device arrays with one common layout; an output may be one of the inputs. */
enum { VOC_BINARY_MULTIPLY = 0, VOC_BINARY_DIVIDE = 1, VOC_BINARY_ADD = 2 };
int32_t voc_binary_frames_dev(voc_ctx *ctx, int32_t op, const double *mag1, const double *inc1, const double *mag2,
                              const double *inc2, double *mag_out, double *inc_out, int64_t count, void *stream);

/* ---- stage level (the batched hot path) -----------------------------------------------
 * voc_analysis_stage  replaces analysisStage par ph frames           Vocoder.hs:149-150
 *                     = mapAccumL (analysisBlock par)                Vocoder.hs:153-154
 * voc_synthesis_stage replaces synthesisStage par ph stft            Vocoder.hs:177-178
 *                     = mapAccumL (synthesisBlock par)               Vocoder.hs:181-182
 * C channels are C independent mapAccumL folds sharing one VocoderParams, as
 * analysisF / synthesisF do (vocoder-conduit/src/Vocoder/Conduit.hs:63-68).
 *   phase_io [C][K]: in = phase before the first frame, out = phase after the last
 *   frames   [C][F][W], mag / ph_inc [C][F][K].  C = 1 is the reference signature.
 */
int32_t voc_analysis_stage(voc_ctx *ctx, double *phase_io, const double *frames, double *mag, double *ph_inc,
                           int64_t channels, int64_t nframes);
int32_t voc_synthesis_stage(voc_ctx *ctx, double *phase_io, const double *mag, const double *ph_inc, double *frames,
                            int64_t channels, int64_t nframes);
int32_t voc_analysis_stage_dev(voc_ctx *ctx, double *phase_io, const double *frames, double *mag, double *ph_inc,
                               int64_t channels, int64_t nframes, void *stream);
int32_t voc_synthesis_stage_dev(voc_ctx *ctx, double *phase_io, const double *mag, const double *ph_inc,
                                double *frames, int64_t channels, int64_t nframes, void *stream);

/* ---- framing and overlap-add --------------------------------------------------------------
 * voc_num_frames     frames genFramesOfE yields for `length` buffered samples: offsets
 *                    0, H, .. <= length - W          vocoder-conduit/src/Vocoder/Conduit/Frames.hs:34-35
 * voc_leftover_length samples genFramesOfE keeps as state: drop ((len-W) div H)*H + H   Frames.hs:36-37
 * voc_ola_length     samples sumFramesE chunk H emits for F frames: the last chunk is
 *                    zero-padded to `chunk`                          Frames.hs:45-47,57
 * voc_frames_of      replaces genFramesOfE W H q on one whole buffer per channel   Frames.hs:26-38
 *                    x [C][x_stride] (first `length` used) -> frames [C][F][W]
 * voc_sum_frames     replaces sumFramesE chunk H                     Frames.hs:41-58
 *                    frames [C][F][W] -> y [C][y_stride] (first voc_ola_length used)
 */
int64_t voc_num_frames(const voc_ctx *ctx, int64_t length);
int64_t voc_leftover_length(const voc_ctx *ctx, int64_t length);
int64_t voc_ola_length(const voc_ctx *ctx, int64_t nframes, int64_t chunk);
int32_t voc_frames_of(voc_ctx *ctx, const double *x, int64_t x_stride, int64_t length, int64_t channels,
                      double *frames);
int32_t voc_sum_frames(voc_ctx *ctx, const double *frames, int64_t nframes, int64_t channels, int64_t chunk,
                       double *y, int64_t y_stride);

/* ---- fused round trip (the metric path) --------------------------------------------------------
 * voc_roundtrip replaces, per channel, Vocoder.Conduit's
 *     genFramesOfE W H q  `fuseBoth`  processFrames par phs idFilter  `fuseUpstream`  sumFramesE chunk H
 * i.e. framing, analysis, volumeFix (mag * volumeCoeff), identity filter, synthesis and
 * overlap-add (vocoder-conduit/src/Vocoder/Conduit.hs:34-35,46-53; with chunk = H this is
 * `process`, with the caller's q / phases it is the `processWith` loop that
 * vocoder-audio threads between sources, vocoder-audio/src/Vocoder/Audio.hs:31-33,86-91).
 *   x            [C][x_stride]: previous leftover ++ new samples, `length` per channel
 *   y            [C][y_stride]: receives voc_ola_length(voc_num_frames(length), chunk) samples
 *   ana_phase_io, syn_phase_io [C][K]: in = phases before, out = phases after
 *   leftover     [C][leftover_stride] or NULL: receives the voc_leftover_length(length) trailing samples
 * The STFT (mag, ph_inc) of every frame is written once to and read once from device
 * workspace between the two halves, where a frequency-domain filter would sit.
 * voc_roundtrip_workspace_bytes: device bytes the _dev form needs to run all channels in one
 * pass; with a smaller (or NULL) workspace the library processes the channels in groups
 * (NULL: it keeps a grow-only workspace of its own in the context).
 */
int32_t voc_roundtrip(voc_ctx *ctx, const double *x, int64_t x_stride, int64_t length, int64_t channels,
                      int64_t chunk, double *y, int64_t y_stride, double *ana_phase_io, double *syn_phase_io,
                      double *leftover, int64_t leftover_stride);
int64_t voc_roundtrip_workspace_bytes(const voc_ctx *ctx, int64_t length, int64_t channels);
int32_t voc_roundtrip_dev(voc_ctx *ctx, const double *x, int64_t x_stride, int64_t length, int64_t channels,
                          int64_t chunk, double *y, int64_t y_stride, double *ana_phase_io, double *syn_phase_io,
                          double *leftover, int64_t leftover_stride, void *workspace, int64_t workspace_bytes,
                          void *stream);

/* ---- fused round trip that allocates nothing ---------------------------------------------------
 * voc_roundtrip_static_dev has no counterpart in the reference.  It is voc_roundtrip_dev for
 * callers that cannot allocate device memory or let a library do so: code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  The STFT, the phase snapshots
 * and the tile carries of the call live in an array of VOC_STATIC_SCRATCH_BYTES that is part of
 * the library's code object (one copy per GPU, held from the first context created on that GPU
 * until the process ends); the call makes no hipMalloc / hipFree, creates no event and calls no
 * allocating routine.  Same arguments, results and bits as voc_roundtrip_dev without its
 * workspace pair.  Channels run in groups that fit the array; if ONE channel does not fit --
 * voc_roundtrip_workspace_bytes(ctx, length, 1) > VOC_STATIC_SCRATCH_BYTES -- the call returns
 * VOC_ERR_SCRATCH and touches nothing (N = 4096, H = 1024: about 2000 frames per channel,
 * 43 s at 48 kHz).  Every call on one GPU, from any context or thread, uses that GPU's one
 * array: calls that overlap in time on one GPU race; order them on one HIP stream or
 * synchronise between them.  Nothing in the array has to be cleared between calls.  Frame lengths
 * above 4096 that are not powers of two return VOC_ERR_UNSUPPORTED here: their transforms go through a
 * buffer that the library allocates.  Launch timing (voc_profile_enable) records its launches with events
 * it creates, so it is off for the duration of this call: voc_profile_read does not count them.
 */
#define VOC_STATIC_SCRATCH_BYTES (64u << 20)
int32_t voc_roundtrip_static_dev(voc_ctx *ctx, const double *x, int64_t x_stride, int64_t length, int64_t channels,
                                 int64_t chunk, double *y, int64_t y_stride, double *ana_phase_io,
                                 double *syn_phase_io, double *leftover, int64_t leftover_stride, void *stream);

/* ---- fused path, single precision (contexts created with VOC_PRECISION_F32) --------------------
 * Same contract as voc_roundtrip / voc_roundtrip_dev with float buffers; the phases returned
 * are wrapped.  voc_roundtrip_workspace_bytes accounts for the context's precision.
 */
int32_t voc_roundtrip_f32(voc_ctx *ctx, const float *x, int64_t x_stride, int64_t length, int64_t channels,
                          int64_t chunk, float *y, int64_t y_stride, float *ana_phase_io, float *syn_phase_io,
                          float *leftover, int64_t leftover_stride);
int32_t voc_roundtrip_f32_dev(voc_ctx *ctx, const float *x, int64_t x_stride, int64_t length, int64_t channels,
                              int64_t chunk, float *y, int64_t y_stride, float *ana_phase_io, float *syn_phase_io,
                              float *leftover, int64_t leftover_stride, void *workspace, int64_t workspace_bytes,
                              void *stream);

/* ---- the two halves of the fused path, device pointers ---------------------------------------
 * For callers that put something between analysis and synthesis (a frequency-domain filter,
 * Conduit.hs:47; a change of frame rate, Conduit/Filter.hs:122-136) or that cut a long signal
 * into time segments.
 * voc_analysis_signal_dev   genFramesOfE W H (Frames.hs:26-38) .| analysisStage (Vocoder.hs:149-154),
 *                           times volumeCoeff (Conduit.hs:34-35) when volume_fix != 0: frame j of
 *                           channel c is read in place from x[c][j*H .. j*H+W).
 * voc_synthesis_signal_dev  synthesisStage (Vocoder.hs:177-182) .| sumFramesE (Frames.hs:41-58):
 *                           y[c][0 .. y_len) receives the overlap-add of the F frames, zeros
 *                           past sample (F-1)*H+W; y_len >= (F-1)*H+W.
 * voc_phase_advance_dev     the phase half of synthesisStage alone: phase_io <- the accumulated
 *                           phase after the F frames (Vocoder.hs:190), nothing else computed.
 *   mag, ph_inc [C][F][ldk], ldk >= K (row pitch in doubles); phase_io [C][K] in/out.
 */
int32_t voc_analysis_signal_dev(voc_ctx *ctx, const double *x, int64_t x_stride, int64_t nframes, int64_t channels,
                                double *phase_io, double *mag, double *ph_inc, int64_t ldk, int32_t volume_fix,
                                void *stream);
int32_t voc_synthesis_signal_dev(voc_ctx *ctx, const double *mag, const double *ph_inc, int64_t ldk, int64_t nframes,
                                 int64_t channels, double *phase_io, double *y, int64_t y_stride, int64_t y_len,
                                 void *stream);
int32_t voc_phase_advance_dev(voc_ctx *ctx, const double *ph_inc, int64_t ldk, int64_t nframes, int64_t channels,
                              double *phase_io, void *stream);

/* voc_synthesis_signal_dev reading its frames through row maps: output frame j of a channel takes
 * its moduli from row mag_rows[j] of that channel's mag_frames rows of `mag`, its increments from
 * row inc_rows[j] of the inc_frames rows of `ph_inc` (device arrays of nframes indices; entries are
 * clamped into their array).  This is playSpeed (Conduit/Filter.hs:122-136: every frame emitted
 * zero or more times) feeding synthesis without the copy voc_gather_frames_dev makes.  A NULL map
 * means row j, and that array then holds exactly nframes rows per channel.  The two maps are
 * separate because a filter that follows playSpeed may rewrite one array per output frame and
 * leave the other alone (randomPhaseFilter, Filter.hs:156-157).
 */
int32_t voc_synthesis_mapped_dev(voc_ctx *ctx, const double *mag, int64_t mag_frames, const int64_t *mag_rows,
                                 const double *ph_inc, int64_t inc_frames, const int64_t *inc_rows, int64_t ldk,
                                 int64_t nframes, int64_t channels, double *phase_io, double *y, int64_t y_stride,
                                 int64_t y_len, void *stream);

/* ---- frequency-domain filters (vocoder/src/Vocoder/Filter.hs) -----------------------------------
 * A voc_filter is a chain in composeFilters order (Filter.hs:54-55; vocoder-file folds its
 * command-line filters the same way, vocoder-audio/example/VocoderFile.hs:72); an empty chain
 * is idFilter (Filter.hs:62-63).  freq_step is the reference's FreqStep, sample rate / frame
 * length (vocoder-audio/src/Vocoder/Audio.hs:49).  voc_filter_append(kind, params, nparams):
 *   VOC_FILTER_AMPLIFY                 k           amplify                 Filter.hs:80-81
 *   VOC_FILTER_LOWPASS_BRICKWALL       t           lowpassBrickwall        Filter.hs:84-85
 *   VOC_FILTER_HIGHPASS_BRICKWALL      t           highpassBrickwall       Filter.hs:88-89
 *   VOC_FILTER_BANDPASS_BRICKWALL      t,u         bandpassBrickwall       Filter.hs:92-93
 *   VOC_FILTER_BANDSTOP_BRICKWALL      t,u         bandstopBrickwall       Filter.hs:96-97
 *   VOC_FILTER_LOWPASS_BUTTERWORTH     n,t         lowpassButterworth      Filter.hs:103-104
 *   VOC_FILTER_HIGHPASS_BUTTERWORTH    n,t         highpassButterworth     Filter.hs:107-108
 *   VOC_FILTER_BANDPASS_BUTTERWORTH    n,t,u       bandpassButterworth     Filter.hs:111-112
 *   VOC_FILTER_BANDSTOP_BUTTERWORTH    n,t,u       bandstopButterworth     Filter.hs:115-116
 *   VOC_FILTER_GAIN_TABLE              g[0..K)     linearAmplitudeFilter f with g[k] = f (step*k),
 *                                                  evaluated by the caller    Filter.hs:76-77
 *   VOC_FILTER_PITCH_SHIFT_INTERPOLATE n > 0       pitchShiftInterpolate   Filter.hs:118-129
 *   VOC_FILTER_CONVOLUTION             ker[0..len) convolutionFilter       Filter.hs:132-141
 *   VOC_FILTER_ENVELOPE                ksize       envelopeFilter          Filter.hs:144-152
 *   VOC_FILTER_RANDOM_PHASE            seed < 2^53 randomPhaseFilter       Filter.hs:156-157
 * The reference draws randomPhaseFilter's increments from the unseeded global generator, so
 * its values are not reproducible; here they are Philox-4x32-10 words keyed by the seed with
 * counter (bin | position in the chain << 16, channel, frame low word, frame high word), where
 * channel and frame are positions in the WHOLE signal: a signal filtered in pieces (chunks of a
 * stream, runs of a time-sharded segment, blocks of channels on different GPUs) draws the numbers
 * one call over all of it would draw, and every frame draws fresh ones as in the reference.
 * voc_filter_apply[_dev] runs the chain in place on mag / ph_inc ([C][F][K] host arrays, or
 * device arrays with row pitch ldk), taking row 0 for frame 0 of channel 0;
 * voc_filter_apply_at_dev says where in the signal the rows sit (first_frame, first_channel).
 * A voc_stream passes its own frame count.  At most 12 operations per chain (VOC_ERR_SIZE beyond).
 * addFilters (Filter.hs:58-59): run the two chains on copies and add them with voc_add_frames[_dev].
 */
typedef struct voc_filter voc_filter;
enum {
    VOC_FILTER_AMPLIFY = 0,
    VOC_FILTER_LOWPASS_BRICKWALL = 1,
    VOC_FILTER_HIGHPASS_BRICKWALL = 2,
    VOC_FILTER_BANDPASS_BRICKWALL = 3,
    VOC_FILTER_BANDSTOP_BRICKWALL = 4,
    VOC_FILTER_LOWPASS_BUTTERWORTH = 5,
    VOC_FILTER_HIGHPASS_BUTTERWORTH = 6,
    VOC_FILTER_BANDPASS_BUTTERWORTH = 7,
    VOC_FILTER_BANDSTOP_BUTTERWORTH = 8,
    VOC_FILTER_GAIN_TABLE = 9,
    VOC_FILTER_PITCH_SHIFT_INTERPOLATE = 10,
    VOC_FILTER_CONVOLUTION = 11,
    VOC_FILTER_ENVELOPE = 12,
    VOC_FILTER_RANDOM_PHASE = 13
};
int32_t voc_filter_create(voc_ctx *ctx, double freq_step, voc_filter **out);
int32_t voc_filter_destroy(voc_filter *f);
int32_t voc_filter_length(const voc_filter *f);
/* flags4 <- {reads mag, writes mag, reads ph_inc, writes ph_inc} for the chain as a whole (an array
 * the chain neither reads nor writes is never touched by voc_filter_apply_dev) */
int32_t voc_filter_touches(const voc_filter *f, int32_t *flags4);
int32_t voc_filter_append(voc_filter *f, int32_t kind, const double *params, int32_t nparams);
int32_t voc_filter_apply(voc_filter *f, double *mag, double *ph_inc, int64_t nframes, int64_t channels);
int32_t voc_filter_apply_dev(voc_filter *f, double *mag, double *ph_inc, int64_t ldk, int64_t nframes,
                             int64_t channels, void *stream);
int32_t voc_filter_apply_at_dev(voc_filter *f, double *mag, double *ph_inc, int64_t ldk, int64_t nframes,
                                int64_t channels, int64_t first_frame, int64_t first_channel, void *stream);

/* filter .| synthesisStage .| sumFramesE in one call (the middle of processWith, Conduit.hs:46-47, for a chain of
 * per-frame filters): y as voc_filter_apply_dev followed by voc_synthesis_signal_dev would leave it, bit for bit.
 * Where the chain is pitchShiftInterpolate and / or one linearAmplitudeFilter-based filter, in that order
 * (Filter.hs:76-129), and the context runs the specialised double-precision kernels (N = 2048 ... 8192, W = N,
 * hop a multiple of N/8), synthesis applies the chain to each row as it reads it and the pass over the STFT is
 * saved; every other chain is applied first, in place.  mag / ph_inc are therefore left either untouched or
 * filtered: a caller must not rely on their contents afterwards. */
int32_t voc_synthesis_filtered_dev(voc_filter *f, double *mag, double *ph_inc, int64_t ldk, int64_t nframes, int64_t channels,
                                   double *phase_io, double *y, int64_t y_stride, int64_t y_len, void *stream);

/* ---- playSpeed (vocoder-conduit/src/Vocoder/Conduit/Filter.hs:122-136) --------------------------
 * playSpeed coeff replicates or drops STFT frames: after i input frames floor(i * coeff) frames
 * have been emitted, each emission repeating the newest input frame.  coeff = num / den exactly
 * (the reference takes a Rational: toRational of the command-line double, VocoderFile.hs:119).
 * voc_play_speed_frames  number of frames emitted for nframes_in input frames
 * voc_play_speed_map     src[j] = input frame that output frame j repeats (host array)
 * voc_gather_frames_dev  mag_out/inc_out[c][j][:] = mag_in/inc_in[c][src[j]][:], src on the device
 */
int64_t voc_play_speed_frames(int64_t num, int64_t den, int64_t nframes_in);
int32_t voc_play_speed_map(int64_t num, int64_t den, int64_t nframes_in, int64_t *src, int64_t capacity);
int32_t voc_gather_frames_dev(voc_ctx *ctx, const double *mag_in, const double *inc_in, int64_t ldk, int64_t nframes_in,
                              const int64_t *src_dev, int64_t nframes_out, double *mag_out, double *inc_out,
                              int64_t channels, void *stream);

/* ---- interleaved <-> planar samples ---------------------------------------------------------------
 * vocoder-audio frames interleaved multi-channel audio (frames of W*channels samples at hops of
 * H*channels) and de-interleaves every frame (vocoder-audio/src/Vocoder/Audio.hs:51-55); the
 * library is planar, so a caller with interleaved device buffers converts once per buffer:
 * inter[i*channels + c] <-> planar[c*stride + i], i < n.
 */
int32_t voc_deinterleave_dev(voc_ctx *ctx, const double *inter, int64_t n, int64_t channels, double *planar,
                             int64_t stride, void *stream);
int32_t voc_interleave_dev(voc_ctx *ctx, const double *planar, int64_t stride, int64_t n, int64_t channels,
                           double *inter, void *stream);

/* ---- 16-bit PCM at the file edge ------------------------------------------------------------------
 * vocoder-file reads its sources with sourceSnd and writes WAV / PCM16 with sinkSnd
 * (vocoder-audio/example/VocoderFile.hs:158,168-173), i.e. through libsndfile's double interface.
 * These two calls do that conversion on interleaved device buffers, fused with the
 * (de)interleave: reading, planar[c*stride + i] = inter[i*channels + c] / 32768; writing,
 * inter[i*channels + c] = lrint(x * 32767) kept to its low 16 bits (clip == 0, libsndfile's default) or
 * lrint(x * 32768) saturated to [-32768, 32767] (clip != 0, its SFC_SET_CLIPPING mode); halves round to
 * even.  libsndfile is outside the reference tree: the convention is restated from its published
 * source and is not pinned by any vector of the reference.
 */
int32_t voc_pcm16_to_planar_dev(voc_ctx *ctx, const int16_t *inter, int64_t n, int64_t channels, double *planar,
                                int64_t stride, void *stream);
int32_t voc_planar_to_pcm16_dev(voc_ctx *ctx, const double *planar, int64_t stride, int64_t n, int64_t channels,
                                int32_t clip, int16_t *inter, void *stream);

/* ---- chunk-at-a-time streaming (Vocoder.Conduit.process as a conduit, Conduit.hs:46-53) --------
 * A voc_stream holds what the reference's conduits hold between chunks: the samples
 * genFramesOfE has not consumed (Frames.hs:36-37), the analysis and synthesis phases
 * (Conduit.hs:46-47) and sumFramesE's queue of frames that still reach unpublished output
 * (Frames.hs:48).  Pushing a signal in pieces of any sizes and flushing yields, concatenated,
 * the samples of one voc_roundtrip call over the whole signal, bit for bit, and publishes
 * them when sumFramesE would: after the frame with index j, floor(j*H / chunk) chunks are out
 * (Frames.hs:49-53); voc_stream_flush publishes the rest, zero padded to a whole chunk
 * (publishRest, Frames.hs:46-47), after which the stream only answers queries.
 *   filter       chain applied to every frame between analysis and synthesis, or NULL
 *                (idFilter); must outlive the stream.  playSpeed is not a per-frame chain:
 *                see voc_stream_set_play_speed.
 *   x            [C][x_stride], n new samples per channel (n = 0 is allowed)
 *   y            [C][y_stride], receives *n_out <= y_cap samples per channel;
 *                voc_stream_push_length / _flush_length tell *n_out beforehand
 *                (VOC_ERR_SIZE if y_cap is too small; nothing is consumed then).
 *   phases       [C][K] host arrays; set before the first push (default zeroPhase), get at
 *                any time: the phases after the last frame so far.
 *   voc_stream_leftover  copies the unconsumed samples ([C][out_stride]) and returns their
 *                count (out = NULL: count only) -- with the two phases this is the triple
 *                vocoder-audio threads from one source to the next (Audio.hs:31-33).
 * voc_stream_create_mode(.., VOC_STREAM_DUNAI, ..) follows Vocoder.Dunai.process instead
 * (vocoder-dunai/src/Vocoder/Dunai.hs:46-48): the stream starts from ((W-1) div H)*H zeros
 * (framesOfS, Dunai.hs:58-69), every push must be a multiple of the hop and returns exactly as
 * many samples as it was given (sumFramesWithLengthS, Dunai.hs:76-85), the volume correction is
 * applied to the summed samples, not to the moduli (Dunai.hs:41-42,48); `chunk` is ignored and
 * there is nothing to flush.
 * voc_stream_set_filter swaps the chain between pushes (each source of a concatenation has its
 * own filter, vocoder-audio/src/Vocoder/Audio.hs:48-55,64-69).
 * Destroy streams and filters before the context they were created on.
 */
typedef struct voc_stream voc_stream;
enum { VOC_STREAM_CONDUIT = 0, VOC_STREAM_DUNAI = 1 };
int32_t voc_stream_create(voc_ctx *ctx, int64_t channels, int64_t chunk, voc_filter *filter, voc_stream **out);
int32_t voc_stream_create_mode(voc_ctx *ctx, int64_t channels, int64_t chunk, voc_filter *filter, int32_t mode,
                               voc_stream **out);
int32_t voc_stream_set_filter(voc_stream *s, voc_filter *filter);
/* playSpeed inside the stream (Vocoder.Conduit.Filter.playSpeed, Conduit/Filter.hs:122-136): the frames that
 * leave the stream's filter chain are repeated or dropped at the exact rational rate num / den (frame i is
 * yielded floor((i+1) num / den) - floor(i num / den) times), and `after` (may be NULL) runs on the yielded
 * frames, so that `chain .| playSpeed r .| chain'` is one stream.  The concatenated output equals the batch
 * path's on the whole signal.  To be called before the first push; conduit conventions only; num, den <= 2^20. */
int32_t voc_stream_set_play_speed(voc_stream *s, int64_t num, int64_t den, voc_filter *after);
int32_t voc_stream_destroy(voc_stream *s);
int32_t voc_stream_set_phases(voc_stream *s, const double *ana_phase, const double *syn_phase);
int32_t voc_stream_get_phases(voc_stream *s, double *ana_phase, double *syn_phase);
int64_t voc_stream_leftover(voc_stream *s, double *out, int64_t out_stride);
int64_t voc_stream_push_length(const voc_stream *s, int64_t n);
int64_t voc_stream_flush_length(const voc_stream *s);
int32_t voc_stream_push(voc_stream *s, const double *x, int64_t x_stride, int64_t n, double *y, int64_t y_stride,
                        int64_t y_cap, int64_t *n_out);
int32_t voc_stream_push_dev(voc_stream *s, const double *x, int64_t x_stride, int64_t n, double *y, int64_t y_stride,
                            int64_t y_cap, int64_t *n_out, void *stream);
int32_t voc_stream_flush(voc_stream *s, double *y, int64_t y_stride, int64_t y_cap, int64_t *n_out);
int32_t voc_stream_flush_dev(voc_stream *s, double *y, int64_t y_stride, int64_t y_cap, int64_t *n_out, void *stream);

/* ---- single-precision forms of the FFT-compatible layer, the step level and addFrames (voc_r2c, voc_c2r, voc_do_fft, voc_do_ifft,
 * voc_analysis_step, voc_synthesis_step, voc_add_frames: Vocoder.hs:143-145,194-196,158-163,186-190,220-222) on
 * float data, for contexts created with VOC_PRECISION_F32: same arguments, every array float (a complex
 * element is two floats, {re, im}); synthesis_step returns its phases wrapped, like every form of the variant. */
int32_t voc_r2c_f32(voc_ctx *ctx, const float *in, float *out, int64_t batch);
int32_t voc_c2r_f32(voc_ctx *ctx, const float *in, float *out, int64_t batch);
int32_t voc_do_fft_f32(voc_ctx *ctx, const float *frames, float *spec, int64_t batch);
int32_t voc_do_ifft_f32(voc_ctx *ctx, const float *spec, float *frames, int64_t batch);
int32_t voc_analysis_step_f32(voc_ctx *ctx, const float *prev_ph, const float *spec, float *ph_out, float *mag,
                              float *ph_inc, int64_t batch);
int32_t voc_synthesis_step_f32(voc_ctx *ctx, const float *ph, const float *mag, const float *ph_inc, float *ph_out,
                               float *spec, int64_t batch);
int32_t voc_add_frames_f32(voc_ctx *ctx, const float *mag1, const float *inc1, const float *mag2, const float *inc2,
                           float *mag_out, float *inc_out, int64_t batch);

/* ---- single-precision forms (contexts created with VOC_PRECISION_F32) -------------------------------
 * The stage-level and signal-level entry points on float data: the same arguments as their double
 * forms, every array float.  They run the kernel templates of the fused single-precision round trip
 * (float samples, STFT and phases; the synthesis phase kept wrapped), to that variant's own bar. */
int32_t voc_analysis_stage_f32(voc_ctx *ctx, float *phase_io, const float *frames, float *mag, float *ph_inc,
                               int64_t channels, int64_t nframes);
int32_t voc_synthesis_stage_f32(voc_ctx *ctx, float *phase_io, const float *mag, const float *ph_inc, float *frames,
                                int64_t channels, int64_t nframes);
int32_t voc_analysis_stage_f32_dev(voc_ctx *ctx, float *phase_io, const float *frames, float *mag, float *ph_inc,
                                   int64_t channels, int64_t nframes, void *stream);
int32_t voc_synthesis_stage_f32_dev(voc_ctx *ctx, float *phase_io, const float *mag, const float *ph_inc, float *frames,
                                    int64_t channels, int64_t nframes, void *stream);
int32_t voc_analysis_signal_f32_dev(voc_ctx *ctx, const float *x, int64_t x_stride, int64_t nframes, int64_t channels,
                                    float *phase_io, float *mag, float *ph_inc, int64_t ldk, int32_t volume_fix,
                                    void *stream);
int32_t voc_synthesis_signal_f32_dev(voc_ctx *ctx, const float *mag, const float *ph_inc, int64_t ldk, int64_t nframes,
                                     int64_t channels, float *phase_io, float *y, int64_t y_stride, int64_t y_len,
                                     void *stream);
int32_t voc_synthesis_mapped_f32_dev(voc_ctx *ctx, const float *mag, int64_t mag_frames, const int64_t *mag_rows,
                                     const float *ph_inc, int64_t inc_frames, const int64_t *inc_rows, int64_t ldk,
                                     int64_t nframes, int64_t channels, float *phase_io, float *y, int64_t y_stride,
                                     int64_t y_len, void *stream);
/* A filter chain (voc_filter_create / voc_filter_append on the single-precision context) on a float STFT: the
 * chain's arithmetic is the reference's, in double; rows are widened on load and rounded once on store.
 * first_frame / first_channel as in voc_filter_apply_at_dev. */
int32_t voc_filter_apply_f32_dev(voc_filter *f, float *mag, float *ph_inc, int64_t ldk, int64_t nframes, int64_t channels,
                                 int64_t first_frame, int64_t first_channel, void *stream);
int32_t voc_filter_apply_f32(voc_filter *f, float *mag, float *ph_inc, int64_t nframes, int64_t channels);
/* A voc_stream created on a single-precision context holds float state; its data calls are these (the queries
 * voc_stream_push_length / _flush_length, voc_stream_set_filter and voc_stream_set_play_speed serve both). */
int32_t voc_stream_set_phases_f32(voc_stream *s, const float *ana_phase, const float *syn_phase);
int32_t voc_stream_get_phases_f32(voc_stream *s, float *ana_phase, float *syn_phase);
int64_t voc_stream_leftover_f32(voc_stream *s, float *out, int64_t out_stride);
int32_t voc_stream_push_f32(voc_stream *s, const float *x, int64_t x_stride, int64_t n, float *y, int64_t y_stride,
                            int64_t y_cap, int64_t *n_out);
int32_t voc_stream_push_f32_dev(voc_stream *s, const float *x, int64_t x_stride, int64_t n, float *y, int64_t y_stride,
                                int64_t y_cap, int64_t *n_out, void *stream);
int32_t voc_stream_flush_f32(voc_stream *s, float *y, int64_t y_stride, int64_t y_cap, int64_t *n_out);
int32_t voc_stream_flush_f32_dev(voc_stream *s, float *y, int64_t y_stride, int64_t y_cap, int64_t *n_out, void *stream);

/* ---- measurement hook (no counterpart in the reference) ------------------------------------------
 * With profiling on, every kernel launch made through the context is bracketed by HIP events
 * on its launch stream.  voc_profile_read waits for them, adds the elapsed milliseconds and
 * launch counts per kernel kind into ms[4] / launches[4] -- 0: analysis kernel, 1: phase-carry
 * kernel, 2: synthesis kernel, 3: FFT-only kernels -- and forgets them.
 */
int32_t voc_profile_enable(voc_ctx *ctx, int32_t on);
int32_t voc_profile_read(voc_ctx *ctx, double *ms, int64_t *launches);

#ifdef __cplusplus
}
#endif
#endif /* VOCODER_HIP_H */
