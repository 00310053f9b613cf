{-# LANGUAGE ForeignFunctionInterface #-}
{-|
Module      : Vocoder.HIP.FFI
Description : Raw bindings to libvocoder_hip.so (include/vocoder_hip.h)

The binding a maintainer of the @vocoder@ package adds to move the STFT path to an MI355X.  It
stands where @Numeric.FFT.Vector.Invertible@ / @Numeric.FFT.Vector.Plan@ stand today
(vocoder/src/Vocoder.hs:49-50).  This file could not be compiled where it was written (no GHC
there); the C ABI it binds is exercised by the C++ and Python hosts instead.

Stage and round-trip calls block for milliseconds, so they are imported @safe@; the accessors are
@unsafe@.  A @voc_ctx@ is internally locked, so calls may come from any capability.
-}
module Vocoder.HIP.FFI where

import Data.Int
import Foreign.C.String
import Foreign.C.Types
import Foreign.Ptr

data VocCtx

foreign import ccall unsafe "voc_strerror"        c_voc_strerror :: Int32 -> IO CString
foreign import ccall unsafe "voc_last_hip_error"  c_voc_last_hip_error :: Ptr VocCtx -> IO CString
foreign import ccall safe   "voc_ctx_create"      c_voc_ctx_create
    :: Int32 -> Int32 -> Ptr Double -> Int32 -> Int32 -> Int32 -> Ptr (Ptr VocCtx) -> IO Int32
foreign import ccall safe   "voc_ctx_destroy"     c_voc_ctx_destroy :: Ptr VocCtx -> IO Int32
foreign import ccall unsafe "&voc_ctx_destroy"    p_voc_ctx_destroy :: FunPtr (Ptr VocCtx -> IO ())
foreign import ccall unsafe "voc_frame_length"       c_voc_frame_length :: Ptr VocCtx -> IO Int32
foreign import ccall unsafe "voc_input_frame_length" c_voc_input_frame_length :: Ptr VocCtx -> IO Int32
foreign import ccall unsafe "voc_freq_frame_length"  c_voc_freq_frame_length :: Ptr VocCtx -> IO Int32
foreign import ccall unsafe "voc_hop_size"           c_voc_hop_size :: Ptr VocCtx -> IO Int32
foreign import ccall unsafe "voc_volume_coeff"       c_voc_volume_coeff :: Ptr VocCtx -> IO Double

foreign import ccall safe "voc_r2c"     c_voc_r2c     :: Ptr VocCtx -> Ptr Double -> Ptr Double -> Int64 -> IO Int32
foreign import ccall safe "voc_c2r"     c_voc_c2r     :: Ptr VocCtx -> Ptr Double -> Ptr Double -> Int64 -> IO Int32
foreign import ccall safe "voc_do_fft"  c_voc_do_fft  :: Ptr VocCtx -> Ptr Double -> Ptr Double -> Int64 -> IO Int32
foreign import ccall safe "voc_do_ifft" c_voc_do_ifft :: Ptr VocCtx -> Ptr Double -> Ptr Double -> Int64 -> IO Int32

foreign import ccall safe "voc_analysis_step" c_voc_analysis_step
    :: Ptr VocCtx -> Ptr Double -> Ptr Double -> Ptr Double -> Ptr Double -> Ptr Double -> Int64 -> IO Int32
foreign import ccall safe "voc_synthesis_step" c_voc_synthesis_step
    :: Ptr VocCtx -> Ptr Double -> Ptr Double -> Ptr Double -> Ptr Double -> Ptr Double -> Int64 -> IO Int32
foreign import ccall safe "voc_add_frames" c_voc_add_frames
    :: Ptr VocCtx -> Ptr Double -> Ptr Double -> Ptr Double -> Ptr Double -> Ptr Double -> Ptr Double -> Int64 -> IO Int32

-- The reference's argument lists (no VocoderParams): hop, frame length, K, ... / hop, K, ...
foreign import ccall safe "voc_analysis_step_hn" c_voc_analysis_step_hn
    :: Int32 -> Int32 -> Int64 -> Ptr Double -> Ptr Double -> Ptr Double -> Ptr Double -> Ptr Double -> Int64 -> IO Int32
foreign import ccall safe "voc_synthesis_step_h" c_voc_synthesis_step_h
    :: Int32 -> Int64 -> Ptr Double -> Ptr Double -> Ptr Double -> Ptr Double -> Ptr Double -> Int64 -> IO Int32
-- | spec (count complex), mag, ph, count
foreign import ccall safe "voc_frame_from_complex" c_voc_frame_from_complex
    :: Ptr Double -> Ptr Double -> Ptr Double -> Int64 -> IO Int32
-- | mag, ph, spec (count complex), count
foreign import ccall safe "voc_frame_to_complex" c_voc_frame_to_complex
    :: Ptr Double -> Ptr Double -> Ptr Double -> Int64 -> IO Int32
foreign import ccall safe "voc_add_frames_n" c_voc_add_frames_n
    :: Ptr Double -> Ptr Double -> Ptr Double -> Ptr Double -> Ptr Double -> Ptr Double -> Int64 -> IO Int32

-- | phase_io[C*K], frames[C*F*W], mag[C*F*K], ph_inc[C*F*K], channels, frames
foreign import ccall safe "voc_analysis_stage" c_voc_analysis_stage
    :: Ptr VocCtx -> Ptr Double -> Ptr Double -> Ptr Double -> Ptr Double -> Int64 -> Int64 -> IO Int32
foreign import ccall safe "voc_synthesis_stage" c_voc_synthesis_stage
    :: Ptr VocCtx -> Ptr Double -> Ptr Double -> Ptr Double -> Ptr Double -> Int64 -> Int64 -> IO Int32

foreign import ccall unsafe "voc_num_frames"      c_voc_num_frames :: Ptr VocCtx -> Int64 -> IO Int64
foreign import ccall unsafe "voc_leftover_length" c_voc_leftover_length :: Ptr VocCtx -> Int64 -> IO Int64
foreign import ccall unsafe "voc_ola_length"      c_voc_ola_length :: Ptr VocCtx -> Int64 -> Int64 -> IO Int64
foreign import ccall safe "voc_frames_of" c_voc_frames_of
    :: Ptr VocCtx -> Ptr Double -> Int64 -> Int64 -> Int64 -> Ptr Double -> IO Int32
foreign import ccall safe "voc_sum_frames" c_voc_sum_frames
    :: Ptr VocCtx -> Ptr Double -> Int64 -> Int64 -> Int64 -> Ptr Double -> Int64 -> IO Int32

-- | x, x_stride, length, channels, chunk, y, y_stride, ana_phase_io, syn_phase_io, leftover, leftover_stride
foreign import ccall safe "voc_roundtrip" c_voc_roundtrip
    :: Ptr VocCtx -> Ptr Double -> Int64 -> Int64 -> Int64 -> Int64 -> Ptr Double -> Int64
    -> Ptr Double -> Ptr Double -> Ptr Double -> Int64 -> IO Int32

-- * Filters (Vocoder.Filter) on host arrays mag[C*F*K], ph_inc[C*F*K]

data VocFilter

foreign import ccall safe   "voc_filter_create"  c_voc_filter_create
    :: Ptr VocCtx -> Double -> Ptr (Ptr VocFilter) -> IO Int32
foreign import ccall unsafe "&voc_filter_destroy" p_voc_filter_destroy :: FunPtr (Ptr VocFilter -> IO ())
-- | kind (VOC_FILTER_* in vocoder_hip.h), params, nparams
foreign import ccall safe   "voc_filter_append"  c_voc_filter_append
    :: Ptr VocFilter -> Int32 -> Ptr Double -> Int32 -> IO Int32
foreign import ccall safe   "voc_filter_apply"   c_voc_filter_apply
    :: Ptr VocFilter -> Ptr Double -> Ptr Double -> Int64 -> Int64 -> IO Int32

-- * playSpeed (Vocoder.Conduit.Filter): numerator, denominator, input frames
foreign import ccall unsafe "voc_play_speed_frames" c_voc_play_speed_frames :: Int64 -> Int64 -> Int64 -> IO Int64
foreign import ccall unsafe "voc_play_speed_map"    c_voc_play_speed_map
    :: Int64 -> Int64 -> Int64 -> Ptr Int64 -> Int64 -> IO Int32

-- * Interleaved and 16-bit PCM device buffers at the file edge (device pointers; Audio.hs:51-55,
--   VocoderFile.hs:158,168-173)
foreign import ccall safe   "voc_deinterleave_dev"    c_voc_deinterleave_dev
    :: Ptr VocCtx -> Ptr Double -> Int64 -> Int64 -> Ptr Double -> Int64 -> Ptr () -> IO Int32
foreign import ccall safe   "voc_interleave_dev"      c_voc_interleave_dev
    :: Ptr VocCtx -> Ptr Double -> Int64 -> Int64 -> Int64 -> Ptr Double -> Ptr () -> IO Int32
foreign import ccall safe   "voc_pcm16_to_planar_dev" c_voc_pcm16_to_planar_dev
    :: Ptr VocCtx -> Ptr Int16 -> Int64 -> Int64 -> Ptr Double -> Int64 -> Ptr () -> IO Int32
foreign import ccall safe   "voc_planar_to_pcm16_dev" c_voc_planar_to_pcm16_dev
    :: Ptr VocCtx -> Ptr Double -> Int64 -> Int64 -> Int64 -> Int32 -> Ptr Int16 -> Ptr () -> IO Int32

-- * Chunk-at-a-time streaming (Vocoder.Conduit.process as a conduit)

data VocStream

foreign import ccall safe   "voc_stream_create"  c_voc_stream_create
    :: Ptr VocCtx -> Int64 -> Int64 -> Ptr VocFilter -> Ptr (Ptr VocStream) -> IO Int32
foreign import ccall unsafe "&voc_stream_destroy" p_voc_stream_destroy :: FunPtr (Ptr VocStream -> IO ())
foreign import ccall safe   "voc_stream_set_phases" c_voc_stream_set_phases
    :: Ptr VocStream -> Ptr Double -> Ptr Double -> IO Int32
foreign import ccall safe   "voc_stream_get_phases" c_voc_stream_get_phases
    :: Ptr VocStream -> Ptr Double -> Ptr Double -> IO Int32
foreign import ccall safe   "voc_stream_leftover"   c_voc_stream_leftover
    :: Ptr VocStream -> Ptr Double -> Int64 -> IO Int64
foreign import ccall unsafe "voc_stream_push_length"  c_voc_stream_push_length :: Ptr VocStream -> Int64 -> IO Int64
foreign import ccall unsafe "voc_stream_flush_length" c_voc_stream_flush_length :: Ptr VocStream -> IO Int64
-- | x, x_stride, n, y, y_stride, y_cap, n_out
foreign import ccall safe   "voc_stream_push" c_voc_stream_push
    :: Ptr VocStream -> Ptr Double -> Int64 -> Int64 -> Ptr Double -> Int64 -> Int64 -> Ptr Int64 -> IO Int32
foreign import ccall safe   "voc_stream_flush" c_voc_stream_flush
    :: Ptr VocStream -> Ptr Double -> Int64 -> Int64 -> Ptr Int64 -> IO Int32

-- * Device-resident entry points (pointers into GPU memory, a HIP stream as @Ptr ()@; for hosts
--   that keep their buffers on the device through a HIP binding of their own).  They enqueue and
--   return; the caller synchronises the stream.

-- | x, x_stride, nframes, channels, phase_io, mag, ph_inc, ldk, volume_fix, stream
foreign import ccall safe "voc_analysis_signal_dev" c_voc_analysis_signal_dev
    :: Ptr VocCtx -> Ptr Double -> Int64 -> Int64 -> Int64 -> Ptr Double -> Ptr Double -> Ptr Double -> Int64 -> Int32
    -> Ptr () -> IO Int32
-- | mag, ph_inc, ldk, nframes, channels, phase_io, y, y_stride, y_len, stream
foreign import ccall safe "voc_synthesis_signal_dev" c_voc_synthesis_signal_dev
    :: Ptr VocCtx -> Ptr Double -> Ptr Double -> Int64 -> Int64 -> Int64 -> Ptr Double -> Ptr Double -> Int64 -> Int64
    -> Ptr () -> IO Int32
-- | mag, mag_frames, mag_rows, ph_inc, inc_frames, inc_rows, ldk, nframes, channels, phase_io, y, y_stride, y_len,
--   stream: synthesis through the row maps of 'playSpeedMap' (no gathered copy of the STFT)
foreign import ccall safe "voc_synthesis_mapped_dev" c_voc_synthesis_mapped_dev
    :: Ptr VocCtx -> Ptr Double -> Int64 -> Ptr Int64 -> Ptr Double -> Int64 -> Ptr Int64 -> Int64 -> Int64 -> Int64
    -> Ptr Double -> Ptr Double -> Int64 -> Int64 -> Ptr () -> IO Int32
-- | ph_inc, ldk, nframes, channels, phase_io, stream
foreign import ccall safe "voc_phase_advance_dev" c_voc_phase_advance_dev
    :: Ptr VocCtx -> Ptr Double -> Int64 -> Int64 -> Int64 -> Ptr Double -> Ptr () -> IO Int32
-- | mag, ph_inc, ldk, nframes, channels, stream
foreign import ccall safe "voc_filter_apply_dev" c_voc_filter_apply_dev
    :: Ptr VocFilter -> Ptr Double -> Ptr Double -> Int64 -> Int64 -> Int64 -> Ptr () -> IO Int32
-- | flags4 <- {reads mag, writes mag, reads ph_inc, writes ph_inc}
foreign import ccall unsafe "voc_filter_touches" c_voc_filter_touches :: Ptr VocFilter -> Ptr Int32 -> IO Int32
-- | x, x_stride, length, channels, chunk, y, y_stride, ana_phase_io, syn_phase_io, leftover, leftover_stride,
--   workspace, workspace_bytes, stream
foreign import ccall safe "voc_roundtrip_dev" c_voc_roundtrip_dev
    :: Ptr VocCtx -> Ptr Double -> Int64 -> Int64 -> Int64 -> Int64 -> Ptr Double -> Int64 -> Ptr Double -> Ptr Double
    -> Ptr Double -> Int64 -> Ptr () -> Int64 -> Ptr () -> IO Int32
foreign import ccall unsafe "voc_roundtrip_workspace_bytes" c_voc_roundtrip_workspace_bytes
    :: Ptr VocCtx -> Int64 -> Int64 -> IO Int64
-- | as voc_roundtrip_dev without workspace arguments: allocates nothing, refuses inputs whose per-channel
--   scratch exceeds the library's static buffer
foreign import ccall safe "voc_roundtrip_static_dev" c_voc_roundtrip_static_dev
    :: Ptr VocCtx -> Ptr Double -> Int64 -> Int64 -> Int64 -> Int64 -> Ptr Double -> Int64 -> Ptr Double -> Ptr Double
    -> Ptr Double -> Int64 -> Ptr () -> IO Int32
-- | filter, mag, ph_inc, ldk, nframes, channels, phase_io, y, y_stride, y_len, stream: the chain, synthesisStage and
--   sumFramesE in one call (a pitch shift and / or one gain table is applied by synthesis as it reads the rows)
foreign import ccall safe "voc_synthesis_filtered_dev" c_voc_synthesis_filtered_dev
    :: Ptr VocFilter -> Ptr Double -> Ptr Double -> Int64 -> Int64 -> Int64 -> Ptr Double -> Ptr Double -> Int64 -> Int64
    -> Ptr () -> IO Int32
-- | mag, ph_inc, ldk, nframes, channels, first_frame, first_channel, stream: the chain on a piece of a
--   longer STFT (position-dependent filters number frames and channels from the given offsets)
foreign import ccall safe "voc_filter_apply_at_dev" c_voc_filter_apply_at_dev
    :: Ptr VocFilter -> Ptr Double -> Ptr Double -> Int64 -> Int64 -> Int64 -> Int64 -> Int64 -> Ptr () -> IO Int32
-- | op (0 multiply, 1 divide, 2 add), mag1, inc1, mag2, inc2, mag_out, inc_out, count, stream
foreign import ccall safe "voc_binary_frames_dev" c_voc_binary_frames_dev
    :: Ptr VocCtx -> Int32 -> Ptr Double -> Ptr Double -> Ptr Double -> Ptr Double -> Ptr Double -> Ptr Double -> Int64
    -> Ptr () -> IO Int32

-- * Single-precision forms (contexts created with precision 1; every array is Float).  Argument lists as the
--   double forms above.
foreign import ccall safe "voc_roundtrip_f32" c_voc_roundtrip_f32
    :: Ptr VocCtx -> Ptr Float -> Int64 -> Int64 -> Int64 -> Int64 -> Ptr Float -> Int64
    -> Ptr Float -> Ptr Float -> Ptr Float -> Int64 -> IO Int32
foreign import ccall safe "voc_analysis_stage_f32" c_voc_analysis_stage_f32
    :: Ptr VocCtx -> Ptr Float -> Ptr Float -> Ptr Float -> Ptr Float -> Int64 -> Int64 -> IO Int32
foreign import ccall safe "voc_synthesis_stage_f32" c_voc_synthesis_stage_f32
    :: Ptr VocCtx -> Ptr Float -> Ptr Float -> Ptr Float -> Ptr Float -> Int64 -> Int64 -> IO Int32
foreign import ccall safe "voc_r2c_f32"     c_voc_r2c_f32     :: Ptr VocCtx -> Ptr Float -> Ptr Float -> Int64 -> IO Int32
foreign import ccall safe "voc_c2r_f32"     c_voc_c2r_f32     :: Ptr VocCtx -> Ptr Float -> Ptr Float -> Int64 -> IO Int32
foreign import ccall safe "voc_do_fft_f32"  c_voc_do_fft_f32  :: Ptr VocCtx -> Ptr Float -> Ptr Float -> Int64 -> IO Int32
foreign import ccall safe "voc_do_ifft_f32" c_voc_do_ifft_f32 :: Ptr VocCtx -> Ptr Float -> Ptr Float -> Int64 -> IO Int32
foreign import ccall safe "voc_analysis_step_f32" c_voc_analysis_step_f32
    :: Ptr VocCtx -> Ptr Float -> Ptr Float -> Ptr Float -> Ptr Float -> Ptr Float -> Int64 -> IO Int32
foreign import ccall safe "voc_synthesis_step_f32" c_voc_synthesis_step_f32
    :: Ptr VocCtx -> Ptr Float -> Ptr Float -> Ptr Float -> Ptr Float -> Ptr Float -> Int64 -> IO Int32
foreign import ccall safe "voc_add_frames_f32" c_voc_add_frames_f32
    :: Ptr VocCtx -> Ptr Float -> Ptr Float -> Ptr Float -> Ptr Float -> Ptr Float -> Ptr Float -> Int64 -> IO Int32
foreign import ccall safe "voc_filter_apply_f32" c_voc_filter_apply_f32
    :: Ptr VocFilter -> Ptr Float -> Ptr Float -> Int64 -> Int64 -> IO Int32
foreign import ccall safe "voc_stream_set_phases_f32" c_voc_stream_set_phases_f32
    :: Ptr VocStream -> Ptr Float -> Ptr Float -> IO Int32
foreign import ccall safe "voc_stream_get_phases_f32" c_voc_stream_get_phases_f32
    :: Ptr VocStream -> Ptr Float -> Ptr Float -> IO Int32
foreign import ccall safe "voc_stream_leftover_f32" c_voc_stream_leftover_f32
    :: Ptr VocStream -> Ptr Float -> Int64 -> IO Int64
foreign import ccall safe "voc_stream_push_f32" c_voc_stream_push_f32
    :: Ptr VocStream -> Ptr Float -> Int64 -> Int64 -> Ptr Float -> Int64 -> Int64 -> Ptr Int64 -> IO Int32
foreign import ccall safe "voc_stream_flush_f32" c_voc_stream_flush_f32
    :: Ptr VocStream -> Ptr Float -> Int64 -> Int64 -> Ptr Int64 -> IO Int32
-- | playSpeed inside a stream: num, den, chain that runs on the yielded frames (or nullPtr)
foreign import ccall safe "voc_stream_set_play_speed" c_voc_stream_set_play_speed
    :: Ptr VocStream -> Int64 -> Int64 -> Ptr VocFilter -> IO Int32
