{-|
Module      : Vocoder.HIP.Conduit
Description : Vocoder.Conduit.process over the streaming state of libvocoder_hip.so

@Vocoder.Conduit.process par c@ (vocoder-conduit/src/Vocoder/Conduit.hs:50-53) takes an arbitrary
conduit of STFT frames; that generality keeps working through the one-frame 'analysisBlock' /
'synthesisBlock' of "Vocoder.HIP".  When the frame conduit is a chain of the library's filters
(or the identity), the whole pipeline -- framing, analysis, volumeFix, filter, synthesis,
overlap-add -- runs on the device with one FFI call per awaited chunk: that is 'processOnDevice'.
What it yields, chunk by chunk, is what @process par (runFilter f step)@ yields, and its result
is the state @processWith@ returns (leftover samples and the two phases, Conduit.hs:53), so
sources can be concatenated as vocoder-audio does (Audio.hs:60-69).

Not compiled where it was written (no GHC there); it binds the C ABI exercised by the C++ and
Python hosts (voc_stream_*, include/vocoder_hip.h).
-}
module Vocoder.HIP.Conduit (processOnDevice) where

import Control.Monad
import Control.Monad.IO.Class
import Data.Conduit
import Data.Int
import qualified Data.Vector.Storable as V
import qualified Data.Vector.Storable.Mutable as MV
import Foreign.ForeignPtr
import Foreign.Marshal.Alloc
import Foreign.Ptr
import Foreign.Storable

import Vocoder.HIP
import Vocoder.HIP.FFI

-- | One channel, chunks of any sizes in, chunks of @chunk@ samples out.  @flt@ is a filter chain
--   built with @voc_filter_create@ / @voc_filter_append@ (see "Vocoder.HIP.Filter"), or 'nullPtr'.
processOnDevice :: MonadIO m
                => VocoderParams -> Int -> Ptr VocFilter -> (Phase, Phase)
                -> ConduitT Frame Frame m (Frame, (Phase, Phase))
processOnDevice par chunk flt (aph, sph) = do
    st <- liftIO $ withForeignPtr (vocCtx par) $ \ctx -> alloca $ \pout -> do
        check "stream" $ c_voc_stream_create ctx 1 (fromIntegral chunk) flt pout
        s <- peek pout >>= newForeignPtr p_voc_stream_destroy
        withForeignPtr s $ \ps -> V.unsafeWith aph $ \pa -> V.unsafeWith sph $ \pb ->
            check "stream" $ c_voc_stream_set_phases ps pa pb
        return s
    let emit v = forM_ [0, chunk .. V.length v - 1] $ \i -> yield (V.slice i chunk v)
    awaitForever $ \x -> do
        out <- liftIO $ withForeignPtr st $ \ps -> do
            m <- fromIntegral <$> c_voc_stream_push_length ps (fromIntegral (V.length x))
            y <- MV.new (max 1 m)
            V.unsafeWith x $ \px -> MV.unsafeWith y $ \py -> alloca $ \pn ->
                check "stream push" $ c_voc_stream_push ps px (fromIntegral (V.length x)) (fromIntegral (V.length x))
                                                        py (fromIntegral m) (fromIntegral m) (pn :: Ptr Int64)
            V.take m <$> V.unsafeFreeze y
        emit out
    (rest, left, phs) <- liftIO $ withForeignPtr st $ \ps -> do
        m <- fromIntegral <$> c_voc_stream_flush_length ps
        y <- MV.new (max 1 m)
        MV.unsafeWith y $ \py -> alloca $ \pn ->
            check "stream flush" $ c_voc_stream_flush ps py (fromIntegral m) (fromIntegral m) (pn :: Ptr Int64)
        k <- fromIntegral <$> c_voc_stream_leftover ps nullPtr 0
        l <- MV.new (max 1 k)
        _ <- MV.unsafeWith l $ \pl -> c_voc_stream_leftover ps pl (fromIntegral k)
        let nk = vocFreqFrameLength par
        a <- MV.new nk
        b <- MV.new nk
        MV.unsafeWith a $ \pa -> MV.unsafeWith b $ \pb -> check "stream" $ c_voc_stream_get_phases ps pa pb
        (,,) <$> (V.take m <$> V.unsafeFreeze y) <*> (V.take k <$> V.unsafeFreeze l)
             <*> ((,) <$> V.unsafeFreeze a <*> V.unsafeFreeze b)
    emit rest
    return (left, phs)
