{-|
Module      : Vocoder.HIP
Description : The @Vocoder@ core API on top of libvocoder_hip.so

Drop-in for vocoder/src/Vocoder.hs: every name of its export list (Vocoder.hs:12-42) is exported
here with the reference's type, so a package that says @import Vocoder@ builds against this module
renamed (or re-exported) as @Vocoder@.  Only the bodies change:

  * 'VocoderParams' holds a 'ForeignPtr' to the library context instead of two FFTW plans
    (Vocoder.hs:89-99); 'vocoderParams' has the same signature (Vocoder.hs:114-115).
  * 'analysisStage' / 'synthesisStage' (Vocoder.hs:149-150,177-178) marshal the whole
    @Traversable@ of frames into one contiguous buffer and cross the FFI once;
    'analysisBlock' / 'synthesisBlock' (Vocoder.hs:153-154,181-182) are the one-frame case.
  * a non-zero status becomes 'error', as the reference does for bad sizes (Vocoder.hs:132) and as
    vector-fftw's @execute@ does for a length mismatch.

  * 'doFFT' / 'doIFFT' (Vocoder.hs:143-145,194-196) are one-frame calls on the context;
    'analysisStep', 'synthesisStep', 'frameFromComplex', 'frameToComplex' and 'addFrames'
    (Vocoder.hs:158,186,213,217,221) take no 'VocoderParams' in the reference and none here: they
    bind the entry points of the library that need no context.

vocoder-conduit and vocoder-dunai need no source change: they only use names exported here
(Conduit.hs:39,43,64,68; Dunai.hs:30,34).  Could not be compiled where it was written (no GHC).
-}
module Vocoder.HIP
    ( -- the export list of Vocoder.hs:12-42, in its order
      Moduli, Phase, PhaseInc, Frame, Window, HopSize, Length, STFTFrame, FFTOutput
    , VocoderParams, vocoderParams, vocFrameLength, vocInputFrameLength, vocFreqFrameLength
    , vocHopSize, vocWindow, doFFT, doIFFT, analysisBlock, analysisStep, analysisStage
    , synthesisBlock, synthesisStep, synthesisStage, zeroPhase, volumeCoeff
    , frameFromComplex, frameToComplex, addFrames
      -- beyond the reference: the fused round trip on whole channels
    , processChannels
    , vocCtx, check   -- for the sibling modules Vocoder.HIP.Filter and Vocoder.HIP.Conduit
    ) where

import Control.Monad (when)
import Data.Complex (Complex)
import Data.Int (Int32)
import Data.Foldable (toList)
import Data.Traversable (mapAccumL)
import qualified Data.Vector.Storable as V
import qualified Data.Vector.Storable.Mutable as MV
import Foreign.C.String (peekCString)
import Foreign.ForeignPtr
import Foreign.Marshal.Alloc (alloca)
import Foreign.Ptr
import Foreign.Storable (peek)
import System.IO.Unsafe (unsafePerformIO)
import Vocoder.HIP.FFI

type Moduli = V.Vector Double
type Phase = V.Vector Double
type PhaseInc = V.Vector Double
type Frame = V.Vector Double
type Window = Frame
type HopSize = Int
type Length = Int
type STFTFrame = (Moduli, PhaseInc)
type FFTOutput = V.Vector (Complex Double)

data VocoderParams = VocoderParams
    { vocCtx :: ForeignPtr VocCtx
    , vocFrameLength :: Int
    , vocHopSize :: Int
    , vocWindow :: Window
    }

vocInputFrameLength :: VocoderParams -> Int
vocInputFrameLength = V.length . vocWindow

vocFreqFrameLength :: VocoderParams -> Int
vocFreqFrameLength par = vocFrameLength par `div` 2 + 1

-- | A non-zero status becomes `error`, the reference's own convention (Vocoder.hs:132).
check :: String -> IO Int32 -> IO ()
check what act = do
    st <- act
    when (st /= 0) $ do
        msg <- c_voc_strerror st >>= peekCString
        error (what ++ ": " ++ msg)

-- | vocoderParams len hop window (Vocoder.hs:114-115)
vocoderParams :: Int -> Int -> Window -> VocoderParams
vocoderParams len hs wnd = unsafePerformIO $
    V.unsafeWith wnd $ \pw -> alloca $ \pout -> do
        check "vocoderParams" $
            c_voc_ctx_create (fromIntegral len) (fromIntegral hs) pw (fromIntegral (V.length wnd)) 0 0 pout
        ctx <- peek pout >>= newForeignPtr p_voc_ctx_destroy
        return (VocoderParams ctx len hs wnd)
{-# NOINLINE vocoderParams #-}

zeroPhase :: VocoderParams -> Phase
zeroPhase par = V.replicate (vocFreqFrameLength par) 0

volumeCoeff :: VocoderParams -> Double
volumeCoeff par = unsafePerformIO $ withForeignPtr (vocCtx par) c_voc_volume_coeff

-- | doFFT (Vocoder.hs:143-145): window, zero padding, rewind, real-to-complex transform.
doFFT :: VocoderParams -> Frame -> FFTOutput
doFFT par fr = unsafePerformIO $ do
    when (V.length fr /= vocInputFrameLength par) $ error "doFFT: frame length mismatch"
    out <- MV.new (2 * vocFreqFrameLength par)
    withForeignPtr (vocCtx par) $ \ctx -> V.unsafeWith fr $ \pf -> MV.unsafeWith out $ \po ->
        check "doFFT" $ c_voc_do_fft ctx pf po 1
    V.unsafeCast <$> V.unsafeFreeze out

-- | doIFFT (Vocoder.hs:194-196): complex-to-real transform, rewind, cutCenter, window.
doIFFT :: VocoderParams -> FFTOutput -> Frame
doIFFT par spec = unsafePerformIO $ do
    when (V.length spec /= vocFreqFrameLength par) $ error "doIFFT: spectrum length mismatch"
    out <- MV.new (vocInputFrameLength par)
    withForeignPtr (vocCtx par) $ \ctx -> V.unsafeWith (V.unsafeCast spec) $ \ps -> MV.unsafeWith out $ \po ->
        check "doIFFT" $ c_voc_do_ifft ctx ps po 1
    V.unsafeFreeze out

-- | analysisStep h eN prev_ph vec (Vocoder.hs:158-163).  The two vectors have one length in every
--   use of the reference; if they differ, all three results are cut to the shorter (the reference
--   cuts only ph_inc, through its zipWith).
analysisStep :: HopSize -> Length -> Phase -> FFTOutput -> (Phase, STFTFrame)
analysisStep h eN prev_ph vec = unsafePerformIO $ do
    let k = min (V.length prev_ph) (V.length vec)
    ph <- MV.new k
    mag <- MV.new k
    inc <- MV.new k
    V.unsafeWith prev_ph $ \pp -> V.unsafeWith (V.unsafeCast vec) $ \pv ->
        MV.unsafeWith ph $ \po -> MV.unsafeWith mag $ \pm -> MV.unsafeWith inc $ \pin ->
            check "analysisStep" $
                c_voc_analysis_step_hn (fromIntegral h) (fromIntegral eN) (fromIntegral k) pp pv po pm pin 1
    phv <- V.unsafeFreeze ph
    magv <- V.unsafeFreeze mag
    incv <- V.unsafeFreeze inc
    return (phv, (magv, incv))

-- | synthesisStep hop ph (mag, ph_inc) (Vocoder.hs:186-190)
synthesisStep :: HopSize -> Phase -> STFTFrame -> (Phase, FFTOutput)
synthesisStep hop ph (mag, ph_inc) = unsafePerformIO $ do
    let k = minimum [V.length ph, V.length mag, V.length ph_inc]
    new_ph <- MV.new k
    spec <- MV.new (2 * k)
    V.unsafeWith ph $ \pp -> V.unsafeWith mag $ \pm -> V.unsafeWith ph_inc $ \pin ->
        MV.unsafeWith new_ph $ \po -> MV.unsafeWith spec $ \ps ->
            check "synthesisStep" $ c_voc_synthesis_step_h (fromIntegral hop) (fromIntegral k) pp pm pin po ps 1
    (,) <$> V.unsafeFreeze new_ph <*> (V.unsafeCast <$> V.unsafeFreeze spec)

-- | frameFromComplex (Vocoder.hs:217-218)
frameFromComplex :: FFTOutput -> STFTFrame
frameFromComplex vec = unsafePerformIO $ do
    let k = V.length vec
    mag <- MV.new k
    ph <- MV.new k
    V.unsafeWith (V.unsafeCast vec) $ \pv -> MV.unsafeWith mag $ \pm -> MV.unsafeWith ph $ \pp ->
        check "frameFromComplex" $ c_voc_frame_from_complex pv pm pp (fromIntegral k)
    (,) <$> V.unsafeFreeze mag <*> V.unsafeFreeze ph

-- | frameToComplex (Vocoder.hs:213-214)
frameToComplex :: STFTFrame -> FFTOutput
frameToComplex (mag, ph) = unsafePerformIO $ do
    let k = min (V.length mag) (V.length ph)
    spec <- MV.new (2 * k)
    V.unsafeWith mag $ \pm -> V.unsafeWith ph $ \pp -> MV.unsafeWith spec $ \ps ->
        check "frameToComplex" $ c_voc_frame_to_complex pm pp ps (fromIntegral k)
    V.unsafeCast <$> V.unsafeFreeze spec

-- | analysisStage (Vocoder.hs:149-150): one FFI call for the whole container of frames.
analysisStage :: Traversable t => VocoderParams -> Phase -> t Frame -> (Phase, t STFTFrame)
analysisStage par ph frs = unsafePerformIO $ do
    let fl = toList frs
        f = length fl
        k = vocFreqFrameLength par
    phm <- V.thaw ph
    mag <- MV.new (f * k)
    inc <- MV.new (f * k)
    withForeignPtr (vocCtx par) $ \ctx ->
        V.unsafeWith (V.concat fl) $ \pfr -> MV.unsafeWith phm $ \pph ->
        MV.unsafeWith mag $ \pm -> MV.unsafeWith inc $ \pin ->
            check "analysisStage" $ c_voc_analysis_stage ctx pph pfr pm pin 1 (fromIntegral f)
    ph' <- V.freeze phm
    magv <- V.unsafeFreeze mag
    incv <- V.unsafeFreeze inc
    let slice v i = V.slice (i * k) k v
        (_, out) = mapAccumL (\i _ -> (i + 1, (slice magv i, slice incv i))) 0 frs
    return (ph', out)

analysisBlock :: VocoderParams -> Phase -> Frame -> (Phase, STFTFrame)
analysisBlock par ph fr = let (ph', [r]) = analysisStage par ph [fr] in (ph', r)

-- | synthesisStage (Vocoder.hs:177-178)
synthesisStage :: Traversable t => VocoderParams -> Phase -> t STFTFrame -> (Phase, t Frame)
synthesisStage par ph frs = unsafePerformIO $ do
    let fl = toList frs
        f = length fl
        w = vocInputFrameLength par
    phm <- V.thaw ph
    out <- MV.new (f * w)
    withForeignPtr (vocCtx par) $ \ctx ->
        V.unsafeWith (V.concat (map fst fl)) $ \pm -> V.unsafeWith (V.concat (map snd fl)) $ \pin ->
        MV.unsafeWith phm $ \pph -> MV.unsafeWith out $ \po ->
            check "synthesisStage" $ c_voc_synthesis_stage ctx pph pm pin po 1 (fromIntegral f)
    ph' <- V.freeze phm
    outv <- V.unsafeFreeze out
    let (_, res) = mapAccumL (\i _ -> (i + 1, V.slice (i * w) w outv)) 0 frs
    return (ph', res)

synthesisBlock :: VocoderParams -> Phase -> STFTFrame -> (Phase, Frame)
synthesisBlock par ph fr = let (ph', [r]) = synthesisStage par ph [fr] in (ph', r)

-- | The fused path of Vocoder.Conduit.process with the identity filter, for C planar channels
--   of equal length: (leftover ++ input per channel, phases) -> (output, leftover', phases').
processChannels :: VocoderParams -> Int -> [V.Vector Double] -> ([Phase], [Phase])
                -> ([V.Vector Double], [V.Vector Double], ([Phase], [Phase]))
processChannels par chunk xs (aphs, sphs) = unsafePerformIO $ do
    let c = length xs
        len = V.length (head xs)
        k = vocFreqFrameLength par
    withForeignPtr (vocCtx par) $ \ctx -> do
        nf <- c_voc_num_frames ctx (fromIntegral len)
        ol <- fromIntegral <$> c_voc_ola_length ctx nf (fromIntegral chunk)
        keep <- fromIntegral <$> c_voc_leftover_length ctx (fromIntegral len)
        y <- MV.new (c * ol)
        left <- MV.new (c * max 1 keep)
        am <- V.thaw (V.concat aphs)
        sm <- V.thaw (V.concat sphs)
        V.unsafeWith (V.concat xs) $ \px -> MV.unsafeWith y $ \py -> MV.unsafeWith am $ \pa ->
            MV.unsafeWith sm $ \ps -> MV.unsafeWith left $ \pl ->
                check "processChannels" $
                    c_voc_roundtrip ctx px (fromIntegral len) (fromIntegral len) (fromIntegral c)
                        (fromIntegral chunk) py (fromIntegral ol) pa ps pl (fromIntegral keep)
        yv <- V.unsafeFreeze y
        lv <- V.unsafeFreeze left
        av <- V.unsafeFreeze am
        sv <- V.unsafeFreeze sm
        let rows n v = [V.slice (i * n) n v | i <- [0 .. c - 1]]
        return (rows ol yv, rows keep lv, (rows k av, rows k sv))

-- | Adds STFT frames (Vocoder.hs:220-222): polar to complex, sum, complex to polar, on the device.
addFrames :: STFTFrame -> STFTFrame -> STFTFrame
addFrames (m1, i1) (m2, i2) = unsafePerformIO $ do
    let k = minimum (map V.length [m1, i1, m2, i2])   -- zipWith, as in the reference
    mo <- MV.new k
    io <- MV.new k
    V.unsafeWith m1 $ \pm1 -> V.unsafeWith i1 $ \pi1 -> V.unsafeWith m2 $ \pm2 -> V.unsafeWith i2 $ \pi2 ->
        MV.unsafeWith mo $ \pmo -> MV.unsafeWith io $ \pio ->
            check "addFrames" $ c_voc_add_frames_n pm1 pi1 pm2 pi2 pmo pio (fromIntegral k)
    (,) <$> V.unsafeFreeze mo <*> V.unsafeFreeze io
