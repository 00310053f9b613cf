{-|
Module      : Vocoder.HIP.Filter
Description : Vocoder.Filter over libvocoder_hip.so

Drop-in for the filter constructors of @Vocoder.Filter@ (vocoder/src/Vocoder/Filter.hs:67-157).
The reference's @Filter m@ is a function on one STFT frame; here a filter is a description (a
chain in 'composeFilters' order) that 'applyFilter' runs on a whole container of frames in one
FFI call, on the device.  Not compiled where it was written (no GHC there); it binds the same C
ABI the C++ and Python hosts exercise.
-}
module Vocoder.HIP.Filter
    ( FilterDesc, idFilter, composeFilters, amplify, linearAmplitudeFilter
    , lowpassBrickwall, highpassBrickwall, bandpassBrickwall, bandstopBrickwall
    , lowpassButterworth, highpassButterworth, bandpassButterworth, bandstopButterworth
    , pitchShiftInterpolate, convolutionFilter, envelopeFilter, randomPhaseFilter
    , applyFilter, applyAddFilters, playSpeedMap
    ) where

import Control.Monad
import Data.Foldable (toList)
import Data.Int
import Data.Ratio
import Data.Traversable (mapAccumL)
import qualified Data.Vector.Storable as V
import qualified Data.Vector.Storable.Mutable as MV
import Foreign.ForeignPtr
import Foreign.Marshal.Alloc
import Foreign.Ptr
import Foreign.Storable
import System.IO.Unsafe (unsafePerformIO)

import Vocoder.HIP
import Vocoder.HIP.FFI

-- | One stage: Left (kind, params) is a named filter of the library; Right f is
--   @linearAmplitudeFilter f@, tabulated at @step * k@ when the chain is built.
type Stage = Either (Int32, [Double]) (Double -> Double)
newtype FilterDesc = FilterDesc [Stage]

idFilter :: FilterDesc
idFilter = FilterDesc []

-- | composeFilters f1 f2: f1 first (Filter.hs:54-55)
composeFilters :: FilterDesc -> FilterDesc -> FilterDesc
composeFilters (FilterDesc a) (FilterDesc b) = FilterDesc (a ++ b)

named :: Int32 -> [Double] -> FilterDesc
named k q = FilterDesc [Left (k, q)]

amplify, lowpassBrickwall, highpassBrickwall, pitchShiftInterpolate :: Double -> FilterDesc
amplify k = named 0 [k]
lowpassBrickwall t = named 1 [t]
highpassBrickwall t = named 2 [t]
pitchShiftInterpolate n = named 10 [n]

bandpassBrickwall, bandstopBrickwall, lowpassButterworth, highpassButterworth :: Double -> Double -> FilterDesc
bandpassBrickwall t u = named 3 [t, u]
bandstopBrickwall t u = named 4 [t, u]
lowpassButterworth n t = named 5 [n, t]
highpassButterworth n t = named 6 [n, t]

bandpassButterworth, bandstopButterworth :: Double -> Double -> Double -> FilterDesc
bandpassButterworth n t u = named 7 [n, t, u]
bandstopButterworth n t u = named 8 [n, t, u]

linearAmplitudeFilter :: (Double -> Double) -> FilterDesc
linearAmplitudeFilter f = FilterDesc [Right f]

convolutionFilter :: V.Vector Double -> FilterDesc
convolutionFilter ker = named 11 (V.toList ker)

envelopeFilter :: Int -> FilterDesc
envelopeFilter ksize = named 12 [fromIntegral ksize]

-- | The reference draws from the global generator; the library takes a seed.
randomPhaseFilter :: Int -> FilterDesc
randomPhaseFilter seed = named 13 [fromIntegral seed]

-- | Runs the chain on every frame of the container (one FFI call).
applyFilter :: Traversable t => VocoderParams -> FreqStep -> FilterDesc -> t STFTFrame -> t STFTFrame
applyFilter par step (FilterDesc stages) frs = unsafePerformIO $ do
    let fl = toList frs
        f = length fl
        k = vocFreqFrameLength par
    mag <- V.thaw (V.concat (map fst fl))
    inc <- V.thaw (V.concat (map snd fl))
    withForeignPtr (vocCtx par) $ \ctx -> alloca $ \pout -> do
        check "filter" $ c_voc_filter_create ctx step pout
        flt <- peek pout >>= newForeignPtr p_voc_filter_destroy
        withForeignPtr flt $ \pf -> do
            forM_ stages $ \st -> do
                let (kind, q) = case st of
                        Left kq -> kq
                        Right g -> (9, [g (step * fromIntegral i) | i <- [0 .. k - 1]])
                V.unsafeWith (V.fromList q) $ \pq ->
                    check "filter" $ c_voc_filter_append pf kind pq (fromIntegral (length q))
            MV.unsafeWith mag $ \pm -> MV.unsafeWith inc $ \pin ->
                check "filter" $ c_voc_filter_apply pf pm pin (fromIntegral f) 1
    magv <- V.unsafeFreeze mag
    incv <- V.unsafeFreeze inc
    let slice v i = V.slice (i * k) k v
        (_, out) = mapAccumL (\i _ -> (i + 1 :: Int, (slice magv i, slice incv i))) 0 frs
    return out

-- | addFilters f1 f2 (Filter.hs:58-59) on a container of frames: both chains see the same
--   frames, their outputs are added frame by frame with 'addFrames'.
applyAddFilters :: Traversable t => VocoderParams -> FreqStep -> FilterDesc -> FilterDesc -> t STFTFrame -> t STFTFrame
applyAddFilters par step f1 f2 frs = snd (mapAccumL pick (toList (applyFilter par step f2 frs)) (applyFilter par step f1 frs))
    where pick (b : bs) a = (bs, addFrames a b)
          pick [] _ = error "applyAddFilters: containers of different shape"

type FreqStep = Double

-- | playSpeed coeff on n input frames: the input frame each output frame repeats
--   (Vocoder.Conduit.Filter.playSpeed, Conduit/Filter.hs:122-136).
playSpeedMap :: Rational -> Int -> [Int]
playSpeedMap coeff n = unsafePerformIO $ do
    let nu = fromIntegral (numerator coeff) :: Int64
        de = fromIntegral (denominator coeff) :: Int64
    m <- fromIntegral <$> c_voc_play_speed_frames nu de (fromIntegral n)
    buf <- MV.new (max 1 m)
    MV.unsafeWith buf $ \p -> check "playSpeed" $ c_voc_play_speed_map nu de (fromIntegral n) p (fromIntegral m)
    map fromIntegral . take m . V.toList <$> V.unsafeFreeze buf
