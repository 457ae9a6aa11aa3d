"""h264decode_amd -- MI355X-native H.264 Annex-B decode path (HIP, gfx950) behind the API names of
the Go package `h264` of mrmod/h264decode.

The pixel path exists only on the GPU: importing works anywhere (header parsing is host code), but
creating a Decoder without a HIP device or without the built libh264mi.so raises."""
from ._lib import H264MIError, build, lib, load, load_hooks  # noqa: F401
from .h264 import (  # noqa: F401
    NALU_TYPE_NAMES, PPS, SPS, Decoder, NalUnit, NewNalUnit, NewPPS, NewSPS, NewSliceContext, SliceContext, SliceHeader,
    VideoStream, read_nal_units, MapUnitToSliceGroupMap, MbToSliceGroupMap, nextMbAddress, AccessUnitSplitter, DisplayOrder, H264Reader, handleConnection, ByteStreamReader, BatchServer)
from .mbtype import (  # noqa: F401
    MB_TYPE_INFERRED, ISliceMbType, SISliceMbType, PSliceMbType, BSliceMbType, MbTypeName, MbPartPredMode, NumMbPart, PicWidthInMbs,
    PicHeightInMapUnits, PicSizeInMapUnits, FrameHeightInMbs, PicHeightInMbs, PicSizeInMbs, SubWidthC, SubHeightC, MbWidthC, MbHeightC,
    MbaffFrameFlag, CodedBlockPatternLuma, CodedBlockPatternChroma, Clip3, Clip1y, Clipc, SliceQPy, PreCtxState, SliceData, NewSliceData, MbPred)
