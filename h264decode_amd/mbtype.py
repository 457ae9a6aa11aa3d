"""Macroblock-type tables and the pure helper functions of the reference's package `h264`, under the reference's names.

Reference                          -> here
  ISliceMbType ... BSliceMbType      h264/mbType.go:8-72     -> same names (Tables 7-11, 7-12, 7-13, 7-14)
  MB_TYPE_INFERRED                   h264/mbType.go:5        -> MB_TYPE_INFERRED
  MbTypeName(sliceType, mbType)      h264/mbType.go:75-88    -> MbTypeName
  MbPartPredMode(data, sliceType, mbType, partition)  h264/mbType.go:90-163 -> MbPartPredMode
  NumMbPart                          h264/slice.go:236-250   -> NumMbPart(mbTypeName)  (the reference passes an address: App. A27)
  PicWidthInMbs ... PicSizeInMbs     h264/slice.go:159-176   -> same names
  SubWidthC / SubHeightC             h264/slice.go:179-219   -> same names (Table 6-1)
  MbWidthC / MbHeightC               h264/cabac.go:31-44     -> same names
  CodedBlockPatternLuma / Chroma     h264/slice.go:222-227   -> same names (7-36)
  MbaffFrameFlag                     h264/slice.go:563-568   -> MbaffFrameFlag
  SliceQPy, PreCtxState, Clip3, Clip1y, Clipc   h264/cabac.go:113-139 -> same names
  SliceData / NewSliceData / MbPred  h264/slice.go:77-102, 570, 252 -> SliceData, NewSliceData (from the GPU's macroblock records)

Where the reference is wrong the values here follow the spec: MbPartPredMode answers for both
partitions and gets B macroblocks right (A32: the reference overwrites "BiPred" with "Direct"); "SP" exists in MbTypeName.
The macroblock layer itself is decoded on the GPU: NewSliceData does not parse bits, it presents the records the entropy
kernel wrote (h264mi_frame_read_mbrecs) under the reference's SliceData field names."""
import numpy as np

MB_TYPE_INFERRED = 1000  # h264/mbType.go:5

ISliceMbType = {0: "I_NxN", 25: "I_PCM"}
for _t in range(1, 25):  # Table 7-11: I_16x16_<predMode>_<cbpChroma>_<cbpLuma != 0>
    ISliceMbType[_t] = "I_16x16_%d_%d_%d" % ((_t - 1) % 4, ((_t - 1) // 4) % 3, (_t - 1) // 12)
SISliceMbType = {0: "SI"}
PSliceMbType = {0: "P_L0_16x16", 1: "P_L0_L0_16x8", 2: "P_L0_L0_8x16", 3: "P_8x8", 4: "P_8x8ref0", MB_TYPE_INFERRED: "P_Skip"}
BSliceMbType = {0: "B_Direct_16x16", 1: "B_L0_16x16", 2: "B_L1_16x16", 3: "B_Bi_16x16", 22: "B_8x8", MB_TYPE_INFERRED: "B_Skip"}
_B_PAIRS = [("L0", "L0"), ("L1", "L1"), ("L0", "L1"), ("L1", "L0"), ("L0", "Bi"), ("L1", "Bi"), ("Bi", "L0"), ("Bi", "L1"), ("Bi", "Bi")]
for _i, (_a, _b) in enumerate(_B_PAIRS):  # Table 7-14, mb_type 4..21: two partitions, 16x8 (even) / 8x16 (odd)
    BSliceMbType[4 + 2 * _i] = "B_%s_%s_16x8" % (_a, _b)
    BSliceMbType[5 + 2 * _i] = "B_%s_%s_8x16" % (_a, _b)


def MbTypeName(sliceType: str, mbType: int) -> str:
    """h264/mbType.go:75.  In P / SP / B slices mb_type 5.. (23.. for B) are the intra types of Table 7-11 (7.4.5)."""
    if sliceType in ("P", "SP"):
        return PSliceMbType.get(mbType) or ISliceMbType.get(mbType - 5, "NaSliceType")
    if sliceType == "B":
        return BSliceMbType.get(mbType) or ISliceMbType.get(mbType - 23, "NaSliceType")
    if sliceType == "I":
        return ISliceMbType.get(mbType, "NaSliceType")
    if sliceType == "SI":
        return SISliceMbType.get(mbType) or ISliceMbType.get(mbType - 1, "NaSliceType")
    return "NaSliceType"


def NumMbPart(mbTypeName: str) -> int:
    """Tables 7-13 / 7-14 (h264/slice.go:236-250 means this)."""
    if mbTypeName.endswith(("16x8", "8x16")):
        return 2
    if mbTypeName in ("P_8x8", "P_8x8ref0", "B_8x8"):
        return 4
    return 1


def MbPartPredMode(data, sliceType: str, mbType: int, partition: int) -> str:
    """h264/mbType.go:90 -- Tables 7-11, 7-13, 7-14.  `data` supplies TransformSize8x8Flag for I_NxN (may be None)."""
    name = MbTypeName(sliceType, mbType)
    if name == "I_NxN":
        return "Intra_8x8" if (data is not None and getattr(data, "TransformSize8x8Flag", False)) else "Intra_4x4"
    if name.startswith("I_16x16"):
        return "Intra_16x16"
    if name == "I_PCM":
        return "I_PCM"
    if name == "SI":
        return "Intra_4x4"
    if name in ("P_8x8", "P_8x8ref0", "B_8x8"):
        return "Na%sSliceMode" % sliceType  # the prediction mode comes from sub_mb_type
    if name in ("B_Direct_16x16", "B_Skip"):
        return "Direct"
    if name == "P_Skip":
        return "Pred_L0"
    parts = name.split("_")[1:-1]  # e.g. B_L0_Bi_16x8 -> ["L0", "Bi"]
    if partition >= len(parts):
        return "UnknownPartPredMode"
    return {"L0": "Pred_L0", "L1": "Pred_L1", "Bi": "BiPred"}[parts[partition]]


def _flag(v) -> int:
    return 1 if v else 0


def PicWidthInMbs(sps) -> int:  # h264/slice.go:159 (7-13)
    return sps.PicWidthInMbsMinus1 + 1


def PicHeightInMapUnits(sps) -> int:  # (7-16)
    return sps.PicHeightInMapUnitsMinus1 + 1


def PicSizeInMapUnits(sps) -> int:  # (7-17)
    return PicWidthInMbs(sps) * PicHeightInMapUnits(sps)


def FrameHeightInMbs(sps) -> int:  # (7-18)
    return (2 - _flag(sps.FrameMbsOnly)) * PicHeightInMapUnits(sps)


def PicHeightInMbs(sps, header) -> int:  # (7-26)
    return FrameHeightInMbs(sps) // (1 + _flag(header.FieldPic))


def PicSizeInMbs(sps, header) -> int:  # (7-29)
    return PicWidthInMbs(sps) * PicHeightInMbs(sps, header)


def SubWidthC(sps) -> int:  # Table 6-1; 17 = undefined, as in h264/slice.go:179
    if sps.UseSeparateColorPlane and sps.ChromaFormat == 3:
        return 17
    return {0: 17, 1: 2, 2: 2, 3: 1}.get(sps.ChromaFormat, 17)


def SubHeightC(sps) -> int:
    if sps.UseSeparateColorPlane and sps.ChromaFormat == 3:
        return 17
    return {0: 17, 1: 2, 2: 1, 3: 1}.get(sps.ChromaFormat, 17)


def MbWidthC(sps) -> int:  # h264/cabac.go:31 (6-1)
    return 0 if (sps.ChromaFormat == 0 or sps.UseSeparateColorPlane) else 16 // SubWidthC(sps)


def MbHeightC(sps) -> int:
    return 0 if (sps.ChromaFormat == 0 or sps.UseSeparateColorPlane) else 16 // SubHeightC(sps)


def MbaffFrameFlag(sps, header) -> int:  # h264/slice.go:563
    return 1 if (sps.MbAdaptiveFrameField and not header.FieldPic) else 0


def CodedBlockPatternLuma(data) -> int:  # h264/slice.go:222 (7-36)
    return data.CodedBlockPattern % 16


def CodedBlockPatternChroma(data) -> int:
    return data.CodedBlockPattern // 16


def Clip3(x: int, y: int, z: int) -> int:  # h264/cabac.go:131 (5-8)
    return x if z < x else (y if z > y else z)


def Clip1y(x: int, bitDepthY: int = 8) -> int:  # h264/cabac.go:123
    return Clip3(0, (1 << bitDepthY) - 1, x)


def Clipc(x: int, bitDepthC: int = 8) -> int:  # h264/cabac.go:126
    return Clip3(0, (1 << bitDepthC) - 1, x)


def SliceQPy(pps, header) -> int:  # h264/cabac.go:113 (7-30)
    return 26 + pps.PicInitQpMinus26 + header.SliceQpDelta


def PreCtxState(m: int, n: int, sliceQPy: int) -> int:  # h264/cabac.go:118 (9-5)
    return Clip3(1, 126, ((m * Clip3(0, 51, sliceQPy)) >> 4) + n)


# ---------------------------------------------------------------------------------------------------
_MBT = {0: "none", 1: "I4x4", 2: "I8x8", 3: "I16x16", 4: "IPCM", 5: "P16x16", 6: "P16x8", 7: "P8x16", 8: "P8x8", 9: "PSKIP", 10: "B", 11: "BDIRECT",
        12: "BSKIP"}  # mi_types.h MBT_*


class SliceData:
    """One macroblock as the reference's SliceData would hold it (h264/slice.go:77-102), filled from the 128-byte MbRec
    the GPU entropy kernel wrote (mi_types.h) and, for B slices, the macroblock's list-1 vectors.  Final motion vectors
    replace the reference's MvdL0 / MvdL1 (the kernel adds the prediction of 8.4.1.3 right away); residual coefficients
    stay on the device.  mb_type is the value as coded (Tables 7-11 / 7-13 / 7-14): the record carries it for inter
    macroblocks, intra ones are rebuilt from prediction mode and coded block pattern."""

    def __init__(self, rec: np.ndarray, sliceType: str, mv1=None):
        t = int(rec[0])
        self.SliceTypeName = sliceType
        self.TransformSize8x8Flag = bool(rec[1])
        self.QPY = int(rec[2])
        cbp = int(rec[5])
        self.CodedBlockPattern = cbp
        self.IntraChromaPredMode = int(rec[6])
        i16mode = int(rec[7])
        intra, inter = 1 <= t <= 4, t >= 5
        if t in (1, 2):
            raw = 0
        elif t == 3:
            raw = 1 + i16mode + 4 * (cbp >> 4) + (12 if (cbp & 15) else 0)
        elif t == 4:
            raw = 25
        elif t in (9, 12):
            raw = MB_TYPE_INFERRED  # P_Skip / B_Skip: mb_type is inferred, nothing is coded
        else:
            raw = int(rec[20])  # ipm[4]: mb_type as coded (B_Direct_16x16 = 0)
        # intra macroblocks of P / B slices: mb_type = 5 / 23 + the I-slice value (7.3.5)
        self.MbType = raw + ((5 if sliceType in ("P", "SP") else 23 if sliceType == "B" else 0) if intra else 0)
        self.MbTypeName = MbTypeName(sliceType, self.MbType) if t else "NotDecoded"
        self.MbSkipFlag = t in (9, 12)
        self.SubMbType = rec[21:25].view(np.int8).tolist() if (inter and not self.MbSkipFlag and raw == (3 if sliceType != "B" else 22)) else []
        self.Intra4x4PredMode = rec[16:32].view(np.int8).tolist() if t in (1, 2) else []
        self.RefIdxL0 = rec[32:36].view(np.int8).tolist() if inter else []
        self.MvL0 = rec[48:112].view(np.int16).reshape(16, 2).tolist() if inter else []
        # list 1 exists in B slices only: reference indices share the bytes of the intra modes, -1 = the 8x8 block does not use the list
        isb = inter and sliceType == "B"
        self.RefIdxL1 = [r if s >= 0 else -1 for r, s in zip(rec[16:20].view(np.int8).tolist(), rec[120:128].view(np.int16).tolist())] if isb else []
        self.MvL1 = np.asarray(mv1).reshape(16, 2).tolist() if (isb and mv1 is not None) else []
        self.CodedBlockFlagsLuma4x4 = int(rec[8:10].view(np.uint16)[0])


def NewSliceData(sliceContext, b=None, decoder=None, stream: int = 0, frame: int = 0):
    """h264/slice.go:570 NewSliceData(sliceContext, bitReader).  There is no CPU macroblock parser in this package (the
    macroblock layer is decoded by the HIP entropy kernel), so instead of a bit reader this takes the Decoder that
    decoded the stream and returns the picture's macroblocks as a list of SliceData, in macroblock address order.
    (The slice type is taken from the given slice: for pictures that mix slice types call it once per slice.)"""
    if decoder is None:
        raise NotImplementedError("slice_data() is decoded on the GPU: pass decoder=, stream=, frame= of a decoded batch "
                                  "(there is no CPU fallback for the macroblock layer)")
    sps = sliceContext.SPS
    n = PicSizeInMbs(sps, sliceContext.Slice.Header)
    recs = decoder.read_mbrecs(stream, frame, n)
    st = {0: "P", 1: "B", 2: "I", 3: "SP", 4: "SI"}[sliceContext.Slice.Header.SliceType % 5]
    mv1 = decoder.read_mbmv1(stream, frame, n) if st == "B" else None
    return [SliceData(recs[i], st, None if mv1 is None else mv1[i]) for i in range(n)]


def MbPred(sliceData, b=None, rbsp=None):
    """h264/slice.go:252 MbPred fills the mb_pred() fields of a SliceData from the bit stream.  Here mb_pred() / sub_mb_pred() are
    decoded by the GPU entropy kernel (k_entropy.hip decode_mb) together with the rest of the macroblock, so this returns those
    fields of an already decoded SliceData (from NewSliceData): the prediction modes of an intra macroblock, the reference indices
    and final vectors of an inter one."""
    if not isinstance(sliceData, SliceData):
        raise TypeError("MbPred takes a SliceData returned by NewSliceData(..., decoder=...): mb_pred() is decoded on the GPU")
    return {"Intra4x4PredMode": sliceData.Intra4x4PredMode, "IntraChromaPredMode": sliceData.IntraChromaPredMode, "SubMbType": sliceData.SubMbType,
            "RefIdxL0": sliceData.RefIdxL0, "RefIdxL1": sliceData.RefIdxL1, "MvL0": sliceData.MvL0, "MvL1": sliceData.MvL1}
