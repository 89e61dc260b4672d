"""Python mirror of the reference's host interface for the hot path.

* :class:`LumaQuantizer` -- ``include/luma/luma_quantizer.h:89-126``: ``setQuantizer``, array ``quantize`` /
  ``dequantize``, ``transformColorSpace``, ``getMapping``, ``getSize``.
* :class:`LumaFrameCodec` -- the hot-path halves of ``LumaEncoder::encode(LumaFrame*)``
  (``include/luma/luma_encoder.h:142-148``) and ``LumaDecoder::decode()`` (``include/luma/luma_decoder.h:143-161``):
  everything except ``run()`` (VP9 + Matroska, unchanged downstream stages).

Every frame-level method runs on the GPU through the C ABI; nothing here touches pixels on the CPU.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import capi
from .capi import Context, LumaHipError


@dataclass
class LumaEncoderParams:
    """include/luma/luma_encoder.h:59-70,110-118 (defaults identical)"""
    quantizerScale: int = 2
    ptfBitDepth: int = 11
    colorBitDepth: int = 8
    preScaling: float = 1.0
    minLum: float = 0.005
    maxLum: float = 1e4
    fps: float = 25.0
    ptf: int = capi.PTF_PQ
    colorSpace: int = capi.CS_LUV
    bitrate: int = 10000
    profile: int = 2
    keyframeInterval: int = 0
    bitDepth: int = 12
    lossLess: bool = False


@dataclass
class LumaDecoderParams:
    """include/luma/luma_decoder.h:60-69,112-120"""
    ptf: int = capi.PTF_PSI
    colorSpace: int = capi.CS_LUV
    preScaling: float = 1.0
    minLum: float = 0.005
    maxLum: float = 1e4
    ptfBitDepth: int = 11
    colorBitDepth: int = 8
    highBitDepth: bool = True
    profile: int = 2


class LumaQuantizer:
    def __init__(self, ctx: Context | None = None, device: int = -1):
        self.ctx = ctx if ctx is not None else Context(device)
        self._lut = None
        self._cfg = None

    def setQuantizer(self, ptf, bitdepth, cs, bitdepthC, maxLum=1e4, minLum=0.005, mapping_override=None):
        """src/luma_quantizer.cpp:172-212.  ``mapping_override`` = the first getSize() floats of attachment 434,
        which LumaDecoder::initialize memcpy's over the freshly built table (src/luma_decoder.cpp:121-122)."""
        lut = capi.build_lut(ptf, bitdepth, maxLum, minLum)
        if mapping_override is not None:
            ov = np.asarray(mapping_override, dtype=np.float32).ravel()
            if ov.size > lut.size:
                raise LumaHipError(capi.ERR_ARG, "mapping override longer than the table")
            lut[:ov.size] = ov
        self.ctx.set_quantizer(ptf, bitdepth, cs, bitdepthC, maxLum, minLum, lut)
        self._lut = lut
        self._cfg = (ptf, bitdepth, cs, bitdepthC, maxLum, minLum)

    def getMapping(self) -> np.ndarray:
        return self._lut

    def getSize(self) -> int:
        """m_maxVal, i.e. one less than the table length (include/luma/luma_quantizer.h:109)"""
        return self._lut.size - 1

    def getMaxLum(self):
        return self._cfg[4]

    def getMinLum(self):
        return self._cfg[5]

    def quantize(self, values, ch=0):
        return self.ctx.quantize_array(values, ch)

    def dequantize(self, values, ch=0):
        return self.ctx.dequantize_array(values, ch)

    def transformColorSpace(self, frame: np.ndarray, toCs: bool, sc: float) -> bool:
        """in place; False where the reference returns false (unknown colour space)"""
        try:
            self.ctx.transform_color_space(frame, toCs, sc)
        except LumaHipError as e:
            if e.code == capi.ERR_UNSUPPORTED:
                return False
            raise
        return True


class LumaFrameCodec:
    """The per-frame hot path either side of the (out-of-scope) VP9 stage."""

    def __init__(self, params: LumaEncoderParams | None = None, ctx: Context | None = None, device: int = -1):
        self.params = params or LumaEncoderParams()
        self.quant = LumaQuantizer(ctx, device)
        p = self.params
        # profile adjustment for the container bit depth, src/luma_encoder.cpp:68-72
        if p.profile > 1 and p.bitDepth == 8:
            p.profile -= 2
        if p.profile < 2 and p.bitDepth > 8:
            p.profile += 2
        self.quant.setQuantizer(p.ptf, p.ptfBitDepth, p.colorSpace, p.colorBitDepth, p.maxLum, p.minLum)

    def encode(self, frame: np.ndarray):
        """(3,h,w) float32 -> (planes, strides, mean_luminance).  Raises for odd sizes like the reference
        ("Invalid frame size", src/luma_encoder.cpp:118-119)."""
        p = self.params
        return self.quant.ctx.encode_frame(frame, p.preScaling, p.profile)

    def decode(self, planes, strides, w, h):
        p = self.params
        return self.quant.ctx.decode_frame(planes, strides, w, h, p.preScaling, p.profile)
