"""lumahdrv_amd -- MI355X-native Luma HDRv quantize / dequantize hot path.

The product is ``lumahdrv_amd/lib/liblumahip.so`` (hand-written HIP for gfx950 behind the C ABI of
``include/lumahip.h``) plus the C++ facade in ``include/luma``.  This package is the thin Python host
side: ``capi`` binds the C ABI with ctypes, ``quantizer`` mirrors the reference's ``LumaQuantizer`` /
``LumaEncoder::encode`` / ``LumaDecoder::decode`` interface on top of it, ``sharding`` splits a batch of
frames across the GPUs of a node (one process per GPU, LUT broadcast over RCCL).

There is no CPU fallback anywhere in this package: if the HIP library is missing or no GPU is present,
the entry points raise.
"""
from .capi import (CS_LUV, CS_RGB, CS_XYZ, CS_YCBCR, PTF_JND_HDRVDP, PTF_LINEAR, PTF_LOG, PTF_PQ, PTF_PSI, Context,
                   LumaHipError, build_library, build_lut, library_path, plane_geometry)
from .quantizer import LumaDecoderParams, LumaEncoderParams, LumaFrameCodec, LumaQuantizer

__all__ = ["Context", "LumaHipError", "build_library", "build_lut", "library_path", "plane_geometry",
           "LumaQuantizer", "LumaFrameCodec", "LumaEncoderParams", "LumaDecoderParams",
           "PTF_PSI", "PTF_PQ", "PTF_LOG", "PTF_JND_HDRVDP", "PTF_LINEAR", "CS_LUV", "CS_RGB", "CS_YCBCR", "CS_XYZ"]
