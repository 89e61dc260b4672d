"""ctypes binding of include/lumahip.h (the C ABI of liblumahip.so).  Nothing here computes pixels."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# LUMAHIP_LIB: load another build of the same ABI (A/B measurements, the -DLH_NO_FAST_DIV comparison build of the tests)
_LIB_PATH = os.environ.get("LUMAHIP_LIB") or os.path.join(HERE, "lib", "liblumahip.so")

# include/luma/luma_quantizer.h:95-96 (values are serialised in the stream metadata)
PTF_PSI, PTF_PQ, PTF_LOG, PTF_JND_HDRVDP, PTF_LINEAR = range(5)
CS_LUV, CS_RGB, CS_YCBCR, CS_XYZ = range(4)

OK, ERR_ARG, ERR_HIP, ERR_STATE, ERR_UNSUPPORTED = range(5)

SYMBOLS = [
    "lumahip_abi_version", "lumahip_device_count", "lumahip_create", "lumahip_destroy", "lumahip_last_error",
    "lumahip_set_stream", "lumahip_reset_stream", "lumahip_sync", "lumahip_tune", "lumahip_set_quantizer", "lumahip_build_lut", "lumahip_thresh_index_host", "lumahip_lin_index_host", "lumahip_ycbcr_luma_index_host", "lumahip_ycbcr_ytab_host", "lumahip_ycbcr_half_table_host", "lumahip_half_table_info", "lumahip_rb_table_info", "lumahip_quantize_value_host", "lumahip_dequantize_value_host", "lumahip_half_upload_info", "lumahip_numa_info", "lumahip_numa_pin_current_thread", "lumahip_numa_plan_host", "lumahip_quantizer_info",
    "lumahip_encode_stream_push", "lumahip_encode_stream_pop", "lumahip_encode_stream_pending",
    "lumahip_decode_stream_push", "lumahip_decode_stream_pop", "lumahip_decode_stream_pending",
    "lumahip_encode_frame_host", "lumahip_decode_frame_host", "lumahip_encode_frames_host", "lumahip_decode_frames_host", "lumahip_pack_frame_host", "lumahip_unpack_frame_host", "lumahip_transform_color_space_host",
    "lumahip_quantize_array_host", "lumahip_dequantize_array_host", "lumahip_quantize_array_device", "lumahip_dequantize_array_device",
    "lumahip_encode_frames_device", "lumahip_mean_luminance_reference_device",
    "lumahip_decode_frames_device", "lumahip_decode_frames_device_rotating", "lumahip_decode_display_frames_device", "lumahip_transform_color_space_device", "lumahip_synth_frames_device",
    "lumahip_device", "lumahip_encode_frames_device_planar", "lumahip_decode_frames_device_planar", "lumahip_begin_unordered", "lumahip_end_unordered",
    "lumahip_probe_decode_traffic_device",
    "lumahip_decoded_ring_create", "lumahip_decoded_ring_destroy", "lumahip_decoded_ring_info", "lumahip_decoded_ring_frame", "lumahip_decode_frames_device_ring",
    "lumahip_pool_create", "lumahip_pool_create_small", "lumahip_pool_destroy", "lumahip_pool_alloc", "lumahip_pool_release", "lumahip_pool_available", "lumahip_pool_group_of",
    "lumahip_pool_stats_json", "lumahip_pool_find_groups",
    "lumahip_multi_create", "lumahip_multi_destroy", "lumahip_multi_shards", "lumahip_multi_ctx", "lumahip_multi_last_error", "lumahip_multi_used_rccl", "lumahip_multi_set_transport", "lumahip_multi_transport_note",
    "lumahip_shard_range", "lumahip_multi_set_quantizer", "lumahip_multi_encode_frames_host", "lumahip_multi_decode_frames_host",
    "lumahip_multi_encode_frames_device", "lumahip_multi_decode_frames_device", "lumahip_multi_sync",
    "lumahip_time_launches", "lumahip_probe_encode_traffic_device", "lumahip_powf_probe_device", "lumahip_quantize_probe_device", "lumahip_ycbcr_luma_probe_device", "lumahip_host_register", "lumahip_host_unregister", "lumahip_malloc", "lumahip_free", "lumahip_memcpy_h2d", "lumahip_memcpy_d2h",
]


PROBE_FN = C.CFUNCTYPE(C.c_double, C.c_int, C.c_int, C.c_void_p)


class LumaHipError(RuntimeError):
    """Counterpart of the reference's LumaException (include/luma/luma_exception.h:53-71)."""

    def __init__(self, code, msg):
        super().__init__("lumahip error %d: %s" % (code, msg))
        self.code = code


def library_path() -> str:
    return _LIB_PATH


def build_library(force: bool = False, nofastdiv: bool = False) -> str:
    """Compile every HIP source for gfx950 into lumahdrv_amd/lib/liblumahip.so (hipcc cross-compiles
    without a GPU).  nofastdiv: additionally the -DLH_NO_FAST_DIV comparison build of the C ABI library
    (lumahdrv_amd/lib_nofastdiv/liblumahip.so) that tests/test_gpu_parity.py loads through LUMAHIP_LIB."""
    cmd = ["make", "-s", "-j", str(min(8, os.cpu_count() or 1)), "-C", os.path.join(HERE, "csrc")]
    if force:
        subprocess.run(cmd + ["clean"], check=True)
    subprocess.run(cmd, check=True)
    if nofastdiv:
        out = os.path.join(HERE, "lib_nofastdiv")
        subprocess.run(cmd + ["OUT=" + out, "EXTRA=-DLH_NO_FAST_DIV", os.path.join(out, "liblumahip.so")], check=True)
    return os.path.join(HERE, "lib", "liblumahip.so")


# device code + launch geometry + compiler flags (NOT the host plumbing: lumahip_core / _host / _pool / _multi, lumahip_internal.hpp)
KERNEL_SOURCES = ("luma_device.hpp", "luma_kernels.hpp", "pow_glibc.hpp", "lumahip_launch.hip", "lumahip_encode.hip",
                  "lumahip_decode.hip", "lumahip_misc.hip", "lut_index.cpp", "lut_index.hpp", "flags.mk")


def kernel_source_sha() -> str:
    """short SHA-1 over the device sources; profiles/*.json captured by rocprofv3 carry it so that bench.py never
    reports a counter-derived figure measured on different kernels"""
    import hashlib
    hsh = hashlib.sha1()
    for f in KERNEL_SOURCES:
        with open(os.path.join(HERE, "csrc", f), "rb") as fh:
            hsh.update(fh.read())
    return hsh.hexdigest()[:12]


_lib = None


def lib():
    """Load liblumahip.so.  torch, when installed, is imported FIRST: its wheel bundles its own
    libamdhip64.so (same SONAME as /opt/rocm's), and the process must end up with exactly one HIP
    runtime so that device pointers from torch tensors are valid in our launches."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise LumaHipError(ERR_STATE, "%s not built; run `python -c 'import __graft_entry__ as g; g.build()'` "
                                      "(there is no CPU fallback)" % _LIB_PATH)
    if "torch" not in sys.modules and not os.environ.get("LUMAHIP_NO_TORCH"):
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = C.CDLL(_LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, u, i, f, sz = C.c_void_p, C.c_uint, C.c_int, C.c_float, C.c_size_t
    pp3 = C.POINTER(C.c_void_p)
    ip3 = C.POINTER(C.c_int)
    sp3 = C.POINTER(C.c_size_t)
    L.lumahip_abi_version.restype = i
    L.lumahip_device_count.argtypes = [C.POINTER(i)]
    L.lumahip_create.argtypes = [C.POINTER(vp), i]
    L.lumahip_destroy.argtypes = [vp]
    L.lumahip_destroy.restype = None
    L.lumahip_last_error.argtypes = [vp]
    L.lumahip_last_error.restype = C.c_char_p
    L.lumahip_set_stream.argtypes = [vp, vp]
    L.lumahip_reset_stream.argtypes = [vp]
    L.lumahip_sync.argtypes = [vp]
    L.lumahip_tune.argtypes = [vp, C.c_char_p, C.c_long]
    L.lumahip_set_quantizer.argtypes = [vp, i, u, i, u, f, f, vp, sz]
    L.lumahip_build_lut.argtypes = [i, u, f, f, vp, sz]
    L.lumahip_thresh_index_host.argtypes = [vp, sz, C.POINTER(i), vp, sz]
    L.lumahip_lin_index_host.argtypes = [vp, sz, C.POINTER(i), vp, sz]
    L.lumahip_quantizer_info.argtypes = [vp, C.POINTER(i)]
    L.lumahip_ycbcr_luma_index_host.argtypes = [vp, sz, f, C.POINTER(i), vp, sz]
    L.lumahip_ycbcr_ytab_host.argtypes = [vp, sz, f, vp]
    L.lumahip_ycbcr_half_table_host.argtypes = [f, f, vp, sz]
    L.lumahip_half_table_info.argtypes = [vp, f, C.POINTER(i)]
    L.lumahip_rb_table_info.argtypes = [vp, f, C.POINTER(i)]
    L.lumahip_half_upload_info.argtypes = [vp, C.POINTER(C.c_long)]
    L.lumahip_numa_info.argtypes = [vp, C.POINTER(i)]
    L.lumahip_numa_pin_current_thread.argtypes = [vp]
    L.lumahip_numa_plan_host.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(i), C.POINTER(i), i, C.POINTER(i)]
    L.lumahip_quantize_value_host.argtypes = [vp, sz, i, u, f, u, C.POINTER(f)]
    L.lumahip_dequantize_value_host.argtypes = [vp, sz, i, u, f, u, C.POINTER(f)]
    L.lumahip_encode_frame_host.argtypes = [vp, vp, u, u, f, i, pp3, ip3, C.POINTER(f), vp]
    L.lumahip_decode_frame_host.argtypes = [vp, pp3, ip3, u, u, i, f, vp]
    L.lumahip_encode_frames_host.argtypes = [vp, pp3, u, u, u, f, i, pp3, ip3, C.POINTER(f)]
    L.lumahip_decode_frames_host.argtypes = [vp, pp3, ip3, u, u, u, i, f, pp3]
    L.lumahip_pack_frame_host.argtypes = [vp, vp, u, u, i, pp3, ip3, C.POINTER(f)]
    L.lumahip_unpack_frame_host.argtypes = [vp, pp3, ip3, u, u, i, vp]
    L.lumahip_transform_color_space_host.argtypes = [vp, vp, u, u, i, f]
    L.lumahip_quantize_array_host.argtypes = [vp, vp, vp, sz, u]
    L.lumahip_dequantize_array_host.argtypes = [vp, vp, vp, sz, u]
    L.lumahip_quantize_array_device.argtypes = [vp, vp, vp, sz, u]
    L.lumahip_dequantize_array_device.argtypes = [vp, vp, vp, sz, u]
    L.lumahip_encode_frames_device.argtypes = [vp, vp, sz, u, u, u, f, i, pp3, ip3, sp3, vp]
    L.lumahip_mean_luminance_reference_device.argtypes = [vp, vp, u, u, f, C.POINTER(f)]
    L.lumahip_decode_frames_device.argtypes = [vp, pp3, ip3, sp3, u, u, u, i, f, vp, sz]
    L.lumahip_decode_display_frames_device.argtypes = [vp, pp3, ip3, sp3, u, u, u, i, f, vp, sz, vp, i, sz, f, f, i, i]
    L.lumahip_transform_color_space_device.argtypes = [vp, vp, sz, u, u, u, i, f]
    L.lumahip_synth_frames_device.argtypes = [vp, vp, sz, u, u, u, C.c_uint64, C.c_uint64]
    L.lumahip_time_launches.argtypes = [vp, i, i, vp, sz, u, u, u, f, i, pp3, ip3, sp3, C.POINTER(f)]
    L.lumahip_probe_encode_traffic_device.argtypes = [vp, vp, sz, u, u, u, pp3, ip3, sp3, i, C.POINTER(f)]
    L.lumahip_powf_probe_device.argtypes = [vp, vp, C.c_uint32, sz, f, i]
    L.lumahip_quantize_probe_device.argtypes = [vp, vp, C.c_uint32, sz, i]
    L.lumahip_ycbcr_luma_probe_device.argtypes = [vp, vp, C.c_uint32, sz, i]
    L.lumahip_host_register.argtypes = [vp, vp, sz]
    L.lumahip_host_unregister.argtypes = [vp, vp]
    L.lumahip_malloc.argtypes = [vp, C.POINTER(vp), sz]
    L.lumahip_free.argtypes = [vp, vp]
    L.lumahip_memcpy_h2d.argtypes = [vp, vp, vp, sz]
    L.lumahip_memcpy_d2h.argtypes = [vp, vp, vp, sz]
    L.lumahip_device.argtypes = [vp]
    L.lumahip_encode_frames_device_planar.argtypes = [vp, pp3, sz, u, u, u, f, i, pp3, ip3, sp3, vp]
    L.lumahip_decode_frames_device_planar.argtypes = [vp, pp3, ip3, sp3, u, u, u, i, f, pp3, sz]
    L.lumahip_decode_frames_device_rotating.argtypes = [vp, pp3, ip3, sp3, u, u, u, i, f, pp3, sz]
    L.lumahip_begin_unordered.argtypes = [vp, i]
    L.lumahip_end_unordered.argtypes = [vp]
    L.lumahip_probe_decode_traffic_device.argtypes = [vp, pp3, ip3, sp3, u, u, u, pp3, sz, i, C.POINTER(f)]
    L.lumahip_pool_create.argtypes = [vp, C.POINTER(PoolConfig), C.POINTER(vp)]
    L.lumahip_pool_create_small.argtypes = [vp, i, i, i, i, C.POINTER(vp)]
    L.lumahip_decoded_ring_create.argtypes = [vp, u, u, u, u, C.POINTER(vp)]
    L.lumahip_decoded_ring_destroy.argtypes = [vp]
    L.lumahip_decoded_ring_destroy.restype = None
    L.lumahip_decoded_ring_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_size_t)]
    L.lumahip_decoded_ring_frame.argtypes = [vp, u, u]
    L.lumahip_decoded_ring_frame.restype = vp
    L.lumahip_decode_frames_device_ring.argtypes = [vp, pp3, ip3, sp3, u, i, f, vp, u]
    L.lumahip_pool_destroy.argtypes = [vp]
    L.lumahip_pool_destroy.restype = None
    L.lumahip_pool_alloc.argtypes = [vp, i, i, C.POINTER(vp)]
    L.lumahip_pool_release.argtypes = [vp, vp]
    L.lumahip_pool_available.argtypes = [vp, i, i]
    L.lumahip_pool_group_of.argtypes = [vp, vp]
    L.lumahip_pool_stats_json.argtypes = [vp]
    L.lumahip_pool_stats_json.restype = C.c_char_p
    L.lumahip_pool_find_groups.argtypes = [i, PROBE_FN, vp, vp, C.POINTER(i), C.POINTER(C.c_double), C.POINTER(i)]
    L.lumahip_multi_create.argtypes = [C.POINTER(vp), ip3, i]
    L.lumahip_multi_destroy.argtypes = [vp]
    L.lumahip_multi_destroy.restype = None
    L.lumahip_multi_shards.argtypes = [vp]
    L.lumahip_multi_ctx.argtypes = [vp, i]
    L.lumahip_multi_ctx.restype = vp
    L.lumahip_multi_last_error.argtypes = [vp]
    L.lumahip_multi_last_error.restype = C.c_char_p
    L.lumahip_multi_used_rccl.argtypes = [vp]
    L.lumahip_multi_set_transport.argtypes = [vp, i]
    L.lumahip_multi_transport_note.argtypes = [vp]
    L.lumahip_multi_transport_note.restype = C.c_char_p
    L.lumahip_shard_range.argtypes = [u, i, i, C.POINTER(u), C.POINTER(u)]
    L.lumahip_multi_set_quantizer.argtypes = [vp, i, u, i, u, f, f, vp, sz]
    L.lumahip_multi_encode_frames_host.argtypes = [vp, pp3, u, u, u, f, i, pp3, ip3, C.POINTER(f)]
    L.lumahip_multi_decode_frames_host.argtypes = [vp, pp3, ip3, u, u, u, i, f, pp3]
    L.lumahip_multi_encode_frames_device.argtypes = [vp, pp3, sz, C.POINTER(u), u, u, f, i, pp3, ip3, sp3]
    L.lumahip_multi_decode_frames_device.argtypes = [vp, pp3, ip3, sp3, C.POINTER(u), u, u, i, f, pp3, sz]
    L.lumahip_multi_sync.argtypes = [vp]
    _lib = L
    return L


class PoolConfig(C.Structure):
    """lumahip_pool_config (include/lumahip.h)"""
    _fields_ = [("chunk_bytes", C.c_size_t), ("n_float", C.c_int), ("n_y", C.c_int), ("n_uv", C.c_int), ("n_striped", C.c_int),
                ("keep_free_bytes", C.c_size_t), ("max_chunks", C.c_int), ("probe_iters", C.c_int)]


POOL_FLOAT, POOL_Y, POOL_UV, POOL_STRIPED, POOL_ROTATING = range(5)


def shard_range(nframes: int, shard: int, nshards: int) -> range:
    """lumahip_shard_range: the contiguous block of frames shard `shard` of `nshards` owns"""
    a, b = C.c_uint(0), C.c_uint(0)
    rc = lib().lumahip_shard_range(nframes, shard, nshards, C.byref(a), C.byref(b))
    if rc != OK:
        raise LumaHipError(rc, "lumahip_shard_range(%d, %d, %d)" % (nframes, shard, nshards))
    return range(a.value, a.value + b.value)


def pool_find_groups(n: int, probe):
    """host-only: the chunk pool's grouping step (lumahip_pool_find_groups) with a Python measurement probe(i, r) -> time of
    reading chunk i while writing chunk r.  Returns (groups or None when there is no contrast, fastest pair time, probes)."""
    g = np.full(n, -1, dtype=np.int32)
    ng, npr, fast = C.c_int(0), C.c_int(0), C.c_double(0.0)
    cb = PROBE_FN(lambda i, r, _u: float(probe(i, r)))
    rc = lib().lumahip_pool_find_groups(n, cb, None, g.ctypes.data, C.byref(ng), C.byref(fast), C.byref(npr))
    if rc != OK:
        raise LumaHipError(rc, "lumahip_pool_find_groups failed")
    groups = None if ng.value == 0 else [[int(i) for i in np.nonzero(g == k)[0]] for k in range(ng.value)]
    return groups, fast.value, npr.value


def plane_geometry(w: int, h: int, profile: int, align: int = 32):
    """(widths, heights, strides-in-bytes, bytes-per-sample) of the Y/U/V planes as
    vpx_img_alloc(fmt(profile), w, h, 32) lays them out (src/luma_encoder.cpp:121-128)."""
    sub = profile in (0, 2)
    bps = 2 if profile > 1 else 1
    s0 = (w + align - 1) // align * align * bps
    cw, ch = ((w + 1) // 2, (h + 1) // 2) if sub else (w, h)
    s1 = s0 // 2 if sub else s0
    return (w, cw, cw), (h, ch, ch), (s0, s1, s1), bps


def build_lut(ptf: int, bitdepth: int, max_lum: float = 1e4, min_lum: float = 0.005) -> np.ndarray:
    """The transfer-function table of LumaQuantizer::setQuantizer, built by the library's host code with the
    host libm exactly as the reference does (no GPU needed)."""
    out = np.empty(1 << bitdepth, dtype=np.float32)
    rc = lib().lumahip_build_lut(ptf, bitdepth, max_lum, min_lum, out.ctypes.data, out.size)
    if rc != OK:
        raise LumaHipError(rc, "lumahip_build_lut(ptf=%d, bitdepth=%d) failed" % (ptf, bitdepth))
    return out


def thresh_index(lut: np.ndarray):
    """host-only: the threshold records of a table (include/lumahip.h lumahip_thresh_index_host); no GPU needed"""
    lut = np.ascontiguousarray(lut, dtype=np.float32)
    info = (C.c_int * 5)()
    rc = lib().lumahip_thresh_index_host(lut.ctypes.data, lut.size, info, None, 0)
    if rc != OK:
        raise LumaHipError(rc, "lumahip_thresh_index_host failed")
    d = dict(ok=bool(info[0]), mant_bits=info[1], shift=info[2], kmin=info[3], nbuckets=info[4], rec=None)
    if d["ok"]:
        rec = np.zeros(info[4], dtype=np.uint32)
        rc = lib().lumahip_thresh_index_host(lut.ctypes.data, lut.size, info, rec.ctypes.data, rec.size)
        if rc != OK:
            raise LumaHipError(rc, "lumahip_thresh_index_host failed")
        d["rec"] = rec
    return d


def lin_index(lut: np.ndarray):
    """host-only: the value-keyed records of an evenly spaced table (include/lumahip.h lumahip_lin_index_host); no GPU needed.
    dict(ok, nbuckets, kscale (np.float32), rec (nbuckets x 2 uint32: bits(T) - 1, start) | None)"""
    lut = np.ascontiguousarray(lut, dtype=np.float32)
    info = (C.c_int * 3)()
    rc = lib().lumahip_lin_index_host(lut.ctypes.data, lut.size, info, None, 0)
    if rc != OK:
        raise LumaHipError(rc, "lumahip_lin_index_host failed")
    d = dict(ok=bool(info[0]), nbuckets=info[1], kscale=np.array([info[2]], dtype=np.int32).view(np.float32)[0], rec=None)
    if d["ok"]:
        rec = np.zeros((info[1], 2), dtype=np.uint32)
        rc = lib().lumahip_lin_index_host(lut.ctypes.data, lut.size, info, rec.ctypes.data, rec.size)
        if rc != OK:
            raise LumaHipError(rc, "lumahip_lin_index_host failed")
        d["rec"] = rec
    return d


def lin_lookup(ix, v: np.ndarray) -> np.ndarray:
    """numpy evaluation of value-keyed records, as the kernels evaluate them (luma_device.hpp quantize_linkey)"""
    v = np.ascontiguousarray(v, dtype=np.float32)
    with np.errstate(all="ignore"):
        p = np.fmin(v * np.float32(ix["kscale"]), np.float32(ix["nbuckets"] - 1))      # fmin: a NaN product -> the top bucket
        k = np.where(p > 0, p, np.float32(0)).astype(np.int64)                            # truncation; negatives and -0 -> 0
    rec = ix["rec"]
    return rec[k, 1].astype(np.int64) + (v.view(np.int32) > rec[k, 0].view(np.int32)).astype(np.int64)


def ycbcr_luma_index(lut: np.ndarray, max_lum: float):
    """host-only: the composite luma -> luminance code records of the YCbCr encode kernels (same dict as thresh_index)"""
    lut = np.ascontiguousarray(lut, dtype=np.float32)
    info = (C.c_int * 5)()
    rc = lib().lumahip_ycbcr_luma_index_host(lut.ctypes.data, lut.size, max_lum, info, None, 0)
    if rc != OK:
        raise LumaHipError(rc, "lumahip_ycbcr_luma_index_host failed")
    d = dict(ok=bool(info[0]), mant_bits=info[1], shift=info[2], kmin=info[3], nbuckets=info[4], rec=None)
    if d["ok"]:
        rec = np.zeros(info[4], dtype=np.uint32)
        rc = lib().lumahip_ycbcr_luma_index_host(lut.ctypes.data, lut.size, max_lum, info, rec.ctypes.data, rec.size)
        if rc != OK:
            raise LumaHipError(rc, "lumahip_ycbcr_luma_index_host failed")
        d["rec"] = rec
    return d


def ycbcr_ytab(lut: np.ndarray, max_lum: float) -> np.ndarray:
    """host-only: the per-stream y table of the YCbCr decode kernels"""
    lut = np.ascontiguousarray(lut, dtype=np.float32)
    out = np.empty_like(lut)
    rc = lib().lumahip_ycbcr_ytab_host(lut.ctypes.data, lut.size, max_lum, out.ctypes.data)
    if rc != OK:
        raise LumaHipError(rc, "lumahip_ycbcr_ytab_host failed")
    return out


def quantize_value(lut: np.ndarray, cs: int, bitdepth_c: int, val: float, ch: int = 0, dequantize: bool = False) -> float:
    """host-only: LumaQuantizer::quantize / dequantize for one value (include/lumahip.h lumahip_quantize_value_host)"""
    lut = np.ascontiguousarray(lut, dtype=np.float32)
    out = C.c_float(0)
    fn = lib().lumahip_dequantize_value_host if dequantize else lib().lumahip_quantize_value_host
    rc = fn(lut.ctypes.data, lut.size, cs, bitdepth_c, val, ch, C.byref(out))
    if rc != OK:
        raise LumaHipError(rc, "lumahip_(de)quantize_value_host: bad argument")
    return float(out.value)


def numa_plan(sysfs_root, pci_bus_id: str, allowed: str = ""):
    """host-only: (node, [cpus]) the library would place the host side of a context on for the GPU at `pci_bus_id`"""
    node, n = C.c_int(-1), C.c_int(0)
    cpus = (C.c_int * 4096)()
    rc = lib().lumahip_numa_plan_host(sysfs_root.encode() if sysfs_root else None, pci_bus_id.encode(), allowed.encode(), C.byref(node), cpus,
                                      4096, C.byref(n))
    if rc != OK:
        raise LumaHipError(rc, "lumahip_numa_plan_host: bad argument")
    return node.value, list(cpus[:min(n.value, 4096)])


HALF_TABLE_LEN = 0x7C00 + 1


def ycbcr_half_table(sc: float, max_lum: float):
    """host-only: the half-input table of the YCbCr encode kernels for (preScaling, maxLum), or None when the pair does not qualify"""
    out = np.empty(HALF_TABLE_LEN, dtype=np.float32)
    rc = lib().lumahip_ycbcr_half_table_host(sc, max_lum, out.ctypes.data, out.size)
    if rc == ERR_UNSUPPORTED:
        return None
    if rc != OK:
        raise LumaHipError(rc, "lumahip_ycbcr_half_table_host failed")
    return out


def thresh_lookup(ix, v: np.ndarray) -> np.ndarray:
    """numpy evaluation of the record table exactly as the kernels do it (sign-set NaNs excluded by the caller)"""
    b = np.ascontiguousarray(v, dtype=np.float32).view(np.int32)
    k = np.clip(b >> ix["shift"], ix["kmin"], ix["kmin"] + ix["nbuckets"] - 1) - ix["kmin"]
    low = b.view(np.uint32) & np.uint32((1 << ix["shift"]) - 1)
    return ((ix["rec"][k] + low) >> np.uint32(ix["shift"])).astype(np.int64)


def _arr3(ctype, vals):
    return (ctype * 3)(*vals)


class Context:
    """One GPU, one stream, one quantizer configuration (lumahip_ctx)."""

    def __init__(self, device: int = -1):
        self.L = lib()
        h = C.c_void_p()
        rc = self.L.lumahip_create(C.byref(h), device)
        if rc != OK:
            raise LumaHipError(rc, "lumahip_create failed: no usable HIP device (there is no CPU fallback)")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.lumahip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != OK:
            raise LumaHipError(rc, self.L.lumahip_last_error(self.h).decode())

    # ---- configuration
    def set_stream(self, hip_stream: int | None):
        """hip_stream: a hipStream_t handle (0 = the default stream, as torch's default stream reports);
        None = back to the context's own stream"""
        if hip_stream is None:
            self._chk(self.L.lumahip_reset_stream(self.h))
        else:
            self._chk(self.L.lumahip_set_stream(self.h, C.c_void_p(hip_stream)))

    def sync(self):
        self._chk(self.L.lumahip_sync(self.h))

    def tune(self, key: str, value: int):
        """measurement override (include/lumahip.h lumahip_tune); never changes a result"""
        self._chk(self.L.lumahip_tune(self.h, key.encode(), int(value)))

    def set_quantizer(self, ptf, bitdepth, cs, bitdepthC, max_lum, min_lum, lut: np.ndarray):
        lut = np.ascontiguousarray(lut, dtype=np.float32)
        self._chk(self.L.lumahip_set_quantizer(self.h, ptf, bitdepth, cs, bitdepthC, max_lum, min_lum,
                                               lut.ctypes.data, lut.size))

    def quantizer_info(self):
        a = (C.c_int * 5)()
        self._chk(self.L.lumahip_quantizer_info(self.h, a))
        return dict(mode=a[0], mant_bits=a[1], buckets=a[2], shift=a[3], lds_bytes=a[4])

    def half_upload_info(self):
        a = (C.c_long * 3)()
        self._chk(self.L.lumahip_half_upload_info(self.h, a))
        return dict(half_frames=a[0], float_fallbacks=a[1], pause_left=a[2])

    def numa_info(self):
        a = (C.c_int * 3)()
        self._chk(self.L.lumahip_numa_info(self.h, a))
        return dict(node=a[0], cpus=a[1], first_cpu=a[2])

    def rb_table_info(self, sc: float):
        a = (C.c_int * 4)()
        self._chk(self.L.lumahip_rb_table_info(self.h, sc, a))
        return dict(used=bool(a[0]), bytes=a[1], table_launches=a[2], backoff_launches=a[3])

    def half_table_info(self, sc: float):
        a = (C.c_int * 6)()
        self._chk(self.L.lumahip_half_table_info(self.h, sc, a))
        return dict(used=bool(a[0]), lds_bytes=a[1], device_copies=a[2], entries=a[3], table_launches=a[4], backoff_launches=a[5])

    # ---- host entry points (numpy)
    def encode_frame(self, rgb: np.ndarray, sc=1.0, profile=2, align=32, want_transformed=False, strides=None):
        """rgb: (3,h,w) float32 (LumaFrame layout).  Returns (planes, strides, mean_lum[, transformed])."""
        rgb = np.ascontiguousarray(rgb, dtype=np.float32)
        _, h, w = rgb.shape
        _, hs, st, _ = plane_geometry(w, h, profile, align)
        if strides is not None:
            st = tuple(strides)
        planes = [np.zeros((hs[p], st[p]), dtype=np.uint8) for p in range(3)]
        mean = C.c_float(0)
        tr = np.empty_like(rgb) if want_transformed else None
        self._chk(self.L.lumahip_encode_frame_host(self.h, rgb.ctypes.data, w, h, sc, profile,
                                                   _arr3(C.c_void_p, [p.ctypes.data for p in planes]),
                                                   _arr3(C.c_int, st), C.byref(mean),
                                                   tr.ctypes.data if tr is not None else None))
        if want_transformed:
            return planes, st, float(mean.value), tr
        return planes, st, float(mean.value)

    def decode_frame(self, planes, strides, w, h, sc=1.0, profile=2) -> np.ndarray:
        planes = [np.ascontiguousarray(p) for p in planes]
        out = np.empty((3, h, w), dtype=np.float32)
        self._chk(self.L.lumahip_decode_frame_host(self.h, _arr3(C.c_void_p, [p.ctypes.data for p in planes]),
                                                   _arr3(C.c_int, strides), w, h, profile, sc, out.ctypes.data))
        return out

    def encode_frames(self, frames, sc=1.0, profile=2, align=32):
        """pipelined batch form of encode_frame: frames = list of (3,h,w) float32 arrays.  Returns
        (list of plane triplets, strides, list of mean luminances)."""
        frames = [np.ascontiguousarray(f, dtype=np.float32) for f in frames]
        n = len(frames)
        _, h, w = frames[0].shape
        _, hs, st, _ = plane_geometry(w, h, profile, align)
        planes = [[np.zeros((hs[p], st[p]), dtype=np.uint8) for p in range(3)] for _ in range(n)]
        fp = (C.c_void_p * n)(*[f.ctypes.data for f in frames])
        pp = (C.c_void_p * (3 * n))(*[pl.ctypes.data for tri in planes for pl in tri])
        means = (C.c_float * n)()
        self._chk(self.L.lumahip_encode_frames_host(self.h, fp, n, w, h, sc, profile, pp, _arr3(C.c_int, st), means))
        return planes, st, [float(m) for m in means]

    def decode_frames(self, planes_list, strides, w, h, sc=1.0, profile=2):
        n = len(planes_list)
        planes_list = [[np.ascontiguousarray(p) for p in tri] for tri in planes_list]
        outs = [np.empty((3, h, w), dtype=np.float32) for _ in range(n)]
        pp = (C.c_void_p * (3 * n))(*[pl.ctypes.data for tri in planes_list for pl in tri])
        op = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        self._chk(self.L.lumahip_decode_frames_host(self.h, pp, _arr3(C.c_int, strides), n, w, h, profile, sc, op))
        return outs

    def pack_frame(self, transformed: np.ndarray, profile=2, align=32):
        """LumaEncoder::setChannels on its own: quantize + pack an already colour-transformed frame"""
        t = np.ascontiguousarray(transformed, dtype=np.float32)
        _, h, w = t.shape
        _, hs, st, _ = plane_geometry(w, h, profile, align)
        planes = [np.zeros((hs[p], st[p]), dtype=np.uint8) for p in range(3)]
        mean = C.c_float(0)
        self._chk(self.L.lumahip_pack_frame_host(self.h, t.ctypes.data, w, h, profile,
                                                 _arr3(C.c_void_p, [p.ctypes.data for p in planes]),
                                                 _arr3(C.c_int, st), C.byref(mean)))
        return planes, st, float(mean.value)

    def unpack_frame(self, planes, strides, w, h, profile=2) -> np.ndarray:
        """LumaDecoder::getVpxChannels on its own: unpack + dequantize, no inverse colour transform"""
        planes = [np.ascontiguousarray(p) for p in planes]
        out = np.empty((3, h, w), dtype=np.float32)
        self._chk(self.L.lumahip_unpack_frame_host(self.h, _arr3(C.c_void_p, [p.ctypes.data for p in planes]),
                                                   _arr3(C.c_int, strides), w, h, profile, out.ctypes.data))
        return out

    def transform_color_space(self, frame: np.ndarray, to_cs: bool, sc=1.0) -> np.ndarray:
        assert frame.dtype == np.float32 and frame.flags.c_contiguous and frame.shape[0] == 3
        self._chk(self.L.lumahip_transform_color_space_host(self.h, frame.ctypes.data, frame.shape[2], frame.shape[1],
                                                            int(bool(to_cs)), sc))
        return frame

    def quantize_array(self, a, ch=0) -> np.ndarray:
        a = np.ascontiguousarray(a, dtype=np.float32)
        out = np.empty_like(a)
        self._chk(self.L.lumahip_quantize_array_host(self.h, a.ctypes.data, out.ctypes.data, a.size, ch))
        return out

    def dequantize_array(self, a, ch=0) -> np.ndarray:
        a = np.ascontiguousarray(a, dtype=np.float32)
        out = np.empty_like(a)
        self._chk(self.L.lumahip_dequantize_array_host(self.h, a.ctypes.data, out.ctypes.data, a.size, ch))
        return out

    # ---- device entry points (raw device pointers, e.g. torch.Tensor.data_ptr())
    def encode_frames_device(self, rgb_ptr, frame_stride, nframes, w, h, sc, profile, plane_ptrs, strides,
                             plane_frame_strides, stats_ptr=None):
        self._chk(self.L.lumahip_encode_frames_device(self.h, rgb_ptr, frame_stride, nframes, w, h, sc, profile,
                                                      _arr3(C.c_void_p, plane_ptrs), _arr3(C.c_int, strides),
                                                      _arr3(C.c_size_t, plane_frame_strides), stats_ptr))

    def mean_luminance_reference_device(self, rgb_ptr, w, h, sc=1.0) -> float:
        """the reference's sequentially-summed mean of transformed channel 0 (exact; ~25 ms at 4K)"""
        m = C.c_float(0)
        self._chk(self.L.lumahip_mean_luminance_reference_device(self.h, rgb_ptr, w, h, sc, C.byref(m)))
        return float(m.value)

    def decode_frames_device(self, plane_ptrs, strides, plane_frame_strides, nframes, w, h, profile, sc, rgb_ptr,
                             frame_stride):
        self._chk(self.L.lumahip_decode_frames_device(self.h, _arr3(C.c_void_p, plane_ptrs), _arr3(C.c_int, strides),
                                                      _arr3(C.c_size_t, plane_frame_strides), nframes, w, h, profile,
                                                      sc, rgb_ptr, frame_stride))

    def decode_display_frames_device(self, plane_ptrs, strides, plane_frame_strides, nframes, w, h, profile, sc, rgba_ptr,
                                     rgba_stride, rgba_frame_stride, exposure=1.0, gamma=2.2, do_tmo=False,
                                     ldr_sim=False, rgb_ptr=None, frame_stride=0):
        """decode fused with the player's display transform -> RGBA8"""
        self._chk(self.L.lumahip_decode_display_frames_device(
            self.h, _arr3(C.c_void_p, plane_ptrs), _arr3(C.c_int, strides), _arr3(C.c_size_t, plane_frame_strides),
            nframes, w, h, profile, sc, rgb_ptr, frame_stride, rgba_ptr, rgba_stride, rgba_frame_stride, exposure, gamma,
            int(bool(do_tmo)), int(bool(ldr_sim))))

    def quantize_array_device(self, in_ptr, out_ptr, n, ch=0):
        self._chk(self.L.lumahip_quantize_array_device(self.h, in_ptr, out_ptr, n, ch))

    def dequantize_array_device(self, in_ptr, out_ptr, n, ch=0):
        self._chk(self.L.lumahip_dequantize_array_device(self.h, in_ptr, out_ptr, n, ch))

    def transform_frames_device(self, ptr, frame_stride, nframes, w, h, to_cs, sc):
        self._chk(self.L.lumahip_transform_color_space_device(self.h, ptr, frame_stride, nframes, w, h,
                                                              int(bool(to_cs)), sc))

    def synth_frames_device(self, ptr, frame_stride, nframes, w, h, seed=20250929, first_frame=0):
        self._chk(self.L.lumahip_synth_frames_device(self.h, ptr, frame_stride, nframes, w, h, seed, first_frame))

    def time_launches(self, direction, iters, rgb_ptr, frame_stride, nframes, w, h, sc, profile, plane_ptrs, strides,
                      plane_frame_strides) -> float:
        """average milliseconds per launch, measured with hipEvents on the context's stream"""
        ms = C.c_float(0)
        self._chk(self.L.lumahip_time_launches(self.h, direction, iters, rgb_ptr, frame_stride, nframes, w, h, sc,
                                               profile, _arr3(C.c_void_p, plane_ptrs), _arr3(C.c_int, strides),
                                               _arr3(C.c_size_t, plane_frame_strides), C.byref(ms)))
        return float(ms.value)

    def device(self) -> int:
        return int(self.L.lumahip_device(self.h))

    def encode_frames_device_planar(self, rgb_plane_ptrs, frame_stride, nframes, w, h, sc, profile, plane_ptrs, strides,
                                    plane_frame_strides, stats_ptr=None):
        """float frames as three colour-plane base pointers (plane c of frame f at rgb_plane_ptrs[c] + f*frame_stride floats)"""
        self._chk(self.L.lumahip_encode_frames_device_planar(self.h, _arr3(C.c_void_p, rgb_plane_ptrs), frame_stride, nframes, w, h,
                                                             sc, profile, _arr3(C.c_void_p, plane_ptrs), _arr3(C.c_int, strides),
                                                             _arr3(C.c_size_t, plane_frame_strides), stats_ptr))

    def decode_frames_device_rotating(self, plane_ptrs, strides, plane_frame_strides, nframes, w, h, profile, sc, base_ptrs, frame_stride):
        """packed frames over three buffers: frame f at base_ptrs[f % 3] + (f // 3) * frame_stride floats"""
        self._chk(self.L.lumahip_decode_frames_device_rotating(self.h, _arr3(C.c_void_p, plane_ptrs), _arr3(C.c_int, strides),
                                                               _arr3(C.c_size_t, plane_frame_strides), nframes, w, h, profile, sc,
                                                               _arr3(C.c_void_p, base_ptrs), frame_stride))

    def decode_frames_device_planar(self, plane_ptrs, strides, plane_frame_strides, nframes, w, h, profile, sc, rgb_plane_ptrs,
                                    frame_stride):
        self._chk(self.L.lumahip_decode_frames_device_planar(self.h, _arr3(C.c_void_p, plane_ptrs), _arr3(C.c_int, strides),
                                                             _arr3(C.c_size_t, plane_frame_strides), nframes, w, h, profile, sc,
                                                             _arr3(C.c_void_p, rgb_plane_ptrs), frame_stride))

    def begin_unordered(self, lanes: int = 0):
        """open an unordered section: the following _device encode / decode calls are independent of each other"""
        self._chk(self.L.lumahip_begin_unordered(self.h, lanes))

    def end_unordered(self):
        self._chk(self.L.lumahip_end_unordered(self.h))

    def probe_decode_traffic(self, plane_ptrs, strides, plane_frame_strides, nframes, w, h, rgb_plane_ptrs, frame_stride,
                             iters=1) -> float:
        """ms per launch of the decode kernel's loads + stores without arithmetic (overwrites the frames)"""
        ms = C.c_float(0)
        self._chk(self.L.lumahip_probe_decode_traffic_device(self.h, _arr3(C.c_void_p, plane_ptrs), _arr3(C.c_int, strides),
                                                             _arr3(C.c_size_t, plane_frame_strides), nframes, w, h,
                                                             _arr3(C.c_void_p, rgb_plane_ptrs), frame_stride, iters, C.byref(ms)))
        return float(ms.value)

    def probe_encode_traffic(self, rgb_ptr, frame_stride, nframes, w, h, plane_ptrs, strides, plane_frame_strides,
                             iters=1) -> float:
        """ms per launch of the encode kernel's loads + stores without arithmetic (overwrites the planes)"""
        ms = C.c_float(0)
        self._chk(self.L.lumahip_probe_encode_traffic_device(self.h, rgb_ptr, frame_stride, nframes, w, h,
                                                             _arr3(C.c_void_p, plane_ptrs), _arr3(C.c_int, strides),
                                                             _arr3(C.c_size_t, plane_frame_strides), iters, C.byref(ms)))
        return float(ms.value)

    def powf_probe_device(self, out_ptr, first_bits, n, y, regular=1):
        self._chk(self.L.lumahip_powf_probe_device(self.h, out_ptr, first_bits, n, y, int(regular)))

    def quantize_probe_device(self, out_ptr, first_bits, n, nonneg=False):
        """uint16 codes of the n consecutive fp32 bit patterns from first_bits, through quantize_lut<mode, 4, nonneg>"""
        self._chk(self.L.lumahip_quantize_probe_device(self.h, out_ptr, first_bits, n, int(bool(nonneg))))

    def ycbcr_luma_probe_device(self, out_ptr, first_bits, n, direct=False):
        """uint16 luminance codes of the n consecutive fp32 luma bit patterns from first_bits (YCbCr quantizers): through the
        composite records (direct=False) or through the reference's arithmetic on the device (direct=True)"""
        self._chk(self.L.lumahip_ycbcr_luma_probe_device(self.h, out_ptr, first_bits, n, int(bool(direct))))

    def host_register(self, arr: np.ndarray):
        """pin a numpy array's memory for PCIe-rate transfers by the host entry points"""
        self._chk(self.L.lumahip_host_register(self.h, arr.ctypes.data, arr.nbytes))

    def host_unregister(self, arr: np.ndarray):
        self._chk(self.L.lumahip_host_unregister(self.h, arr.ctypes.data))

    # ---- raw device memory (hosts without torch)
    def malloc(self, nbytes) -> int:
        p = C.c_void_p()
        self._chk(self.L.lumahip_malloc(self.h, C.byref(p), nbytes))
        return p.value

    def free(self, ptr):
        self._chk(self.L.lumahip_free(self.h, ptr))

    def h2d(self, dst_ptr, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        self._chk(self.L.lumahip_memcpy_h2d(self.h, dst_ptr, arr.ctypes.data, arr.nbytes))

    def d2h(self, arr: np.ndarray, src_ptr):
        assert arr.flags.c_contiguous
        self._chk(self.L.lumahip_memcpy_d2h(self.h, arr.ctypes.data, src_ptr, arr.nbytes))


class DecodedRing:
    """lumahip_decoded_ring: `nbatches` batches of up to `nframes` packed LumaFrames in buffers the LIBRARY allocates and places
    (the frames of a batch rotate over three buffers in three HBM region groups); include/lumahip.h"""

    def __init__(self, ctx: Context, nbatches, nframes, w, h):
        self.L, self.ctx = lib(), ctx
        h_ = C.c_void_p()
        rc = self.L.lumahip_decoded_ring_create(ctx.h, nbatches, nframes, w, h, C.byref(h_))
        if rc != OK:
            raise LumaHipError(rc, "lumahip_decoded_ring_create failed")
        self.h = h_
        a, fs = (C.c_int * 4)(), C.c_size_t(0)
        self.L.lumahip_decoded_ring_info(self.h, a, C.byref(fs))
        self.placed, self.nbatches, self.nframes, self.groups, self.frame_stride = bool(a[0]), a[1], a[2], a[3], fs.value

    def frame_ptr(self, batch, frame) -> int:
        p = self.L.lumahip_decoded_ring_frame(self.h, batch, frame)
        if not p:
            raise LumaHipError(ERR_ARG, "no frame %d of batch %d in this ring" % (frame, batch))
        return p

    def decode(self, plane_ptrs, strides, plane_frame_strides, nframes, profile, sc, batch):
        """lumahip_decode_frames_device_ring: nframes frames into batch slot `batch` (asynchronous)"""
        self.ctx._chk(self.L.lumahip_decode_frames_device_ring(self.ctx.h, _arr3(C.c_void_p, plane_ptrs), _arr3(C.c_int, strides),
                                                               _arr3(C.c_size_t, plane_frame_strides), nframes, profile, sc, self.h, batch))

    def close(self):
        if self.h:
            self.L.lumahip_decoded_ring_destroy(self.h)
            self.h = None


class Pool:
    """lumahip_pool: device memory in chunks, classified by HBM region group (include/lumahip.h)"""

    def __init__(self, ctx: Context, n_float, n_y, n_uv, n_striped=0, chunk_bytes=0, keep_free=0, max_chunks=0, iters=0, small=False):
        self.L = lib()
        h = C.c_void_p()
        if small:       # lumahip_pool_create_small: a dozen chunks probed for milliseconds instead of all free memory for seconds
            rc = self.L.lumahip_pool_create_small(ctx.h, n_float, n_y, n_uv, n_striped, C.byref(h))
        else:
            cfg = PoolConfig(chunk_bytes, n_float, n_y, n_uv, n_striped, keep_free, max_chunks, iters)
            rc = self.L.lumahip_pool_create(ctx.h, C.byref(cfg), C.byref(h))
        if rc != OK:
            raise LumaHipError(rc, "lumahip_pool_create failed")
        self.h = h
        self.chunk_bytes = chunk_bytes or (2 << 30)

    def stats(self) -> dict:
        import json
        return json.loads(self.L.lumahip_pool_stats_json(self.h).decode())

    def available(self, kind, group=-1) -> int:
        return int(self.L.lumahip_pool_available(self.h, kind, group))

    def alloc(self, kind, group=-1) -> int:
        p = C.c_void_p()
        rc = self.L.lumahip_pool_alloc(self.h, kind, group, C.byref(p))
        if rc != OK:
            raise LumaHipError(rc, "lumahip_pool_alloc: no chunk of kind %d / group %d left" % (kind, group))
        return p.value

    def release(self, ptr: int):
        rc = self.L.lumahip_pool_release(self.h, C.c_void_p(ptr))
        if rc != OK:
            raise LumaHipError(rc, "lumahip_pool_release: not a chunk of this pool")

    def group_of(self, ptr: int) -> int:
        return int(self.L.lumahip_pool_group_of(self.h, C.c_void_p(ptr)))

    def close(self):
        if getattr(self, "h", None):
            self.L.lumahip_pool_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Multi:
    """lumahip_multi: one context per shard, block-sharded batches, quantizer broadcast over RCCL (include/lumahip.h)"""

    def __init__(self, devices=None, nshards=0):
        self.L = lib()
        h = C.c_void_p()
        if devices is None:
            rc = self.L.lumahip_multi_create(C.byref(h), None, nshards)
        else:
            arr = (C.c_int * len(devices))(*devices)
            rc = self.L.lumahip_multi_create(C.byref(h), arr, len(devices))
        if rc != OK:
            raise LumaHipError(rc, "lumahip_multi_create failed (there is no CPU fallback)")
        self.h = h

    def _chk(self, rc):
        if rc != OK:
            raise LumaHipError(rc, self.L.lumahip_multi_last_error(self.h).decode())

    @property
    def shards(self) -> int:
        return int(self.L.lumahip_multi_shards(self.h))

    def used_rccl(self) -> bool:
        return bool(self.L.lumahip_multi_used_rccl(self.h))

    def set_transport(self, mode: int):
        """0 auto (RCCL across several devices, host copies on one), 1 always RCCL, 2 always host copies"""
        self._chk(self.L.lumahip_multi_set_transport(self.h, mode))

    def transport_note(self) -> str:
        return self.L.lumahip_multi_transport_note(self.h).decode()

    def ctx(self, shard: int) -> Context:
        """the shard's context as a (non-owning) Context"""
        raw = self.L.lumahip_multi_ctx(self.h, shard)
        if not raw:
            raise LumaHipError(ERR_ARG, "no such shard")
        return _BorrowedContext(self.L, C.c_void_p(raw))

    def set_quantizer(self, ptf, bitdepth, cs, bitdepthC, max_lum, min_lum, lut: np.ndarray):
        lut = np.ascontiguousarray(lut, dtype=np.float32)
        self._chk(self.L.lumahip_multi_set_quantizer(self.h, ptf, bitdepth, cs, bitdepthC, max_lum, min_lum, lut.ctypes.data,
                                                     lut.size))

    def encode_frames(self, frames, sc=1.0, profile=2, align=32):
        frames = [np.ascontiguousarray(f, dtype=np.float32) for f in frames]
        n = len(frames)
        _, h, w = frames[0].shape
        _, hs, st, _ = plane_geometry(w, h, profile, align)
        planes = [[np.zeros((hs[p], st[p]), dtype=np.uint8) for p in range(3)] for _ in range(n)]
        fp = (C.c_void_p * n)(*[f.ctypes.data for f in frames])
        pp = (C.c_void_p * (3 * n))(*[pl.ctypes.data for tri in planes for pl in tri])
        means = (C.c_float * n)()
        self._chk(self.L.lumahip_multi_encode_frames_host(self.h, fp, n, w, h, sc, profile, pp, _arr3(C.c_int, st), means))
        return planes, st, [float(m) for m in means]

    def decode_frames(self, planes_list, strides, w, h, sc=1.0, profile=2):
        n = len(planes_list)
        planes_list = [[np.ascontiguousarray(p) for p in tri] for tri in planes_list]
        outs = [np.empty((3, h, w), dtype=np.float32) for _ in range(n)]
        pp = (C.c_void_p * (3 * n))(*[pl.ctypes.data for tri in planes_list for pl in tri])
        op = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        self._chk(self.L.lumahip_multi_decode_frames_host(self.h, pp, _arr3(C.c_int, strides), n, w, h, profile, sc, op))
        return outs

    def encode_frames_device(self, rgb_ptrs, frame_stride, counts, w, h, sc, profile, plane_ptrs, strides, plane_frame_strides):
        """rgb_ptrs[s], counts[s], plane_ptrs[s] = (Y, U, V) per shard"""
        ns = self.shards
        rp = (C.c_void_p * ns)(*rgb_ptrs)
        cn = (C.c_uint * ns)(*counts)
        pp = (C.c_void_p * (3 * ns))(*[p for tri in plane_ptrs for p in tri])
        self._chk(self.L.lumahip_multi_encode_frames_device(self.h, rp, frame_stride, cn, w, h, sc, profile, pp,
                                                            _arr3(C.c_int, strides), _arr3(C.c_size_t, plane_frame_strides)))

    def decode_frames_device(self, plane_ptrs, strides, plane_frame_strides, counts, w, h, profile, sc, rgb_ptrs, frame_stride):
        ns = self.shards
        rp = (C.c_void_p * ns)(*rgb_ptrs)
        cn = (C.c_uint * ns)(*counts)
        pp = (C.c_void_p * (3 * ns))(*[p for tri in plane_ptrs for p in tri])
        self._chk(self.L.lumahip_multi_decode_frames_device(self.h, pp, _arr3(C.c_int, strides),
                                                            _arr3(C.c_size_t, plane_frame_strides), cn, w, h, profile, sc, rp,
                                                            frame_stride))

    def sync(self):
        self._chk(self.L.lumahip_multi_sync(self.h))

    def close(self):
        if getattr(self, "h", None):
            self.L.lumahip_multi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _BorrowedContext(Context):
    """a Context view of a lumahip_ctx owned by someone else (a Multi): never destroyed from here"""

    def __init__(self, L, h):
        self.L = L
        self.h = h

    def close(self):
        self.h = None
