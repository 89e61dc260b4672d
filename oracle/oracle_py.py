"""ctypes bindings for the parity oracle.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg import this
module; the product package ``lumahdrv_amd`` never does.

* :class:`Oracle`   -- ``oracle/_build/libluma_oracle.so`` (our C restatement, builds anywhere).
* :class:`RefQuantizer` -- ``oracle/_ref/libluma_ref.so`` (the real reference ``LumaQuantizer`` compiled
  from ``/root/reference`` in the build container; the prebuilt ``.so`` travels to the GPU box).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "_build", "libluma_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libluma_ref.so")
REF_PLANES_TOOL = os.path.join(HERE, "_ref", "ref_planes_tool")

PTF_PSI, PTF_PQ, PTF_LOG, PTF_JND_HDRVDP, PTF_LINEAR = range(5)
CS_LUV, CS_RGB, CS_YCBCR, CS_XYZ = range(4)


def build(ref: bool = True) -> None:
    """(Re)build the oracle; the reference build is attempted only where /root/reference exists."""
    targets = ["all"]
    if ref and os.path.isdir("/root/reference/src"):
        targets += ["ref", "ref_planes", "ref_args", "ref_full"]
    subprocess.run(["make", "-s", "-C", HERE] + targets, check=True)


class _Q(C.Structure):
    _fields_ = [("ptf", C.c_int), ("cs", C.c_int), ("bitdepth", C.c_uint), ("bitdepthC", C.c_uint),
                ("maxVal", C.c_uint), ("maxValColor", C.c_uint), ("Lmax", C.c_float), ("Lmin", C.c_float),
                ("mapping", C.POINTER(C.c_float))]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            build(ref=False)
        L = C.CDLL(ORACLE_SO)
        fp = C.POINTER(C.c_float)
        L.lo_quantizer_init.argtypes = [C.POINTER(_Q)]
        L.lo_quantizer_free.argtypes = [C.POINTER(_Q)]
        L.lo_set_quantizer.argtypes = [C.POINTER(_Q), C.c_int, C.c_uint, C.c_int, C.c_uint, C.c_float, C.c_float,
                                       C.c_void_p, C.c_size_t]
        L.lo_overwrite_mapping.argtypes = [C.POINTER(_Q), C.c_void_p, C.c_size_t]
        L.lo_transform_pq.argtypes = [C.c_float, C.c_float, C.c_int]
        L.lo_transform_pq.restype = C.c_float
        L.lo_transform_log.argtypes = [C.c_float, C.c_float, C.c_float, C.c_int]
        L.lo_transform_log.restype = C.c_float
        L.lo_quantize.argtypes = [C.POINTER(_Q), C.c_float, C.c_uint]
        L.lo_quantize.restype = C.c_float
        L.lo_dequantize.argtypes = [C.POINTER(_Q), C.c_float, C.c_uint]
        L.lo_dequantize.restype = C.c_float
        L.lo_transform_color_space.argtypes = [C.POINTER(_Q), C.c_void_p, C.c_uint, C.c_uint, C.c_int, C.c_float]
        L.lo_pack_plane.argtypes = [C.POINTER(_Q), C.c_void_p, C.c_int, C.c_int, C.c_uint, C.c_uint, C.c_void_p,
                                    C.c_int, fp]
        L.lo_unpack_plane.argtypes = [C.POINTER(_Q), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_uint,
                                      C.c_void_p]
        pp = C.POINTER(C.c_void_p)
        ip = C.POINTER(C.c_int)
        L.lo_encode_frame.argtypes = [C.POINTER(_Q), C.c_void_p, C.c_uint, C.c_uint, C.c_float, C.c_int, pp, ip, fp]
        L.lo_decode_frame.argtypes = [C.POINTER(_Q), pp, ip, C.c_uint, C.c_uint, C.c_int, C.c_float, C.c_void_p]
        L.lo_encode_frame_mt.argtypes = L.lo_encode_frame.argtypes + [C.c_int]
        L.lo_decode_frame_mt.argtypes = L.lo_decode_frame.argtypes + [C.c_int]
        L.lo_powf_compare.argtypes = [C.c_void_p, C.c_uint32, C.c_size_t, C.c_float, C.c_int, C.POINTER(C.c_uint32)]
        L.lo_powf_compare.restype = C.c_size_t
        L.lo_test_frame.argtypes = [C.c_void_p, C.c_uint, C.c_uint]
        L.lo_splitmix64.argtypes = [C.c_uint64]
        L.lo_splitmix64.restype = C.c_uint64
        L.lo_synth_frame.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_uint64, C.c_uint64]
        L.lo_display_transform.argtypes = [C.c_void_p, C.c_size_t, C.c_double, C.c_double, C.c_int, C.c_int, C.c_void_p]
        L.lo_fnv1a64.argtypes = [C.c_void_p, C.c_size_t]
        L.lo_fnv1a64.restype = C.c_uint64
        L.lo_fnv1a64_basis.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64]
        L.lo_fnv1a64_basis.restype = C.c_uint64
        L.lo_fnv1a64_rows.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t]
        L.lo_fnv1a64_rows.restype = C.c_uint64
        _lib = L
    return _lib


def plane_geometry(w: int, h: int, profile: int, align: int = 32):
    """(widths, heights, strides-in-bytes, bytes-per-sample) the way vpx_img_alloc(fmt, w, h, 32) lays a
    frame out (libvpx vpx/src/vpx_image.c: stride = align-rounded width x bytes per sample; chroma
    stride = luma stride >> x_chroma_shift)."""
    sub = profile in (0, 2)
    bps = 2 if profile > 1 else 1
    aw = (w + align - 1) // align * align
    s0 = aw * bps
    cw, ch = ((w + 1) // 2, (h + 1) // 2) if sub else (w, h)
    s1 = s0 // 2 if sub else s0
    return (w, cw, cw), (h, ch, ch), (s0, s1, s1), bps


SURVEY_FNV_BASIS = 1469598103934665603  # the survey probe's (non-standard) offset basis, see luma_oracle.h


def fnv1a64(a, basis: int = 0xCBF29CE484222325) -> int:
    a = np.ascontiguousarray(a)
    return int(lib().lo_fnv1a64_basis(a.ctypes.data, a.nbytes, basis))


def survey_digest(a) -> str:
    """digest exactly as SURVEY.md 8(c) recorded it (16 hex digits)"""
    return "%016x" % fnv1a64(a, SURVEY_FNV_BASIS)


class Oracle:
    """The C restatement, one quantizer configuration."""

    def __init__(self, ptf=PTF_PQ, bitdepth=11, cs=CS_LUV, bitdepthC=8, max_lum=1e4, min_lum=0.005, table=None):
        self.L = lib()
        self.q = _Q()
        self.L.lo_quantizer_init(C.byref(self.q))
        tptr, tlen = None, 0
        if table is not None:
            self._table = np.ascontiguousarray(table, dtype=np.float32)
            tptr, tlen = self._table.ctypes.data, self._table.size
        rc = self.L.lo_set_quantizer(C.byref(self.q), ptf, bitdepth, cs, bitdepthC, max_lum, min_lum, tptr, tlen)
        if rc != 0:
            raise ValueError("lo_set_quantizer rejected the configuration")

    def __del__(self):
        try:
            self.L.lo_quantizer_free(C.byref(self.q))
        except Exception:
            pass

    @property
    def max_val(self):
        return int(self.q.maxVal)

    @property
    def max_val_color(self):
        return int(self.q.maxValColor)

    @property
    def mapping(self) -> np.ndarray:
        return np.ctypeslib.as_array(self.q.mapping, shape=(self.max_val + 1,)).copy()

    def overwrite_mapping(self, lut):
        lut = np.ascontiguousarray(lut, dtype=np.float32)
        if self.L.lo_overwrite_mapping(C.byref(self.q), lut.ctypes.data, lut.size) != 0:
            raise ValueError("bad LUT length")

    def quantize(self, v, ch=0):
        return float(self.L.lo_quantize(C.byref(self.q), v, ch))

    def dequantize(self, v, ch=0):
        return float(self.L.lo_dequantize(C.byref(self.q), v, ch))

    def transform(self, frame: np.ndarray, to_cs: bool, sc: float = 1.0) -> np.ndarray:
        """in place on a (3,h,w) float32 C-contiguous array; returns it"""
        assert frame.dtype == np.float32 and frame.flags.c_contiguous and frame.shape[0] == 3
        ok = self.L.lo_transform_color_space(C.byref(self.q), frame.ctypes.data, frame.shape[2], frame.shape[1],
                                             int(to_cs), sc)
        if not ok:
            raise ValueError("unknown colour space")
        return frame

    def encode(self, frame: np.ndarray, sc=1.0, profile=2, threads=1, align=32):
        """frame (3,h,w) float32 is MUTATED (as the reference does).  Returns (planes, strides, avg):
        planes = three uint8 arrays of shape (rows, stride)."""
        assert frame.dtype == np.float32 and frame.flags.c_contiguous and frame.shape[0] == 3
        h, w = frame.shape[1:]
        _, hs, strides, _ = plane_geometry(w, h, profile, align)
        planes = [np.zeros((hs[p], strides[p]), dtype=np.uint8) for p in range(3)]
        pp = (C.c_void_p * 3)(*[p.ctypes.data for p in planes])
        st = (C.c_int * 3)(*strides)
        avg = C.c_float(0)
        if threads > 1:
            self.L.lo_encode_frame_mt(C.byref(self.q), frame.ctypes.data, w, h, sc, profile, pp, st, C.byref(avg),
                                      threads)
        else:
            self.L.lo_encode_frame(C.byref(self.q), frame.ctypes.data, w, h, sc, profile, pp, st, C.byref(avg))
        return planes, strides, float(avg.value)

    def decode(self, planes, strides, w, h, sc=1.0, profile=2, threads=1) -> np.ndarray:
        out = np.empty((3, h, w), dtype=np.float32)
        planes = [np.ascontiguousarray(p) for p in planes]
        pp = (C.c_void_p * 3)(*[p.ctypes.data for p in planes])
        st = (C.c_int * 3)(*strides)
        if threads > 1:
            self.L.lo_decode_frame_mt(C.byref(self.q), pp, st, w, h, profile, sc, out.ctypes.data, threads)
        else:
            self.L.lo_decode_frame(C.byref(self.q), pp, st, w, h, profile, sc, out.ctypes.data)
        return out


def _oracle_unpack(self, planes, strides, w, h, profile=2) -> np.ndarray:
    """lo_unpack_plane x 3 = LumaDecoder::getVpxChannels on its own (no inverse colour transform)"""
    out = np.empty((3, h, w), dtype=np.float32)
    for p in range(3):
        pl = np.ascontiguousarray(planes[p])
        self.L.lo_unpack_plane(C.byref(self.q), pl.ctypes.data, int(strides[p]), p, profile, w, h, out[p].ctypes.data)
    return out


Oracle.unpack = _oracle_unpack


def powf_compare(got: np.ndarray, first_bits: int, y: float, threads: int = 0):
    """(#mismatches, first mismatching bit pattern) of got[i] vs this host's libm powf(bits(first+i), y)"""
    got = np.ascontiguousarray(got, dtype=np.float32)
    fb = C.c_uint32(0)
    n = lib().lo_powf_compare(got.ctypes.data, first_bits, got.size, y, threads or (os.cpu_count() or 1), C.byref(fb))
    return int(n), int(fb.value)


def test_frame(w=1280, h=720) -> np.ndarray:
    out = np.empty((3, h, w), dtype=np.float32)
    lib().lo_test_frame(out.ctypes.data, w, h)
    return out


def synth_frame(w, h, seed=20250929, frame=0) -> np.ndarray:
    out = np.empty((3, h, w), dtype=np.float32)
    lib().lo_synth_frame(out.ctypes.data, w, h, seed, frame)
    return out


def display_transform(rgb: np.ndarray, exposure=1.0, gamma=2.2, do_tmo=0, ldr_sim=0) -> np.ndarray:
    """the player's display transform (src/lumaplay_dequantizer.frag:145-156) on decoded linear RGB (3, h, w) -> RGBA8 (h, w, 4)"""
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    _, h, w = rgb.shape
    out = np.empty((h, w, 4), dtype=np.uint8)
    lib().lo_display_transform(rgb.ctypes.data, h * w, float(exposure), float(gamma), int(do_tmo), int(ldr_sim), out.ctypes.data)
    return out


def packed_rows(plane: np.ndarray, row_bytes: int) -> np.ndarray:
    """tightly-packed view of a strided plane (what the SURVEY digests hash)"""
    return np.ascontiguousarray(plane[:, :row_bytes])


# ---------------------------------------------------------------------------- the real reference


def have_ref() -> bool:
    return os.path.exists(REF_SO)


class RefQuantizer:
    """The reference's own LumaQuantizer (src/luma_quantizer.cpp compiled unmodified)."""

    _L = None

    def __init__(self, ptf=PTF_PQ, bitdepth=11, cs=CS_LUV, bitdepthC=8, max_lum=1e4, min_lum=0.005):
        if RefQuantizer._L is None:
            L = C.CDLL(REF_SO)
            L.ref_create.restype = C.c_void_p
            L.ref_destroy.argtypes = [C.c_void_p]
            L.ref_set_quantizer.argtypes = [C.c_void_p, C.c_int, C.c_uint, C.c_int, C.c_uint, C.c_float, C.c_float]
            L.ref_get_size.argtypes = [C.c_void_p]
            L.ref_get_size.restype = C.c_uint
            L.ref_get_mapping.argtypes = [C.c_void_p]
            L.ref_get_mapping.restype = C.POINTER(C.c_float)
            L.ref_overwrite_mapping.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
            L.ref_quantize.argtypes = [C.c_void_p, C.c_float, C.c_uint]
            L.ref_quantize.restype = C.c_float
            L.ref_dequantize.argtypes = [C.c_void_p, C.c_float, C.c_uint]
            L.ref_dequantize.restype = C.c_float
            L.ref_quantize_array.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint]
            L.ref_dequantize_array.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint]
            L.ref_transform_color_space.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_int, C.c_float]
            if hasattr(L, "ref_encode_frame"):
                L.ref_encode_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_float, C.c_int,
                                               C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_float)]
            RefQuantizer._L = L
        self.L = RefQuantizer._L
        self.h = self.L.ref_create()
        self.L.ref_set_quantizer(self.h, ptf, bitdepth, cs, bitdepthC, max_lum, min_lum)

    def __del__(self):
        try:
            self.L.ref_destroy(self.h)
        except Exception:
            pass

    @property
    def size(self):
        return int(self.L.ref_get_size(self.h))

    @property
    def mapping(self) -> np.ndarray:
        return np.ctypeslib.as_array(self.L.ref_get_mapping(self.h), shape=(self.size + 1,)).copy()

    def overwrite_mapping(self, lut):
        lut = np.ascontiguousarray(lut, dtype=np.float32)
        self.L.ref_overwrite_mapping(self.h, lut.ctypes.data, lut.size)

    def quantize(self, v, ch=0):
        return float(self.L.ref_quantize(self.h, v, ch))

    def dequantize(self, v, ch=0):
        return float(self.L.ref_dequantize(self.h, v, ch))

    def quantize_array(self, a, ch=0):
        a = np.ascontiguousarray(a, dtype=np.float32)
        out = np.empty_like(a)
        self.L.ref_quantize_array(self.h, a.ctypes.data, out.ctypes.data, a.size, ch)
        return out

    def dequantize_array(self, a, ch=0):
        a = np.ascontiguousarray(a, dtype=np.float32)
        out = np.empty_like(a)
        self.L.ref_dequantize_array(self.h, a.ctypes.data, out.ctypes.data, a.size, ch)
        return out

    def encode(self, frame: np.ndarray, sc=1.0, profile=2, align=32):
        """the real LumaQuantizer (transformColorSpace + per-sample quantize) under the harness's restatement of
        setVpxChannel's loop; mutates `frame`.  Returns (planes, strides, avg)."""
        assert frame.dtype == np.float32 and frame.flags.c_contiguous and frame.shape[0] == 3
        h, w = frame.shape[1:]
        _, hs, strides, _ = plane_geometry(w, h, profile, align)
        planes = [np.zeros((hs[p], strides[p]), dtype=np.uint8) for p in range(3)]
        pp = (C.c_void_p * 3)(*[p.ctypes.data for p in planes])
        st = (C.c_int * 3)(*strides)
        avg = C.c_float(0)
        self.L.ref_encode_frame(self.h, frame.ctypes.data, w, h, sc, profile, pp, st, C.byref(avg))
        return planes, strides, float(avg.value)

    def transform(self, frame: np.ndarray, to_cs: bool, sc: float = 1.0) -> np.ndarray:
        assert frame.dtype == np.float32 and frame.flags.c_contiguous and frame.shape[0] == 3
        ok = self.L.ref_transform_color_space(self.h, frame.ctypes.data, frame.shape[2], frame.shape[1], int(to_cs),
                                              sc)
        if not ok:
            raise ValueError("unknown colour space")
        return frame


def have_ref_planes() -> bool:
    return os.path.exists(REF_PLANES_TOOL)


class RefPlanes:
    """The reference's own plane loops -- LumaEncoder::setChannels / setVpxChannel (src/luma_encoder.cpp:196-201,260-317)
    and LumaDecoder::getVpxChannels (src/luma_decoder.cpp:205-240) -- compiled unmodified into oracle/_ref/ref_planes_tool
    (oracle/ref_planes_harness.cpp, `make -C oracle ref_planes`) and run as a subprocess."""

    def __init__(self, ptf=PTF_PQ, bitdepth=11, cs=CS_LUV, bitdepthC=8, max_lum=1e4, min_lum=0.005, lut_override=None):
        self.cfg = (ptf, bitdepth, cs, bitdepthC, max_lum, min_lum)
        self.lut = None if lut_override is None else np.ascontiguousarray(lut_override, dtype=np.float32)

    def _run(self, mode, profile, w, h, strides, xform, sc, data: bytes):
        import tempfile
        with tempfile.TemporaryDirectory() as d:
            inp, outp = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
            with open(inp, "wb") as f:
                f.write(data)
            ptf, bits, cs, bitsC, mx, mn = self.cfg
            cmd = [REF_PLANES_TOOL, mode, str(ptf), str(bits), str(cs), str(bitsC), repr(float(mx)), repr(float(mn)),
                   str(profile), str(w), str(h)] + [str(int(x)) for x in strides] + [str(int(xform)), repr(float(sc)), inp, outp]
            if self.lut is not None:
                lp = os.path.join(d, "lut.bin")
                self.lut.tofile(lp)
                cmd.append(lp)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("ref_planes_tool failed: %s" % r.stderr[-500:])
            with open(outp, "rb") as f:
                return f.read(), r.stderr

    def encode(self, frame: np.ndarray, sc=1.0, profile=2, strides=None, xform=True, align=32):
        """setChannels (after transformColorSpace when xform) -> (planes, strides, mean luminance the reference's warning
        printed or None).  Bytes the reference does not write keep the harness's 0xA5 fill."""
        frame = np.ascontiguousarray(frame, dtype=np.float32)
        h, w = frame.shape[1:]
        _, hs, st, _ = plane_geometry(w, h, profile, align)
        if strides is not None:
            st = tuple(strides)
        out, err = self._run("enc", profile, w, h, st, xform, sc, frame.tobytes())
        buf = np.frombuffer(out, dtype=np.uint8)
        planes, off = [], 0
        for p in range(3):
            n = hs[p] * st[p]
            planes.append(buf[off:off + n].reshape(hs[p], st[p]).copy())
            off += n
        mean = None
        for line in err.splitlines():
            if "Mean luminance is" in line:
                mean = float(line.split("Mean luminance is")[1].split()[0])
        return planes, st, mean

    def decode(self, planes, strides, w, h, sc=1.0, profile=2, xform=True) -> np.ndarray:
        data = b"".join(np.ascontiguousarray(p).tobytes() for p in planes)
        out, _ = self._run("dec", profile, w, h, strides, xform, sc, data)
        return np.frombuffer(out, dtype=np.float32).reshape(3, h, w).copy()
