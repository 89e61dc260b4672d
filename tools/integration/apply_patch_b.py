#!/usr/bin/env python3
"""tools/integration/apply_patch_b.py <reference_root> <out_dir>

INTEGRATION.md, way B ("keep the reference's classes, replace only the two hot loops"), as an executable recipe: copies
the reference's include/luma/luma_{encoder,decoder}.h and src/luma_{encoder,decoder}.cpp into <out_dir> and makes the
five edits a maintainer would make -- an `#include "lumahip.h"` + a `lumahip_ctx *m_hip` member per class, the two
statements of LumaEncoder::encode / LumaDecoder::decode that run the hot loops replaced by one C-ABI call each, and
lumahip_create / lumahip_set_quantizer after each class's m_quant.setQuantizer(...).  The edits are located by the identifiers they attach to, so this file
contains none of the reference's text.  tests/test_host_side.py compiles the patched sources (build container only)."""
import os
import re
import sys


def sub1(text, pattern, repl, what, flags=re.S):
    new, n = re.subn(pattern, repl, text, count=1, flags=flags)
    if n != 1:
        raise SystemExit("anchor not found: " + what)
    return new


ENC_CALL = """// one fused HIP kernel replaces transformColorSpace(frame, true, preScaling) + setChannels(frame)
        {
            float avg = 0.0f;
            unsigned char *planes[3] = {m_rawFrame.planes[0], m_rawFrame.planes[1], m_rawFrame.planes[2]};
            int rc = lumahip_encode_frame_host(m_hip, frame->buffer, frame->width, frame->height, m_params.preScaling,
                                               (int)m_params.profile, planes, m_rawFrame.stride, &avg,
                                               frame->buffer /* the reference transforms the caller's frame in place */);
            if (rc != LUMAHIP_OK)
                throw LumaException(lumahip_last_error(m_hip));
            if (avg <= 1.0f)
                fprintf(stderr, "\\n\\tWarning! Mean luminance is %f cd/m2. Is input calibrated to physical units? \\n", avg);
        }
"""

DEC_CALL = """// one fused HIP kernel replaces getVpxChannels() + transformColorSpace(&m_frame, false, preScaling)
        {
            const unsigned char *planes[3] = {m_vpxFrame->planes[0], m_vpxFrame->planes[1], m_vpxFrame->planes[2]};
            int rc = lumahip_decode_frame_host(m_hip, planes, m_params.stride, m_vpxFrame->d_w, m_vpxFrame->d_h,
                                               m_params.profile, m_params.preScaling, m_frame.buffer);
            if (rc != LUMAHIP_OK)
                throw LumaException(lumahip_last_error(m_hip));
        }
"""

SETQ = """
    // MI355X hot path: hand the FINAL table to the device (include/lumahip.h)
    if (!m_hip && lumahip_create(&m_hip, -1) != LUMAHIP_OK)
        throw LumaException("No usable HIP device for the Luma HDRv quantizer");
    if (lumahip_set_quantizer(m_hip, (int)m_params.ptf, m_params.ptfBitDepth, (int)m_params.colorSpace,
                              m_params.colorBitDepth, m_params.maxLum, m_params.minLum, m_quant.getMapping(),
                              (size_t)m_quant.getSize() + 1) != LUMAHIP_OK)
        throw LumaException(lumahip_last_error(m_hip));
"""


def main():
    ref, out = sys.argv[1], sys.argv[2]
    os.makedirs(out, exist_ok=True)
    rd = lambda p: open(os.path.join(ref, p)).read()  # noqa: E731

    h = rd("include/luma/luma_encoder.h")
    h = sub1(h, r'(#include "vp8cx\.h"\n)', r'\1#include "lumahip.h"\n#include "luma_exception.h"\n#include <cstdio>\n', "encoder includes")
    # inside the inline encode(): drop the colour-transform statement, turn the setChannels statement into the C-ABI call
    h = sub1(h, r"[ \t]*m_quant\.transformColorSpace\(\s*frame\s*,\s*true[^;]*;\n", "", "encode: transformColorSpace statement")
    h = sub1(h, r"[ \t]*setChannels\(\s*frame\s*\)\s*;\n", lambda m: "        " + ENC_CALL, "encode: setChannels statement")
    h = sub1(h, r"(LumaEncoderParams\s+m_params;\n)", r"\1    lumahip_ctx *m_hip = NULL;\n", "encoder member")
    open(os.path.join(out, "luma_encoder.h"), "w").write(h)

    c = rd("src/luma_encoder.cpp")
    c = sub1(c, r"(m_quant\.setQuantizer\([^;]*;\n)", lambda m: m.group(1) + SETQ, "encoder setQuantizer")
    open(os.path.join(out, "luma_encoder.cpp"), "w").write(c)

    h = rd("include/luma/luma_decoder.h")
    h = sub1(h, r'(#include "vp8dx\.h"\n)', r'\1#include "lumahip.h"\n#include "luma_exception.h"\n', "decoder includes")
    # inside the inline decode(): turn the getVpxChannels statement into the C-ABI call, drop the inverse colour transform
    h = sub1(h, r"[ \t]*getVpxChannels\(\s*\)\s*;\n", lambda m: "        " + DEC_CALL, "decode: getVpxChannels statement")
    h = sub1(h, r"[ \t]*m_quant\.transformColorSpace\(\s*&m_frame\s*,\s*false[^;]*;\n", "", "decode: transformColorSpace statement")
    h = sub1(h, r"(LumaDecoderParams\s+m_params;\n)", r"\1    lumahip_ctx *m_hip = NULL;\n", "decoder member")
    open(os.path.join(out, "luma_decoder.h"), "w").write(h)

    c = rd("src/luma_decoder.cpp")
    # after the attachment-434 overwrite of the table (the memcpy into getMapping())
    c = sub1(c, r"(memcpy\(\(void\*\)m_quant\.getMapping\(\)[^;]*;\n)", lambda m: m.group(1) + SETQ, "decoder table overwrite")
    open(os.path.join(out, "luma_decoder.cpp"), "w").write(c)
    print("patched sources written to", out)


if __name__ == "__main__":
    main()
