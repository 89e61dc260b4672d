// tests/cpp/sink_encode.cpp -- INTEGRATION.md way A, linked and run: this repo's LumaEncoder (fused HIP kernel through the C
// ABI) in front of the reference's own downstream stages -- libvpx VP9 + its MkvInterface -- attached with
// tools/integration/vpx_mkv_sink.h.  Built only in the build container (`make -C oracle ref_full`, which builds the vendored
// libvpx / libebml / libmatroska); the GPU suite runs it next to the complete reference application and compares the .mkv files.
//
//   sink_encode_hipA out.mkv nframes [lossless]
// encodes `nframes` test-pattern frames (ExrInterface::testFrame, 1280x720 -- what `lumaenc -i __test__` feeds) with the default
// parameters of LumaEncoder, as the reference's lumaenc does with no options.
#include <cstdio>
#include <cstdlib>

#include "exr_interface.h"
#include "vpx_mkv_sink.h"

int main(int argc, char **argv)
{
    if (argc < 3) {
        fprintf(stderr, "usage: %s out.mkv nframes [lossless]\n", argv[0]);
        return 2;
    }
    try {
        LumaEncoder enc;
        LumaEncoderParams p = enc.getParams();
        p.lossLess = argc > 3 && atoi(argv[3]) != 0;
        enc.setParams(p);
        VpxMkvSink sink(p);
        enc.setSink(&sink);
        for (int f = 0; f < atoi(argv[2]); f++) {
            LumaFrame frame;
            ExrInterface::testFrame(frame);
            if (!enc.initialized())
                enc.initialize(argv[1], frame.width, frame.height);
            enc.encode(&frame);
        }
        enc.finish();
    } catch (LumaException &e) {
        fprintf(stderr, "sink_encode: %s\n", e.what());
        return 1;
    }
    return 0;
}
