// tools/test_simple_enc.cpp -- counterpart of the reference's test/test_simple_enc.cpp (same arguments,
// same parameter set, same progress lines) on the MI355X facade.
//   test_simple_enc <hdr_frames printf pattern> <start_frame> <end_frame> <output>
// Without arguments five synthetic test frames (1280x720) are encoded into "output.lhs" (a raw Y/U/V plane
// stream with the reference's metadata attachments; the VP9 + Matroska stages are out of scope).
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "exr_interface.h"
#include "luma/luma_encoder.h"

int main(int argc, char *argv[])
{
    if (argc > 1 && (!strcmp(argv[1], "-h") || !strcmp(argv[1], "--help"))) {
        printf("Usage: ./test_simple_enc <hdr_frames> <start_frame> <end_frame> <output>\n");
        return 1;
    }
    const char *hdrFrames = argc > 1 ? argv[1] : NULL;
    const int startFrame = argc > 2 ? atoi(argv[2]) : 1;
    const int endFrame = argc > 3 ? atoi(argv[3]) : 5;
    const char *outputFile = argc > 4 ? argv[4] : "output.lhs";

    try {
        LumaEncoder encoder;
        LumaEncoderParams params = encoder.getParams();
        params.profile = 2;
        params.bitrate = 1000;
        params.keyframeInterval = 0;
        params.bitDepth = 12;
        params.ptfBitDepth = 11;
        params.colorBitDepth = 8;
        params.lossLess = 0;
        params.quantizerScale = 4;
        params.ptf = LumaQuantizer::PTF_PQ;
        params.colorSpace = LumaQuantizer::CS_LUV;
        encoder.setParams(params);

        char name[500];
        for (int f = startFrame; f <= endFrame; f++) {
            printf("Encoding frame %d.\n", f);
            LumaFrame frame;
            if (hdrFrames != NULL) {
                snprintf(name, sizeof name, hdrFrames, f);
                ExrInterface::readFrame(name, frame);
            } else {
                ExrInterface::testFrame(frame);
            }
            if (!encoder.initialized())
                encoder.initialize(outputFile, frame.width, frame.height);
            encoder.encode(&frame);
        }
        encoder.finish();
        printf("Encoding finished. %d frames encoded.\n", endFrame - startFrame + 1);
    } catch (LumaException &e) {
        fprintf(stderr, "\nError: %s\n", e.what());
        return 1;
    }
    return 0;
}
