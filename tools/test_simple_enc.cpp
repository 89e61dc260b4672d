// tools/test_simple_enc.cpp -- the MI355X counterpart of the reference's minimal encoder example
// (test/test_simple_enc.cpp there): same command line, same parameter set, same progress lines.
//
//   test_simple_enc [<hdr_frames printf pattern> [<start_frame> [<end_frame> [<output>]]]]
//
// With no arguments frames 1..5 are the synthetic 1280x720 test pattern and the stream goes to "output.lhs" --
// a raw Y/U/V plane stream carrying the reference's metadata attachments (VP9 + Matroska are out of scope).
#include <cstdio>
#include <cstdlib>
#include <string>

#include "exr_interface.h"
#include "luma/luma_encoder.h"

namespace {

struct Options {
    std::string pattern;  // empty: synthetic test frames
    int first = 1, last = 5;
    std::string output = "output.lhs";
};

bool wantsHelp(int argc, char **argv)
{
    return argc > 1 && (std::string(argv[1]) == "-h" || std::string(argv[1]) == "--help");
}

Options parse(int argc, char **argv)
{
    Options o;
    if (argc > 1) o.pattern = argv[1];
    if (argc > 2) o.first = std::atoi(argv[2]);
    if (argc > 3) o.last = std::atoi(argv[3]);
    if (argc > 4) o.output = argv[4];
    return o;
}

// the parameter set of the reference example: 12-bit VP9 profile 2, PQ 11 bits, Lu'v' with 8-bit chroma
LumaEncoderParams exampleParams(LumaEncoderParams p)
{
    p.profile = 2;
    p.bitDepth = 12;
    p.ptf = LumaQuantizer::PTF_PQ;
    p.ptfBitDepth = 11;
    p.colorSpace = LumaQuantizer::CS_LUV;
    p.colorBitDepth = 8;
    p.quantizerScale = 4;
    p.bitrate = 1000;
    p.keyframeInterval = 0;
    p.lossLess = false;
    return p;
}

void loadFrame(const Options &o, int index, LumaFrame &frame)
{
    if (o.pattern.empty()) {
        ExrInterface::testFrame(frame);
        return;
    }
    char path[1024];
    std::snprintf(path, sizeof path, o.pattern.c_str(), index);
    ExrInterface::readFrame(path, frame);
}

}  // namespace

int main(int argc, char *argv[])
{
    if (wantsHelp(argc, argv)) {
        std::printf("Usage: ./test_simple_enc <hdr_frames> <start_frame> <end_frame> <output>\n");
        return 1;
    }
    const Options opt = parse(argc, argv);
    try {
        LumaEncoder encoder;
        encoder.setParams(exampleParams(encoder.getParams()));
        for (int f = opt.first; f <= opt.last; ++f) {
            std::printf("Encoding frame %d.\n", f);
            LumaFrame frame;
            loadFrame(opt, f, frame);
            if (!encoder.initialized())
                encoder.initialize(opt.output.c_str(), frame.width, frame.height);
            encoder.encode(&frame);
        }
        encoder.finish();
        std::printf("Encoding finished. %d frames encoded.\n", opt.last - opt.first + 1);
    } catch (const LumaException &e) {
        std::fprintf(stderr, "\nError: %s\n", e.what());
        return 1;
    }
    return 0;
}
