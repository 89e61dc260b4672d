/*
 * ref_argparser_harness.cpp -- the reference's own command-line parser (src/arg_parser.cpp, include/arg_parser.h, compiled
 * unmodified where they lie by `make -C oracle ref_args`) behind the option table of the reference's lumaenc
 * (lumaenc.cpp:125-143, registered here through the parser's public add() calls).  TEST INFRASTRUCTURE ONLY: lets
 * tests/test_host_side.py compare tools/lumaenc's option handling (tools/luma_cli.h, an independent implementation) with
 * the reference parser's accept / reject decisions, messages and parsed values on the same command lines.
 * lumaenc.cpp itself cannot be built here (it includes a CMake-generated config.h and needs OpenEXR).
 *
 * Output: "OK <key>=<value> ..." on success, "HELP" when the parser showed its usage text, "ERR <message>" on a
 * ParserException.
 */
#include <cstdio>
#include <string>

#include "arg_parser.h"

int main(int argc, char **argv)
{
    std::string input, output, frames, ptf, cs;
    std::string ptfValues[] = {"PSI", "PQ", "LOG", "HDRVDP", "LINEAR"}, csValues[] = {"LUV", "RGB", "YCBCR", "XYZ"};
    unsigned int bdValues[] = {8, 10, 12};
    // defaults of LumaEncoderParams (include/luma/luma_encoder.h:59-69,118-126)
    float fps = 25.0f, preScaling = 1.0f, maxLum = 1e4f, minLum = 0.005f;
    unsigned int profile = 2, quantizerScale = 2, ptfBitDepth = 11, colorBitDepth = 8, bitrate = 10000, keyframeInterval = 0, bitDepth = 12;
    bool lossLess = false, verbose = false;
    try {
        ArgParser p("usage", "post");
        p.add(&input, "--input", "-i", "");
        p.add(&output, "--output", "-o", "", 0);
        p.add(&frames, "--frames", "-f", "");
        p.add(&fps, "--framerate", "-fps", "");
        p.add(&profile, "--profile", "-p", "", (unsigned int)(0), (unsigned int)(3));
        p.add(&quantizerScale, "--quantizer-scaling", "-q", "", (unsigned int)(0), (unsigned int)(63));
        p.add(&preScaling, "--pre-scaling", "-sc", "", 0.0f, 1e20f);
        p.add(&ptfBitDepth, "--ptf-bitdepth", "-pb", "", (unsigned int)(0), (unsigned int)(16));
        p.add(&colorBitDepth, "--color-bitdepth", "-cb", "", (unsigned int)(0), (unsigned int)(16));
        p.add(&ptf, "--transfer-function", "-ptf", "", ptfValues, 5);
        p.add(&cs, "--color-space", "-cs", "", csValues, 4);
        p.add(&maxLum, "--max-luminance", "-ma", "", 100.0f, 1e5f);
        p.add(&minLum, "--min-luminance", "-mi", "", 1e-10f, 99.99f);
        p.add(&bitrate, "--bitrate", "-b", "", (unsigned int)(0), (unsigned int)(9999));
        p.add(&keyframeInterval, "--keyframe-interval", "-k", "", (unsigned int)(0), (unsigned int)(9999));
        p.add(&bitDepth, "--encoding-bitdepth", "-eb", "", bdValues, 3);
        p.add(&lossLess, "--lossless", "-l", "");
        p.add(&verbose, "--verbose", "-v", "");
        if (!p.read(argc, argv)) {
            printf("HELP\n");
            return 0;
        }
    } catch (ParserException &e) {
        printf("ERR %s\n", e.what());
        return 0;
    }
    printf("OK input=%s output=%s frames=%s fps=%.9g profile=%u q=%u sc=%.9g pb=%u cb=%u ptf=%s cs=%s ma=%.9g mi=%.9g b=%u k=%u eb=%u l=%d v=%d\n",
           input.c_str(), output.c_str(), frames.c_str(), fps, profile, quantizerScale, preScaling, ptfBitDepth, colorBitDepth, ptf.c_str(),
           cs.c_str(), maxLum, minLum, bitrate, keyframeInterval, bitDepth, (int)lossLess, (int)verbose);
    return 0;
}
