// tools/lumadec.cpp -- the MI355X counterpart of the reference's `lumadec` application (lumadec.cpp there) on the C++
// facade: --input <stream> --output <EXR printf pattern> [--verbose]; every frame comes out of LumaDecoder::decode (ONE
// fused HIP kernel) and is written with ExrInterface::writeFrame.  The input is this build's raw Luma plane stream
// (.lhs) instead of VP9 in Matroska (out of scope; a LumaPlaneSource for it attaches upstream of the decoder).
#include <cstdio>
#include <cstdlib>
#include <string>

#include "exr_interface.h"
#include "luma/luma_decoder.h"
#include "luma_cli.h"

int main(int argc, char *argv[])
{
    std::string pattern, input;
    bool verbose = false;
    try {
        lumacli::Options opt("lumadec -- Decode a high dynamic range (HDR) video that has been encoded with the HDRv codec\n\n"
                             "Usage: lumadec --input <hdr_video> --output <hdr_frames>\n",
                             "\nExample: lumadec -i hdr_video.lhs -o hdr_frame_%05d.exr\n");
        opt.text(&input, "--input", "-i", "Input HDR video", true);
        opt.text(&pattern, "--output", "-o", "Output location of decoded HDR frames");
        opt.flag(&verbose, "--verbose", "-v", "Verbose mode");
        if (!opt.parse(argc, argv))
            return 1;
        LumaDecoder decoder(input.c_str(), verbose);
        // the reference's loop as it stands, with the decoder reading one frame ahead (the download of frame i runs under the read,
        // upload and kernel of frame i+1; same frames in the same order; LUMADEC_PIPELINED=0 switches it off)
        const char *pipeEnv = std::getenv("LUMADEC_PIPELINED");
        decoder.setPipelined(!(pipeEnv && std::atoi(pipeEnv) == 0));
        int done = 0;
        for (int f = 1;; f++) {
            std::fprintf(stderr, "Decoding frame %d... ", f);
            LumaFrame *frame = decoder.decode();
            if (!frame)
                break;  // end of stream
            std::fprintf(stderr, "done\n");
            done++;
            if (pattern.empty() || lumacli::endsWithNoCase(pattern, "pfs"))
                throw LumaException("Compiled without pfstools support");
            char path[500];
            std::snprintf(path, sizeof path - 1, pattern.c_str(), f);
            if (!ExrInterface::writeFrame(path, *frame))
                break;
        }
        std::fprintf(stderr, "\n\nDecoding finished. %d frames decoded.\n", done);
    } catch (const lumacli::UsageError &e) {
        std::fprintf(stderr, "\nlumadec input error: %s\n", e.what());
        return 1;
    } catch (const LumaException &e) {
        std::fprintf(stderr, "\nlumadec decoding error: %s\n", e.what());
        return 1;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "\nlumadec error: %s\n", e.what());
        return 1;
    }
    return 0;
}
