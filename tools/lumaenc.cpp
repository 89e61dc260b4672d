// tools/lumaenc.cpp -- the MI355X counterpart of the reference's `lumaenc` application (lumaenc.cpp there) on the C++
// facade: the same options, defaults, ranges and messages; frames come from an EXR printf pattern or the built-in
// `__test__` pattern; every frame goes through LumaEncoder::encode (ONE fused HIP kernel).  With more than one GPU visible
// (or LUMAENC_SHARDS=<n> set) the frames are read in groups and go through LumaBatchEncoder instead: the group is split into
// contiguous blocks, one per GPU, transformed on all GPUs at once and written out in frame order (the reference's loop,
// lumaenc.cpp:205-243 there, is strictly one frame at a time).
//
// Difference, by scope: the reference's encoder hands the Y/U/V planes to libvpx + Matroska and insists on an .mkv
// output name; this build's downstream is a LumaPlaneSink and, by default, the raw plane stream (.lhs) that carries the
// reference's metadata attachments 430-436.  The VP9-only options (--bitrate, --quantizer-scaling, --keyframe-interval,
// --encoding-bitdepth, --lossless) are parsed, range-checked and stored in LumaEncoderParams exactly as the reference
// does; nothing in the hot path reads them.  PFS streams are not supported (the reference needs pfstools for them, too).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "exr_interface.h"
#include "luma/luma_batch_encoder.h"
#include "luma/luma_encoder.h"
#include "lumahip.h"
#include "luma_cli.h"

namespace {

struct Job {
    std::string frames, output, range, ptfText, csText;  // the last three as given, for LUMAENC_PRINT_ARGS
    unsigned int first = 1, last = 9999, step = 1;  // lumaenc.cpp:51-52
    bool verbose = false;
};

bool configure(int argc, char **argv, LumaEncoderParams &p, Job &job)
{
    std::string &range = job.range, &ptf = job.ptfText, &cs = job.csText;
    lumacli::Options opt(
        "lumaenc -- Compress a sequence of high dyncamic range (HDR) frames in to a Luma HDRv plane stream (.lhs) on an MI355X\n\n"
        "Usage: lumaenc --input <hdr_frames> \\\n"
        "               --frames <start_frame:step:end_frame> \\\n"
        "               --output <output>\n",
        "\nExample: lumaenc -i hdr_frame_%05d.exr -f 1:100 -o hdr_video.lhs\n");
    opt.text(&job.frames, "--input", "-i", "Input HDR video sequence");
    opt.text(&job.output, "--output", "-o", "Output location of the compressed HDR video", true);
    opt.text(&range, "--frames", "-f", "Input frames, formatted as startframe:step:endframe");
    opt.real(&p.fps, "--framerate", "-fps", "Framerate of video stream, specified as frames/s");
    opt.number(&p.profile, "--profile", "-p", "VP9 encoding profile", 0u, 3u);
    opt.number(&p.quantizerScale, "--quantizer-scaling", "-q", "Scaling of the encoding quantization", 0u, 63u);
    opt.real(&p.preScaling, "--pre-scaling", "-sc", "Scaling of pixels to apply before tranformation and encoding", 0.0f, 1e20f);
    opt.number(&p.ptfBitDepth, "--ptf-bitdepth", "-pb", "Bit depth of the perceptual transfer function", 0u, 16u);
    opt.number(&p.colorBitDepth, "--color-bitdepth", "-cb", "Bit depth of the color channels", 0u, 16u);
    opt.choice(&ptf, "--transfer-function", "-ptf", "The perceptual transfer function used for encoding", {"PSI", "PQ", "LOG", "HDRVDP", "LINEAR"});
    opt.choice(&cs, "--color-space", "-cs", "Color space for encoding", {"LUV", "RGB", "YCBCR", "XYZ"});
    opt.real(&p.maxLum, "--max-luminance", "-ma", "Maximum luminance in encoding (for PQ and LOG transfer function)", 100.0f, 1e5f);
    opt.real(&p.minLum, "--min-luminance", "-mi", "Minimum luminance in encoding (for PQ and LOG transfer function)", 1e-10f, 99.99f);
    opt.number(&p.bitrate, "--bitrate", "-b", "HDR video stream target bandwidth, in Kb/s", 0u, 9999u);
    opt.number(&p.keyframeInterval, "--keyframe-interval", "-k", "Interval between keyframes. 0 for automatic keyframes", 0u, 9999u);
    opt.numberOneOf(&p.bitDepth, "--encoding-bitdepth", "-eb", "Encoding at 8, 10 or 12 bits", {8u, 10u, 12u});
    opt.flag(&p.lossLess, "--lossless", "-l", "Enable lossless encoding mode");
    opt.flag(&job.verbose, "--verbose", "-v", "Verbose mode");
    if (!opt.parse(argc, argv))
        return false;

    if (!lumacli::endsWithNoCase(job.output, ".lhs"))
        throw lumacli::UsageError("Unsupported output format. This build stores the HDR video as a raw Luma plane stream (.lhs); "
                                  "VP9 + Matroska (.mkv) attach downstream through a LumaPlaneSink");
    if (!range.empty() && !lumacli::parseFrameRange(range, job.first, job.step, job.last))
        throw lumacli::UsageError("Unable to parse frame range from '" + range + "'. Valid format is startframe:step:endframe");
    if (job.last < job.first)
        throw lumacli::UsageError("Invalid frame range '" + range + "'. End frame should be >= start frame");
    if (job.step == 0)
        throw lumacli::UsageError("Invalid frame range '" + range + "'. Step should be >= 1");   // (the reference would loop forever)

    const char *ptfNames[] = {"PSI", "PQ", "LOG", "HDRVDP", "LINEAR"};  // = LumaQuantizer::ptf_t order
    for (int i = 0; i < 5; i++)
        if (ptf == ptfNames[i])
            p.ptf = (LumaQuantizer::ptf_t)i;
    const char *csNames[] = {"LUV", "RGB", "YCBCR", "XYZ"};             // = LumaQuantizer::colorSpace_t order
    for (int i = 0; i < 4; i++)
        if (cs == csNames[i])
            p.colorSpace = (LumaQuantizer::colorSpace_t)i;
    return true;
}

void fetch(const Job &job, unsigned int index, LumaFrame &frame)
{
    if (job.frames.empty() || lumacli::endsWithNoCase(job.frames, "pfs"))
        throw LumaException("Compiled without pfstools support");
    if (job.frames == "__test__") {
        ExrInterface::testFrame(frame);
        return;
    }
    char path[500];
    std::snprintf(path, sizeof path - 1, job.frames.c_str(), index);
    ExrInterface::readFrame(path, frame);
}

}  // namespace

int main(int argc, char *argv[])
{
    Job job;
    LumaEncoder encoder;
    LumaEncoderParams params = encoder.getParams();
    try {
        if (!configure(argc, argv, params, job))
            return 1;
        encoder.setParams(params);
        if (std::getenv("LUMAENC_PRINT_ARGS")) {  // tests: what the command line was understood as (same line format as
                                                  // oracle/ref_argparser_harness.cpp prints for the reference's parser)
            std::printf("OK input=%s output=%s frames=%s fps=%.9g profile=%u q=%u sc=%.9g pb=%u cb=%u ptf=%s cs=%s ma=%.9g mi=%.9g "
                        "b=%u k=%u eb=%u l=%d v=%d\n",
                        job.frames.c_str(), job.output.c_str(), job.range.c_str(), params.fps, params.profile, params.quantizerScale,
                        params.preScaling, params.ptfBitDepth, params.colorBitDepth, job.ptfText.c_str(), job.csText.c_str(),
                        params.maxLum, params.minLum, params.bitrate, params.keyframeInterval, params.bitDepth, (int)params.lossLess,
                        (int)job.verbose);
            return 0;
        }
        int done = 0;
        int gpus = 0;
        (void)lumahip_device_count(&gpus);
        const char *shardsEnv = std::getenv("LUMAENC_SHARDS");   // n shards, round-robin over the visible GPUs
        const int shards = shardsEnv ? std::atoi(shardsEnv) : (gpus > 1 ? gpus : 0);
        if (shards > 0) {
            // many GPUs: groups of (frames per shard) x shards frames through LumaBatchEncoder, written in frame order
            LumaBatchEncoder batch;
            batch.setParams(params);
            const char *perEnv = std::getenv("LUMAENC_FRAMES_PER_SHARD");
            const size_t group = (size_t)shards * (size_t)(perEnv && std::atoi(perEnv) > 0 ? std::atoi(perEnv) : 4);
            std::vector<std::unique_ptr<LumaFrame>> held;
            auto flush = [&]() {
                if (held.empty())
                    return;
                std::vector<LumaFrame *> ptrs;
                for (auto &fr : held)
                    ptrs.push_back(fr.get());
                std::fprintf(stderr, "Encoding %zu frames on %u shard(s)... ", held.size(), batch.shards());
                batch.encode(ptrs.data(), (unsigned int)ptrs.size());
                done += (int)held.size();
                held.clear();
                std::fprintf(stderr, "done\n");
            };
            for (unsigned int f = job.first; f <= job.last; f += job.step) {
                std::unique_ptr<LumaFrame> frame(new LumaFrame());
                fetch(job, f, *frame);
                if (!batch.initialized())
                    batch.initialize(job.output.empty() ? "output.lhs" : job.output.c_str(), frame->width, frame->height, job.verbose,
                                     NULL, shards);
                held.push_back(std::move(frame));
                if (held.size() == group)
                    flush();
            }
            flush();
            batch.finish();
        } else {
            // one GPU: the reference's loop as it stands, with the encoder in pipelined mode (frame i+1 is read from disk and
            // uploaded while frame i is still being finished; the stream written is the same, LUMAENC_PIPELINED=0 switches it off)
            const char *pipeEnv = std::getenv("LUMAENC_PIPELINED");
            encoder.setPipelined(!(pipeEnv && std::atoi(pipeEnv) == 0));
            for (unsigned int f = job.first; f <= job.last; f += job.step) {
                LumaFrame frame;
                fetch(job, f, frame);
                if (!encoder.initialized())
                    encoder.initialize(job.output.empty() ? "output.lhs" : job.output.c_str(), frame.width, frame.height, job.verbose);
                std::fprintf(stderr, "Encoding frame %d... ", f);
                encoder.encode(&frame);
                done++;
                std::fprintf(stderr, "done\n");
            }
            encoder.finish();
        }
        std::fprintf(stderr, "\n\nEncoding finished. %d frames encoded.\n", done);
    } catch (const lumacli::UsageError &e) {
        std::fprintf(stderr, "\nlumaenc input error: %s\n", e.what());
        return 1;
    } catch (const LumaException &e) {
        std::fprintf(stderr, "\nlumaenc encoding error: %s\n", e.what());
        return 1;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "\nlumaenc error: %s\n", e.what());
        return 1;
    }
    return 0;
}
