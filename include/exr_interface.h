// exr_interface.h -- ExrInterface with the reference's interface (include/exr_interface.h:54-61 there):
// readFrame / writeFrame / testFrame on LumaFrame, without OpenEXR.
//
// A self-contained reader / writer for the subset of OpenEXR the reference's I/O path uses
// (src/exr_interface.cpp:73-187 goes through Imf::RgbaInputFile / RgbaOutputFile):
//   * single-part scan-line files, channels named R, G, B, A of type HALF, FLOAT or UINT, sampling 1;
//   * compression NONE, RLE, ZIPS, ZIP, PIZ, PXR24 on read (B44 / DWA files are rejected with a LumaException);
//     PIZ / PXR24 are validated against an independent Python restatement only (no OpenEXR in the build image);
//   * pixels pass through HALF exactly as Imf::Rgba does: FLOAT channels are rounded to half
//     (round-to-nearest-even, overflow to infinity) on read, and writeFrame stores HALF R,G,B (WRITE_RGB);
//   * channel handling as the reference: RGB / RGBA -> three planes; a file with only R, only G or only B
//     replicates that channel; anything else throws "Reading of luminance only frames not yet supported".
#ifndef LUMA_HIP_EXR_INTERFACE_H
#define LUMA_HIP_EXR_INTERFACE_H

#include <cstddef>
#include <cstdint>

#include "luma/luma_exception.h"
#include "luma/luma_frame.h"

class ExrInterface {
public:
    // OpenEXR's numbering; writeFrame produces 0..3, readFrame additionally decodes PIZ and PXR24
    enum Compression { NO_COMPRESSION = 0, RLE_COMPRESSION = 1, ZIPS_COMPRESSION = 2, ZIP_COMPRESSION = 3, PIZ_COMPRESSION = 4, PXR24_COMPRESSION = 5 };

    static bool readFrame(const char *inputFile, LumaFrame &frame);
    static bool writeFrame(const char *outputFile, LumaFrame &frame);
    static bool testFrame(LumaFrame &frame, unsigned int w = 1280, unsigned int h = 720);

    // additions: choose the compression writeFrame uses (default ZIP), FLOAT output for lossless tests
    static bool writeFrame(const char *outputFile, LumaFrame &frame, Compression c, bool asFloat);

    // IEEE half <-> float exactly as Imath's `half` does it
    static uint16_t floatToHalf(float f);
    static float halfToFloat(uint16_t h);
};

#endif
