// exr_codecs.h -- chunk decoders used by exr_interface.cpp (PIZ, PXR24); see exr_codecs.cpp
#ifndef LUMA_HIP_EXR_CODECS_H
#define LUMA_HIP_EXR_CODECS_H

#include <cstddef>
#include <vector>

namespace lumaexr {

struct ChannelLayout {
    int type;   // 0 UINT, 1 HALF, 2 FLOAT
    int bytes;  // bytes per sample: 2 or 4
};

// Both produce `lines` scan lines in the standard uncompressed chunk layout (per line: every channel's `width` samples in
// channel order, little-endian) and throw LumaException on malformed input.
void piz_decode_block(const unsigned char *in, size_t nIn, const std::vector<ChannelLayout> &chans, int width, int lines,
                      std::vector<unsigned char> &raw_out);
void pxr24_decode_block(const unsigned char *in, size_t nIn, const std::vector<ChannelLayout> &chans, int width, int lines,
                        std::vector<unsigned char> &raw_out);

}  // namespace lumaexr

#endif
