// luma_frame.h -- planar fp32 frame, the boundary type of the hot path.
//
// Layout contract (the reference's LumaFrame, include/luma/luma_frame.h:51-90 there, which every caller and both
// I/O adapters rely on): public `width`, `height`, `channels`, `buffer`; `buffer` holds channels*height*width floats,
// channel c starts at buffer + c*height*width, rows are `width` floats, no padding; init() (re)allocates for the
// current dimensions, clear() releases, getChannel(c) returns the plane pointer.  Written from scratch here.
#ifndef LUMA_HIP_FRAME_H
#define LUMA_HIP_FRAME_H

#include <cstddef>

struct LumaFrame {
    unsigned int height = 0, width = 0, channels = 3;
    float *buffer = NULL;

    LumaFrame() {}
    LumaFrame(unsigned int w, unsigned int h, unsigned int c = 3) : height(h), width(w), channels(c)
    {
        if (pixelCount() != 0)
            init();
    }
    ~LumaFrame() { clear(); }

    // a frame owns its storage; a copy would free it twice (as it would in the reference), so copying is disabled
    LumaFrame(const LumaFrame &) = delete;
    LumaFrame &operator=(const LumaFrame &) = delete;

    size_t planeSize() const { return (size_t)height * width; }
    size_t pixelCount() const { return planeSize() * channels; }

    void clear()
    {
        delete[] buffer;
        buffer = NULL;
    }

    // The reference refuses only when width, height AND channels are all zero; a frame with a
    // single zero dimension gets an empty allocation.  Same here.
    bool init()
    {
        const bool nothingSet = (height | width | channels) == 0;
        if (nothingSet)
            return false;
        clear();
        buffer = new float[pixelCount()];
        return true;
    }

    float *getChannel(unsigned int c) { return buffer + c * planeSize(); }
    const float *getChannel(unsigned int c) const { return buffer + c * planeSize(); }
};

#endif
