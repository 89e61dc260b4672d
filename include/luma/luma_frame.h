// luma_frame.h -- planar fp32 frame, the boundary type of the hot path.
// Same public members and layout as the reference's LumaFrame (include/luma/luma_frame.h:51-90 there):
// `buffer` holds channels*height*width floats, channel c starts at buffer + c*height*width, rows are
// `width` floats, no padding.  Written from scratch for this repo.
#ifndef LUMA_HIP_FRAME_H
#define LUMA_HIP_FRAME_H

#include <cstddef>

struct LumaFrame {
    explicit LumaFrame(unsigned int w = 0, unsigned int h = 0, unsigned int c = 3)
        : height(h), width(w), channels(c), buffer(NULL)
    {
        if (w != 0 && h != 0 && c != 0)
            init();
    }
    ~LumaFrame() { clear(); }

    // frames own their storage; copying would double-free, exactly as in the reference, so forbid it here
    LumaFrame(const LumaFrame &) = delete;
    LumaFrame &operator=(const LumaFrame &) = delete;

    void clear()
    {
        delete[] buffer;
        buffer = NULL;
    }

    // (re)allocates for the current width/height/channels.  The reference refuses only when ALL three are
    // zero (its guard uses &&); a frame with one zero dimension gets an empty buffer.
    bool init()
    {
        if (height == 0 && width == 0 && channels == 0)
            return false;
        clear();
        buffer = new float[(size_t)channels * height * width];
        return true;
    }

    float *getChannel(unsigned int c) { return buffer + (size_t)c * height * width; }
    const float *getChannel(unsigned int c) const { return buffer + (size_t)c * height * width; }

    unsigned int height, width, channels;
    float *buffer;
};

#endif
