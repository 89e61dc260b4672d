// tests/cpp/exr_tool.cpp -- thin CLI over ExrInterface for the Python tests.
//   exr_tool read  in.exr out.f32      -> writes w,h (2 x uint32) + 3*w*h floats
//   exr_tool write in.f32 out.exr comp asFloat
//   exr_tool half                       -> reads floats from stdin (binary), writes half bits (uint16) to stdout
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "exr_interface.h"

int main(int argc, char **argv)
{
    try {
        if (argc >= 4 && !strcmp(argv[1], "read")) {
            LumaFrame f;
            ExrInterface::readFrame(argv[2], f);
            FILE *o = fopen(argv[3], "wb");
            unsigned dims[2] = {f.width, f.height};
            fwrite(dims, 4, 2, o);
            fwrite(f.buffer, 4, (size_t)3 * f.width * f.height, o);
            fclose(o);
            return 0;
        }
        if (argc >= 6 && !strcmp(argv[1], "write")) {
            FILE *i = fopen(argv[2], "rb");
            unsigned dims[2];
            if (fread(dims, 4, 2, i) != 2)
                return 2;
            LumaFrame f(dims[0], dims[1], 3);
            if (fread(f.buffer, 4, (size_t)3 * dims[0] * dims[1], i) != (size_t)3 * dims[0] * dims[1])
                return 2;
            fclose(i);
            ExrInterface::writeFrame(argv[3], f, (ExrInterface::Compression)atoi(argv[4]), atoi(argv[5]) != 0);
            return 0;
        }
        if (argc >= 2 && !strcmp(argv[1], "half")) {
            float v;
            while (fread(&v, 4, 1, stdin) == 1) {
                unsigned short h = ExrInterface::floatToHalf(v);
                fwrite(&h, 2, 1, stdout);
            }
            return 0;
        }
    } catch (LumaException &e) {
        fprintf(stderr, "LumaException: %s\n", e.what());
        return 1;
    }
    fprintf(stderr, "usage: exr_tool read|write|half ...\n");
    return 2;
}
