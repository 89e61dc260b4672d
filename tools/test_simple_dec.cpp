// tools/test_simple_dec.cpp -- counterpart of the reference's test/test_simple_dec.cpp: decode every frame of
// a stream and write numbered EXR files.
//   test_simple_dec <input stream> <output printf pattern, e.g. out_%05d.exr>
#include <cstdio>
#include <cstring>

#include "exr_interface.h"
#include "luma/luma_decoder.h"

int main(int argc, char *argv[])
{
    if (argc < 3 || !strcmp(argv[1], "-h") || !strcmp(argv[1], "--help")) {
        printf("Usage: ./test_simple_dec <input> <output_frames>\n");
        return 1;
    }
    try {
        LumaDecoder decoder(argv[1]);
        LumaFrame *frame;
        char name[500];
        int n = 0;
        while ((frame = decoder.decode()) != NULL) {
            snprintf(name, sizeof name, argv[2], ++n);
            ExrInterface::writeFrame(name, *frame);
            printf("Decoded frame %d (%ux%u) -> %s\n", n, frame->width, frame->height, name);
        }
        printf("Decoding finished. %d frames decoded.\n", n);
    } catch (LumaException &e) {
        fprintf(stderr, "\nError: %s\n", e.what());
        return 1;
    }
    return 0;
}
