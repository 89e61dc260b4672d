// tests/cpp/source_decode.cpp -- INTEGRATION.md way A, decode side, linked and run: the reference's MkvInterface + libvpx VP9
// decoder (tools/integration/vpx_mkv_source.h) feeding this repo's LumaDecoder (fused HIP decode kernel through the C ABI).
// Built by `make -C oracle ref_full` in the build container.
//
//   source_decode_hipA in.mkv out_%05d.exr      (the frames go out through ExrInterface::writeFrame, as lumadec writes them)
#include <cstdio>

#include "exr_interface.h"
#include "vpx_mkv_source.h"

int main(int argc, char **argv)
{
    if (argc < 3) {
        fprintf(stderr, "usage: %s in.mkv out_%%05d.exr\n", argv[0]);
        return 2;
    }
    try {
        VpxMkvSource src;
        LumaDecoder dec;
        dec.setSource(&src);
        dec.initialize(argv[1]);
        int n = 0;
        char name[600];
        while (LumaFrame *f = dec.decode()) {
            snprintf(name, sizeof name, argv[2], ++n);
            ExrInterface::writeFrame(name, *f);
        }
        fprintf(stderr, "%d frames decoded\n", n);
    } catch (LumaException &e) {
        fprintf(stderr, "source_decode: %s\n", e.what());
        return 1;
    }
    return 0;
}
