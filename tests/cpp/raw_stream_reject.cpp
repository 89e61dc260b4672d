// tests/cpp/raw_stream_reject.cpp -- LumaRawStreamReader::open on crafted headers: implausible geometry, a geometry the
// file is too short for (would otherwise allocate gigabytes), truncated attachment tables.  Every case must raise
// LumaException (what the reference's applications catch), never std::bad_alloc / a crash.  Exit code 0 = all rejected
// and a well-formed one-frame stream accepted.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "luma/luma_exception.h"
#include "luma/luma_planes.h"

static void put_u32(std::vector<unsigned char> &v, unsigned x)
{
    for (int i = 0; i < 4; i++)
        v.push_back((unsigned char)(x >> (8 * i)));
}

static std::vector<unsigned char> header(unsigned w, unsigned h, unsigned profile, unsigned natt)
{
    std::vector<unsigned char> v;
    const char *m = "LHIPSTR1";
    v.insert(v.end(), m, m + 8);
    put_u32(v, w);
    put_u32(v, h);
    put_u32(v, profile);
    const float fps = 25.0f;
    unsigned f;
    memcpy(&f, &fps, 4);
    put_u32(v, f);
    put_u32(v, natt);
    return v;
}

static bool rejected(const std::string &path, const std::vector<unsigned char> &bytes)
{
    FILE *fp = fopen(path.c_str(), "wb");
    fwrite(bytes.data(), 1, bytes.size(), fp);
    fclose(fp);
    try {
        LumaRawStreamReader r;
        r.open(path.c_str());
    } catch (const LumaException &) {
        return true;
    }
    return false;
}

int main(int argc, char **argv)
{
    const std::string path = std::string(argc > 1 ? argv[1] : "/tmp") + "/crafted.lhs";
    int bad = 0;
    bad += !rejected(path, header(65536, 65536, 2, 0));               // 25 GB of planes announced, no data
    bad += !rejected(path, header(65534, 65534, 3, 0));
    bad += !rejected(path, header(3840, 2160, 2, 0));                  // plausible size, but not one frame of data
    bad += !rejected(path, header(64, 32, 7, 0));                      // unknown profile
    bad += !rejected(path, header(63, 32, 2, 0));                      // odd width
    bad += !rejected(path, header(64, 32, 2, 5));                      // attachment table missing
    {
        std::vector<unsigned char> v = header(64, 32, 2, 1);           // attachment announcing 64 MB + 1
        put_u32(v, 430);
        put_u32(v, 0);
        put_u32(v, (64u << 20) + 1);
        bad += !rejected(path, v);
    }
    {
        std::vector<unsigned char> v = header(64, 32, 2, 0);           // one whole frame (4:2:0 16-bit = 6144 bytes): accepted
        v.resize(v.size() + 64 * 32 * 2 + 2 * 32 * 16 * 2, 0);
        if (rejected(path, v))
            bad += 100;
        v.resize(v.size() - 1);                                        // one byte short: rejected
        bad += !rejected(path, v);
    }
    printf(bad ? "FAILED %d\n" : "ok\n", bad);
    return bad ? 1 : 0;
}
