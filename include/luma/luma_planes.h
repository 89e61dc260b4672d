// luma_planes.h -- the seam between the hot path and the unchanged downstream stages (libvpx VP9 +
// Matroska in the reference).
//
// LumaPlanes carries exactly the vpx_image_t fields the reference's hot path touches
// (src/luma_encoder.cpp:260-273, src/luma_decoder.cpp:151-162): planes[3], stride[3], d_w, d_h,
// x/y_chroma_shift and the high-bit-depth flag.  A libvpx-backed sink fills one from its vpx_image_t
// (field for field) and hands the SAME memory to vpx_codec_encode afterwards; this repo ships a
// self-describing raw plane stream (LumaRawStreamWriter / Reader) so that encode -> decode round trips
// run without libvpx / libmatroska, carrying the same metadata attachments 430..436 as the reference
// (src/luma_encoder.cpp:78-104).
#ifndef LUMA_HIP_PLANES_H
#define LUMA_HIP_PLANES_H

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

struct LumaPlanes {
    unsigned char *planes[3];
    int stride[3];
    unsigned int d_w, d_h;
    unsigned int x_chroma_shift, y_chroma_shift;
    bool highBitDepth;  // VPX_IMG_FMT_HIGHBITDEPTH: 16-bit little-endian samples

    unsigned int planeWidth(int p) const { return (p > 0 && x_chroma_shift > 0) ? (d_w + 1) >> x_chroma_shift : d_w; }
    unsigned int planeHeight(int p) const { return (p > 0 && y_chroma_shift > 0) ? (d_h + 1) >> y_chroma_shift : d_h; }
    unsigned int bytesPerSample() const { return highBitDepth ? 2 : 1; }
    // the VP9 profile that selects this layout (src/luma_decoder.cpp:157)
    int profile() const { return x_chroma_shift ? (highBitDepth ? 2 : 0) : (highBitDepth ? 3 : 1); }
};

// Owns plane memory laid out like vpx_img_alloc(fmt(profile), w, h, 32) does (stride = 32-aligned width x
// bytes per sample, chroma stride = luma stride >> x_chroma_shift).
class LumaPlaneBuffer {
public:
    LumaPlaneBuffer() {}
    void allocate(unsigned int w, unsigned int h, int profile, unsigned int align = 32);
    LumaPlanes &image() { return m_img; }
    const LumaPlanes &image() const { return m_img; }
    bool allocated() const { return !m_store.empty(); }

private:
    LumaPlanes m_img{};
    std::vector<unsigned char> m_store;
};

struct LumaAttachment {
    unsigned int id;
    std::string description;
    std::vector<unsigned char> data;
};

// Downstream of the encoder hot path: receives the metadata attachments and every filled frame.
class LumaPlaneSink {
public:
    virtual ~LumaPlaneSink() {}
    // container-level luminance range (the reference's MkvInterface::openWrite(file, w, h, ma, mi) records it as HDR10
    // mastering metadata, src/mkv_interface.cpp:155-180 there); called before open(); sinks without such metadata ignore it
    virtual void setLuminanceRange(float maxLum, float minLum)
    {
        (void)maxLum;
        (void)minLum;
    }
    virtual void open(const char *file, unsigned int w, unsigned int h, int profile, float fps) = 0;
    virtual void addAttachment(unsigned int id, const void *data, size_t size, const char *description) = 0;
    virtual void writeAttachments() = 0;
    virtual bool addFrame(const LumaPlanes &img) = 0;
    virtual void close() = 0;
};

// Upstream of the decoder hot path.
class LumaPlaneSource {
public:
    virtual ~LumaPlaneSource() {}
    virtual void open(const char *file) = 0;
    // index-th attachment; false when there is none (same calling convention as MkvInterface::getAttachment)
    virtual bool getAttachment(unsigned int index, unsigned char **buffer, unsigned int &id, unsigned int &size) = 0;
    // next frame's planes (valid until the next call); false at end of stream
    virtual bool readFrame(const LumaPlanes **img) = 0;
    virtual bool seekToFrame(unsigned int index) = 0;
    // what the reference's player asks its MkvInterface through LumaDecoder::getReader() (lumaplay.cpp:200,443):
    // stream duration and the duration of one frame, in seconds; 0 when unknown
    virtual float getDuration() { return 0.0f; }
    virtual float getFrameDuration() { return 0.0f; }
};

// Raw plane stream: "LHIPSTR1" | w h profile fps | n_attachments { id, desc, bytes } | frames (tight rows).
class LumaRawStreamWriter : public LumaPlaneSink {
public:
    LumaRawStreamWriter() : m_f(NULL), m_w(0), m_h(0), m_profile(2), m_fps(25.0f), m_headerDone(false) {}
    ~LumaRawStreamWriter() { close(); }
    void open(const char *file, unsigned int w, unsigned int h, int profile, float fps);
    void addAttachment(unsigned int id, const void *data, size_t size, const char *description);
    void writeAttachments();
    bool addFrame(const LumaPlanes &img);
    void close();

private:
    FILE *m_f;
    unsigned int m_w, m_h;
    int m_profile;
    float m_fps;
    bool m_headerDone;
    std::vector<LumaAttachment> m_att;
};

class LumaRawStreamReader : public LumaPlaneSource {
public:
    LumaRawStreamReader() : m_f(NULL), m_w(0), m_h(0), m_profile(2), m_fps(25.0f), m_dataStart(0), m_frameBytes(0) {}
    ~LumaRawStreamReader();
    void open(const char *file);
    bool getAttachment(unsigned int index, unsigned char **buffer, unsigned int &id, unsigned int &size);
    bool readFrame(const LumaPlanes **img);
    bool seekToFrame(unsigned int index);
    unsigned int width() const { return m_w; }
    unsigned int height() const { return m_h; }
    float fps() const { return m_fps; }
    float getDuration();
    float getFrameDuration() { return m_fps > 0.0f ? 1.0f / m_fps : 0.0f; }

private:
    FILE *m_f;
    unsigned int m_w, m_h;
    int m_profile;
    float m_fps;
    long m_dataStart;
    size_t m_frameBytes;
    std::vector<LumaAttachment> m_att;
    LumaPlaneBuffer m_buf;
};

#endif
