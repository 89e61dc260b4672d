// Plane buffers and the raw plane stream (the stand-in for the out-of-scope VP9 + Matroska stages).
#include "../../../include/luma/luma_planes.h"

#include <cstring>
#include <exception>

#include "../../../include/luma/luma_exception.h"

void LumaPlaneBuffer::allocate(unsigned int w, unsigned int h, int profile, unsigned int align)
{
    const bool sub = (profile == 0 || profile == 2);
    const bool hbd = profile > 1;
    m_img.d_w = w;
    m_img.d_h = h;
    m_img.x_chroma_shift = m_img.y_chroma_shift = sub ? 1 : 0;
    m_img.highBitDepth = hbd;
    const unsigned int aw = (w + align - 1) / align * align;
    const int s0 = (int)(aw * (hbd ? 2u : 1u));
    m_img.stride[0] = s0;
    m_img.stride[1] = m_img.stride[2] = sub ? s0 / 2 : s0;
    size_t off[3], total = 0;
    for (int p = 0; p < 3; p++) {
        off[p] = total;
        total += ((size_t)m_img.planeHeight(p) * m_img.stride[p] + 63) & ~(size_t)63;
    }
    m_store.assign(total + 64, 0);
    unsigned char *base = m_store.data();
    base += (64 - ((uintptr_t)base & 63)) & 63;
    for (int p = 0; p < 3; p++)
        m_img.planes[p] = base + off[p];
}

// ---------------------------------------------------------------------------------------- writer

static void put_u32(FILE *f, uint32_t v) { fwrite(&v, 4, 1, f); }
static bool get_u32(FILE *f, uint32_t &v) { return fread(&v, 4, 1, f) == 1; }

void LumaRawStreamWriter::open(const char *file, unsigned int w, unsigned int h, int profile, float fps)
{
    close();
    m_f = fopen(file, "wb");
    if (!m_f)
        throw LumaException(std::string("Cannot open '") + file + "' for writing");
    m_w = w;
    m_h = h;
    m_profile = profile;
    m_fps = fps;
    m_headerDone = false;
    m_att.clear();
}

void LumaRawStreamWriter::addAttachment(unsigned int id, const void *data, size_t size, const char *description)
{
    LumaAttachment a;
    a.id = id;
    a.description = description ? description : "";
    a.data.assign((const unsigned char *)data, (const unsigned char *)data + size);
    m_att.push_back(a);
}

void LumaRawStreamWriter::writeAttachments()
{
    if (!m_f || m_headerDone)
        return;
    fwrite("LHIPSTR1", 1, 8, m_f);
    put_u32(m_f, m_w);
    put_u32(m_f, m_h);
    put_u32(m_f, (uint32_t)m_profile);
    fwrite(&m_fps, 4, 1, m_f);
    put_u32(m_f, (uint32_t)m_att.size());
    for (size_t i = 0; i < m_att.size(); i++) {
        put_u32(m_f, m_att[i].id);
        put_u32(m_f, (uint32_t)m_att[i].description.size());
        fwrite(m_att[i].description.data(), 1, m_att[i].description.size(), m_f);
        put_u32(m_f, (uint32_t)m_att[i].data.size());
        fwrite(m_att[i].data.data(), 1, m_att[i].data.size(), m_f);
    }
    m_headerDone = true;
}

bool LumaRawStreamWriter::addFrame(const LumaPlanes &img)
{
    if (!m_f)
        return false;
    writeAttachments();
    for (int p = 0; p < 3; p++) {
        const size_t rb = (size_t)img.planeWidth(p) * img.bytesPerSample();
        for (unsigned int y = 0; y < img.planeHeight(p); y++)
            if (fwrite(img.planes[p] + (size_t)y * img.stride[p], 1, rb, m_f) != rb)
                return false;
    }
    return true;
}

void LumaRawStreamWriter::close()
{
    if (m_f) {
        writeAttachments();
        fclose(m_f);
        m_f = NULL;
    }
}

// ---------------------------------------------------------------------------------------- reader

LumaRawStreamReader::~LumaRawStreamReader()
{
    if (m_f)
        fclose(m_f);
}

void LumaRawStreamReader::open(const char *file)
{
    if (m_f)
        fclose(m_f);
    m_att.clear();
    m_f = fopen(file, "rb");
    if (!m_f)
        throw LumaException(std::string("Cannot open '") + file + "' for reading");
    char magic[8];
    uint32_t prof = 0, natt = 0;
    if (fread(magic, 1, 8, m_f) != 8 || memcmp(magic, "LHIPSTR1", 8) != 0 || !get_u32(m_f, m_w) || !get_u32(m_f, m_h) ||
        !get_u32(m_f, prof) || fread(&m_fps, 4, 1, m_f) != 1 || !get_u32(m_f, natt))
        throw LumaException(std::string("'") + file + "' is not a Luma HIP plane stream");
    m_profile = (int)prof;
    if (m_w == 0 || m_h == 0 || (m_w & 1) || (m_h & 1) || m_w > 65536 || m_h > 65536 || prof > 3 || natt > 1024)
        throw LumaException(std::string("'") + file + "' has an implausible plane stream header");
    for (uint32_t i = 0; i < natt; i++) {
        LumaAttachment a;
        uint32_t dl = 0, sz = 0;
        if (!get_u32(m_f, a.id) || !get_u32(m_f, dl) || dl > 4096)
            throw LumaException("truncated attachment table");
        a.description.resize(dl);
        if (dl && fread(&a.description[0], 1, dl, m_f) != dl)
            throw LumaException("truncated attachment table");
        if (!get_u32(m_f, sz) || sz > (64u << 20))
            throw LumaException("truncated attachment table");
        a.data.resize(sz);
        if (sz && fread(a.data.data(), 1, sz, m_f) != sz)
            throw LumaException("truncated attachment table");
        m_att.push_back(a);
    }
    m_dataStart = ftell(m_f);
    // A corrupt header must not make us allocate gigabytes (65536 x 65536 would be ~25 GB of planes plus a 51 GB float
    // frame in the decoder): the file has to hold at least one whole frame of the announced geometry.
    {
        const bool sub = (m_profile == 0 || m_profile == 2);
        const size_t bps = m_profile > 1 ? 2 : 1;
        const size_t cw = sub ? (m_w + 1) / 2 : m_w, chh = sub ? (m_h + 1) / 2 : m_h;
        const size_t need = ((size_t)m_w * m_h + 2 * cw * chh) * bps;
        long end = -1;
        if (fseek(m_f, 0, SEEK_END) == 0)
            end = ftell(m_f);
        if (end < 0 || fseek(m_f, m_dataStart, SEEK_SET) != 0)
            throw LumaException(std::string("cannot determine the size of '") + file + "'");
        if ((size_t)(end - m_dataStart) < need)
            throw LumaException(std::string("'") + file + "' is too short for one frame of the announced size");
    }
    try {
        m_buf.allocate(m_w, m_h, m_profile);
    } catch (const std::exception &e) {
        throw LumaException(std::string("cannot allocate the plane buffer: ") + e.what());
    }
    m_frameBytes = 0;
    const LumaPlanes &im = m_buf.image();
    for (int p = 0; p < 3; p++)
        m_frameBytes += (size_t)im.planeWidth(p) * im.bytesPerSample() * im.planeHeight(p);
}

bool LumaRawStreamReader::getAttachment(unsigned int index, unsigned char **buffer, unsigned int &id, unsigned int &size)
{
    if (index >= m_att.size())
        return false;
    *buffer = m_att[index].data.data();
    id = m_att[index].id;
    size = (unsigned int)m_att[index].data.size();
    return true;
}

bool LumaRawStreamReader::readFrame(const LumaPlanes **img)
{
    if (!m_f)
        return false;
    LumaPlanes &im = m_buf.image();
    for (int p = 0; p < 3; p++) {
        const size_t rb = (size_t)im.planeWidth(p) * im.bytesPerSample();
        for (unsigned int y = 0; y < im.planeHeight(p); y++)
            if (fread(im.planes[p] + (size_t)y * im.stride[p], 1, rb, m_f) != rb)
                return false;
    }
    *img = &im;
    return true;
}

float LumaRawStreamReader::getDuration()
{
    if (!m_f || !m_frameBytes || !(m_fps > 0.0f))
        return 0.0f;
    const long here = ftell(m_f);
    float d = 0.0f;
    if (fseek(m_f, 0, SEEK_END) == 0)
        d = (float)((size_t)(ftell(m_f) - m_dataStart) / m_frameBytes) / m_fps;
    fseek(m_f, here, SEEK_SET);
    return d;
}

bool LumaRawStreamReader::seekToFrame(unsigned int index)
{
    if (!m_f)
        return false;
    return fseek(m_f, m_dataStart + (long)((size_t)index * m_frameBytes), SEEK_SET) == 0;
}
