// bgzf.cpp -- BGZF framing of the BAM byte stream (SAM/BAM specification section 4.1; the reference goes through htslib's
// bgzf_write, source/BAMoutput.cpp:60-84).  Every block is an independent gzip member of at most 0xff00 input bytes, so the
// post-map threads compress their own part of a batch and the blocks are simply concatenated in read order.
#include "host.h"
#include <zlib.h>
#include <cstring>

namespace staramd {

static const size_t BGZF_INPUT_MAX = 0xff00;

static bool oneBlock(const uint8_t *in, size_t n, int level, std::string &out) {
    uint8_t buf[0x10000 + 64];
    z_stream zs; memset(&zs, 0, sizeof(zs));
    if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
    zs.next_in = const_cast<uint8_t *>(in); zs.avail_in = (uInt)n;
    zs.next_out = buf + 18; zs.avail_out = sizeof(buf) - 18 - 8;
    int rc = deflate(&zs, Z_FINISH);
    if (rc != Z_STREAM_END) { deflateEnd(&zs); return false; }
    size_t clen = zs.total_out;
    deflateEnd(&zs);
    size_t total = 18 + clen + 8;
    if (total > 0x10000) return false;
    static const uint8_t hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
    memcpy(buf, hdr, 16);
    buf[16] = (uint8_t)((total - 1) & 0xff); buf[17] = (uint8_t)((total - 1) >> 8);
    uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), in, (uInt)n), isize = (uint32_t)n;
    memcpy(buf + 18 + clen, &crc, 4); memcpy(buf + 18 + clen + 4, &isize, 4);
    out.append((const char *)buf, total);
    return true;
}

// compress `raw` into BGZF blocks appended to `out`; returns false on a zlib failure
bool bgzfCompress(const std::string &raw, int level, std::string &out) {
    for (size_t off = 0; off < raw.size(); off += BGZF_INPUT_MAX) {
        size_t n = std::min(BGZF_INPUT_MAX, raw.size() - off);
        // level 0 output can exceed the block limit for a full block: fall back to two halves
        if (!oneBlock((const uint8_t *)raw.data() + off, n, level, out)) {
            size_t h = n / 2;
            if (!oneBlock((const uint8_t *)raw.data() + off, h, level, out) || !oneBlock((const uint8_t *)raw.data() + off + h, n - h, level, out)) return false;
        }
    }
    return true;
}

// the 28-byte end-of-file marker block
void bgzfEof(std::string &out) {
    static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    out.append((const char *)eof, 28);
}

} // namespace staramd
