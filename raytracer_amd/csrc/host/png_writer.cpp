// writeImage: 8-bit RGB PNG encoder over zlib (no libpng).  Same signature, pixel packing
// (0xRRGGBB per int, row-major) and output format (8-bit RGB, non-interlaced) as the reference's
// /root/reference/src/pngUtil.c:4-29.  PNG is lossless, so a decoder returns exactly the bytes packed
// here; the compressed byte stream itself depends on the zlib build and is not part of any contract.
#include <zlib.h>

#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>

#include "rt_host.hpp"

namespace {

void be32(std::vector<unsigned char>& v, uint32_t x) {
    v.push_back((unsigned char)(x >> 24)); v.push_back((unsigned char)(x >> 16));
    v.push_back((unsigned char)(x >> 8)); v.push_back((unsigned char)x);
}

bool chunk(FILE* f, const char type[4], const std::vector<unsigned char>& data) {
    std::vector<unsigned char> hdr;
    be32(hdr, (uint32_t)data.size());
    uLong crc = crc32(0L, Z_NULL, 0);
    crc = crc32(crc, (const Bytef*)type, 4);
    if (!data.empty()) crc = crc32(crc, data.data(), (uInt)data.size());
    std::vector<unsigned char> tail;
    be32(tail, (uint32_t)crc);
    return fwrite(hdr.data(), 1, 4, f) == 4 && fwrite(type, 1, 4, f) == 4 &&
           (data.empty() || fwrite(data.data(), 1, data.size(), f) == data.size()) && fwrite(tail.data(), 1, 4, f) == 4;
}

}  // namespace

extern "C" int writeImage(char* filename, int width, int height, int* image) {
    if (!filename || !image || width <= 0 || height <= 0) return -1;
    std::vector<unsigned char> raw((size_t)height * (1 + (size_t)width * 3));
    size_t o = 0;
    for (int y = 0; y < height; y++) {
        raw[o++] = 0;   // filter type None
        for (int x = 0; x < width; x++) {
            const int p = image[(size_t)y * width + x];
            raw[o++] = (unsigned char)((p >> 16) & 0xff);
            raw[o++] = (unsigned char)((p >> 8) & 0xff);
            raw[o++] = (unsigned char)(p & 0xff);
        }
    }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<unsigned char> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return -1;
    comp.resize(clen);
    FILE* f = fopen(filename, "wb");
    if (!f) return -1;
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    bool ok = fwrite(sig, 1, 8, f) == 8;
    std::vector<unsigned char> ihdr;
    be32(ihdr, (uint32_t)width); be32(ihdr, (uint32_t)height);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);   // 8-bit, RGB
    ok = ok && chunk(f, "IHDR", ihdr) && chunk(f, "IDAT", comp) && chunk(f, "IEND", {});
    ok = (fclose(f) == 0) && ok;
    return ok ? 0 : -1;
}
