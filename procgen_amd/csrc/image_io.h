// image_io.h -- host-side asset decoding for the HIP libenv (product code, no Qt, no libpng).
//
// Replaces the reference's QImage(path).convertToFormat(fmt) (reference src/resources.cpp:19-28):
// sprites are converted to premultiplied ARGB32 (QImage::Format_ARGB32_Premultiplied), backgrounds to
// RGB32 (alpha forced to 0xFF).  Pixel word layout is Qt's: 0xAARRGGBB.
//
// Also defines the ".atlas" pack: a zlib-compressed cache of already-decoded images keyed by their
// path relative to resource_root, so a box without the PNG tree (the GPU box) can still run.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace pgamd {

enum ImageFormat : uint32_t {
    IMG_ARGB32_PM = 0,  // sprites   (reference src/resources.cpp:815)
    IMG_RGB32 = 1,      // backgrounds (reference src/resources.cpp:945)
};

struct Image {
    int w = 0, h = 0;
    ImageFormat format = IMG_ARGB32_PM;
    std::vector<uint32_t> px;  // row-major, 0xAARRGGBB
};

// Decode one PNG file.  Supports what the asset tree uses: non-interlaced, 8-bit RGB / RGBA and
// 1/2/4/8-bit palette images with optional tRNS; gAMA/sRGB/iCCP are ignored, as Qt ignores them
// when no display gamma is requested.  Returns false with a message in *err on failure.
bool decode_png(const std::string &path, ImageFormat format, Image *out, std::string *err);

// Qt's qPremultiply() (qrgb.h) for one 0xAARRGGBB word.
inline uint32_t premultiply_argb(uint32_t x) {
    const uint32_t a = x >> 24;
    uint32_t t = (x & 0xff00ffu) * a;
    t = (t + ((t >> 8) & 0xff00ffu) + 0x800080u) >> 8;
    t &= 0xff00ffu;
    x = ((x >> 8) & 0xffu) * a;
    x = (x + ((x >> 8) & 0xffu) + 0x80u);
    x &= 0xff00u;
    return x | t | (a << 24);
}

// ---- .atlas pack --------------------------------------------------------------------------------
// file  := magic "PGATLAS1" u32 count { entry }
// entry := u32 name_len, name bytes, u32 w, u32 h, u32 format, u32 zlen, zlib(deflate) of w*h*4 bytes
struct AtlasPack {
    std::map<std::string, Image> images;
    bool load(const std::string &path, std::string *err);
    bool save(const std::string &path, std::string *err) const;
};

uint32_t crc32_bytes(const void *data, size_t n);

}  // namespace pgamd
