// image_io.cpp -- PNG decoder (zlib inflate + PNG unfiltering written here) and the .atlas pack.
#include "image_io.h"

#include <zlib.h>

#include <cstdio>
#include <cstring>

namespace pgamd {

namespace {

uint32_t be32(const uint8_t *p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }

bool read_file(const std::string &path, std::vector<uint8_t> *out) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    out->resize(n > 0 ? n : 0);
    size_t got = n > 0 ? fread(out->data(), 1, n, f) : 0;
    fclose(f);
    return got == size_t(n);
}

int paeth(int a, int b, int c) {
    int p = a + b - c;
    int pa = p > a ? p - a : a - p;
    int pb = p > b ? p - b : b - p;
    int pc = p > c ? p - c : c - p;
    if (pa <= pb && pa <= pc) return a;
    if (pb <= pc) return b;
    return c;
}

}  // namespace

uint32_t crc32_bytes(const void *data, size_t n) { return (uint32_t)crc32(0L, (const Bytef *)data, (uInt)n); }

bool decode_png(const std::string &path, ImageFormat format, Image *out, std::string *err) {
    std::vector<uint8_t> d;
    if (!read_file(path, &d) || d.size() < 33 || memcmp(d.data(), "\x89PNG\r\n\x1a\n", 8) != 0) {
        if (err) *err = "cannot read PNG " + path;
        return false;
    }
    uint32_t w = 0, h = 0;
    int bit_depth = 0, color_type = 0, interlace = 0;
    std::vector<uint8_t> idat;
    uint32_t palette[256];
    int npal = 0;
    for (int i = 0; i < 256; i++) palette[i] = 0xff000000u;
    bool have_key = false;
    uint16_t key[3] = {0, 0, 0};
    size_t off = 8;
    while (off + 12 <= d.size()) {
        uint32_t len = be32(&d[off]);
        const uint8_t *type = &d[off + 4];
        const uint8_t *body = &d[off + 8];
        if (off + 12 + len > d.size()) break;
        if (!memcmp(type, "IHDR", 4)) {
            w = be32(body);
            h = be32(body + 4);
            bit_depth = body[8];
            color_type = body[9];
            interlace = body[12];
        } else if (!memcmp(type, "PLTE", 4)) {
            npal = len / 3;
            for (int i = 0; i < npal && i < 256; i++)
                palette[i] = (palette[i] & 0xff000000u) | (uint32_t(body[3 * i]) << 16) | (uint32_t(body[3 * i + 1]) << 8) | body[3 * i + 2];
        } else if (!memcmp(type, "tRNS", 4)) {
            if (color_type == 3) {
                for (uint32_t i = 0; i < len && i < 256; i++) palette[i] = (palette[i] & 0x00ffffffu) | (uint32_t(body[i]) << 24);
            } else if (color_type == 2 && len >= 6) {
                have_key = true;
                for (int i = 0; i < 3; i++) key[i] = uint16_t((body[2 * i] << 8) | body[2 * i + 1]);
            }
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!memcmp(type, "IEND", 4)) {
            break;
        }
        off += 12 + len;
    }
    int channels = color_type == 6 ? 4 : color_type == 2 ? 3 : color_type == 3 ? 1 : 0;
    bool ok_depth = (color_type == 3) ? (bit_depth == 1 || bit_depth == 2 || bit_depth == 4 || bit_depth == 8) : bit_depth == 8;
    if (!w || !h || !channels || !ok_depth || interlace) {
        if (err) *err = "unsupported PNG layout in " + path;
        return false;
    }
    size_t bpp_bits = size_t(channels) * bit_depth;
    size_t stride = (size_t(w) * bpp_bits + 7) / 8;
    size_t fb = bpp_bits >= 8 ? bpp_bits / 8 : 1;  // filter byte distance
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf rawlen = raw.size();
    if (uncompress(raw.data(), &rawlen, idat.data(), idat.size()) != Z_OK || rawlen != raw.size()) {
        if (err) *err = "bad IDAT stream in " + path;
        return false;
    }
    std::vector<uint8_t> prev(stride, 0), cur(stride);
    out->w = int(w);
    out->h = int(h);
    out->format = format;
    out->px.resize(size_t(w) * h);
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t *line = &raw[(stride + 1) * y];
        int ft = line[0];
        const uint8_t *s = line + 1;
        for (size_t i = 0; i < stride; i++) {
            int a = i >= fb ? cur[i - fb] : 0, b = prev[i], c = i >= fb ? prev[i - fb] : 0;
            int v = s[i];
            switch (ft) {
                case 0: break;
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) >> 1; break;
                case 4: v += paeth(a, b, c); break;
                default:
                    if (err) *err = "bad PNG filter in " + path;
                    return false;
            }
            cur[i] = uint8_t(v);
        }
        uint32_t *dst = &out->px[size_t(y) * w];
        for (uint32_t x = 0; x < w; x++) {
            uint32_t argb;
            if (color_type == 6) {
                const uint8_t *p = &cur[4 * x];
                argb = (uint32_t(p[3]) << 24) | (uint32_t(p[0]) << 16) | (uint32_t(p[1]) << 8) | p[2];
            } else if (color_type == 2) {
                const uint8_t *p = &cur[3 * x];
                argb = 0xff000000u | (uint32_t(p[0]) << 16) | (uint32_t(p[1]) << 8) | p[2];
                if (have_key && p[0] == key[0] && p[1] == key[1] && p[2] == key[2]) argb &= 0x00ffffffu;
            } else {
                int idx;
                if (bit_depth == 8) {
                    idx = cur[x];
                } else {
                    int per = 8 / bit_depth;
                    int sh = (per - 1 - int(x % per)) * bit_depth;
                    idx = (cur[x / per] >> sh) & ((1 << bit_depth) - 1);
                }
                argb = palette[idx];
            }
            if (format == IMG_RGB32)
                argb |= 0xff000000u;  // Qt ARGB32 -> RGB32 conversion only masks the alpha in
            else
                argb = premultiply_argb(argb);
            dst[x] = argb;
        }
        prev.swap(cur);
    }
    return true;
}

bool AtlasPack::save(const std::string &path, std::string *err) const {
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) {
        if (err) *err = "cannot write " + path;
        return false;
    }
    fwrite("PGATLAS1", 1, 8, f);
    uint32_t cnt = uint32_t(images.size());
    fwrite(&cnt, 4, 1, f);
    for (const auto &kv : images) {
        const Image &im = kv.second;
        uint32_t nl = uint32_t(kv.first.size());
        fwrite(&nl, 4, 1, f);
        fwrite(kv.first.data(), 1, nl, f);
        uint32_t hdr[3] = {uint32_t(im.w), uint32_t(im.h), uint32_t(im.format)};
        fwrite(hdr, 4, 3, f);
        uLongf zl = compressBound(im.px.size() * 4);
        std::vector<uint8_t> z(zl);
        compress2(z.data(), &zl, (const Bytef *)im.px.data(), im.px.size() * 4, 6);
        uint32_t zlen = uint32_t(zl);
        fwrite(&zlen, 4, 1, f);
        fwrite(z.data(), 1, zlen, f);
    }
    fclose(f);
    return true;
}

bool AtlasPack::load(const std::string &path, std::string *err) {
    std::vector<uint8_t> d;
    if (!read_file(path, &d) || d.size() < 12 || memcmp(d.data(), "PGATLAS1", 8) != 0) {
        if (err) *err = "cannot read atlas pack " + path;
        return false;
    }
    uint32_t cnt;
    memcpy(&cnt, &d[8], 4);
    size_t off = 12;
    for (uint32_t i = 0; i < cnt; i++) {
        uint32_t nl;
        memcpy(&nl, &d[off], 4);
        off += 4;
        std::string name((const char *)&d[off], nl);
        off += nl;
        uint32_t hdr[4];
        memcpy(hdr, &d[off], 16);
        off += 16;
        Image im;
        im.w = int(hdr[0]);
        im.h = int(hdr[1]);
        im.format = ImageFormat(hdr[2]);
        im.px.resize(size_t(im.w) * im.h);
        uLongf rl = im.px.size() * 4;
        if (off + hdr[3] > d.size() || uncompress((Bytef *)im.px.data(), &rl, &d[off], hdr[3]) != Z_OK || rl != im.px.size() * 4) {
            if (err) *err = "corrupt atlas pack " + path;
            return false;
        }
        off += hdr[3];
        images.emplace(std::move(name), std::move(im));
    }
    return true;
}

}  // namespace pgamd
