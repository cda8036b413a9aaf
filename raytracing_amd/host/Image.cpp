// Image.cpp — load_hdr (Image.cpp:6-30 of the reference) without the vendored decoder.
//
// The reference hands the file to stbi_loadf and keeps the floats; the only HDR format its scenes use is Radiance
// RGBE (main.cpp:1028 loads "noon_grass.hdr").  This is a self-contained reader of that format — header, the
// "-Y h +X w" resolution line, flat or new-style run-length-encoded scanlines — with the decoder's conversion
// rgb = mantissa * 2^(e - 136) (no half-step offset), so a file yields the floats the reference would hold.
// Errors follow the reference's convention: a message on stdout and nullptr.
#include "Scene.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>

namespace rt {
namespace {

bool read_line(std::istream& in, std::string& line) {
    line.clear();
    char c;
    while (in.get(c)) {
        if (c == '\n') return true;
        line.push_back(c);
        if (line.size() > 1024) return false;
    }
    return !line.empty();
}

void rgbe_to_float(const unsigned char* p, float* out) {
    if (p[3] == 0) {
        out[0] = out[1] = out[2] = 0.f;
        return;
    }
    const float f = std::ldexp(1.0f, (int)p[3] - (128 + 8));
    out[0] = p[0] * f;
    out[1] = p[1] * f;
    out[2] = p[2] * f;
}

} // namespace

HDRImage* load_hdr(const std::string& filename) {
    auto fail = [&](const char* why) -> HDRImage* {
        std::cout << "Couldn't load image: " << filename << " (" << why << ")" << std::endl;
        return nullptr;
    };
    std::ifstream in(filename, std::ios::binary);
    if (!in) return fail("cannot open");
    std::string line;
    if (!read_line(in, line) || (line != "#?RADIANCE" && line != "#?RGBE")) return fail("not a Radiance file");
    bool format_ok = false;
    for (;;) {
        if (!read_line(in, line)) return fail("truncated header");
        if (line.empty()) break;
        if (line == "FORMAT=32-bit_rle_rgbe") format_ok = true;
    }
    if (!format_ok) return fail("unsupported FORMAT");
    if (!read_line(in, line)) return fail("no resolution line");
    int w = 0, h = 0;
    if (std::sscanf(line.c_str(), "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0 || w > (1 << 16) || h > (1 << 16))
        return fail("unsupported resolution line");

    HDRImage* img = new HDRImage;
    img->width = w;
    img->height = h;
    img->internalFormat = Image::Format::RGB_32f;
    img->imageData.resize((size_t)w * h * 3 * sizeof(float));
    float* out = (float*)img->imageData.data();
    std::vector<unsigned char> scan((size_t)w * 4);
    auto bad = [&](const char* why) -> HDRImage* {
        delete img;
        return fail(why);
    };
    for (int y = 0; y < h; y++) {
        unsigned char head[4];
        if (!in.read((char*)head, 4)) return bad("truncated pixel data");
        const bool rle = w >= 8 && w < 32768 && head[0] == 2 && head[1] == 2 && !(head[2] & 0x80);
        if (!rle) { // flat: w RGBE quadruples, the first already read
            std::memcpy(scan.data(), head, 4);
            if (w > 1 && !in.read((char*)scan.data() + 4, (std::streamsize)(w - 1) * 4)) return bad("truncated pixel data");
            for (int x = 0; x < w; x++) rgbe_to_float(&scan[(size_t)x * 4], out + ((size_t)y * w + x) * 3);
            continue;
        }
        if (((int)head[2] << 8 | head[3]) != w) return bad("scanline width mismatch");
        // four runs of w bytes, one per component
        std::vector<unsigned char> comp((size_t)w * 4);
        for (int c = 0; c < 4; c++) {
            int x = 0;
            while (x < w) {
                unsigned char count, value;
                if (!in.read((char*)&count, 1)) return bad("truncated run");
                if (count > 128) {
                    count -= 128;
                    if (count == 0 || x + count > w || !in.read((char*)&value, 1)) return bad("bad run");
                    std::memset(&comp[(size_t)c * w + x], value, count);
                } else {
                    if (count == 0 || x + count > w || !in.read((char*)&comp[(size_t)c * w + x], count)) return bad("bad run");
                }
                x += count;
            }
        }
        for (int x = 0; x < w; x++) {
            const unsigned char px[4] = {comp[x], comp[(size_t)w + x], comp[(size_t)2 * w + x], comp[(size_t)3 * w + x]};
            rgbe_to_float(px, out + ((size_t)y * w + x) * 3);
        }
    }
    return img;
}

} // namespace rt
