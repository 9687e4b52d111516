// CPU-only sanitizer harness for host/y2_codec.cpp (GPU sanitizers are not available; this is host code anyway): decodes every file given on
// the command line, then mutated copies of it (bit flips, truncations, overwritten bytes, runs of 0xff) - the decoders must return an error or
// an image, never touch memory they do not own, never inflate without bound.  Built and run by tests/test_host_codec.py with
//   g++ -std=c++17 -O1 -g -fwrapv -fsanitize=address,undefined -fno-sanitize-recover=undefined tools/fuzz_codec.cpp host/y2_codec.cpp host/y2_host.cpp
// usage: fuzz_codec <mutations per file> <files...>
#include "../host/y2_host.hpp"
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <random>
#include <vector>
int main(int argc, char **argv)
{
    long ok = 0, bad = 0;
    std::mt19937 rng(12345);
    const int rounds = argc > 1 ? std::atoi(argv[1]) : 200;
    for (int f = 2; f < argc; ++f) {
        std::ifstream in(argv[f], std::ios::binary);
        std::vector<unsigned char> data((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        if (data.empty()) continue;
        for (int r = 0; r <= rounds; ++r) {
            std::vector<unsigned char> d = data;
            if (r > 0) {
                const int kind = rng() % 4;
                if (kind == 0) { const int n = 1 + rng() % 8; for (int k = 0; k < n; ++k) d[rng() % d.size()] ^= (unsigned char)(1u << (rng() % 8)); }
                else if (kind == 1) d.resize(rng() % d.size());
                else if (kind == 2) { const size_t p = rng() % d.size(); d[p] = (unsigned char)rng(); if (p + 1 < d.size()) d[p + 1] = (unsigned char)rng(); }
                else { const size_t p = rng() % d.size(), n = std::min<size_t>(d.size() - p, 1 + rng() % 64); for (size_t k = 0; k < n; ++k) d[p + k] = 0xff; }
            }
            try {
                y2h::ImageU8 im = y2h::decode_image(d.data(), d.size(), "fuzz");
                if (im.w > 0 && im.h > 0 && (size_t)im.w * im.h * 3 == im.rgb.size()) ++ok; else ++bad;
            } catch (const std::exception &) { ++bad; }
        }
    }
    std::printf("decoded %ld, rejected %ld\n", ok, bad);
    return 0;
}
