// yolov2_weight_gen -- offline tool: natural-order conv weights ([N][C][K][K] per layer, layers
// concatenated) -> the accelerator's tiled stream weights_reorg[_int16].bin.
//
// Same role, flags and on-disk formats as the reference tool
// (src/models/yolov2/yolov2_weight_gen.cpp:34-68,137-276): per conv layer, blocks of up to 32 output
// x 4 input channels, each block stored tap-major [k*k][TM_MIN][TN_MIN] with no padding of partial
// tiles; int16 files carry one pad element after every odd-length layer on both sides
// (hls/models/yolov2/yolo2_model.cpp:198-224).  Output is checked byte-for-byte against the
// reference tool in tests/test_host_side.py.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "y2_host.hpp"

namespace {
constexpr int kTm = 32, kTn = 4;  // hls/core/params.hpp

template <typename T>
std::vector<T> read_all(const std::string &path)
{
    FILE *fp = std::fopen(path.c_str(), "rb");
    if (!fp) throw std::runtime_error("Couldn't open file: " + path);
    std::fseek(fp, 0, SEEK_END);
    const long sz = std::ftell(fp);
    std::fseek(fp, 0, SEEK_SET);
    if (sz <= 0 || sz % (long)sizeof(T) != 0) { std::fclose(fp); throw std::runtime_error("Invalid weight file size: " + path); }
    std::vector<T> buf((size_t)sz / sizeof(T));
    const size_t rd = std::fread(buf.data(), sizeof(T), buf.size(), fp);
    std::fclose(fp);
    if (rd != buf.size()) throw std::runtime_error("Failed to read weights: " + path);
    return buf;
}

template <typename T>
void reorg_layer(const T *w, std::vector<T> &out, int C, int N, int K)
{
    const int KK = K * K;
    for (int m0 = 0; m0 < N; m0 += kTm) {
        const int tm_min = std::min(kTm, N - m0);
        for (int n0 = 0; n0 < C; n0 += kTn) {
            const int tn_min = std::min(kTn, C - n0);
            for (int tk = 0; tk < KK; ++tk)
                for (int tm = 0; tm < tm_min; ++tm)
                    for (int tn = 0; tn < tn_min; ++tn)
                        out.push_back(w[((size_t)(m0 + tm) * C + (n0 + tn)) * KK + tk]);
        }
    }
}

template <typename T>
int run(const y2h::Network &net, const std::string &in, const std::string &out_path, bool layer_pad)
{
    const std::vector<T> w = read_all<T>(in);
    std::vector<T> out;
    size_t off = 0;
    for (const y2h::Layer &l : net.layers) {
        if (l.type != y2h::CONV) continue;
        const size_t len = (size_t)l.n * l.c * l.size * l.size;
        if (off + len > w.size()) throw std::runtime_error("weights file too small for the cfg");
        reorg_layer(w.data() + off, out, l.c, l.n, l.size);
        off += len;
        if (layer_pad && (len & 1)) { off += 1; out.push_back(T(0)); }
    }
    FILE *fp = std::fopen(out_path.c_str(), "wb");
    if (!fp) throw std::runtime_error("Couldn't open file for write: " + out_path);
    const size_t wr = std::fwrite(out.data(), sizeof(T), out.size(), fp);
    std::fclose(fp);
    if (wr != out.size()) throw std::runtime_error("Failed to write weights: " + out_path);
    std::printf("wrote %zu elements to %s\n", out.size(), out_path.c_str());
    return 0;
}
}  // namespace

int main(int argc, char **argv)
{
    std::string cfg = "config/yolov2.cfg", in, out, prec = "fp32";
    for (int i = 1; i < argc; ++i) {
        const std::string a(argv[i]);
        if ((a == "--cfg" || a == "-c") && i + 1 < argc) cfg = argv[++i];
        else if ((a == "--weights" || a == "-w") && i + 1 < argc) in = argv[++i];
        else if ((a == "--out" || a == "-o") && i + 1 < argc) out = argv[++i];
        else if ((a == "--precision" || a == "-p") && i + 1 < argc) prec = argv[++i];
        else if (a == "--int16") prec = "int16";
        else if (a == "--fp32") prec = "fp32";
        else if (a == "--help" || a == "-h") {
            std::printf("Usage: %s [--cfg <cfg>] [--weights <weights.bin>] [--out <weights_reorg.bin>] [--precision fp32|int16]\n", argv[0]);
            return 0;
        }
    }
    try {
        const bool i16 = prec == "int16" || prec == "i16" || prec == "fixed";
        if (!i16 && prec != "fp32" && prec != "float" && prec != "f32") throw std::runtime_error("Unknown precision: " + prec);
        if (in.empty()) in = i16 ? "weights/weight_int16.bin" : "weights/weights.bin";
        if (out.empty()) out = i16 ? "weights/weights_reorg_int16.bin" : "weights/weights_reorg.bin";
        const y2h::Network net = y2h::parse_cfg(cfg);
        return i16 ? run<int16_t>(net, in, out, true) : run<float>(net, in, out, false);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "Fatal error: %s\n", e.what());
        return 1;
    }
}
