// yolov2_detect -- the repo's own detection CLI with `--backend hip`.
//
// Same role and flags as the reference CLI (src/models/yolov2/yolov2_main.cpp:83-132, 234-335):
// parse cfg, load names, load + letterbox the image, run the accelerator path, dump the region
// tensors, decode boxes, NMS, draw, save.  The accelerator path is the HIP library behind its C ABI
// (include/yolo2_hip.h); there is no CPU backend here -- `--backend hls|cpu` names the reference's
// own binaries and is rejected with a pointer to them.
//
// Differences that are deliberate: images are binary PPM/PGM (tools/img2ppm.py converts JPEG/PNG),
// the annotated result is written as PPM, boxes are also printed as text and (--json) JSON lines;
// `--batch N` runs the same frame N times through one batched call to show the batched entry.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/yolo2_hip.h"
#include "y2_host.hpp"

namespace {

struct AppConfig {
    std::string cfg_path = "config/yolov2.cfg";
    std::string names_path = "config/coco.names";
    std::string input_path = "examples/test_images/dog.ppm";
    std::string weights_dir = "weights";
    std::string output_prefix;
    float thresh = 0.25f;  // the reference's code default (its usage text says 0.5: yolov2_main.cpp:36,69)
    float nms = 0.45f;
    float hier_thresh = 0.5f;
    std::string backend = "hip";
    std::string precision = "int16";
    int batch = 1;
    int device = 0;
    bool json = false;
};

void print_usage(const char *prog)
{
    std::printf(
        "Usage: %s [options] [image.ppm]\n"
        "  --cfg <path>          Network cfg file (default: config/yolov2.cfg)\n"
        "  --names <path>        Class names file (default: config/coco.names)\n"
        "  --input <path>        Input image, binary PPM/PGM\n"
        "  --weights <dir>       Directory with weights_reorg_int16.bin, bias_int16.bin, *_Q.bin (default: weights)\n"
        "  --output <prefix>     Output file prefix without extension (default: results/<input>_prediction)\n"
        "  --thresh <float>      Confidence threshold (default: 0.25)\n"
        "  --nms <float>         NMS IoU threshold (default: 0.45)\n"
        "  --hier <float>        Hierarchical threshold (accepted, unused by region layers)\n"
        "  --backend <hip>       Backend selector (hip = MI355X library; hls/cpu live in the reference build)\n"
        "  --precision <int16|fp32>  int16 = the batched fixed-point path; fp32 = the exact fp32 pass (one frame)\n"
        "  --batch <n>           Frames per accelerator call (default 1)\n"
        "  --device <n>          HIP device (default 0)\n"
        "  --json                Also print detections as JSON lines\n",
        prog);
}

AppConfig parse_args(int argc, char **argv)
{
    AppConfig cfg;
    for (int i = 1; i < argc; ++i) {
        const std::string arg(argv[i]);
        auto need = [&](const char *) { return i + 1 < argc; };
        if (arg == "--help" || arg == "-h") { print_usage(argv[0]); std::exit(0); }
        else if (arg == "--cfg" && need("")) cfg.cfg_path = argv[++i];
        else if (arg == "--names" && need("")) cfg.names_path = argv[++i];
        else if (arg == "--input" && need("")) cfg.input_path = argv[++i];
        else if (arg == "--weights" && need("")) cfg.weights_dir = argv[++i];
        else if (arg == "--output" && need("")) cfg.output_prefix = argv[++i];
        else if (arg == "--thresh" && need("")) cfg.thresh = std::strtof(argv[++i], nullptr);
        else if (arg == "--nms" && need("")) cfg.nms = std::strtof(argv[++i], nullptr);
        else if (arg == "--hier" && need("")) cfg.hier_thresh = std::strtof(argv[++i], nullptr);
        else if (arg == "--batch" && need("")) cfg.batch = std::atoi(argv[++i]);
        else if (arg == "--device" && need("")) cfg.device = std::atoi(argv[++i]);
        else if (arg == "--json") cfg.json = true;
        else if (arg == "--backend" && need("")) {
            cfg.backend = argv[++i];
            if (cfg.backend != "hip") {
                std::fprintf(stderr, "Unsupported backend '%s'. This build provides 'hip' only; 'hls'/'cpu' are the reference's "
                             "own yolov2_detect.\n", cfg.backend.c_str());
                std::exit(1);
            }
        } else if (arg == "--precision" && need("")) {
            cfg.precision = argv[++i];
            if (cfg.precision == "float" || cfg.precision == "f32") cfg.precision = "fp32";
            if (cfg.precision == "i16" || cfg.precision == "fixed") cfg.precision = "int16";
            if (cfg.precision != "int16" && cfg.precision != "fp32") {
                std::fprintf(stderr, "Unsupported precision: %s (the hip backend runs int16 and fp32)\n", cfg.precision.c_str());
                std::exit(1);
            }
        } else if (arg.rfind("--", 0) == 0) {
            std::fprintf(stderr, "Unknown option: %s\n", arg.c_str());
            print_usage(argv[0]);
            std::exit(1);
        } else cfg.input_path = arg;
    }
    return cfg;
}

std::string default_output_prefix(const std::string &input_path)
{
    std::string base = std::filesystem::path(input_path).stem().string();
    return base + "_prediction";
}

void dump_floats(const char *path, const float *data, size_t n)
{
    FILE *fp = std::fopen(path, "w");
    if (!fp) { std::fprintf(stderr, "Warning: cannot open dump file %s\n", path); return; }
    for (size_t i = 0; i < n; ++i) std::fprintf(fp, "%.9g\n", data[i]);  // same format as yolo2_model.cpp:47
    std::fclose(fp);
    std::printf("Dumped %zu floats to %s\n", n, path);
}

// the parsed .cfg must be the network the batched entry implements
void check_topology(const y2h::Network &net)
{
    if ((int)net.layers.size() != yolo2_hip_num_layers()) throw std::runtime_error("cfg does not describe the 32-layer YOLOv2 network the hip backend implements");
    for (int i = 0; i < (int)net.layers.size(); ++i) {
        int d[9];
        yolo2_hip_layer_desc(i, d);
        const y2h::Layer &l = net.layers[i];
        bool ok = d[0] == (int)l.type;
        if (l.type == y2h::CONV) ok = ok && d[1] == l.c && d[2] == l.h && d[3] == l.w && d[4] == l.n && d[5] == l.size && d[6] == l.stride && d[7] == l.pad && d[8] == (int)l.leaky;
        if (l.type == y2h::MAXPOOL) ok = ok && d[1] == l.c && d[2] == l.h && d[3] == l.w && d[5] == l.size && d[6] == l.stride;
        if (l.type == y2h::REORG) ok = ok && d[1] == l.c && d[2] == l.h && d[6] == l.stride;
        if (!ok) throw std::runtime_error("cfg layer " + std::to_string(i) + " differs from the network the hip backend implements");
    }
}

void run_detector(AppConfig cfg)
{
    std::setbuf(stdout, nullptr);
    namespace fs = std::filesystem;
    if (cfg.output_prefix.empty()) cfg.output_prefix = default_output_prefix(cfg.input_path);
    {
        fs::path prefix(cfg.output_prefix);
        if (!prefix.has_parent_path()) { fs::create_directories("results"); prefix = fs::path("results") / prefix; }
        else fs::create_directories(prefix.parent_path());
        cfg.output_prefix = prefix.string();
    }
    std::printf("YOLOv2 Object Detection - Starting\n  cfg:    %s\n  names:  %s\n  input:  %s\n  precision: %s\n  backend: hip (device %d, batch %d)\n  output: %s[.ppm]\n",
                cfg.cfg_path.c_str(), cfg.names_path.c_str(), cfg.input_path.c_str(), cfg.precision.c_str(), cfg.device, cfg.batch,
                cfg.output_prefix.c_str());

    const y2h::Network net = y2h::parse_cfg(cfg.cfg_path);
    check_topology(net);
    const std::vector<std::string> names = y2h::load_names(cfg.names_path);
    y2h::Image im = y2h::load_pnm(cfg.input_path);
    std::printf("Input img: %s (w=%d, h=%d, c=%d)\n", cfg.input_path.c_str(), im.w, im.h, im.c);
    const y2h::Image sized = y2h::letterbox_image(im, net.w, net.h);

    const y2h::Layer &last = net.layers.back();
    std::vector<float> raw(YOLO2_REGION_ELEMS), proc(YOLO2_REGION_ELEMS);
    yolo2_hip_ctx *ctx = nullptr;
    if (yolo2_hip_create(cfg.device, &ctx) != YOLO2_SUCCESS) throw std::runtime_error(yolo2_hip_last_error());
    double elapsed = 0;
    int frames_run = cfg.batch;
    if (cfg.precision == "fp32") {
        // weights/weights_reorg.bin + weights/bias.bin, the files load_weights() reads at Precision::FP32 (yolo2_model.cpp:171-183)
        auto read_floats = [](const std::string &path, size_t want) {
            std::ifstream f(path, std::ios::binary | std::ios::ate);
            if (!f) throw std::runtime_error("Cannot open " + path);
            const size_t n = (size_t)f.tellg() / sizeof(float);
            if (n < want) throw std::runtime_error(path + " is too small (" + std::to_string(n) + " floats, need " + std::to_string(want) + ")");
            std::vector<float> v(n);
            f.seekg(0);
            f.read(reinterpret_cast<char *>(v.data()), (std::streamsize)(n * sizeof(float)));
            return v;
        };
        const std::vector<float> w = read_floats(cfg.weights_dir + "/weights_reorg.bin", (size_t)YOLO2_N_WEIGHTS);
        const std::vector<float> b = read_floats(cfg.weights_dir + "/bias.bin", (size_t)YOLO2_N_BIAS);
        if (yolo2_hip_load_weights_fp32(ctx, w.data(), w.size(), b.data(), b.size()) != YOLO2_SUCCESS) throw std::runtime_error(yolo2_hip_last_error());
        const auto t0 = std::chrono::high_resolution_clock::now();
        if (yolo2_hip_run_frame_fp32_host(ctx, sized.data.data(), raw.data()) != YOLO2_SUCCESS) throw std::runtime_error(yolo2_hip_last_error());
        elapsed = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
        frames_run = 1;
    } else {
        std::vector<int> wlen(yolo2_weight_len, yolo2_weight_len + YOLO2_N_CONV), blen(yolo2_bias_len, yolo2_bias_len + YOLO2_N_CONV);
        const y2h::WeightsI16 wp = y2h::load_weights_int16(cfg.weights_dir, wlen, blen);
        if (yolo2_hip_load_weights_int16(ctx, wp.weights.data(), wp.weights.size(), wp.bias.data(), wp.bias.size(), wp.weight_q.data(),
                                         (int)wp.weight_q.size(), wp.bias_q.data(), (int)wp.bias_q.size(), wp.act_q.data(),
                                         (int)wp.act_q.size()) != YOLO2_SUCCESS)
            throw std::runtime_error(yolo2_hip_last_error());
        if (yolo2_hip_set_batch(ctx, cfg.batch) != YOLO2_SUCCESS) throw std::runtime_error(yolo2_hip_last_error());
        std::vector<float> frames((size_t)cfg.batch * YOLO2_FRAME_ELEMS);
        for (int b = 0; b < cfg.batch; ++b) std::memcpy(frames.data() + (size_t)b * YOLO2_FRAME_ELEMS, sized.data.data(), sizeof(float) * YOLO2_FRAME_ELEMS);
        std::vector<int16_t> region((size_t)cfg.batch * YOLO2_REGION_ELEMS);
        int q = 0;
        const auto t0 = std::chrono::high_resolution_clock::now();
        if (yolo2_hip_run_batch_int16_host(ctx, frames.data(), cfg.batch, region.data(), &q) != YOLO2_SUCCESS)
            throw std::runtime_error(yolo2_hip_last_error());
        elapsed = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
        // dequantise (yolo2_model.cpp:415-421)
        const float scale = std::ldexp(1.0f, -q);
        for (int t = 0; t < YOLO2_REGION_ELEMS; ++t) raw[t] = (float)region[t] * scale;
    }
    std::printf("%s: Predicted in %.3f seconds.\n", cfg.input_path.c_str(), elapsed);
    std::printf("inference time: %.2f ms\n", elapsed * 1e3 / frames_run);  // line format parsed by scripts/yolo2_report.py:685-729
    yolo2_hip_destroy(ctx);

    // dumps with the reference's env-var names (yolo2_model.cpp:426-439, yolov2_main.cpp:297-306)
    const char *nd = std::getenv("YOLO2_NO_DUMP");
    const bool do_dump = !(nd && nd[0] && nd[0] != '0');
    const char *raw_path = std::getenv("YOLO2_DUMP_REGION_RAW");
    if (!raw_path || !raw_path[0]) raw_path = "yolov2_region_raw_hip.txt";
    if (do_dump) dump_floats(raw_path, raw.data(), raw.size());
    y2h::region_forward(last, raw.data(), proc.data());
    const char *proc_path = std::getenv("YOLO2_DUMP_REGION");
    if (!proc_path || !proc_path[0]) proc_path = "yolov2_region_proc_hip.txt";
    if (do_dump) dump_floats(proc_path, proc.data(), proc.size());

    std::vector<y2h::Detection> dets = y2h::region_boxes(last, proc.data(), im.w, im.h, net.w, net.h, cfg.thresh);
    int total = (int)dets.size();
    if (cfg.nms > 0.0f) total = y2h::nms_sort(dets, last.classes, cfg.nms);
    if ((int)names.size() < last.classes)
        std::fprintf(stderr, "Warning: names file provides %d labels, but network expects %d classes.\n", (int)names.size(), last.classes);

    int shown = 0;
    for (int i = 0; i < total; ++i) {
        for (int j = 0; j < last.classes; ++j) {
            if (dets[i].prob[j] <= cfg.thresh) continue;
            const y2h::Box &b = dets[i].bbox;
            const char *label = j < (int)names.size() ? names[j].c_str() : "?";
            std::printf("%s: %.0f%%  (x=%.4f y=%.4f w=%.4f h=%.4f)\n", label, dets[i].prob[j] * 100, b.x, b.y, b.w, b.h);
            if (cfg.json)
                std::printf("{\"label\":\"%s\",\"class\":%d,\"prob\":%.6f,\"x\":%.6f,\"y\":%.6f,\"w\":%.6f,\"h\":%.6f}\n", label, j, dets[i].prob[j], b.x, b.y, b.w, b.h);
            const int x1 = (int)((b.x - b.w / 2.) * im.w), x2 = (int)((b.x + b.w / 2.) * im.w);
            const int y1 = (int)((b.y - b.h / 2.) * im.h), y2 = (int)((b.y + b.h / 2.) * im.h);
            const float hue = (float)((j * 123457) % last.classes) / last.classes;
            y2h::draw_box(im, x1, y1, x2, y2, std::max(1, (int)(im.h * .006)), hue, 1.f - hue, 0.5f);
            ++shown;
        }
    }
    std::printf("%d detection(s) above %.2f\n", shown, cfg.thresh);
    y2h::save_ppm(im, cfg.output_prefix + ".ppm");
    std::printf("Output written to %s.ppm\nYOLOv2 Object Detection - Complete\n", cfg.output_prefix.c_str());
}

}  // namespace

int main(int argc, char **argv)
{
    try {
        run_detector(parse_args(argc, argv));
    } catch (const std::exception &ex) {
        std::fprintf(stderr, "Fatal error: %s\n", ex.what());  // yolov2_main.cpp:339-347
        return 1;
    }
    return 0;
}
