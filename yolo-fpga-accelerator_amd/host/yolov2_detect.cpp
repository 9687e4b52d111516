// yolov2_detect -- the repo's own detection CLI with `--backend hip`.
//
// Same role and flags as the reference CLI (src/models/yolov2/yolov2_main.cpp:83-132, 234-335):
// parse cfg, load names, load + letterbox the image, run the accelerator path, dump the region
// tensors, decode boxes, NMS, draw, save.  The accelerator path is the HIP library behind its C ABI
// (include/yolo2_hip.h); there is no CPU backend here -- `--backend hls|cpu` names the reference's
// own binaries and is rejected with a pointer to them.
//
// Differences that are deliberate: images are decoded by the host's own JPEG / PNG decoder (y2_codec.cpp: the bytes the
// reference's stb_image produces, reproduced byte for byte; binary PPM/PGM are read too),
// the annotated result is written as PPM, boxes are also printed as text and (--json) JSON lines;
// `--batch N` runs the same frame N times through one batched call to show the batched entry.
//
// Streaming frontend (the role of the reference's yolo2_linux --video/--camera loop, linux_app/src/main.c:878-1288):
//   --input-list <file> | --input-dir <dir> | --video-raw <file|-> --video-width W --video-height H
// Frames are taken in chunks of --batch per device, go to the GPU as BYTES (letterbox on the GPU), through the
// int16 network on --devices a,b,... (contiguous frame shards, weights broadcast once by the library) and through the
// region + boxes + NMS tail (--post gpu: yolo2_hip_postprocess_int16; --post host: the threaded host code).  Per frame it
// prints the reference's "Frame %d (infer %d) inference time: %.2f ms" line (parsed by scripts/yolo2_report.py:685-729)
// and, with --jsonl <path> (--output-json is accepted too, the reference's name), writes one record per frame with the
// reference's fields (main.c:1028-1077): mode, source, frame_index, inference_index, width, height, detections[class_id,
// label, prob, bbox_norm{x,y,w,h}, bbox_px{x0,y0,x1,y1}].
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <map>
#include <fstream>
#include <algorithm>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/yolo2_hip.h"
#include "y2_host.hpp"

namespace {

struct AppConfig {
    std::string cfg_path = "config/yolov2.cfg";
    std::string names_path = "config/coco.names";
    std::string input_path = "examples/test_images/dog.jpg";
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
    // streaming frontend
    std::vector<int> devices;          // --devices a,b,...: frame shards over several GPUs
    std::string input_list, input_dir, video_raw, jsonl_path, save_dir;
    int video_w = 640, video_h = 480;
    int max_frames = 0;                // 0 = all
    int infer_every = 1;
    std::string post = "gpu";          // region + boxes + NMS: gpu | host
    int chunk_batches = 4;             // batches per device and accelerator call (streaming)
    int decode_threads = 0;            // image decode pool; 0 = all host cores
    bool strict = false;               // streaming: stop at the first undecodable input instead of skipping the frame
    bool plan_cache = true;            // int16: keep bounds / forms / timed conv plans of this weight set in <weights>/weights_reorg_int16.bin.y2plan
    bool streaming() const { return !input_list.empty() || !input_dir.empty() || !video_raw.empty(); }
};

void print_usage(const char *prog)
{
    std::printf(
        "Usage: %s [options] [image]\n"
        "  --cfg <path>          Network cfg file (default: config/yolov2.cfg)\n"
        "  --names <path>        Class names file (default: config/coco.names)\n"
        "  --input <path>        Input image: JPEG (baseline / progressive), PNG, or binary PPM/PGM\n"
        "  --weights <dir>       Directory with weights_reorg_int16.bin, bias_int16.bin, *_Q.bin (default: weights)\n"
        "  --output <prefix>     Output file prefix without extension (default: results/<input>_prediction)\n"
        "  --thresh <float>      Confidence threshold (default: 0.25)\n"
        "  --nms <float>         NMS IoU threshold (default: 0.45)\n"
        "  --hier <float>        Hierarchical threshold (accepted, unused by region layers)\n"
        "  --backend <hip>       Backend selector (hip = MI355X library; hls/cpu live in the reference build)\n"
        "  --precision <int16|fp32|fp32fast|fp16>  int16 = the batched fixed-point path; fp32 = the exact fp32 pass (tiled, batched);\n"
        "                            fp32fast = the fp32 weight files on the matrix cores in split-fp16 form (hi + lo halves, three MFMAs per\n"
        "                            product: boxes within 1e-3 of fp32's, ~7x the exact pass's rate); fp16 = fp16 operands, fp32 accumulate (approximate)\n"
        "  --batch <n>           Frames per accelerator call (default 1)\n"
        "  --device <n>          HIP device (default 0)\n"
        "  --devices <a,b,..>    Several HIP devices: frames shard contiguously, weights are broadcast once (RCCL)\n"
        "  --json                Also print detections as JSON lines\n"
        "streaming (frames in chunks of --batch per device, bytes to the GPU, letterbox + network + NMS there):\n"
        "  --input-list <file>   One image path per line\n"
        "  --input-dir <dir>     Every *.jpg / *.jpeg / *.png / *.ppm / *.pgm of a directory, sorted by name\n"
        "  --video-raw <file|->  Raw RGB24 frames (e.g. from `ffmpeg -f rawvideo -pix_fmt rgb24 -`), with\n"
        "  --video-width <w> --video-height <h>   frame size (default 640x480)\n"
        "  --max-frames <n>      Stop after n inference frames (default: all)\n"
        "  --infer-every <n>     Run inference on every n-th frame (default 1)\n"
        "  --jsonl <path>        One JSON record per frame (fields of the reference's --output-json)\n"
        "  --save-annotated-dir <dir>   Write annotated frames as PPM\n"
        "  --chunk-batches <n>   Batches per device and accelerator call (default 4)\n"
        "  --decode-threads <n>  Host threads decoding images ahead of the accelerators (default: all cores; one pool feeds every device)\n"
        "  --strict              Stop at the first input that cannot be read or decoded (default: log it, skip the frame, go on)\n"
        "  --no-plan-cache       int16: do not read / write <weights>/weights_reorg_int16.bin.y2plan (the weight set's bounds, forms and\n"
        "                        timed conv plans; with it a weight set is timed once and every later run uses the same kernels)\n"
        "  --post <gpu|host>     Where region + boxes + NMS run (default gpu)\n",
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
        else if (arg == "--devices" && need("")) {
            std::string v = argv[++i];
            size_t pos = 0;
            while (pos <= v.size()) {
                const size_t c = v.find(',', pos);
                const std::string tok = v.substr(pos, c == std::string::npos ? std::string::npos : c - pos);
                if (!tok.empty()) cfg.devices.push_back(std::atoi(tok.c_str()));
                if (c == std::string::npos) break;
                pos = c + 1;
            }
        }
        else if (arg == "--input-list" && need("")) cfg.input_list = argv[++i];
        else if (arg == "--input-dir" && need("")) cfg.input_dir = argv[++i];
        else if (arg == "--video-raw" && need("")) cfg.video_raw = argv[++i];
        else if (arg == "--video-width" && need("")) cfg.video_w = std::atoi(argv[++i]);
        else if (arg == "--video-height" && need("")) cfg.video_h = std::atoi(argv[++i]);
        else if (arg == "--max-frames" && need("")) cfg.max_frames = std::atoi(argv[++i]);
        else if (arg == "--infer-every" && need("")) cfg.infer_every = std::max(1, std::atoi(argv[++i]));
        else if ((arg == "--jsonl" || arg == "--output-json") && need("")) cfg.jsonl_path = argv[++i];
        else if (arg == "--save-annotated-dir" && need("")) cfg.save_dir = argv[++i];
        else if (arg == "--chunk-batches" && need("")) cfg.chunk_batches = std::max(1, std::atoi(argv[++i]));
        else if (arg == "--decode-threads" && need("")) cfg.decode_threads = std::atoi(argv[++i]);
        else if (arg == "--strict") cfg.strict = true;
        else if (arg == "--no-plan-cache") cfg.plan_cache = false;
        else if (arg == "--post" && need("")) {
            cfg.post = argv[++i];
            if (cfg.post != "gpu" && cfg.post != "host") { std::fprintf(stderr, "Unsupported --post %s (gpu | host)\n", cfg.post.c_str()); std::exit(1); }
        }
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
            if (cfg.precision == "half" || cfg.precision == "f16") cfg.precision = "fp16";
            if (cfg.precision == "f32fast" || cfg.precision == "fp32tol") cfg.precision = "fp32fast";
            if (cfg.precision != "int16" && cfg.precision != "fp32" && cfg.precision != "fp16" && cfg.precision != "fp32fast") {
                std::fprintf(stderr, "Unsupported precision: %s (the hip backend runs int16, fp32, fp32fast and fp16)\n", cfg.precision.c_str());
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

// ---- JSONL records with the reference's fields (linux_app/src/main.c:1028-1077)
struct OutDet {
    int class_id;
    float prob;
    y2h::Box box;
};

void json_escaped(FILE *fp, const std::string &sv)
{
    std::fputc('"', fp);
    for (unsigned char ch : sv) {
        if (ch == '"' || ch == '\\') { std::fputc('\\', fp); std::fputc(ch, fp); }
        else if (ch < 0x20) std::fprintf(fp, "\\u%04x", ch);
        else std::fputc(ch, fp);
    }
    std::fputc('"', fp);
}

void write_jsonl(FILE *fp, const char *mode, const std::string &source, int frame_index, int infer_index, int w, int h,
                 const std::vector<OutDet> &dets, const std::vector<std::string> &names)
{
    std::fprintf(fp, "{\"mode\":\"%s\",\"source\":", mode);
    json_escaped(fp, source);
    std::fprintf(fp, ",\"frame_index\":%d,\"inference_index\":%d,\"width\":%d,\"height\":%d,\"detections\":[", frame_index, infer_index, w, h);
    bool first = true;
    for (const OutDet &d : dets) {
        const y2h::Box &b = d.box;
        const int x0 = (int)((b.x - b.w * 0.5f) * (float)w), y0 = (int)((b.y - b.h * 0.5f) * (float)h);
        const int x1 = (int)((b.x + b.w * 0.5f) * (float)w), y1 = (int)((b.y + b.h * 0.5f) * (float)h);
        if (!first) std::fputc(',', fp);
        first = false;
        std::fprintf(fp, "{\"class_id\":%d,\"label\":", d.class_id);
        json_escaped(fp, d.class_id < (int)names.size() ? names[(size_t)d.class_id] : "unknown");
        std::fprintf(fp, ",\"prob\":%.6f,\"bbox_norm\":{\"x\":%.6f,\"y\":%.6f,\"w\":%.6f,\"h\":%.6f},", d.prob, b.x, b.y, b.w, b.h);
        std::fprintf(fp, "\"bbox_px\":{\"x0\":%d,\"y0\":%d,\"x1\":%d,\"y1\":%d}}", x0, y0, x1, y1);
    }
    std::fprintf(fp, "]}\n");
    std::fflush(fp);
}

// the reference's rule for a record (main.c:1040-1052): a detection's best class, if its prob exceeds the threshold
std::vector<OutDet> best_class_dets(const std::vector<y2h::Detection> &dets, int total, int classes, float thresh)
{
    std::vector<OutDet> out;
    for (int i = 0; i < total; ++i) {
        int best = -1;
        float bp = 0.f;
        for (int j = 0; j < classes; ++j)
            if (dets[(size_t)i].prob[(size_t)j] > bp) { bp = dets[(size_t)i].prob[(size_t)j]; best = j; }
        if (bp <= thresh || best < 0) continue;
        out.push_back({best, bp, dets[(size_t)i].bbox});
    }
    return out;
}

struct SrcFrame {
    std::string source;
    int frame_index = 0;    // 1-based position in the stream
    y2h::ImageU8 img;
};

// Frame source: a list of image files, a directory, or a raw RGB24 stream.  next_ref() enumerates the frames selected for
// inference (--infer-every) WITHOUT decoding image files: the reader decodes a whole chunk in parallel (decode()).
class FrameSource {
  public:
    explicit FrameSource(const AppConfig &cfg) : cfg_(cfg)
    {
        namespace fs = std::filesystem;
        if (!cfg.input_list.empty()) {
            std::ifstream in(cfg.input_list);
            if (!in) throw std::runtime_error("Cannot open " + cfg.input_list);
            std::string line;
            while (std::getline(in, line)) {
                while (!line.empty() && (line.back() == '\r' || line.back() == ' ')) line.pop_back();
                if (!line.empty() && line[0] != '#') files_.push_back(line);
            }
            mode_ = "list";
        } else if (!cfg.input_dir.empty()) {
            for (const auto &e : fs::directory_iterator(cfg.input_dir)) {
                const std::string ext = e.path().extension().string();
                if (ext == ".ppm" || ext == ".pgm" || ext == ".jpg" || ext == ".jpeg" || ext == ".png") files_.push_back(e.path().string());
            }
            std::sort(files_.begin(), files_.end());
            mode_ = "dir";
        } else {
            if (cfg.video_w <= 0 || cfg.video_h <= 0) throw std::runtime_error("--video-width / --video-height must be positive");
            raw_ = cfg.video_raw == "-" ? stdin : std::fopen(cfg.video_raw.c_str(), "rb");
            if (!raw_) throw std::runtime_error("Cannot open " + cfg.video_raw);
            mode_ = "video";
        }
        if (mode_ != "video" && files_.empty()) throw std::runtime_error("no input images");
    }
    ~FrameSource() { if (raw_ && raw_ != stdin) std::fclose(raw_); }
    const char *mode() const { return mode_.c_str(); }
    bool is_video() const { return mode_ == "video"; }
    // next frame selected for inference (honours --infer-every); false at the end of the stream.  Video frames arrive with
    // their pixels, image files with their path only.
    bool next_ref(SrcFrame &out)
    {
        for (;;) {
            SrcFrame f;
            if (mode_ == "video") {
                f.img.w = cfg_.video_w; f.img.h = cfg_.video_h;
                f.img.rgb.resize((size_t)cfg_.video_w * cfg_.video_h * 3);
                const size_t rd = std::fread(f.img.rgb.data(), 1, f.img.rgb.size(), raw_);
                if (rd != f.img.rgb.size()) return false;    // EOF (a trailing partial frame is dropped like the reference's reader)
                f.source = cfg_.video_raw;
            } else {
                if (pos_ >= files_.size()) return false;
                f.source = files_[pos_++];
            }
            const bool take = (count_ % cfg_.infer_every) == 0;
            ++count_;
            if (!take) continue;
            f.frame_index = count_;
            out = std::move(f);
            return true;
        }
    }
    static void decode(SrcFrame &f) { f.img = y2h::load_image_u8(f.source); }

  private:
    const AppConfig &cfg_;
    std::vector<std::string> files_;
    size_t pos_ = 0;
    int count_ = 0;
    FILE *raw_ = nullptr;
    std::string mode_;
};

// a bounded hand-off between two pipeline stages
template <typename T>
class Channel {
  public:
    explicit Channel(size_t cap) : cap_(cap) {}
    void push(T v)
    {
        std::unique_lock<std::mutex> lk(mu_);
        cv_space_.wait(lk, [&] { return q_.size() < cap_ || closed_; });
        if (closed_) return;
        q_.push_back(std::move(v));
        cv_item_.notify_one();
    }
    bool pop(T &out)     // false once the channel is closed and drained
    {
        std::unique_lock<std::mutex> lk(mu_);
        cv_item_.wait(lk, [&] { return !q_.empty() || closed_; });
        if (q_.empty()) return false;
        out = std::move(q_.front());
        q_.pop_front();
        cv_space_.notify_one();
        return true;
    }
    void close()
    {
        std::lock_guard<std::mutex> lk(mu_);
        closed_ = true;
        cv_item_.notify_all();
        cv_space_.notify_all();
    }

  private:
    std::mutex mu_;
    std::condition_variable cv_item_, cv_space_;
    std::deque<T> q_;
    size_t cap_;
    bool closed_ = false;
};

struct Chunk {
    long seq = 0;                            // position in the stream: the writer emits chunks in this order
    int device_slot = 0;                     // which device lane ran it
    std::vector<SrcFrame> frames;            // frames whose image could not be decoded are dropped before the accelerator (see skipped)
    std::vector<std::pair<int, std::string>> skipped;   // (frame_index, reason)
    std::vector<std::vector<OutDet>> dets;   // filled by the accelerator stage
    double seconds = 0;                      // wall time of the accelerator call for this chunk
};

// A pool of decode threads that lives as long as the stream (round 3 created and joined the threads per chunk).  decode()
// spreads the frames of one chunk over the pool and returns when all are done; a frame whose file cannot be read or decoded
// keeps an empty image and its reason in `error` - one bad file must not end a stream of thousands.
class DecodePool {
  public:
    explicit DecodePool(int n)
    {
        for (int i = 0; i < std::max(1, n); ++i) th_.emplace_back([this] { work(); });
    }
    ~DecodePool()
    {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_job_.notify_all();
        for (auto &t : th_) t.join();
    }
    int size() const { return (int)th_.size(); }
    void decode(std::vector<SrcFrame> &frames, std::vector<std::string> &errors)
    {
        errors.assign(frames.size(), std::string());
        std::unique_lock<std::mutex> lk(mu_);
        frames_ = &frames; errors_ = &errors; next_ = 0; left_ = frames.size();
        ++gen_;
        cv_job_.notify_all();
        cv_done_.wait(lk, [&] { return left_ == 0; });
        frames_ = nullptr;
    }

  private:
    void work()
    {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            cv_job_.wait(lk, [&] { return stop_ || (frames_ && next_ < frames_->size()); });
            if (stop_) return;
            const size_t i = next_++;
            std::vector<SrcFrame> *fr = frames_;
            std::vector<std::string> *er = errors_;
            lk.unlock();
            try { FrameSource::decode((*fr)[i]); }
            catch (const std::exception &e) { (*er)[i] = e.what(); if ((*er)[i].empty()) (*er)[i] = "decode failed"; }
            lk.lock();
            if (--left_ == 0) cv_done_.notify_all();
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_job_, cv_done_;
    std::vector<SrcFrame> *frames_ = nullptr;
    std::vector<std::string> *errors_ = nullptr;
    size_t next_ = 0, left_ = 0;
    unsigned long gen_ = 0;
    bool stop_ = false;
};

// The streaming frontend as overlapped stages (the reference's loop, linux_app/src/main.c:878-1288, does them in turn per
// frame):  reader (file I/O + JPEG / PNG decode of the next chunk on a persistent pool of host threads)  ->  one ACCELERATOR LANE
// PER DEVICE (each pops the next decoded chunk from the shared queue: bytes in, letterbox + network + region / boxes / NMS on its
// device, detection records out - the region tensor never leaves HBM)  ->  writer (log lines, JSONL, annotated frames, in stream
// order).  Round 3 drove all devices from one call per chunk (every chunk was split over all devices and joined): a chunk was as
// slow as its slowest device and the reader fed them in lock step.  Now a chunk belongs to ONE device, devices take chunks as
// they become free (work-conserving, no join across devices), and the writer restores the order by chunk number.
void run_stream(AppConfig cfg)
{
    namespace fs = std::filesystem;
    static char outbuf[1 << 16];
    std::setvbuf(stdout, outbuf, _IOFBF, sizeof(outbuf));    // thousands of frames a second: no write() per line
    if (cfg.precision != "int16") throw std::runtime_error("the streaming frontend runs the int16 path");
    if (cfg.devices.empty()) cfg.devices.push_back(cfg.device);
    if (cfg.batch <= 0) throw std::runtime_error("--batch must be positive");
    const y2h::Network net = y2h::parse_cfg(cfg.cfg_path);
    check_topology(net);
    const std::vector<std::string> names = y2h::load_names(cfg.names_path);
    const y2h::Layer &last = net.layers.back();
    FrameSource src(cfg);
    const int ndev = (int)cfg.devices.size();
    std::printf("YOLOv2 Object Detection - streaming (%s)\n  devices:", src.mode());
    for (int d : cfg.devices) std::printf(" %d", d);
    std::printf("\n  batch per device: %d\n  post-processing: %s\n", cfg.batch, cfg.post.c_str());

    yolo2_hip_multi *m = nullptr;
    if (yolo2_hip_multi_create(cfg.devices.data(), ndev, &m) != YOLO2_SUCCESS) throw std::runtime_error(yolo2_hip_last_error());
    struct Guard { yolo2_hip_multi *m; ~Guard() { yolo2_hip_multi_destroy(m); } } guard{m};
    {
        // the weight-side cache (include/yolo2_hip.h): beside the interchange file, one per weight set, shared by every device
        if (cfg.plan_cache)
            for (int i = 0; i < yolo2_hip_multi_num_devices(m); ++i)
                if (yolo2_hip_set_plan_cache(yolo2_hip_multi_ctx(m, i), (cfg.weights_dir + "/weights_reorg_int16.bin.y2plan").c_str()) != YOLO2_SUCCESS)
                    throw std::runtime_error(yolo2_hip_last_error());
        std::vector<int> wlen(yolo2_weight_len, yolo2_weight_len + YOLO2_N_CONV), blen(yolo2_bias_len, yolo2_bias_len + YOLO2_N_CONV);
        const y2h::WeightsI16 wp = y2h::load_weights_int16(cfg.weights_dir, wlen, blen);
        if (yolo2_hip_multi_load_weights_int16(m, wp.weights.data(), wp.weights.size(), wp.bias.data(), wp.bias.size(), wp.weight_q.data(),
                                               (int)wp.weight_q.size(), wp.bias_q.data(), (int)wp.bias_q.size(), wp.act_q.data(),
                                               (int)wp.act_q.size()) != YOLO2_SUCCESS)
            throw std::runtime_error(yolo2_hip_last_error());
    }
    std::printf("  weights on %d device(s)%s\n", yolo2_hip_multi_num_devices(m), yolo2_hip_multi_uses_rccl(m) ? " (RCCL broadcast)" : "");
    auto plan_source_name = [](int s) { return s == 1 ? "plan table" : s == 2 ? "timed in this process" : s == 3 ? "static defaults" : s == 4 ? "forced" : s == 5 ? "weight cache" : "none"; };
    FILE *jf = nullptr;
    if (!cfg.jsonl_path.empty()) {
        jf = std::fopen(cfg.jsonl_path.c_str(), "w");
        if (!jf) throw std::runtime_error("Failed to open JSON output " + cfg.jsonl_path);
    }
    if (!cfg.save_dir.empty()) fs::create_directories(cfg.save_dir);

    // a chunk = what one accelerator call on ONE device takes: `chunk_batches` batches, so that the fill / drain of the call's
    // internal upload-compute-download pipeline is a small part of it
    const int chunk = cfg.batch * std::max(1, cfg.chunk_batches);
    const int threads = cfg.decode_threads > 0 ? cfg.decode_threads : (int)std::max(1u, std::thread::hardware_concurrency());
    std::printf("  decode threads: %d (host feed: one pool for all devices; the host side, not the GPUs, bounds a multi-device run - DESIGN.md 6)\n", threads);
    Channel<std::unique_ptr<Chunk>> to_run((size_t)2 * ndev), to_write((size_t)2 * ndev + 2);
    std::mutex err_mu;
    std::string err;
    auto fail_with = [&](const std::string &what) {
        { std::lock_guard<std::mutex> lk(err_mu); if (err.empty()) err = what; }
        to_run.close(); to_write.close();
    };
    const auto t_start = std::chrono::steady_clock::now();

    std::thread reader([&] {
        try {
            DecodePool pool(threads);
            int taken = 0;
            long seq = 0;
            bool more = true;
            // the first chunks are short (one batch, then doubling): the accelerators start after `batch` decoded images each
            int want = std::min(chunk, cfg.batch);
            std::vector<std::string> errors;
            while (more) {
                auto ck = std::make_unique<Chunk>();
                const int this_chunk = want;
                if (seq % ndev == ndev - 1) want = std::min(chunk, want * 2);     // every device has had a chunk of this size
                while ((int)ck->frames.size() < this_chunk && (cfg.max_frames <= 0 || taken + (int)ck->frames.size() < cfg.max_frames)) {
                    SrcFrame f;
                    if (!src.next_ref(f)) { more = false; break; }
                    ck->frames.push_back(std::move(f));
                }
                taken += (int)ck->frames.size();
                if (cfg.max_frames > 0 && taken >= cfg.max_frames) more = false;
                if (ck->frames.empty()) break;
                if (!src.is_video()) {     // decode the chunk's files on the pool
                    pool.decode(ck->frames, errors);
                    std::vector<SrcFrame> good;
                    good.reserve(ck->frames.size());
                    for (size_t i = 0; i < ck->frames.size(); ++i) {
                        if (errors[i].empty()) { good.push_back(std::move(ck->frames[i])); continue; }
                        if (cfg.strict) throw std::runtime_error(ck->frames[i].source + ": " + errors[i]);
                        ck->skipped.emplace_back(ck->frames[i].frame_index, ck->frames[i].source + ": " + errors[i]);
                    }
                    ck->frames = std::move(good);
                }
                ck->seq = seq++;
                to_run.push(std::move(ck));
            }
        } catch (const std::exception &e) { fail_with(e.what()); }
        to_run.close();
    });

    int infer_idx = 0, skipped_total = 0;
    std::thread writer([&] {
        try {
            std::unique_ptr<Chunk> ck;
            std::map<long, std::unique_ptr<Chunk>> held;      // chunks that arrived ahead of their turn (another device was faster)
            long next_seq = 0;
            auto emit = [&](Chunk &c) {
                for (const auto &sk : c.skipped) {
                    ++skipped_total;
                    std::printf("Frame %d skipped: %s\n", sk.first, sk.second.c_str());
                    std::fprintf(stderr, "Warning: frame %d skipped: %s\n", sk.first, sk.second.c_str());
                }
                const int n = (int)c.frames.size();
                for (int f = 0; f < n; ++f) {
                    ++infer_idx;
                    const SrcFrame &fr = c.frames[(size_t)f];
                    // the per-frame share of the chunk's wall time (the frames of a chunk run as one batched call)
                    std::printf("Frame %d (infer %d) inference time: %.2f ms\n", fr.frame_index, infer_idx, c.seconds * 1e3 / n);
                    for (const OutDet &d : c.dets[(size_t)f])
                        std::printf("  %s: %.0f%%  (x=%.4f y=%.4f w=%.4f h=%.4f)\n", d.class_id < (int)names.size() ? names[(size_t)d.class_id].c_str() : "?",
                                    d.prob * 100, d.box.x, d.box.y, d.box.w, d.box.h);
                    if (jf) write_jsonl(jf, src.mode(), fr.source, fr.frame_index, infer_idx, fr.img.w, fr.img.h, c.dets[(size_t)f], names);
                    if (!cfg.save_dir.empty()) {
                        y2h::Image im = y2h::make_image(fr.img.w, fr.img.h, 3);
                        for (int k = 0; k < 3; ++k)
                            for (int y = 0; y < im.h; ++y)
                                for (int x = 0; x < im.w; ++x) im.at(x, y, k) = (float)fr.img.rgb[((size_t)y * im.w + x) * 3 + k] / 255.f;
                        for (const OutDet &d : c.dets[(size_t)f]) {
                            const y2h::Box &b = d.box;
                            const float hue = (float)((d.class_id * 123457) % last.classes) / last.classes;
                            y2h::draw_box(im, (int)((b.x - b.w / 2.) * im.w), (int)((b.y - b.h / 2.) * im.h), (int)((b.x + b.w / 2.) * im.w),
                                          (int)((b.y + b.h / 2.) * im.h), std::max(1, (int)(im.h * .006)), hue, 1.f - hue, 0.5f);
                        }
                        char name[64];
                        std::snprintf(name, sizeof(name), "frame_%06d.ppm", infer_idx);
                        y2h::save_ppm(im, (fs::path(cfg.save_dir) / name).string());
                    }
                }
                std::fflush(stdout);
            };
            while (to_write.pop(ck)) {
                held[ck->seq] = std::move(ck);
                for (auto it = held.find(next_seq); it != held.end(); it = held.find(next_seq)) {
                    emit(*it->second);
                    held.erase(it);
                    ++next_seq;
                }
            }
            if (!held.empty() && err.empty()) throw std::runtime_error("stream ended with chunks out of order");
        } catch (const std::exception &e) { fail_with(e.what()); }
    });

    // ---- one accelerator lane per device
    std::vector<double> accel_s((size_t)ndev, 0.0);
    auto lane = [&](int slot) {
        try {
            yolo2_hip_ctx *ctx = yolo2_hip_multi_ctx(m, slot);
            // while the reader decodes its first chunk: plan the batch (activation tensors, lanes) on this device
            if (yolo2_hip_set_batch(ctx, cfg.batch) != YOLO2_SUCCESS) throw std::runtime_error(yolo2_hip_last_error());
            if (slot == 0) std::printf("  conv plans: %s\n", plan_source_name(yolo2_hip_plan_source(ctx)));
            std::unique_ptr<Chunk> ck;
            std::vector<int16_t> region;
            std::vector<yolo2_hip_det> recs;
            const int post_threads = std::max(1, threads / ndev);
            while (to_run.pop(ck)) {
                const int n = (int)ck->frames.size();
                ck->device_slot = slot;
                ck->dets.assign((size_t)n, {});
                if (n > 0) {
                    std::vector<const uint8_t *> ptrs((size_t)n);
                    std::vector<int> ws((size_t)n), hs((size_t)n);
                    for (int i = 0; i < n; ++i) { ptrs[(size_t)i] = ck->frames[(size_t)i].img.rgb.data(); ws[(size_t)i] = ck->frames[(size_t)i].img.w; hs[(size_t)i] = ck->frames[(size_t)i].img.h; }
                    int q = 0;
                    const auto t0 = std::chrono::steady_clock::now();
                    if (cfg.post == "gpu") {
                        // one record per detection (its best class, main.c:1040-1052): at most 845 per frame, never truncated
                        const int cap = 845;
                        recs.resize((size_t)n * cap);
                        std::vector<int> counts((size_t)n);
                        if (yolo2_hip_run_images_u8_dets(ctx, ptrs.data(), ws.data(), hs.data(), 3, n, cfg.batch, cfg.thresh, cfg.nms, YOLO2_DETS_BEST_CLASS,
                                                         recs.data(), cap, counts.data(), &q) != YOLO2_SUCCESS)
                            throw std::runtime_error(yolo2_hip_last_error());
                        for (int f = 0; f < n; ++f) {
                            if (counts[(size_t)f] > cap) throw std::runtime_error("detection records truncated");   // cannot happen in best-class mode
                            for (int k = 0; k < counts[(size_t)f]; ++k) {
                                const yolo2_hip_det &r = recs[(size_t)f * cap + k];
                                if (r.prob > cfg.thresh) ck->dets[(size_t)f].push_back({r.cls, r.prob, {r.x, r.y, r.w, r.h}});
                            }
                        }
                    } else {
                        region.resize((size_t)n * YOLO2_REGION_ELEMS);
                        if (yolo2_hip_run_images_u8_host(ctx, ptrs.data(), ws.data(), hs.data(), 3, n, cfg.batch, region.data(), &q) != YOLO2_SUCCESS)
                            throw std::runtime_error(yolo2_hip_last_error());
                        auto all = y2h::postprocess_batch(region.data(), n, q, ws.data(), hs.data(), cfg.thresh, cfg.nms, post_threads);
                        for (int f = 0; f < n; ++f) ck->dets[(size_t)f] = best_class_dets(all[(size_t)f], (int)all[(size_t)f].size(), last.classes, cfg.thresh);
                    }
                    ck->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                    accel_s[(size_t)slot] += ck->seconds;
                }
                to_write.push(std::move(ck));
            }
        } catch (const std::exception &e) { fail_with(e.what()); }
    };
    std::vector<std::thread> lanes;
    for (int i = 1; i < ndev; ++i) lanes.emplace_back(lane, i);
    lane(0);                                   // device 0's lane runs on this thread
    for (auto &t : lanes) t.join();
    to_write.close();
    to_run.close();
    reader.join();
    writer.join();
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    if (jf) std::fclose(jf);
    if (!err.empty()) throw std::runtime_error(err);
    if (infer_idx == 0) throw std::runtime_error("No inference frames processed");
    const double busiest = *std::max_element(accel_s.begin(), accel_s.end());
    if (skipped_total) std::printf("\n%d frame(s) skipped: undecodable input (use --strict to stop at the first one)\n", skipped_total);
    std::printf("\nStreaming inference completed successfully (%d inference frames; %.1f frames/s end to end incl. file I/O, decode and "
                "output; %.1f frames/s inside the accelerator calls)\n", infer_idx, infer_idx / std::max(wall, 1e-9), infer_idx / std::max(busiest, 1e-9));
    std::fflush(stdout);
}

void run_detector(AppConfig cfg)
{
    if (cfg.streaming()) { run_stream(cfg); return; }
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
    y2h::Image im = y2h::load_image(cfg.input_path);   // load_image_stb(path, 3) of the reference, own decoders
    std::printf("Input img: %s (w=%d, h=%d, c=%d)\n", cfg.input_path.c_str(), im.w, im.h, im.c);
    const y2h::Image sized = y2h::letterbox_image(im, net.w, net.h);

    const y2h::Layer &last = net.layers.back();
    std::vector<float> raw(YOLO2_REGION_ELEMS), proc(YOLO2_REGION_ELEMS);
    yolo2_hip_ctx *ctx = nullptr;
    if (yolo2_hip_create(cfg.device, &ctx) != YOLO2_SUCCESS) throw std::runtime_error(yolo2_hip_last_error());
    double elapsed = 0;
    int frames_run = cfg.batch;
    if (cfg.precision == "fp32" || cfg.precision == "fp16" || cfg.precision == "fp32fast") {
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
        // the tiled exact fp32 pass (bit-identical to the reference's fp32 path); --batch N repeats the frame like the int16 branch
        std::vector<float> frames((size_t)cfg.batch * YOLO2_FRAME_ELEMS), regions((size_t)cfg.batch * YOLO2_REGION_ELEMS);
        for (int b = 0; b < cfg.batch; ++b) std::memcpy(frames.data() + (size_t)b * YOLO2_FRAME_ELEMS, sized.data.data(), sizeof(float) * YOLO2_FRAME_ELEMS);
        const auto t0 = std::chrono::high_resolution_clock::now();
        // (fp16: the MFMA path on the same weights - not the reference's arithmetic, max |error| ~0.006 on a +-4.7 region tensor)
        const int rc = cfg.precision == "fp16" ? yolo2_hip_run_batch_fp16_host(ctx, frames.data(), cfg.batch, regions.data())
                       : cfg.precision == "fp32fast" ? yolo2_hip_run_batch_f32tol_host(ctx, frames.data(), cfg.batch, regions.data())
                                                     : yolo2_hip_run_batch_fp32_host(ctx, frames.data(), cfg.batch, regions.data());
        if (rc != YOLO2_SUCCESS) throw std::runtime_error(yolo2_hip_last_error());
        elapsed = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
        std::memcpy(raw.data(), regions.data(), sizeof(float) * YOLO2_REGION_ELEMS);
    } else {
        std::vector<int> wlen(yolo2_weight_len, yolo2_weight_len + YOLO2_N_CONV), blen(yolo2_bias_len, yolo2_bias_len + YOLO2_N_CONV);
        const y2h::WeightsI16 wp = y2h::load_weights_int16(cfg.weights_dir, wlen, blen);
        if (cfg.plan_cache && yolo2_hip_set_plan_cache(ctx, (cfg.weights_dir + "/weights_reorg_int16.bin.y2plan").c_str()) != YOLO2_SUCCESS)
            throw std::runtime_error(yolo2_hip_last_error());
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
    if (!cfg.jsonl_path.empty()) {
        FILE *jf = std::fopen(cfg.jsonl_path.c_str(), "w");
        if (!jf) throw std::runtime_error("Failed to open JSON output " + cfg.jsonl_path);
        write_jsonl(jf, "image", cfg.input_path, 1, 1, im.w, im.h, best_class_dets(dets, total, last.classes, cfg.thresh), names);
        std::fclose(jf);
    }
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
