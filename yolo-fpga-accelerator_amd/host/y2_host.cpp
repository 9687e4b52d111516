// y2_host.cpp -- host side of the YOLOv2 path (see y2_host.hpp).  Our own code; the behaviours it
// reproduces are cited per function (paths relative to the reference repository).
#include "y2_host.hpp"

#include <algorithm>
#include <thread>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>

namespace y2h {

// ------------------------------------------------------------------------------------------ cfg

namespace {
std::string strip(const std::string &s)
{
    size_t a = 0, b = s.size();
    while (a < b && isspace((unsigned char)s[a])) ++a;
    while (b > a && isspace((unsigned char)s[b - 1])) --b;
    return s.substr(a, b - a);
}

struct Section {
    std::string name;
    std::map<std::string, std::string> kv;
    int get_int(const char *k, int def) const
    {
        auto it = kv.find(k);
        return it == kv.end() ? def : atoi(it->second.c_str());
    }
    std::string get_str(const char *k, const char *def) const
    {
        auto it = kv.find(k);
        return it == kv.end() ? std::string(def) : it->second;
    }
};

// Darknet cfg: "[section]" lines, "key=value" lines, '#' / ';' comments (src/core/yolo_cfg.cpp)
std::vector<Section> read_sections(const std::string &path)
{
    std::ifstream in(path);
    if (!in) throw std::runtime_error("Couldn't open file: " + path);
    std::vector<Section> secs;
    std::string line;
    while (std::getline(in, line)) {
        line = strip(line);
        if (line.empty() || line[0] == '#' || line[0] == ';') continue;
        if (line[0] == '[') {
            Section s;
            s.name = line;
            secs.push_back(s);
            continue;
        }
        const size_t eq = line.find('=');
        if (eq == std::string::npos || secs.empty()) continue;
        secs.back().kv[strip(line.substr(0, eq))] = strip(line.substr(eq + 1));
    }
    return secs;
}
}  // namespace

Network parse_cfg(const std::string &path)
{
    const std::vector<Section> secs = read_sections(path);
    if (secs.empty() || (secs[0].name != "[net]" && secs[0].name != "[network]"))
        throw std::runtime_error("First section must be [net] or [network]");
    Network net;
    net.w = secs[0].get_int("width", 0);
    net.h = secs[0].get_int("height", 0);
    net.c = secs[0].get_int("channels", 0);
    int h = net.h, w = net.w, c = net.c;
    for (size_t si = 1; si < secs.size(); ++si) {
        const Section &s = secs[si];
        Layer l;
        l.c = c; l.h = h; l.w = w;
        const int index = (int)net.layers.size();
        if (s.name == "[convolutional]" || s.name == "[conv]") {
            // parse_convolutional, src/core/yolo_layers.cpp:90-117: pad=1 means padding = size/2
            l.type = CONV;
            l.n = s.get_int("filters", 1);
            l.size = s.get_int("size", 1);
            l.stride = s.get_int("stride", 1);
            l.pad = s.get_int("padding", 0);
            if (s.get_int("pad", 0)) l.pad = l.size / 2;
            l.leaky = s.get_str("activation", "logistic") == "leaky";
            l.batch_normalize = s.get_int("batch_normalize", 0) != 0;
            if (!(h && w && c)) throw std::runtime_error("Layer before convolutional layer must output image.");
            l.out_c = l.n;
            l.out_h = (h + 2 * l.pad - l.size) / l.stride + 1;
            l.out_w = (w + 2 * l.pad - l.size) / l.stride + 1;
        } else if (s.name == "[maxpool]" || s.name == "[max]") {
            // parse_maxpool / make_maxpool_layer, yolo_layers.cpp:289-326: padding defaults to size-1
            l.type = MAXPOOL;
            l.stride = s.get_int("stride", 1);
            l.size = s.get_int("size", l.stride);
            l.pad = s.get_int("padding", l.size - 1);
            l.n = c;
            l.out_c = c;
            l.out_h = (h + l.pad - l.size) / l.stride + 1;
            l.out_w = (w + l.pad - l.size) / l.stride + 1;
        } else if (s.name == "[reorg]") {
            l.type = REORG;
            l.stride = s.get_int("stride", 1);
            l.out_c = c * l.stride * l.stride;
            l.out_h = h / l.stride;
            l.out_w = w / l.stride;
            l.n = l.out_c;
        } else if (s.name == "[route]") {
            // parse_route, yolo_layers.cpp:119-...: negative indices are relative to this layer
            l.type = ROUTE;
            std::stringstream ss(s.get_str("layers", ""));
            std::string tok;
            l.out_c = 0;
            while (std::getline(ss, tok, ',')) {
                int idx = atoi(tok.c_str());
                if (idx < 0) idx = index + idx;
                if (idx < 0 || idx >= index) throw std::runtime_error("route layer index out of range");
                l.route_layers.push_back(idx);
                const Layer &src = net.layers[idx];
                l.out_h = src.out_h;
                l.out_w = src.out_w;
                l.out_c += src.out_c;
            }
            if (l.route_layers.empty()) throw std::runtime_error("Route Layer must specify input layers");
            l.c = l.out_c; l.h = l.out_h; l.w = l.out_w;
        } else if (s.name == "[region]") {
            l.type = REGION;
            l.coords = s.get_int("coords", 4);
            l.classes = s.get_int("classes", 20);
            l.num = s.get_int("num", 1);
            l.softmax = s.get_int("softmax", 0) != 0;
            l.background = s.get_int("background", 0) != 0;
            l.anchors.assign((size_t)l.num * 2, 0.5f);
            std::stringstream ss(s.get_str("anchors", ""));
            std::string tok;
            size_t k = 0;
            while (std::getline(ss, tok, ',') && k < l.anchors.size()) l.anchors[k++] = (float)atof(tok.c_str());
            l.out_c = c; l.out_h = h; l.out_w = w;
            l.n = l.num;
        } else {
            throw std::runtime_error("Type not recognized: " + s.name);  // yolo_net.cpp:253-265 knows only these
        }
        net.layers.push_back(l);
        c = l.out_c; h = l.out_h; w = l.out_w;
    }
    if (net.layers.empty()) throw std::runtime_error("Config file has no layers");
    return net;
}

// --------------------------------------------------------------------------------------- images

Image make_image(int w, int h, int c)
{
    Image im;
    im.w = w; im.h = h; im.c = c;
    im.data.assign((size_t)w * h * c, 0.f);
    return im;
}

namespace {
int pnm_int(FILE *f)
{
    int ch;
    for (;;) {
        ch = fgetc(f);
        if (ch == '#') {
            while (ch != '\n' && ch != EOF) ch = fgetc(f);
        } else if (!isspace(ch)) break;
    }
    int v = 0;
    while (ch != EOF && isdigit(ch)) {
        v = v * 10 + (ch - '0');
        ch = fgetc(f);
    }
    return v;
}
}  // namespace

ImageU8 load_pnm_u8(const std::string &path)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("Cannot load image \"" + path + "\"");
    char magic[3] = {0, 0, 0};
    if (fread(magic, 1, 2, f) != 2 || magic[0] != 'P' || (magic[1] != '6' && magic[1] != '5')) {
        fclose(f);
        throw std::runtime_error("Unsupported image format (binary PPM P6 / PGM P5 expected): " + path);
    }
    const int chans = magic[1] == '6' ? 3 : 1;
    const int w = pnm_int(f), h = pnm_int(f), maxv = pnm_int(f);
    if (w <= 0 || h <= 0 || maxv != 255 || (long long)w * h > (1LL << 28)) {   // (the letterbox entry's own size limit)
        fclose(f);
        throw std::runtime_error("Bad PNM header: " + path);
    }
    std::vector<unsigned char> raw((size_t)w * h * chans);
    const size_t rd = fread(raw.data(), 1, raw.size(), f);
    fclose(f);
    if (rd != raw.size()) throw std::runtime_error("Short read: " + path);
    ImageU8 im;
    im.w = w; im.h = h;
    if (chans == 3) im.rgb = std::move(raw);
    else {
        im.rgb.resize((size_t)w * h * 3);
        for (size_t i = 0; i < (size_t)w * h; ++i) im.rgb[3 * i] = im.rgb[3 * i + 1] = im.rgb[3 * i + 2] = raw[i];
    }
    return im;
}

Image load_pnm(const std::string &path)
{
    const ImageU8 u = load_pnm_u8(path);
    // like load_image_stb (src/core/yolo_image.cpp): always 3 planes, HWC bytes -> CHW floats /255
    Image im = make_image(u.w, u.h, 3);
    for (int k = 0; k < 3; ++k)
        for (int y = 0; y < u.h; ++y)
            for (int x = 0; x < u.w; ++x) im.at(x, y, k) = (float)u.rgb[((size_t)y * u.w + x) * 3 + k] / 255.f;
    return im;
}

void save_ppm(const Image &im, const std::string &path)
{
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("Cannot write " + path);
    fprintf(f, "P6\n%d %d\n255\n", im.w, im.h);
    std::vector<unsigned char> row((size_t)im.w * 3);
    for (int y = 0; y < im.h; ++y) {
        for (int x = 0; x < im.w; ++x)
            for (int k = 0; k < 3; ++k) row[(size_t)x * 3 + k] = (unsigned char)(255 * std::min(1.f, std::max(0.f, im.at(x, y, std::min(k, im.c - 1)))));
        fwrite(row.data(), 1, row.size(), f);
    }
    fclose(f);
}

// resize_image, src/core/yolo_image.cpp:84-126: horizontal pass into `part`, then vertical pass
// accumulating the two row contributions separately (the order of the float ops is kept).
Image resize_image(const Image &im, int w, int h)
{
    Image resized = make_image(w, h, im.c), part = make_image(w, im.h, im.c);
    const float w_scale = (float)(im.w - 1) / (w - 1);
    const float h_scale = (float)(im.h - 1) / (h - 1);
    for (int k = 0; k < im.c; ++k)
        for (int r = 0; r < im.h; ++r)
            for (int c = 0; c < w; ++c) {
                float val;
                if (c == w - 1 || im.w == 1) {
                    val = im.at(im.w - 1, r, k);
                } else {
                    const float sx = c * w_scale;
                    const int ix = (int)sx;
                    const float dx = sx - ix;
                    val = (1 - dx) * im.at(ix, r, k) + dx * im.at(ix + 1, r, k);
                }
                part.at(c, r, k) = val;
            }
    for (int k = 0; k < im.c; ++k)
        for (int r = 0; r < h; ++r) {
            const float sy = r * h_scale;
            const int iy = (int)sy;
            const float dy = sy - iy;
            for (int c = 0; c < w; ++c) resized.at(c, r, k) = (1 - dy) * part.at(c, iy, k);
            if (r == h - 1 || im.h == 1) continue;
            for (int c = 0; c < w; ++c) resized.at(c, r, k) += dy * part.at(c, iy + 1, k);
        }
    return resized;
}

// letterbox_image, yolo_image.cpp:148-165
Image letterbox_image(const Image &im, int w, int h)
{
    int new_w = im.w, new_h = im.h;
    if (((float)w / im.w) < ((float)h / im.h)) {
        new_w = w;
        new_h = (im.h * w) / im.w;
    } else {
        new_h = h;
        new_w = (im.w * h) / im.h;
    }
    const Image resized = resize_image(im, new_w, new_h);
    Image boxed = make_image(w, h, im.c);
    std::fill(boxed.data.begin(), boxed.data.end(), .5f);
    const int dx = (w - new_w) / 2, dy = (h - new_h) / 2;
    for (int k = 0; k < resized.c; ++k)
        for (int y = 0; y < new_h; ++y)
            for (int x = 0; x < new_w; ++x) boxed.at(dx + x, dy + y, k) = resized.at(x, y, k);
    return boxed;
}

void draw_box(Image &im, int x1, int y1, int x2, int y2, int thick, float r, float g, float b)
{
    const float col[3] = {r, g, b};
    auto clampi = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
    for (int t = 0; t < thick; ++t) {
        const int a1 = clampi(x1 + t, 0, im.w - 1), a2 = clampi(x2 - t, 0, im.w - 1);
        const int b1 = clampi(y1 + t, 0, im.h - 1), b2 = clampi(y2 - t, 0, im.h - 1);
        for (int k = 0; k < std::min(3, im.c); ++k) {
            for (int x = a1; x <= a2; ++x) { im.at(x, b1, k) = col[k]; im.at(x, b2, k) = col[k]; }
            for (int y = b1; y <= b2; ++y) { im.at(a1, y, k) = col[k]; im.at(a2, y, k) = col[k]; }
        }
    }
}

// -------------------------------------------------------------------------------------- weights

namespace {
template <typename T>
std::vector<T> read_binary(const std::string &path)  // read_binary, yolo2_model.cpp:132-148
{
    FILE *fp = fopen(path.c_str(), "rb");
    if (!fp) throw std::runtime_error("Failed to open file: " + path);
    fseek(fp, 0, SEEK_END);
    const long sz = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    if (sz < 0 || sz % (long)sizeof(T) != 0) {
        fclose(fp);
        throw std::runtime_error("Invalid size for file: " + path);
    }
    std::vector<T> buf((size_t)sz / sizeof(T));
    const size_t rd = fread(buf.data(), sizeof(T), buf.size(), fp);
    fclose(fp);
    if (rd != buf.size()) throw std::runtime_error("Short read: " + path);
    return buf;
}

std::vector<int16_t> strip_pads(const std::vector<int16_t> &file, const std::vector<int> &len, const char *what)
{
    std::vector<int16_t> out;
    size_t fo = 0;
    for (size_t l = 0; l < len.size(); ++l) {
        if (fo + (size_t)len[l] > file.size()) throw std::runtime_error(std::string("int16 ") + what + " truncated at layer " + std::to_string(l));
        out.insert(out.end(), file.begin() + fo, file.begin() + fo + len[l]);
        fo += (size_t)len[l] + (len[l] & 1);  // yolo2_model.cpp:215-220
    }
    return out;
}
}  // namespace

WeightsI16 load_weights_int16(const std::string &dir, const std::vector<int> &wlen, const std::vector<int> &blen)
{
    WeightsI16 wp;
    wp.weights = strip_pads(read_binary<int16_t>(dir + "/weights_reorg_int16.bin"), wlen, "weight");
    wp.bias = strip_pads(read_binary<int16_t>(dir + "/bias_int16.bin"), blen, "bias");
    wp.weight_q = read_binary<int32_t>(dir + "/weight_int16_Q.bin");
    wp.bias_q = read_binary<int32_t>(dir + "/bias_int16_Q.bin");
    if (wp.weight_q.size() < wlen.size() || wp.bias_q.size() < wlen.size())
        throw std::runtime_error("Q tables too small for conv layers");
    try {
        wp.act_q = read_binary<int32_t>(dir + "/iofm_Q.bin");  // optional in the loader, required for int16
    } catch (...) {
        wp.act_q.clear();
    }
    if (wp.act_q.empty()) throw std::runtime_error("Activation Q table (iofm_Q.bin) is required for int16 inference.");
    return wp;
}

// --------------------------------------------------------------------------------------- region

namespace {
inline int entry_index(const Layer &l, int location, int entry)  // yolo_region.cpp:11-16, batch 0
{
    const int wh = l.w * l.h;
    const int n = location / wh, loc = location % wh;
    return n * wh * (4 + l.classes + 1) + entry * wh + loc;
}
inline float logistic(float x) { return (float)(1. / (1. + std::exp(-(double)x))); }  // yolo_math.cpp:19
}  // namespace

void region_forward(const Layer &l, const float *in, float *out)
{
    const int wh = l.w * l.h, total = wh * l.num * (l.coords + l.classes + 1);
    memcpy(out, in, sizeof(float) * total);  // yolo_region.cpp:125
    for (int n = 0; n < l.num; ++n) {
        int index = entry_index(l, n * wh, 0);
        for (int i = 0; i < 2 * wh; ++i) out[index + i] = logistic(out[index + i]);
        index = entry_index(l, n * wh, l.coords);
        if (!l.background)
            for (int i = 0; i < wh; ++i) out[index + i] = logistic(out[index + i]);
    }
    if (l.softmax) {  // softmax_cpu(net_input + index, classes+background, num, inputs/num, wh, 1, wh, 1, out + index)
        const int index = entry_index(l, 0, l.coords + !l.background);
        const int n = l.classes + l.background, batch_offset = total / l.num;
        for (int b = 0; b < l.num; ++b)
            for (int g = 0; g < wh; ++g) {
                const float *ip = in + index + b * batch_offset + g;
                float *op = out + index + b * batch_offset + g;
                float sum = 0, largest = -FLT_MAX;  // softmax(), yolo_math.cpp:226-241
                for (int i = 0; i < n; ++i)
                    if (ip[i * wh] > largest) largest = ip[i * wh];
                for (int i = 0; i < n; ++i) {
                    const float e = (float)std::exp((double)(ip[i * wh] / 1.f - largest / 1.f));
                    sum += e;
                    op[i * wh] = e;
                }
                for (int i = 0; i < n; ++i) op[i * wh] /= sum;
            }
    }
}

std::vector<Detection> region_boxes(const Layer &l, const float *out, int im_w, int im_h, int net_w, int net_h, float thresh)
{
    const int wh = l.w * l.h;
    std::vector<Detection> dets((size_t)wh * l.num);  // make_network_boxes: all slots, zero-initialised
    for (auto &d : dets) d.prob.assign(l.classes, 0.f);
    int count = 0;
    for (int i = 0; i < wh; ++i) {  // get_region_detections, yolo_region.cpp:170-197
        const int row = i / l.w, col = i % l.w;
        for (int n = 0; n < l.num; ++n) {
            const int obj_index = entry_index(l, n * wh + i, l.coords);
            if (out[obj_index] <= thresh) continue;
            const int box_index = entry_index(l, n * wh + i, 0);
            Detection &d = dets[count];
            // get_region_box, yolo_region.cpp:18-26 (std::exp on float)
            d.bbox.x = (col + out[box_index + 0 * wh]) / l.w;
            d.bbox.y = (row + out[box_index + 1 * wh]) / l.h;
            d.bbox.w = std::exp(out[box_index + 2 * wh]) * l.anchors[2 * n] / l.w;
            d.bbox.h = std::exp(out[box_index + 3 * wh]) * l.anchors[2 * n + 1] / l.h;
            d.objectness = out[obj_index];
            for (int j = 0; j < l.classes; ++j) {
                const float prob = d.objectness * out[entry_index(l, n * wh + i, l.coords + 1 + j)];
                d.prob[j] = prob > thresh ? prob : 0;
            }
            ++count;
        }
    }
    // correct_region_boxes, yolo_region.cpp:28-54, relative = 1
    int new_w, new_h;
    if (((float)net_w / im_w) < ((float)net_h / im_h)) {
        new_w = net_w;
        new_h = (im_h * net_w) / im_w;
    } else {
        new_h = net_h;
        new_w = (im_w * net_h) / im_h;
    }
    for (int i = 0; i < count; ++i) {
        Box b = dets[i].bbox;
        b.x = (b.x - (net_w - new_w) / 2. / net_w) / ((float)new_w / net_w);
        b.y = (b.y - (net_h - new_h) / 2. / net_h) / ((float)new_h / net_h);
        b.w *= (float)net_w / new_w;
        b.h *= (float)net_h / new_h;
        dets[i].bbox = b;
    }
    return dets;
}

namespace {
float overlap(float x1, float w1, float x2, float w2)  // yolo_post.cpp:21-30
{
    const float l1 = x1 - w1 / 2, l2 = x2 - w2 / 2;
    const float left = l1 > l2 ? l1 : l2;
    const float r1 = x1 + w1 / 2, r2 = x2 + w2 / 2;
    const float right = r1 < r2 ? r1 : r2;
    return right - left;
}
float box_intersection(const Box &a, const Box &b)
{
    const float w = overlap(a.x, a.w, b.x, b.w), h = overlap(a.y, a.h, b.y, b.h);
    if (w < 0 || h < 0) return 0;
    return w * h;
}
}  // namespace

float box_iou(const Box &a, const Box &b)
{
    const float i = box_intersection(a, b);
    const float u = a.w * a.h + b.w * b.h - i;
    return i / u;
}

int nms_sort(std::vector<Detection> &dets, int classes, float thresh)
{
    // do_nms_sort, yolo_post.cpp:54-85: move objectness==0 to the back, then per class sort + suppress
    int total = (int)dets.size(), k = total - 1;
    for (int i = 0; i <= k; ++i)
        if (dets[i].objectness == 0) {
            std::swap(dets[i], dets[k]);
            --k;
            --i;
        }
    total = k + 1;
    for (int c = 0; c < classes; ++c) {
        // qsort with nms_comparator (descending prob[c]); stable here, ties keep their order
        std::stable_sort(dets.begin(), dets.begin() + total,
                         [c](const Detection &a, const Detection &b) { return a.prob[c] > b.prob[c]; });
        for (int i = 0; i < total; ++i) {
            if (dets[i].prob[c] == 0) continue;
            for (int j = i + 1; j < total; ++j)
                if (box_iou(dets[i].bbox, dets[j].bbox) > thresh) dets[j].prob[c] = 0;
        }
    }
    return total;
}

// the region layer of config/yolov2.cfg (13x13, 5 anchors, 80 classes, softmax)
Layer yolo2_region_layer()
{
    Layer l;
    l.type = REGION; l.w = 13; l.h = 13; l.num = 5; l.classes = 80; l.coords = 4; l.softmax = true;
    l.anchors = {0.57273f, 0.677385f, 1.87446f, 2.06253f, 3.33843f, 5.47434f, 7.88282f, 3.52778f, 9.77052f, 9.16828f};
    return l;
}

std::vector<std::vector<Detection>> postprocess_batch(const int16_t *region, int batch, int final_q, const int *im_w,
                                                      const int *im_h, float thresh, float nms, int threads)
{
    const Layer l = yolo2_region_layer();
    const size_t elems = (size_t)l.num * (l.coords + 1 + l.classes) * l.h * l.w;
    std::vector<std::vector<Detection>> out((size_t)std::max(batch, 0));
    const float scale = std::ldexp(1.0f, -final_q);
    std::atomic<int> next{0};
    auto worker = [&]() {
        std::vector<float> raw(elems), proc(elems);
        for (int f = next.fetch_add(1); f < batch; f = next.fetch_add(1)) {
            const int16_t *r = region + (size_t)f * elems;
            for (size_t t = 0; t < elems; ++t) raw[t] = (float)r[t] * scale;
            region_forward(l, raw.data(), proc.data());
            std::vector<Detection> d = region_boxes(l, proc.data(), im_w[f], im_h[f], 416, 416, thresh);
            if (nms > 0) d.resize((size_t)nms_sort(d, l.classes, nms));
            out[(size_t)f] = std::move(d);
        }
    };
    threads = std::max(1, std::min(threads, batch));
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(worker);
    worker();
    for (auto &t : pool) t.join();
    return out;
}

std::vector<std::string> load_names(const std::string &path)
{
    std::ifstream in(path);
    if (!in) throw std::runtime_error("Could not open names file: " + path);
    std::vector<std::string> v;
    std::string line;
    while (std::getline(in, line)) {
        while (!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back();
        if (!line.empty()) v.push_back(line);
    }
    if (v.empty()) throw std::runtime_error("Names file " + path + " is empty");
    return v;
}

}  // namespace y2h
