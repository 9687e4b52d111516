// y2_host_capi.cpp -- extern "C" doorway into the host logic for the Python tests
// (libyolo2_host.so).  No GPU involved.
#include <algorithm>
#include <cstring>
#include <exception>
#include <string>

#include "y2_host.hpp"

using namespace y2h;

static thread_local std::string g_err;

extern "C" {
const char *y2h_last_error() { return g_err.c_str(); }

// desc[i*12 + {type,c,h,w,out_c,out_h,out_w,n,size,stride,pad,leaky}]; returns layer count or -1
int y2h_parse_cfg(const char *path, int *net_whc, int *desc, int max_layers, float *anchors10, int *region_params)
{
    try {
        Network n = parse_cfg(path);
        net_whc[0] = n.w; net_whc[1] = n.h; net_whc[2] = n.c;
        int i = 0;
        for (const Layer &l : n.layers) {
            if (i >= max_layers) break;
            int v[12] = {l.type, l.c, l.h, l.w, l.out_c, l.out_h, l.out_w, l.n, l.size, l.stride, l.pad, (int)l.leaky};
            memcpy(desc + i * 12, v, sizeof(v));
            if (l.type == REGION) {
                for (size_t k = 0; k < l.anchors.size() && k < 10; ++k) anchors10[k] = l.anchors[k];
                region_params[0] = l.classes; region_params[1] = l.coords; region_params[2] = l.num; region_params[3] = l.softmax;
            }
            ++i;
        }
        return (int)n.layers.size();
    } catch (const std::exception &e) { g_err = e.what(); return -1; }
}

void y2h_letterbox(const float *chw, int w, int h, int c, int nw, int nh, float *out)
{
    Image im = make_image(w, h, c);
    memcpy(im.data.data(), chw, sizeof(float) * (size_t)w * h * c);
    Image o = letterbox_image(im, nw, nh);
    memcpy(out, o.data.data(), sizeof(float) * (size_t)nw * nh * c);
}

void y2h_region_forward(const float *in, float *out) { region_forward(yolo2_region_layer(), in, out); }

// out rows: [x, y, w, h, objectness, prob[80]] for the `kept` detections in front; returns kept
int y2h_boxes_nms(const float *region_proc, int im_w, int im_h, float thresh, float nms, float *out, int max_rows)
{
    const Layer l = yolo2_region_layer();
    std::vector<Detection> d = region_boxes(l, region_proc, im_w, im_h, 416, 416, thresh);
    int total = (int)d.size();
    if (nms > 0) total = nms_sort(d, l.classes, nms);
    int rows = total < max_rows ? total : max_rows;
    for (int i = 0; i < rows; ++i) {
        float *r = out + (size_t)i * 85;
        r[0] = d[i].bbox.x; r[1] = d[i].bbox.y; r[2] = d[i].bbox.w; r[3] = d[i].bbox.h; r[4] = d[i].objectness;
        memcpy(r + 5, d[i].prob.data(), sizeof(float) * 80);
    }
    return total;
}

// rows_out: [batch][max_rows][85] = {x, y, w, h, objectness, prob[80]}; totals[f] = detections kept for frame f
int y2h_postprocess_batch(const int16_t *region, int batch, int final_q, const int *im_w, const int *im_h, float thresh,
                          float nms, int threads, float *rows_out, int max_rows, int *totals)
{
    try {
        auto all = postprocess_batch(region, batch, final_q, im_w, im_h, thresh, nms, threads);
        for (int f = 0; f < batch; ++f) {
            const auto &d = all[(size_t)f];
            totals[f] = (int)d.size();
            const int rows = std::min((int)d.size(), max_rows);
            for (int i = 0; i < rows; ++i) {
                float *r = rows_out + ((size_t)f * max_rows + i) * 85;
                r[0] = d[i].bbox.x; r[1] = d[i].bbox.y; r[2] = d[i].bbox.w; r[3] = d[i].bbox.h; r[4] = d[i].objectness;
                memcpy(r + 5, d[i].prob.data(), sizeof(float) * 80);
            }
        }
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return -1; }
}

int y2h_load_pnm(const char *path, int *whc, float *out, long capacity)
{
    try {
        Image im = load_pnm(path);
        whc[0] = im.w; whc[1] = im.h; whc[2] = im.c;
        if ((long)im.data.size() > capacity) return -2;
        memcpy(out, im.data.data(), sizeof(float) * im.data.size());
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return -1; }
}

// decode_image (y2_codec.cpp): JPEG / PNG bytes -> RGB bytes [h][w][3]; returns w*h*3, or -1 (y2h_last_error) / -2 (capacity)
long y2h_decode_image(const unsigned char *data, long n, int *w, int *h, unsigned char *rgb, long cap)
{
    try {
        const ImageU8 im = decode_image(data, (size_t)n, "<memory>");
        *w = im.w; *h = im.h;
        if ((long)im.rgb.size() > cap) return -2;
        memcpy(rgb, im.rgb.data(), im.rgb.size());
        return (long)im.rgb.size();
    } catch (const std::exception &e) { g_err = e.what(); return -1; }
}
}
