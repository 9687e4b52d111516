// y2_host.hpp -- the repo's own host side of the YOLOv2 path: Darknet .cfg parser, image I/O and
// letterbox, weight-file loading, region layer + box decode + NMS.  Plain C++17, no dependency on
// the oracle and none on the GPU library except in yolov2_detect.cpp.  Each function cites the
// reference behaviour it reproduces (paths relative to the reference repository).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace y2h {

// ---------------------------------------------------------------- cfg (src/core/yolo_net.cpp:218-291)
enum LayerType { CONV = 0, MAXPOOL = 1, REORG = 2, ROUTE = 3, REGION = 4 };  // yolo2_config.h:118-122
struct Layer {
    LayerType type = CONV;
    int c = 0, h = 0, w = 0;              // input
    int out_c = 0, out_h = 0, out_w = 0;  // output
    int n = 0, size = 0, stride = 1, pad = 0;
    bool leaky = false, batch_normalize = false;
    std::vector<int> route_layers;        // absolute indices
    // region
    int classes = 0, coords = 4, num = 0;
    bool softmax = false, background = false;
    std::vector<float> anchors;
};
struct Network {
    int w = 0, h = 0, c = 0;
    std::vector<Layer> layers;
};
Network parse_cfg(const std::string &path);   // throws std::runtime_error

// ---------------------------------------------------------------- images (src/core/yolo_image.cpp)
struct Image {
    int w = 0, h = 0, c = 0;
    std::vector<float> data;  // CHW, [0,1]
    float &at(int x, int y, int k) { return data[(size_t)k * h * w + (size_t)y * w + x]; }
    float at(int x, int y, int k) const { return data[(size_t)k * h * w + (size_t)y * w + x]; }
};
Image make_image(int w, int h, int c);
Image load_pnm(const std::string &path);                 // binary P6 (RGB) / P5 (grey), 8-bit
// the same file as interleaved bytes (what stbi_load hands load_image_stb before its /255.): RGB, grey replicated
struct ImageU8 {
    int w = 0, h = 0;
    std::vector<uint8_t> rgb;   // [h][w][3]
};
ImageU8 load_pnm_u8(const std::string &path);
// The host's own JPEG (baseline, extended-sequential, progressive) and PNG decoders (y2_codec.cpp): the bytes stbi_load(file, 3)
// of the reference's vendored stb_image v2.19 hands load_image_stb (src/core/yolo_image.cpp:167-189), reproduced byte for byte.
ImageU8 decode_jpeg(const uint8_t *data, size_t n);       // throw std::runtime_error
ImageU8 decode_png(const uint8_t *data, size_t n);
ImageU8 decode_image(const uint8_t *data, size_t n, const std::string &name);   // by signature
ImageU8 load_image_u8(const std::string &path);           // JPEG / PNG / binary PNM by signature
Image load_image(const std::string &path);                // load_image_stb(path, 3): + bytes / 255 into CHW floats
void save_ppm(const Image &im, const std::string &path);
Image resize_image(const Image &im, int w, int h);        // yolo_image.cpp:84-126 (two-pass bilinear)
Image letterbox_image(const Image &im, int w, int h);     // yolo_image.cpp:148-165 (grey 0.5 bars)
void draw_box(Image &im, int x1, int y1, int x2, int y2, int thick, float r, float g, float b);

// ---------------------------------------------------------------- weights (yolo2_model.cpp:158-227)
struct WeightsI16 {
    std::vector<int16_t> weights, bias;      // per-layer file pads stripped
    std::vector<int32_t> weight_q, bias_q, act_q;
};
WeightsI16 load_weights_int16(const std::string &dir, const std::vector<int> &wlen, const std::vector<int> &blen);

// ---------------------------------------------------------------- region + boxes + NMS
struct Box { float x, y, w, h; };
struct Detection {
    Box bbox{};
    float objectness = 0;
    std::vector<float> prob;
    int sort_class = -1;
};
// forward_region_layer (src/core/yolo_region.cpp:123-141): logistic on x,y,obj (double exp),
// softmax over classes with stride w*h reading the RAW input.  in/out: [num][coords+1+classes][h][w]
void region_forward(const Layer &l, const float *in, float *out);
// get_network_boxes -> get_region_detections + correct_region_boxes (yolo_region.cpp:170-236),
// relative = 1.  Returns w*h*num detections, objectness 0 for those at or below thresh.
std::vector<Detection> region_boxes(const Layer &l, const float *out, int im_w, int im_h, int net_w, int net_h, float thresh);
// do_nms_sort (src/core/yolo_post.cpp:54-85): drops objectness==0, per-class sort + IoU suppression.
// Returns the number of detections kept in front.
int nms_sort(std::vector<Detection> &dets, int classes, float thresh);
float box_iou(const Box &a, const Box &b);

Layer yolo2_region_layer();

// The tail of the path for a whole batch (at thousands of frames/s a single
// thread of region + NMS is the bottleneck): int16 region tensors [batch][425][13][13] with final
// Q -> dequantise (yolo2_model.cpp:415-417), region_forward, region_boxes, nms_sort per frame,
// frames spread over `threads` host threads.  Per frame exactly the single-frame functions above,
// so the results are identical to calling them in a loop.  Returns, per frame, the detections kept
// in front by nms_sort (all w*h*num of them if nms <= 0).
std::vector<std::vector<Detection>> postprocess_batch(const int16_t *region, int batch, int final_q, const int *im_w,
                                                      const int *im_h, float thresh, float nms, int threads);

std::vector<std::string> load_names(const std::string &path);

}  // namespace y2h
