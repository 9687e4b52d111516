// ref_shim.cpp -- extern "C" doorway into the REFERENCE's own compiled sources.
//
// TEST INFRASTRUCTURE ONLY.  This file is ours; it contains no reference code.  It is
// compiled together with the reference's source files where they lie under /root/reference
// (see oracle/Makefile) into oracle/_ref/libref_{int16,fp32}.so, which stay out of git.
// It lets the tests call the reference's YOLO2_FPGA (hls/models/yolov2/yolo2_accel.hpp:10-17)
// and yolov2_hls_ps (:23) through ctypes to validate oracle/yolo2_oracle.c and to generate
// the fixtures under tests/golden/.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>

#include <core/yolo.h>
#include <core/precision.hpp>
#include <api.hpp>

extern "C" {

int ref_is_int16(void)
{
#ifdef INT16_MODE
    return 1;
#else
    return 0;
#endif
}

// Thin pass-through: same 27 arguments, bool as int.
void ref_YOLO2_FPGA(void *Input, void *Output, void *Weight, void *Beta, int IFM_num, int OFM_num,
                    int Ksize, int Kstride, int Input_w, int Input_h, int Output_w, int Output_h,
                    int Padding, int IsNL, int IsBN, int TM, int TN, int TR, int TC,
                    int OFM_num_bound, int mLoopsxTM, int mLoops_a1xTM, int LayerType,
                    int Qw, int Qa_in, int Qa_out, int Qb)
{
    YOLO2_FPGA((IO_Dtype *)Input, (IO_Dtype *)Output, (IO_Dtype *)Weight, (IO_Dtype *)Beta, IFM_num,
               OFM_num, Ksize, Kstride, Input_w, Input_h, Output_w, Output_h, Padding, IsNL != 0,
               IsBN != 0, TM, TN, TR, TC, OFM_num_bound, mLoopsxTM, mLoops_a1xTM, LayerType, Qw, Qa_in,
               Qa_out, Qb);
}

// Whole network through the reference's own wrapper.  The reference reads weights/*.bin
// relative to the current directory, so the caller chdir()s into a scratch dir first.
// region_proc receives net->layers[n-1].output (71,825 floats, after forward_region_layer).
// The raw (pre-activation) tensor is written by the reference itself to the file named by
// YOLO2_DUMP_REGION_RAW_CPU.
int ref_yolov2_hls_ps(const char *cfg_path, const float *input, float *region_proc)
{
    try {
        network *net = load_network(const_cast<char *>(cfg_path));
        if (!net) return -1;
        set_batch_network(net, 1);
#ifdef INT16_MODE
        yolov2_hls_ps(net, input, Precision::INT16);
#else
        yolov2_hls_ps(net, input, Precision::FP32);
#endif
        layer last = net->layers[net->n - 1];
        std::memcpy(region_proc, last.output, sizeof(float) * last.outputs);
        return last.outputs;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "ref_yolov2_hls_ps: %s\n", e.what());
        return -2;
    }
}

// ---- host logic of the reference (src/core/*.cpp), for parity tests of our own C++ host

// load_image_stb (src/core/yolo_image.cpp:167-189): decodes a JPEG/PNG with the reference's vendored stb and
// returns the decoded pixels as interleaved RGB bytes (the exact inverse of its data[src]/255. conversion) plus
// the CHW float image it builds from them.  Returns w*h*3, or -1 when the buffers are too small.
long ref_load_image_u8(const char *path, int *w, int *h, unsigned char *rgb, float *chw, long cap_elems)
{
    image im = load_image_stb(const_cast<char *>(path), 3);
    *w = im.w; *h = im.h;
    const long n = (long)im.w * im.h * 3;
    if (n > cap_elems) { free_image(im); return -1; }
    for (int k = 0; k < 3; ++k)
        for (int j = 0; j < im.h; ++j)
            for (int i = 0; i < im.w; ++i) {
                const float v = im.data[i + im.w * j + im.w * im.h * k];
                rgb[k + 3 * i + 3 * im.w * j] = (unsigned char)(v * 255.f + 0.5f);
            }
    std::memcpy(chw, im.data, sizeof(float) * n);
    free_image(im);
    return n;
}

// letterbox_image (src/core/yolo_image.cpp:148-165) on a CHW float image
void ref_letterbox(const float *chw, int w, int h, int c, int nw, int nh, float *out)
{
    image im = make_image(w, h, c);
    std::memcpy(im.data, chw, sizeof(float) * (size_t)w * h * c);
    image boxed = letterbox_image(im, nw, nh);
    std::memcpy(out, boxed.data, sizeof(float) * (size_t)nw * nh * c);
    free_image(im);
    free_image(boxed);
}

// forward_region_layer + get_network_boxes + do_nms_sort (src/core/yolo_region.cpp:123-236,
// src/core/yolo_post.cpp:54-85).  rows: all w*h*n detections as [x,y,w,h,objectness,prob[classes]].
int ref_detect(const char *cfg_path, const float *region_raw, int im_w, int im_h, float thresh, float nms,
               float *proc_out, float *rows, int max_rows)
{
    try {
        network *net = load_network(const_cast<char *>(cfg_path));
        if (!net) return -1;
        set_batch_network(net, 1);
        layer l = net->layers[net->n - 1];
        forward_region_layer(l, const_cast<float *>(region_raw));
        std::memcpy(proc_out, l.output, sizeof(float) * l.outputs);
        int nboxes = 0;
        detection *dets = get_network_boxes(net, im_w, im_h, thresh, 0.5f, 0, 1, &nboxes);
        if (nms > 0) do_nms_sort(dets, nboxes, l.classes, nms);
        const int n = nboxes < max_rows ? nboxes : max_rows;
        for (int i = 0; i < n; ++i) {
            float *r = rows + (size_t)i * (5 + l.classes);
            r[0] = dets[i].bbox.x; r[1] = dets[i].bbox.y; r[2] = dets[i].bbox.w; r[3] = dets[i].bbox.h;
            r[4] = dets[i].objectness;
            for (int j = 0; j < l.classes; ++j) r[5 + j] = dets[i].prob[j];
        }
        free_detections(dets, nboxes);
        return nboxes;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "ref_detect: %s\n", e.what());
        return -2;
    }
}

// the layer table the reference's own parser builds from a .cfg (src/core/yolo_net.cpp:218-291)
// desc[i*12 + {type,c,h,w,out_c,out_h,out_w,n,size,stride,pad,leaky}] with our type codes
int ref_parse_cfg(const char *cfg_path, int *net_whc, int *desc, int max_layers)
{
    network *net = load_network(const_cast<char *>(cfg_path));
    if (!net) return -1;
    net_whc[0] = net->w; net_whc[1] = net->h; net_whc[2] = net->c;
    for (int i = 0; i < net->n && i < max_layers; ++i) {
        layer l = net->layers[i];
        int t = l.type == CONVOLUTIONAL ? 0 : l.type == MAXPOOL ? 1 : l.type == REORG ? 2 : l.type == ROUTE ? 3 : l.type == REGION ? 4 : -1;
        int v[12] = {t, l.c, l.h, l.w, l.out_c, l.out_h, l.out_w, l.n, l.size, l.stride, l.pad, l.activation == LEAKY};
        std::memcpy(desc + i * 12, v, sizeof(v));
    }
    return net->n;
}

}  // extern "C"
