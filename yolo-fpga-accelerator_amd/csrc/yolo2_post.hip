// yolo2_post.hip -- the step after the path on the GPU: region-layer activations, box decode and
// per-class NMS for a whole batch, producing the reference's detection rows.
//
// What it restates (paths relative to the reference repository):
//   forward_region_layer   src/core/yolo_region.cpp:123-141   logistic on x, y, objectness (yolo_math.cpp:19: computed in
//                                                             double), softmax over the 80 classes with stride w*h
//                                                             (yolo_math.cpp:226-241: e = (float)exp(double), float sum in
//                                                             class order, float divide)
//   get_region_detections  src/core/yolo_region.cpp:170-197   cells in raster order, anchors inner; a slot per candidate with
//   get_region_box         :18-26                             objectness > thresh; expf on the raw w/h entries
//   correct_region_boxes   :28-54                             letterbox correction, partly in double
//   do_nms_sort            src/core/yolo_post.cpp:54-85       per class: sort by prob (glibc's qsort is a stable merge sort here:
//                                                             ties keep the order the previous class left), greedy suppression
// Bit-exactness.  The int16 path's region tensor takes only 65,536 values (int16 x 2^-Q), so every transcendental the
// reference applies to it is a TABLE computed on the host with the host's own libm in the reference's own expression -
// the GPU result is identical to the CPU's by construction, not "within an ulp".  Everything else is +, -, *, / in float or
// double in the reference's order (the library is built with -ffp-contract=off; divisions are IEEE).  The float entry
// (fp16 / fp32 region tensors) evaluates exp on the device instead: same formulas, results within 1 ulp of the host's.
//
// Kernels (one workgroup per frame, frames are independent):
//   k_region_rows   activations + candidate compaction (ordered) + box decode + class probabilities -> rows [845][85]
//   k_nms_rows      the exact sort/suppress sequence over the 80 classes, order kept in LDS
//   k_compact_dets  (frame, det, class, prob, box) records for prob > 0, in the order the reference prints them
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <type_traits>
#include <vector>

#include "y2_internal.hpp"

extern "C" int yolo2_hip_set_error(int code, const char *msg);

namespace {

constexpr int kW = 13, kH = 13, kWH = 169, kNum = 5, kClasses = 80, kEntries = 85, kDets = kWH * kNum;   // 845
__constant__ float c_anchors[10] = {0.57273f, 0.677385f, 1.87446f, 2.06253f, 3.33843f, 5.47434f, 7.88282f, 3.52778f, 9.77052f, 9.16828f};

int pfail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    return yolo2_hip_set_error(code, buf);
}
#define HIPP_TRY(expr, code)                                                                      \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return pfail(code, "%s failed: %s", #expr, hipGetErrorString(e_));  \
    } while (0)

struct Luts {          // per (device, Q): device pointers to three 65536-entry float tables
    float *logistic = nullptr;   // [v + 32768] = (float)(1. / (1. + exp(-(double)(v * 2^-Q))))
    float *softexp = nullptr;    // [d]         = (float)exp((double)(-(d * 2^-Q)))        d = largest - value, 0..65535
    float *expf_ = nullptr;      // [v + 32768] = expf(v * 2^-Q)
};
std::mutex g_mu;        // guards the table cache
std::map<std::pair<int, int>, Luts> g_luts;

struct FrameGeom {     // per-frame letterbox correction constants (correct_region_boxes), computed on the host
    double off_x, off_y;         // (netw - new_w) / 2. / netw
    float sx, sy;                // (float)new_w / netw
    float mw, mh;                // (float)netw / new_w
};

template <typename T>
struct In;
template <>
struct In<short> {
    const short *p; const float *lg, *se, *ex; float scale;
    __device__ float raw(int i) const { return (float)p[i] * scale; }                       // yolo2_model.cpp:415-417
    __device__ float logistic(int i) const { return lg[(int)p[i] + 32768]; }
    __device__ float expw(int i) const { return ex[(int)p[i] + 32768]; }
    __device__ int key(int i) const { return p[i]; }
    __device__ float softe(int i, int largest_key, float) const { return se[largest_key - (int)p[i]]; }
};
template <>
struct In<float> {
    const float *p; const float *lg, *se, *ex; float scale;
    __device__ float raw(int i) const { return p[i]; }
    __device__ float logistic(int i) const { return (float)(1. / (1. + exp(-(double)p[i]))); }
    __device__ float expw(int i) const { return expf(p[i]); }
    __device__ int key(int) const { return 0; }
    __device__ float softe(int i, int, float largest) const { return (float)exp((double)(p[i] / 1.f - largest / 1.f)); }
};

// rows: [845][85] = x, y, w, h, objectness, prob[80]; rows at and past `total` are zero (calloc'ed slots of make_network_boxes)
template <typename T>
__global__ __launch_bounds__(256) void k_region_rows(In<T> in0, int batch, const FrameGeom *__restrict__ geom, float thresh,
                                                     float *__restrict__ rows, int *__restrict__ totals, float *__restrict__ proc)
{
    __shared__ int flag[kDets + 3];
    __shared__ int rank[kDets + 3];
    __shared__ int wsum[4];
    const int f = blockIdx.x, tid = threadIdx.x;
    In<T> in = in0;
    in.p += (size_t)f * kEntries * kDets;    // 425 * 169
    float *frows = rows + (size_t)f * kDets * kEntries;
    // candidate d = cell * 5 + anchor (get_region_detections' visiting order); entry e of anchor n at ((n*85 + e)*169 + cell)
    for (int d = tid; d < kDets; d += 256) {
        const int cell = d / kNum, n = d - cell * kNum;
        flag[d] = in.logistic((n * kEntries + 4) * kWH + cell) > thresh ? 1 : 0;
    }
    __syncthreads();
    // ordered exclusive scan over 845 flags: 4 per thread (threads own consecutive quads)
    {
        int v[4], s = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int d = tid * 4 + k; v[k] = d < kDets ? flag[d] : 0; s += v[k]; }
        int incl = s;
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if ((tid & 63) >= o) incl += t; }
        if ((tid & 63) == 63) wsum[tid >> 6] = incl;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < (tid >> 6); ++w) base += wsum[w];
        int run = base + incl - s;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int d = tid * 4 + k; if (d < kDets) rank[d] = run; run += v[k]; }
        if (tid == 255) totals[f] = base + incl;
    }
    __syncthreads();
    const int total = totals[f];
    // zero the unused slots
    for (int i = tid; i < (kDets - total) * kEntries; i += 256) frows[(size_t)total * kEntries + i] = 0.f;
    const FrameGeom g = geom[f];
    // one wavefront per candidate: lanes 0..79 (in two passes of 64) own the classes
    const int lane = tid & 63, wave = tid >> 6;
    for (int d = wave; d < kDets; d += 4) {
        if (!flag[d]) continue;
        const int cell = d / kNum, n = d - cell * kNum, row = cell / kW, col = cell - row * kW;
        const int base = n * kEntries * kWH + cell;
        float *r = frows + (size_t)rank[d] * kEntries;
        const float obj = in.logistic(base + 4 * kWH);
        if (lane == 0) {
            float bx = ((float)col + in.logistic(base + 0 * kWH)) / (float)kW;       // get_region_box
            float by = ((float)row + in.logistic(base + 1 * kWH)) / (float)kH;
            float bw = in.expw(base + 2 * kWH) * c_anchors[2 * n] / (float)kW;
            float bh = in.expw(base + 3 * kWH) * c_anchors[2 * n + 1] / (float)kH;
            bx = (float)(((double)bx - g.off_x) / (double)g.sx);                     // correct_region_boxes, relative = 1
            by = (float)(((double)by - g.off_y) / (double)g.sy);
            bw *= g.mw;
            bh *= g.mh;
            r[0] = bx; r[1] = by; r[2] = bw; r[3] = bh; r[4] = obj;
        }
        // softmax over the 80 classes of this (anchor, cell): max, then e_j in class order with a SEQUENTIAL float sum
        float lv = -FLT_MAX; int lk = -0x7fffffff;
        for (int j = lane; j < kClasses; j += 64) {
            const int idx = base + (5 + j) * kWH;
            lv = fmaxf(lv, in.raw(idx));
            lk = max(lk, in.key(idx));
        }
        for (int o = 32; o > 0; o >>= 1) { lv = fmaxf(lv, __shfl_xor(lv, o)); lk = max(lk, __shfl_xor(lk, o)); }
        float e0 = in.softe(base + (5 + lane) * kWH, lk, lv);                                   // classes 0..63
        float e1 = lane < kClasses - 64 ? in.softe(base + (5 + 64 + lane) * kWH, lk, lv) : 0.f; // classes 64..79
        float sum = 0.f;
        for (int j = 0; j < 64; ++j) sum += __shfl(e0, j);          // the reference's order: sum += e, j = 0 .. 79
        for (int j = 0; j < kClasses - 64; ++j) sum += __shfl(e1, j);
        {
            const float p0 = obj * (e0 / sum);
            r[5 + lane] = p0 > thresh ? p0 : 0.f;
            if (lane < kClasses - 64) {
                const float p1 = obj * (e1 / sum);
                r[5 + 64 + lane] = p1 > thresh ? p1 : 0.f;
            }
        }
    }
    // optional: the whole activated tensor (l.output of forward_region_layer), for dumps and parity checks
    if (proc) {
        float *fp = proc + (size_t)f * kEntries * kDets;
        for (int i = tid; i < kNum * 5 * kWH; i += 256) {          // entries 0..4 of every anchor
            const int n = i / (5 * kWH), rem = i - n * 5 * kWH, e = rem / kWH, cell = rem - e * kWH;
            const int idx = (n * kEntries + e) * kWH + cell;
            fp[idx] = (e == 2 || e == 3) ? in.raw(idx) : in.logistic(idx);
        }
        for (int d = wave; d < kDets; d += 4) {                      // class entries: softmax for every (anchor, cell)
            const int cell = d / kNum, n = d - cell * kNum, base = n * kEntries * kWH + cell;
            float lv = -FLT_MAX; int lk = -0x7fffffff;
            for (int j = lane; j < kClasses; j += 64) { const int idx = base + (5 + j) * kWH; lv = fmaxf(lv, in.raw(idx)); lk = max(lk, in.key(idx)); }
            for (int o = 32; o > 0; o >>= 1) { lv = fmaxf(lv, __shfl_xor(lv, o)); lk = max(lk, __shfl_xor(lk, o)); }
            const float e0 = in.softe(base + (5 + lane) * kWH, lk, lv);
            const float e1 = lane < kClasses - 64 ? in.softe(base + (5 + 64 + lane) * kWH, lk, lv) : 0.f;
            float sum = 0.f;
            for (int j = 0; j < 64; ++j) sum += __shfl(e0, j);
            for (int j = 0; j < kClasses - 64; ++j) sum += __shfl(e1, j);
            fp[base + (5 + lane) * kWH] = e0 / sum;
            if (lane < kClasses - 64) fp[base + (5 + 64 + lane) * kWH] = e1 / sum;
        }
    }
}

__device__ __forceinline__ float overlap1(float x1, float w1, float x2, float w2)   // yolo_post.cpp:21-30
{
    const float l1 = x1 - w1 / 2, l2 = x2 - w2 / 2;
    const float left = l1 > l2 ? l1 : l2;
    const float r1 = x1 + w1 / 2, r2 = x2 + w2 / 2;
    const float right = r1 < r2 ? r1 : r2;
    return right - left;
}
__device__ __forceinline__ float box_iou_dev(const float4 a, const float4 b)        // .x .y = centre, .z .w = size
{
    const float w = overlap1(a.x, a.z, b.x, b.z), h = overlap1(a.y, a.w, b.y, b.w);
    const float inter = (w < 0 || h < 0) ? 0.f : w * h;
    const float uni = a.z * a.w + b.z * b.w - inter;
    return inter / uni;
}

// do_nms_sort for one frame per workgroup.  `order` (LDS) is the permutation the successive stable sorts build up;
// rows stay where k_region_rows put them and are permuted ONCE at the end (rows_out), so that the output array is the
// reference's dets[] after the last class, element for element.
__global__ __launch_bounds__(256) void k_nms_rows(float *__restrict__ rows, const int *__restrict__ totals, float nms,
                                                  float *__restrict__ rows_out)
{
    __shared__ unsigned short order[2][kDets + 3];
    __shared__ float key[kDets + 3];
    __shared__ float4 box[kDets];
    __shared__ int s_nnz;
    const int f = blockIdx.x, tid = threadIdx.x;
    float *frows = rows + (size_t)f * kDets * kEntries;
    const int total = totals[f];      // objectness > thresh >= 0, so the compaction loop of do_nms_sort leaves [0, total) as it is
    for (int p = tid; p < total; p += 256) {
        order[0][p] = (unsigned short)p;
        const float *r = frows + (size_t)p * kEntries;
        box[p] = make_float4(r[0], r[1], r[2], r[3]);
    }
    int cur = 0;
    __syncthreads();
    for (int k = 0; k < kClasses; ++k) {
        if (tid == 0) s_nnz = 0;
        __syncthreads();
        int mine = 0;
        for (int p = tid; p < total; p += 256) {
            const float v = frows[(size_t)order[cur][p] * kEntries + 5 + k];
            key[p] = v;
            mine += v != 0.f;
        }
        if (mine) atomicAdd(&s_nnz, mine);
        __syncthreads();
        const int nnz = s_nnz;
        if (nnz == 0) continue;       // all keys equal: a stable sort changes nothing, nothing to suppress (uniform branch)
        // stable descending sort by rank counting: pos(p) = #{q : key[q] > key[p]} + #{q < p : key[q] == key[p]}
        for (int p = tid; p < total; p += 256) {
            const float kp = key[p];
            int pos = 0;
            for (int q = 0; q < total; ++q) {
                const float kq = key[q];
                pos += (kq > kp) || (kq == kp && q < p);
            }
            order[cur ^ 1][pos] = order[cur][p];
        }
        __syncthreads();
        cur ^= 1;
        // keys in the new order (only the non-zero prefix matters for suppression: zeros sort last)
        for (int p = tid; p < nnz; p += 256) key[p] = frows[(size_t)order[cur][p] * kEntries + 5 + k];
        __syncthreads();
        for (int i = 0; i < nnz; ++i) {
            if (key[i] == 0.f) continue;                       // uniform: key lives in LDS
            const float4 a = box[order[cur][i]];
            for (int j = i + 1 + tid; j < nnz; j += 256)
                if (key[j] != 0.f && box_iou_dev(a, box[order[cur][j]]) > nms) {
                    key[j] = 0.f;
                    frows[(size_t)order[cur][j] * kEntries + 5 + k] = 0.f;
                }
            __syncthreads();
        }
        __syncthreads();
    }
    // the reference's array after the last sort
    float *fout = rows_out + (size_t)f * kDets * kEntries;
    for (int i = tid; i < kDets * kEntries; i += 256) {
        const int p = i / kEntries, e = i - p * kEntries;
        fout[i] = p < total ? frows[(size_t)order[cur][p] * kEntries + e] : 0.f;
    }
}

struct DetRec {   // == yolo2_hip_det
    int frame, det, cls;
    float prob, x, y, w, h;
};

// records for prob > 0 in the reference's print order (dets in array order, classes inner), at most cap per frame.
// best_only: ONE record per detection that has any prob > 0 - its best class, the first one among equals (what the reference's
// streaming app prints, linux_app/src/main.c:1040-1052); then a frame never has more than 845 records.
__global__ __launch_bounds__(256) void k_compact_dets(const float *__restrict__ rows, const int *__restrict__ totals, int cap,
                                                      DetRec *__restrict__ out, int *__restrict__ counts, int best_only)
{
    __shared__ int cnt[kDets + 3];
    __shared__ int wsum[4];
    const int f = blockIdx.x, tid = threadIdx.x;
    const float *frows = rows + (size_t)f * kDets * kEntries;
    const int total = totals[f];
    for (int p = tid; p < kDets; p += 256) {
        int c = 0;
        if (p < total)
            for (int j = 0; j < kClasses; ++j) c += frows[(size_t)p * kEntries + 5 + j] > 0.f;
        cnt[p] = best_only ? (c > 0) : c;
    }
    __syncthreads();
    int v[4], s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int d = tid * 4 + k; v[k] = d < kDets ? cnt[d] : 0; s += v[k]; }
    int incl = s;
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if ((tid & 63) >= o) incl += t; }
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < (tid >> 6); ++w) base += wsum[w];
    int run = base + incl - s;
    if (tid == 255) counts[f] = base + incl;      // may exceed cap: the caller sees the truncation
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int p = tid * 4 + k;
        if (p < total && v[k]) {
            const float *r = frows + (size_t)p * kEntries;
            int at = run;
            if (best_only) {
                int bj = 0;
                float bp = 0.f;
                for (int j = 0; j < kClasses; ++j)
                    if (r[5 + j] > bp) { bp = r[5 + j]; bj = j; }
                if (at < cap) out[(size_t)f * cap + at] = DetRec{f, p, bj, bp, r[0], r[1], r[2], r[3]};
            } else
            for (int j = 0; j < kClasses; ++j)
                if (r[5 + j] > 0.f) {
                    if (at < cap) out[(size_t)f * cap + at] = DetRec{f, p, j, r[5 + j], r[0], r[1], r[2], r[3]};
                    ++at;
                }
        }
        run += v[k];
    }
}

int get_luts(int device, int q, Luts &out)
{
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_luts.find({device, q});
    if (it != g_luts.end()) { out = it->second; return YOLO2_SUCCESS; }
    // the reference's own expressions, evaluated by the host's libm for every value an int16 x 2^-Q tensor can hold
    std::vector<float> lg(65536), se(65536), ex(65536);
    const float scale = std::ldexp(1.0f, -q);
    for (int v = -32768; v < 32768; ++v) {
        const float x = (float)v * scale;
        lg[(size_t)(v + 32768)] = (float)(1. / (1. + std::exp(-(double)x)));    // yolo_math.cpp:19
        ex[(size_t)(v + 32768)] = std::exp(x);                                    // std::exp(float): yolo_region.cpp:23-24
    }
    for (int d = 0; d < 65536; ++d) {
        // input[i]/temp - largest/temp with temp = 1: a float subtraction of two multiples of 2^-Q, exact
        const float diff = -((float)d * scale);
        se[(size_t)d] = (float)std::exp((double)diff);                            // yolo_math.cpp:234
    }
    Luts l;
    HIPP_TRY(hipSetDevice(device), YOLO2_INIT_ERROR);
    HIPP_TRY(hipMalloc((void **)&l.logistic, 65536 * 4), YOLO2_MMAP_ERROR);
    HIPP_TRY(hipMalloc((void **)&l.softexp, 65536 * 4), YOLO2_MMAP_ERROR);
    HIPP_TRY(hipMalloc((void **)&l.expf_, 65536 * 4), YOLO2_MMAP_ERROR);
    HIPP_TRY(hipMemcpy(l.logistic, lg.data(), 65536 * 4, hipMemcpyHostToDevice), YOLO2_DMA_ERROR);
    HIPP_TRY(hipMemcpy(l.softexp, se.data(), 65536 * 4, hipMemcpyHostToDevice), YOLO2_DMA_ERROR);
    HIPP_TRY(hipMemcpy(l.expf_, ex.data(), 65536 * 4, hipMemcpyHostToDevice), YOLO2_DMA_ERROR);
    g_luts[{device, q}] = l;
    out = l;
    return YOLO2_SUCCESS;
}

FrameGeom frame_geom(int w, int h)
{
    const int netw = 416, neth = 416;
    int new_w, new_h;
    if (((float)netw / w) < ((float)neth / h)) { new_w = netw; new_h = (h * netw) / w; }
    else { new_h = neth; new_w = (w * neth) / h; }
    FrameGeom g;
    g.off_x = (netw - new_w) / 2. / netw;
    g.off_y = (neth - new_h) / 2. / neth;
    g.sx = (float)new_w / netw;
    g.sy = (float)new_h / neth;
    g.mw = (float)netw / new_w;
    g.mh = (float)neth / new_h;
    return g;
}

// Device buffers of one post-processing call in flight.  The public synchronous entries keep one set per device (g_scratch, one call
// at a time per device); the streaming entries (yolo2_hip.hip) own one set per pipeline stage (y2_post_alloc / y2_post_free).
int alloc_bufs(Y2PostBufs &s, int batch, int cap)
{
    if (s.cap_frames < batch) {
        for (void *p : {(void *)s.rows, (void *)s.rows2, (void *)s.totals, (void *)s.counts, (void *)s.geom}) (void)hipFree(p);
        s.rows = s.rows2 = nullptr; s.totals = s.counts = nullptr; s.geom = nullptr; s.cap_frames = 0;
        const size_t rb = (size_t)batch * kDets * kEntries * sizeof(float);
        HIPP_TRY(hipMalloc((void **)&s.rows, rb), YOLO2_MMAP_ERROR);
        HIPP_TRY(hipMalloc((void **)&s.rows2, rb), YOLO2_MMAP_ERROR);
        HIPP_TRY(hipMalloc((void **)&s.totals, (size_t)batch * sizeof(int)), YOLO2_MMAP_ERROR);
        HIPP_TRY(hipMalloc((void **)&s.counts, (size_t)batch * sizeof(int)), YOLO2_MMAP_ERROR);
        HIPP_TRY(hipMalloc((void **)&s.geom, (size_t)batch * sizeof(FrameGeom)), YOLO2_MMAP_ERROR);
        s.cap_frames = batch;
    }
    const size_t need = (size_t)batch * (size_t)cap;
    if (s.cap_dets < need) {
        (void)hipFree(s.dets);
        s.dets = nullptr; s.cap_dets = 0;
        HIPP_TRY(hipMalloc((void **)&s.dets, need * sizeof(DetRec)), YOLO2_MMAP_ERROR);
        s.cap_dets = need;
    }
    return YOLO2_SUCCESS;
}
std::map<int, Y2PostBufs> g_scratch;
std::map<int, std::mutex> g_dev_mu;   // one synchronous call at a time PER DEVICE (the shared scratch set); devices do not wait for each other

// The three kernels, enqueued on `st`; nothing is synchronised.  geom_dev: `batch` FrameGeom records already on the device (or on
// their way: copied on `st` in front of this call).  proc_dev optional.
template <typename T>
void enqueue(const T *region, const Luts &l, int final_q, int batch, float thresh, float nms, int cap, int best_only, Y2PostBufs &s,
             float *proc_dev, hipStream_t st, float **final_rows)
{
    In<T> in{region, l.logistic, l.softexp, l.expf_, std::ldexp(1.0f, -final_q)};
    hipLaunchKernelGGL((k_region_rows<T>), dim3(batch), dim3(256), 0, st, in, batch, (const FrameGeom *)s.geom, thresh, s.rows, s.totals, proc_dev);
    float *fr = s.rows;
    if (nms > 0.f) {
        hipLaunchKernelGGL(k_nms_rows, dim3(batch), dim3(256), 0, st, s.rows, s.totals, nms, s.rows2);
        fr = s.rows2;
    }
    if (cap > 0) hipLaunchKernelGGL(k_compact_dets, dim3(batch), dim3(256), 0, st, fr, s.totals, cap, (DetRec *)s.dets, s.counts, best_only);
    if (final_rows) *final_rows = fr;
}

template <typename T>
int postprocess(yolo2_hip_ctx *ctx, uint64_t region_dev, int batch, int final_q, const int *im_w, const int *im_h, float thresh, float nms,
                yolo2_hip_det *dets, int cap_per_frame, int *counts, float *rows_host, int *totals_host, float *proc_host, void *stream)
{
    static_assert(sizeof(DetRec) == sizeof(yolo2_hip_det), "record layout");
    if (!ctx || !region_dev || !im_w || !im_h) return pfail(YOLO2_ERROR, "null argument");
    if (batch <= 0 || batch > 65536) return pfail(YOLO2_ERROR, "batch %d out of range", batch);
    if (dets && (cap_per_frame <= 0 || !counts)) return pfail(YOLO2_ERROR, "detection buffer without capacity / counts");
    if (thresh < 0.f) return pfail(YOLO2_ERROR, "negative threshold");
    if (std::is_same<T, short>::value && (final_q < -15 || final_q > 30)) return pfail(YOLO2_ERROR, "final Q %d out of range", final_q);
    const int device = yolo2_hip_ctx_device(ctx);
    HIPP_TRY(hipSetDevice(device), YOLO2_INIT_ERROR);
    {   // The kernels run on the CONTEXT's device: a region tensor that lives in another GPU's HBM (e.g. allocated with the bare
        // yolo2_hip_alloc while another device was current) would be read across xGMI at best and fault at worst.
        hipPointerAttribute_t attr;
        // (ordinary host memory is reported as hipMemoryTypeUnregistered with hipSuccess on ROCm 6+, as an error before)
        if (hipPointerGetAttributes(&attr, (const void *)(uintptr_t)region_dev) != hipSuccess ||
            (attr.type != hipMemoryTypeDevice && attr.type != hipMemoryTypeHost && attr.type != hipMemoryTypeManaged)) {
            (void)hipGetLastError();
            return pfail(YOLO2_ERROR, "region tensor address %#llx is not device-accessible memory", (unsigned long long)region_dev);
        }
        if (attr.type == hipMemoryTypeDevice && attr.device != device)
            return pfail(YOLO2_ERROR, "region tensor lives on device %d, the context runs on device %d (allocate it with yolo2_hip_alloc_on)",
                         attr.device, device);
    }
    hipStream_t st = (hipStream_t)stream;
    std::mutex *dev_mu;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        dev_mu = &g_dev_mu[device];
    }
    std::lock_guard<std::mutex> lk(*dev_mu);
    Luts l;
    if (std::is_same<T, short>::value) {
        const int rc = get_luts(device, final_q, l);
        if (rc) return rc;
    }
    const int cap = dets ? cap_per_frame : 1;
    Y2PostBufs *s;
    {
        std::lock_guard<std::mutex> lk2(g_mu);
        s = &g_scratch[device];
    }
    int rc = alloc_bufs(*s, batch, cap);
    if (rc) return rc;
    std::vector<FrameGeom> g((size_t)batch);
    for (int f = 0; f < batch; ++f) {
        if (im_w[f] <= 0 || im_h[f] <= 0) return pfail(YOLO2_ERROR, "bad image size for frame %d", f);
        g[(size_t)f] = frame_geom(im_w[f], im_h[f]);
    }
    HIPP_TRY(hipMemcpyAsync(s->geom, g.data(), (size_t)batch * sizeof(FrameGeom), hipMemcpyHostToDevice, st), YOLO2_DMA_ERROR);
    HIPP_TRY(hipStreamSynchronize(st), YOLO2_DMA_ERROR);      // g is a local
    float *proc_dev = nullptr;
    if (proc_host) HIPP_TRY(hipMalloc((void **)&proc_dev, (size_t)batch * YOLO2_REGION_ELEMS * sizeof(float)), YOLO2_MMAP_ERROR);
    float *final_rows = nullptr;
    enqueue<T>((const T *)(uintptr_t)region_dev, l, final_q, batch, thresh, nms, dets ? cap : 0, 0, *s, proc_dev, st, &final_rows);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && dets) e = hipMemcpyAsync(counts, s->counts, (size_t)batch * sizeof(int), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && dets) e = hipMemcpyAsync(dets, s->dets, (size_t)batch * cap * sizeof(DetRec), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && rows_host) e = hipMemcpyAsync(rows_host, final_rows, (size_t)batch * kDets * kEntries * sizeof(float), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && totals_host) e = hipMemcpyAsync(totals_host, s->totals, (size_t)batch * sizeof(int), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && proc_host) e = hipMemcpyAsync(proc_host, proc_dev, (size_t)batch * YOLO2_REGION_ELEMS * sizeof(float), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (proc_dev) (void)hipFree(proc_dev);
    if (e != hipSuccess) return pfail(YOLO2_ERROR, "post-processing failed: %s", hipGetErrorString(e));
    return YOLO2_SUCCESS;
}

}  // namespace

extern "C" int yolo2_hip_postprocess_int16(yolo2_hip_ctx *ctx, uint64_t region_dev, int batch, int final_q, const int *im_w, const int *im_h,
                                           float thresh, float nms, yolo2_hip_det *dets, int cap_per_frame, int *counts, float *rows,
                                           int *totals, float *proc, void *stream)
{
    return postprocess<short>(ctx, region_dev, batch, final_q, im_w, im_h, thresh, nms, dets, cap_per_frame, counts, rows, totals, proc, stream);
}

extern "C" int yolo2_hip_postprocess_f32(yolo2_hip_ctx *ctx, uint64_t region_dev, int batch, const int *im_w, const int *im_h, float thresh,
                                         float nms, yolo2_hip_det *dets, int cap_per_frame, int *counts, float *rows, int *totals,
                                         float *proc, void *stream)
{
    return postprocess<float>(ctx, region_dev, batch, 0, im_w, im_h, thresh, nms, dets, cap_per_frame, counts, rows, totals, proc, stream);
}

// ---------------------------------------------------------------------------- internal: the tail as part of a pipeline (yolo2_hip.hip)

int y2_post_alloc(int device, int batch, int cap, Y2PostBufs *b)
{
    HIPP_TRY(hipSetDevice(device), YOLO2_INIT_ERROR);
    return alloc_bufs(*b, batch, cap);
}

void y2_post_free(Y2PostBufs *b)
{
    for (void *p : {(void *)b->rows, (void *)b->rows2, (void *)b->totals, (void *)b->counts, (void *)b->geom, (void *)b->dets}) (void)hipFree(p);
    *b = Y2PostBufs();
}

size_t y2_post_geom_bytes(void) { return sizeof(FrameGeom); }

int y2_post_fill_geom(void *geom_host, const int *im_w, const int *im_h, int n)
{
    FrameGeom *g = static_cast<FrameGeom *>(geom_host);
    for (int f = 0; f < n; ++f) {
        if (im_w[f] <= 0 || im_h[f] <= 0) return pfail(YOLO2_ERROR, "bad image size for frame %d", f);
        g[f] = frame_geom(im_w[f], im_h[f]);
    }
    return YOLO2_SUCCESS;
}

int y2_post_enqueue_int16(int device, const int16_t *region_dev, int batch, int final_q, float thresh, float nms, int cap, int best_only,
                          Y2PostBufs *b, hipStream_t st)
{
    if (final_q < -15 || final_q > 30) return pfail(YOLO2_ERROR, "final Q %d out of range", final_q);
    if (batch > b->cap_frames || (size_t)batch * (size_t)cap > b->cap_dets) return pfail(YOLO2_ERROR, "post-processing buffers too small for %d frames x %d records", batch, cap);
    Luts l;
    const int rc = get_luts(device, final_q, l);
    if (rc) return rc;
    enqueue<short>(region_dev, l, final_q, batch, thresh, nms, cap, best_only, *b, nullptr, st, nullptr);
    HIPP_TRY(hipGetLastError(), YOLO2_ERROR);
    return YOLO2_SUCCESS;
}

