// kernels_pre.hpp -- GPU pre-processing: camera bytes -> letterboxed float frame.
//
// The step before the hot path in the reference's host:
//   load_image_stb   src/core/yolo_image.cpp:29-63    HWC bytes -> CHW float, v / 255.f
//   resize_image     src/core/yolo_image.cpp:84-146   two-pass bilinear (columns, then rows)
//   letterbox_image  src/core/yolo_image.cpp:148-165  aspect-preserving fit on a 0.5 canvas
// One thread per canvas element recomputes the two "part" values it needs instead of storing the
// intermediate image; every float operation is the reference's, in the reference's order, with
// no contraction (-ffp-contract=off, explicit _rn intrinsics), so the frame is bit-identical to
// the host code's and the int16 network result after it is too.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace y2 {

struct LetterboxArgs {
    int w, h, ch;          // source image: w x h, ch interleaved byte channels (1 or 3)
    int net_w, net_h;      // canvas (416 x 416)
    int new_w, new_h;      // fitted size
    int off_x, off_y;      // where the fitted image sits on the canvas
    float w_scale, h_scale;
};

// the horizontally interpolated value part(c, r, k) of resize_image's first pass
__device__ inline float lb_part(const uint8_t *__restrict__ img, const LetterboxArgs &a, int c, int r, int k)
{
    const int kk = a.ch == 3 ? k : 0;
    const uint8_t *row = img + ((size_t)r * a.w) * a.ch + kk;
    if (c == a.new_w - 1 || a.w == 1) return __fdiv_rn((float)row[(size_t)(a.w - 1) * a.ch], 255.f);
    const float sx = __fmul_rn((float)c, a.w_scale);
    const int ix = (int)sx;
    const float dx = __fsub_rn(sx, (float)ix);
    const float p0 = __fdiv_rn((float)row[(size_t)min(ix, a.w - 1) * a.ch], 255.f);
    const float p1 = __fdiv_rn((float)row[(size_t)min(ix + 1, a.w - 1) * a.ch], 255.f);   // (clamp: memory safety only)
    return __fadd_rn(__fmul_rn(__fsub_rn(1.f, dx), p0), __fmul_rn(dx, p1));
}

__device__ inline float lb_value(const uint8_t *__restrict__ img, const LetterboxArgs &a, int t)
{
    const int plane = a.net_w * a.net_h;
    const int k = t / plane, rem = t - k * plane;
    const int y = rem / a.net_w, x = rem - y * a.net_w;
    const int c = x - a.off_x, r = y - a.off_y;
    float v = .5f;
    if (c >= 0 && c < a.new_w && r >= 0 && r < a.new_h) {
        const float sy = __fmul_rn((float)r, a.h_scale);
        const int iy = (int)sy;
        const float dy = __fsub_rn(sy, (float)iy);
        v = __fmul_rn(__fsub_rn(1.f, dy), lb_part(img, a, c, min(iy, a.h - 1), k));
        if (!(r == a.new_h - 1 || a.h == 1)) v = __fadd_rn(v, __fmul_rn(dy, lb_part(img, a, c, min(iy + 1, a.h - 1), k)));
    }
    return v;
}

__global__ void k_letterbox_u8(const uint8_t *__restrict__ img, float *__restrict__ out, const LetterboxArgs a)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 3 * a.net_w * a.net_h) return;
    out[t] = lb_value(img, a, t);
}

// A whole chunk of images in one launch (the streaming entries: 64 launches of 6 us each were 0.4 ms of every 20 ms batch on
// the stream the network waits on).  `base` starts with the chunk's table - one item per frame: where the image's bytes start
// (relative to base) and its geometry - and the images follow; blockIdx.y = frame.
struct LetterboxItem {
    unsigned long long off;
    LetterboxArgs a;
};

__global__ void k_letterbox_u8_batch(const uint8_t *__restrict__ base, float *__restrict__ out, int frame_elems)
{
    const LetterboxItem *it = reinterpret_cast<const LetterboxItem *>(base) + blockIdx.y;
    const LetterboxArgs a = it->a;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 3 * a.net_w * a.net_h) return;
    out[(size_t)blockIdx.y * frame_elems + t] = lb_value(base + it->off, a, t);
}

}  // namespace y2
