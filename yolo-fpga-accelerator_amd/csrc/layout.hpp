// layout.hpp -- HBM layout of activations and weights on the GPU side of the boundary.
//
// The reference keeps feature maps as [C][H][W8] planes (hls/models/yolov2/yolo2_accel.cpp:89-99)
// and gathers a zero-padded 4-channel halo tile per step (hls/core/core_io.cpp:82-138).  On
// MI355X the numerically fixed unit is the 4-input-channel group (Tn=4), so the
// batched path stores one *item* = 4 channels x int16 = 8 bytes per pixel:
//
//   act[cg][b][row -1 .. H-1][col 0 .. W]      (item = short4, cg = channel/4)
//
// * one shared zero column (col W) between rows and one shared zero row (row -1) between
//   planes replace the halo/padding logic: in flat item space a 3x3 conv is the 9-offset
//   stencil {-Wp-1 .. +Wp+1}, Wp = W+1, and every out-of-image tap reads a stored zero;
// * planes of one channel group are contiguous across the batch, so a workgroup's input
//   tile (a run of consecutive pixels, possibly spanning frames) is ONE contiguous copy;
// * the kernels only ever write real pixels, so the zeros written at allocation stay valid.
//
// Weights keep the reference's stream order (src/models/yolov2/yolov2_weight_gen.cpp:43-67)
// with partial tiles padded to full ones:  wpk[mb][cg][tap][32][4] int16, so the 8 output
// channels x 4 input channels a wavefront needs for one tap are 64 contiguous bytes.
#pragma once
#include <cstdint>

namespace y2 {

constexpr int kLead = 64;     // items before plane 0 (tile halo of the first pixel reaches -1)
constexpr int kTail = 1024;   // items after the last plane (halo of the last pixel + tile slack)
constexpr int kTn = 4;        // hls/core/params.hpp Tn
constexpr int kTm = 32;       // hls/core/params.hpp Tm

struct ActGeom {
    int C, CG, H, W, Wp, PL, B;
    long cg_stride;  // items between channel groups = B * PL
    long items;      // total items incl. lead/tail
};

inline ActGeom make_geom(int C, int H, int W, int B)
{
    ActGeom g;
    g.C = C;
    g.CG = (C + kTn - 1) / kTn;
    g.H = H;
    g.W = W;
    g.Wp = W + 1;
    g.PL = (H + 1) * g.Wp;
    g.B = B;
    g.cg_stride = (long)B * g.PL;
    g.items = kLead + (long)g.CG * g.cg_stride + kTail;
    return g;
}

// item index of pixel (y, x) of frame b in channel group cg
inline long item_index(const ActGeom &g, int cg, int b, int y, int x)
{
    return kLead + (long)cg * g.cg_stride + (long)b * g.PL + (long)(y + 1) * g.Wp + x;
}

// Upper bound of the LDS tile length (items) of a run of T consecutive pixels (+ halo).
inline int tile_items_bound(const ActGeom &g, int T, int halo)
{
    const int rows = (T - 1) / g.W + 1;            // row boundaries crossed: one pad column each
    const int frames = (T - 1) / (g.H * g.W) + 1;  // frame boundaries crossed: one pad row each
    return T + rows + frames * g.Wp + 2 * halo + 1;
}

// Division by a launch constant as multiply-high + shift.  The generic 32-bit division the compiler emits is ~30 VALU
// instructions; a conv workgroup's prologue needs 6-14 of them (pixel index -> frame / row / column), which for the layers with
// few input channels is a fifth of the workgroup's life.
// Granlund-Montgomery round-up division: exact for every 32-bit numerator, divisor >= 2
inline void fast_div_magic(unsigned d, unsigned &m, unsigned &s)
{
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    m = (unsigned)((((1ull << l) - d) << 32) / d + 1);
    s = l - 1;
}
#if defined(__HIPCC__)
__device__ __forceinline__ unsigned fast_div(unsigned n, unsigned m, unsigned s)
{
    const unsigned t = __umulhi(m, n);
    return (t + ((n - t) >> 1)) >> s;
}
#endif

}  // namespace y2
