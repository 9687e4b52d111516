/*
 * yolo2_oracle.c -- CPU restatement of the reference's YOLOv2 accelerator path.
 *
 * TEST INFRASTRUCTURE ONLY (see yolo2_oracle.h).  Plain C, re-entrant, no statics with state
 * (the reference keeps its tile buffers in function-local statics and is not re-entrant:
 * hls/models/yolov2/yolo2_accel.cpp:103-113, hls/core/core_scheduler.cpp:21-31).
 *
 * The convolution is written as the per-output definition of SURVEY.md section 8(a):
 * tile sizes Tm/Tr/Tc and the tile visiting order of the reference do not change results,
 * the input-channel group size Tn=4 and the order n-group -> tap(i,j) do.  That equivalence
 * is not assumed: tests/test_oracle_vs_ref.py checks it bit-for-bit against the reference
 * compiled from its own sources (oracle/_ref).
 *
 * Build with -ffp-contract=off (the reference's x86-64 build has no FMA contraction).
 */
#include "yolo2_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* hls/models/yolov2/model_config.cpp:4-10 */
const int orc_yolo2_weight_len[23] = {864, 18432, 73728, 8192, 73728, 294912, 32768, 294912,
                                      1179648, 131072, 1179648, 131072, 1179648, 4718592, 524288,
                                      4718592, 524288, 4718592, 9437184, 9437184, 32768, 11796480,
                                      435200};
const int orc_yolo2_bias_len[23] = {32, 64, 128, 64, 128, 256, 128, 256, 512, 256, 512, 256,
                                    512, 1024, 512, 1024, 512, 1024, 1024, 1024, 64, 1024, 425};

static inline int imin(int a, int b) { return a < b ? a : b; }

/* Worker threads for the per-output-channel loops (results do not depend on it).
 * 1 = the reference's own single-threaded structure. */
static int g_threads = 1;
void orc_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int orc_get_threads(void) { return g_threads; }

#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
#define ORC_SIMD __attribute__((target_clones("avx2", "default")))
#else
#define ORC_SIMD
#endif

/* One (m, 4-channel group, tap) requantise-and-saturate sweep over a row, stride 1, 32-bit
 * exact case (see the bound at the call site). */
ORC_SIMD static void row_step_i16(int32_t *restrict ar, const int16_t *restrict a0, const int16_t *restrict a1,
                                  const int16_t *restrict a2, const int16_t *restrict a3, int n,
                                  int32_t w0, int32_t w1, int32_t w2, int32_t w3, int32_t r, int s)
{
    for (int x = 0; x < n; ++x) {
        int32_t p = w0 * a0[x] + w1 * a1[x] + w2 * a2[x] + w3 * a3[x];
        int32_t v = ar[x] + ((p + r) >> s);
        v = v > 32767 ? 32767 : v;
        v = v < -32768 ? -32768 : v;
        ar[x] = v;
    }
}

ORC_SIMD static void row_step_f32(float *restrict ar, const float *restrict a0, const float *restrict a1,
                                  const float *restrict a2, const float *restrict a3, int n,
                                  float w0, float w1, float w2, float w3)
{
    for (int x = 0; x < n; ++x) {
        float ps = 0.f;
        ps += w0 * a0[x];
        ps += w1 * a1[x];
        ps += w2 * a2[x];
        ps += w3 * a3[x];
        ar[x] = ar[x] + ps;
    }
}

/* Offset of weight block (m0, n0) in weights_reorg order and its tile extents.
 * src/models/yolov2/yolov2_weight_gen.cpp:43-67 writes blocks m0-major, n0-minor with
 * TM_MIN*TN_MIN*KxK elements each and no padding; hls/core/core_io.cpp:154-198 consumes
 * them sequentially through the running Woffset. */
static inline size_t wblock_offset(int m0, int n0, int C, int N, int KK, int *tm_min, int *tn_min)
{
    *tm_min = imin(ORC_TM, N - m0);
    *tn_min = imin(ORC_TN, C - n0);
    return (size_t)m0 * C * KK + (size_t)(*tm_min) * n0 * KK;
}

/* hls/core/core_compute.cpp:191-197 */
int16_t orc_leaky_i16(int16_t x)
{
    int32_t t = x;
    if (t < 0) t = t / 10; /* C division truncates toward zero */
    if (t > 32767) t = 32767;
    if (t < -32768) t = -32768;
    return (int16_t)t;
}

/* yolo2_model.cpp:257-273 */
void orc_quantize_input(const float *in, int16_t *out, size_t n, int q_in)
{
    const float scale = ldexpf(1.0f, q_in);
    for (size_t i = 0; i < n; ++i) {
        float v = in[i] * scale;
        if (v > 32767.f) v = 32767.f;
        if (v < -32768.f) v = -32768.f;
        long long q = llroundf(v);
        if (q > 32767) q = 32767;
        if (q < -32768) q = -32768;
        out[i] = (int16_t)q;
    }
}

/* ----------------------------------------------------------------- int16 conv */

typedef struct {
    int right, left, mag;
    int64_t round;
} orc_shift;

/* core_compute.cpp:48-63: direction, magnitude capped at 30, rounding constant */
static orc_shift make_shift(int s)
{
    orc_shift sh;
    sh.right = s > 0;
    sh.left = s < 0;
    int a = sh.right ? s : (sh.left ? -s : 0);
    sh.mag = a > 30 ? 30 : a;
    sh.round = (sh.right && sh.mag > 0) ? ((int64_t)1 << (sh.mag - 1)) : 0;
    return sh;
}

static inline int64_t apply_shift(int64_t v, const orc_shift *sh)
{
    if (sh->right) return (v + sh->round) >> sh->mag;
    if (sh->left) return (int64_t)((uint64_t)v << sh->mag); /* reference: signed <<, two's complement */
    return v;
}

void orc_conv_i16(const int16_t *in, int16_t *out, const int16_t *w_reorg, const int16_t *bias,
                  int C, int N, int K, int stride, int W, int H, int OW, int OH, int pad,
                  int leaky, int Qw, int Qa_in, int Qa_out, int Qb)
{
    const int W8 = orc_w8(W), OW8 = orc_w8(OW);
    const int KK = K * K;
    const orc_shift so = make_shift(Qa_in + Qw - Qa_out); /* core_compute.cpp:48 */
    const orc_shift sb = make_shift(Qb - Qa_out);          /* core_compute.cpp:49 */

    /* Zero-extended copy of the input: core_io.cpp:63-70 substitutes pad_value=0 outside the
     * image and for missing channels, so every tap of every output sees a value. */
    const int PH = (OH - 1) * stride + K, PW = (OW - 1) * stride + K;
    const int CP = (C + ORC_TN - 1) / ORC_TN * ORC_TN;
    const size_t plane = (size_t)PH * PW;
    int16_t *ipad = (int16_t *)calloc((size_t)CP * plane, sizeof(int16_t));
    for (int c = 0; c < C; ++c)
        for (int y = 0; y < PH; ++y) {
            int sy = y - pad;
            if (sy < 0 || sy >= H) continue;
            for (int x = 0; x < PW; ++x) {
                int sx = x - pad;
                if (sx < 0 || sx >= W) continue;
                ipad[c * plane + (size_t)y * PW + x] = in[(size_t)c * H * W8 + (size_t)sy * W8 + sx];
            }
        }

#pragma omp parallel for schedule(dynamic, 1) num_threads(g_threads)
    for (int m = 0; m < N; ++m) {
        int32_t *acc = (int32_t *)malloc((size_t)OH * OW * sizeof(int32_t));
        const int m0 = m / ORC_TM * ORC_TM, tm = m - m0;
        /* bias moved to the Qa_out domain, NOT saturated (core_compute.cpp:86-97) */
        const int64_t base0 = apply_shift((int64_t)bias[m], &sb);
        int first = 1;
        for (int n0 = 0; n0 < C; n0 += ORC_TN) {
            int tm_min, tn_min;
            const size_t boff = wblock_offset(m0, n0, C, N, KK, &tm_min, &tn_min);
            for (int i = 0; i < K; ++i)
                for (int j = 0; j < K; ++j) {
                    int32_t w[ORC_TN] = {0, 0, 0, 0};
                    int64_t sumabs = 0;
                    for (int t = 0; t < tn_min; ++t) {
                        w[t] = w_reorg[boff + (size_t)(i * K + j) * tm_min * tn_min + (size_t)tm * tn_min + t];
                        sumabs += w[t] < 0 ? -(int64_t)w[t] : w[t];
                    }
                    const int16_t *a0 = ipad + (size_t)(n0 + 0) * plane + (size_t)i * PW + j;
                    const int16_t *a1 = ipad + (size_t)(n0 + 1) * plane + (size_t)i * PW + j;
                    const int16_t *a2 = ipad + (size_t)(n0 + 2) * plane + (size_t)i * PW + j;
                    const int16_t *a3 = ipad + (size_t)(n0 + 3) * plane + (size_t)i * PW + j;
                    /* 32-bit arithmetic is exact when no intermediate can leave int32:
                     * |p| <= sumabs*32768; right shift adds round; left shift multiplies;
                     * the first step adds the unsaturated bias. */
                    int64_t worst = sumabs * 32768 + so.round;
                    if (so.left) worst <<= so.mag;
                    int64_t base_mag = first ? (base0 < 0 ? -base0 : base0) : 32768;
                    const int fast = (worst + base_mag < 2147483647LL) && (base_mag < 2147483647LL);
                    if (first) {
                        /* very first (n==0,i==0,j==0) step starts from the bias, core_compute.cpp:84-97 */
                        for (int y = 0; y < OH; ++y)
                            for (int x = 0; x < OW; ++x) {
                                size_t o = (size_t)y * stride * PW + (size_t)x * stride;
                                int64_t p = (int64_t)(w[0] * (int32_t)a0[o]) + (int64_t)(w[1] * (int32_t)a1[o]) +
                                            (int64_t)(w[2] * (int32_t)a2[o]) + (int64_t)(w[3] * (int32_t)a3[o]);
                                int64_t v = base0 + apply_shift(p, &so);
                                if (v > 32767) v = 32767;
                                if (v < -32768) v = -32768;
                                acc[(size_t)y * OW + x] = (int32_t)v;
                            }
                        first = 0;
                    } else if (fast && so.right) {
                        const int32_t r = (int32_t)so.round, s = so.mag;
                        for (int y = 0; y < OH; ++y) {
                            int32_t *ar = acc + (size_t)y * OW;
                            const size_t ro = (size_t)y * stride * PW;
                            if (stride == 1) {
                                row_step_i16(ar, a0 + ro, a1 + ro, a2 + ro, a3 + ro, OW, w[0], w[1], w[2], w[3], r, s);
                                continue;
                            }
                            for (int x = 0; x < OW; ++x) {
                                size_t o = ro + (size_t)x * stride;
                                int32_t p = w[0] * a0[o] + w[1] * a1[o] + w[2] * a2[o] + w[3] * a3[o];
                                int32_t v = ar[x] + ((p + r) >> s);
                                v = v > 32767 ? 32767 : v;
                                v = v < -32768 ? -32768 : v;
                                ar[x] = v;
                            }
                        }
                    } else {
                        /* general path, 64-bit like the reference (core_compute.cpp:78-118) */
                        for (int y = 0; y < OH; ++y)
                            for (int x = 0; x < OW; ++x) {
                                size_t o = (size_t)y * stride * PW + (size_t)x * stride;
                                int64_t p = (int64_t)(w[0] * (int32_t)a0[o]) + (int64_t)(w[1] * (int32_t)a1[o]) +
                                            (int64_t)(w[2] * (int32_t)a2[o]) + (int64_t)(w[3] * (int32_t)a3[o]);
                                int64_t v = (int64_t)acc[(size_t)y * OW + x] + apply_shift(p, &so);
                                if (v > 32767) v = 32767;
                                if (v < -32768) v = -32768;
                                acc[(size_t)y * OW + x] = (int32_t)v;
                            }
                    }
                }
        }
        /* write-back with integer leaky (core_compute.cpp:175-264); columns OW..OW8-1 untouched */
        for (int y = 0; y < OH; ++y)
            for (int x = 0; x < OW; ++x) {
                int16_t v = (int16_t)acc[(size_t)y * OW + x];
                out[(size_t)m * OH * OW8 + (size_t)y * OW8 + x] = leaky ? orc_leaky_i16(v) : v;
            }
        free(acc);
    }
    free(ipad);
}

/* ------------------------------------------------------------------ fp32 conv */

void orc_conv_f32(const float *in, float *out, const float *w_reorg, const float *bias,
                  int C, int N, int K, int stride, int W, int H, int OW, int OH, int pad, int leaky)
{
    const int W8 = orc_w8(W), OW8 = orc_w8(OW);
    const int KK = K * K;
    const int PH = (OH - 1) * stride + K, PW = (OW - 1) * stride + K;
    const int CP = (C + ORC_TN - 1) / ORC_TN * ORC_TN;
    const size_t plane = (size_t)PH * PW;
    float *ipad = (float *)calloc((size_t)CP * plane, sizeof(float));
    for (int c = 0; c < C; ++c)
        for (int y = 0; y < PH; ++y) {
            int sy = y - pad;
            if (sy < 0 || sy >= H) continue;
            for (int x = 0; x < PW; ++x) {
                int sx = x - pad;
                if (sx < 0 || sx >= W) continue;
                ipad[c * plane + (size_t)y * PW + x] = in[(size_t)c * H * W8 + (size_t)sy * W8 + sx];
            }
        }
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_threads)
    for (int m = 0; m < N; ++m) {
        float *acc = (float *)malloc((size_t)OH * OW * sizeof(float));
        const int m0 = m / ORC_TM * ORC_TM, tm = m - m0;
        int first = 1;
        for (int n0 = 0; n0 < C; n0 += ORC_TN) {
            int tm_min, tn_min;
            const size_t boff = wblock_offset(m0, n0, C, N, KK, &tm_min, &tn_min);
            for (int i = 0; i < K; ++i)
                for (int j = 0; j < K; ++j) {
                    float w[ORC_TN] = {0.f, 0.f, 0.f, 0.f}; /* core_io.cpp:195 zero-fills partial tiles */
                    for (int t = 0; t < tn_min; ++t)
                        w[t] = w_reorg[boff + (size_t)(i * K + j) * tm_min * tn_min + (size_t)tm * tn_min + t];
                    const float *a0 = ipad + (size_t)(n0 + 0) * plane + (size_t)i * PW + j;
                    const float *a1 = ipad + (size_t)(n0 + 1) * plane + (size_t)i * PW + j;
                    const float *a2 = ipad + (size_t)(n0 + 2) * plane + (size_t)i * PW + j;
                    const float *a3 = ipad + (size_t)(n0 + 3) * plane + (size_t)i * PW + j;
                    for (int y = 0; y < OH; ++y) {
                        float *ar = acc + (size_t)y * OW;
                        const size_t ro = (size_t)y * stride * PW;
                        if (stride == 1 && !first) {
                            row_step_f32(ar, a0 + ro, a1 + ro, a2 + ro, a3 + ro, OW, w[0], w[1], w[2], w[3]);
                            continue;
                        }
                        for (int x = 0; x < OW; ++x) {
                            size_t o = ro + (size_t)x * stride;
                            /* core_compute.cpp:152-166: four products rounded to float, summed in
                             * order starting from 0, then added to bias (first step) or the acc */
                            float m0f = w[0] * a0[o], m1f = w[1] * a1[o], m2f = w[2] * a2[o], m3f = w[3] * a3[o];
                            float ps = 0.f;
                            ps += m0f;
                            ps += m1f;
                            ps += m2f;
                            ps += m3f;
                            float pa = first ? bias[m] : ar[x];
                            ar[x] = pa + ps;
                        }
                    }
                    first = 0;
                }
        }
        for (int y = 0; y < OH; ++y)
            for (int x = 0; x < OW; ++x) {
                float v = acc[(size_t)y * OW + x];
                if (leaky && v < 0.0f) v = v * 0.1f; /* core_compute.cpp:201-205 */
                out[(size_t)m * OH * OW8 + (size_t)y * OW8 + x] = v;
            }
        free(acc);
    }
    free(ipad);
}

/* ------------------------------------------------------------------- maxpool */

void orc_maxpool_i16(const int16_t *in, int16_t *out, int C, int K, int stride, int W, int H, int OW, int OH)
{
    const int W8 = orc_w8(W), OW8 = orc_w8(OW);
    for (int c = 0; c < C; ++c)
        for (int y = 0; y < OH; ++y)
            for (int x = 0; x < OW; ++x) {
                int16_t best = -32768; /* core_compute.cpp:288-289 */
                for (int i = 0; i < K; ++i)
                    for (int j = 0; j < K; ++j) {
                        int sy = y * stride + i, sx = x * stride + j; /* padding forced to 0: core_scheduler.cpp:72 */
                        int16_t v = (sy < H && sx < W) ? in[(size_t)c * H * W8 + (size_t)sy * W8 + sx]
                                                       : (int16_t)-32768; /* core_io.cpp:96-99 */
                        if (v > best) best = v;
                    }
                out[(size_t)c * OH * OW8 + (size_t)y * OW8 + x] = best;
            }
}

void orc_maxpool_f32(const float *in, float *out, int C, int K, int stride, int W, int H, int OW, int OH)
{
    const int W8 = orc_w8(W), OW8 = orc_w8(OW);
    const float padv = -1024 * 1024; /* core_compute.cpp:291, core_io.cpp:101 */
    for (int c = 0; c < C; ++c)
        for (int y = 0; y < OH; ++y)
            for (int x = 0; x < OW; ++x) {
                float best = padv;
                for (int i = 0; i < K; ++i)
                    for (int j = 0; j < K; ++j) {
                        int sy = y * stride + i, sx = x * stride + j;
                        float v = (sy < H && sx < W) ? in[(size_t)c * H * W8 + (size_t)sy * W8 + sx] : padv;
                        if (v > best) best = v;
                    }
                out[(size_t)c * OH * OW8 + (size_t)y * OW8 + x] = best;
            }
}

/* --------------------------------------------------------------------- reorg */

/* yolo2_model.cpp:112-129 called with (w=26, h=32*13, c=4, stride=2): out_c = 1, so
 * out[i + 26*(j + 416*k)] = x[(2*i + k%2) + 52*(2*j + k/2)] */
#define REORG_BODY(T)                                                                          \
    T *dense = (T *)malloc(sizeof(T) * 64 * 26 * 26);                                          \
    T *perm = (T *)malloc(sizeof(T) * 64 * 26 * 26);                                           \
    for (int k = 0; k < 26 * 64; ++k) /* yolo2_model.cpp:370-371: strip 26-of-32 columns */    \
        memcpy(dense + (size_t)k * 26, in + (size_t)k * 32, 26 * sizeof(T));                   \
    for (int k = 0; k < 4; ++k)                                                                \
        for (int j = 0; j < 416; ++j)                                                          \
            for (int i = 0; i < 26; ++i)                                                       \
                perm[i + 26 * (j + 416 * k)] = dense[(2 * i + k % 2) + 52 * (2 * j + k / 2)]; \
    memset(out, 0, sizeof(T) * 13 * 16 * 256); /* :374 */                                      \
    for (int k = 0; k < 13 * 256; ++k) /* :375-376: re-pad rows of 13 to 16 */                 \
        memcpy(out + (size_t)k * 16, perm + (size_t)k * 13, 13 * sizeof(T));                   \
    free(dense);                                                                               \
    free(perm);

void orc_reorg_i16(const int16_t *in, int16_t *out, int shift)
{
    REORG_BODY(int16_t)
    if (shift != 0) { /* yolo2_model.cpp:382-396 */
        for (int idx = 0; idx < 13 * 16 * 256; ++idx) {
            int32_t v = out[idx];
            if (shift > 0) v >>= shift;
            else v = (int32_t)((uint32_t)v << (-shift));
            if (v > 32767) v = 32767;
            if (v < -32768) v = -32768;
            out[idx] = (int16_t)v;
        }
    }
}

void orc_reorg_f32(const float *in, float *out)
{
    REORG_BODY(float)
}

/* ---------------------------------------------------------------- file quirk */

long orc_strip_int16_layer_pad(const int16_t *file, size_t file_elems, const int *layer_len,
                               int n_layers, int16_t *dst)
{
    size_t fo = 0, oo = 0;
    for (int l = 0; l < n_layers; ++l) {
        size_t len = (size_t)layer_len[l];
        if (fo + len > file_elems) return -1;
        memcpy(dst + oo, file + fo, len * sizeof(int16_t));
        fo += len + (len & 1); /* yolo2_model.cpp:215-220 */
        oo += len;
    }
    return (long)oo;
}

/* ------------------------------------------------------------------- network */

enum { L_CONV, L_MAX, L_ROUTE, L_REORG, L_REGION };
typedef struct {
    int type, c, h, w, n, size, stride, pad, leaky;
} orc_layer;

/* config/yolov2.cfg as parsed by src/core/yolo_net.cpp:218-291.
 * 1x1 convs: cfg pad=1 -> padding = size/2 = 0 (src/core/yolo_layers.cpp:98). */
static const orc_layer NET[32] = {
    {L_CONV, 3, 416, 416, 32, 3, 1, 1, 1},     {L_MAX, 32, 416, 416, 32, 2, 2, 0, 0},
    {L_CONV, 32, 208, 208, 64, 3, 1, 1, 1},    {L_MAX, 64, 208, 208, 64, 2, 2, 0, 0},
    {L_CONV, 64, 104, 104, 128, 3, 1, 1, 1},   {L_CONV, 128, 104, 104, 64, 1, 1, 0, 1},
    {L_CONV, 64, 104, 104, 128, 3, 1, 1, 1},   {L_MAX, 128, 104, 104, 128, 2, 2, 0, 0},
    {L_CONV, 128, 52, 52, 256, 3, 1, 1, 1},    {L_CONV, 256, 52, 52, 128, 1, 1, 0, 1},
    {L_CONV, 128, 52, 52, 256, 3, 1, 1, 1},    {L_MAX, 256, 52, 52, 256, 2, 2, 0, 0},
    {L_CONV, 256, 26, 26, 512, 3, 1, 1, 1},    {L_CONV, 512, 26, 26, 256, 1, 1, 0, 1},
    {L_CONV, 256, 26, 26, 512, 3, 1, 1, 1},    {L_CONV, 512, 26, 26, 256, 1, 1, 0, 1},
    {L_CONV, 256, 26, 26, 512, 3, 1, 1, 1},    {L_MAX, 512, 26, 26, 512, 2, 2, 0, 0},
    {L_CONV, 512, 13, 13, 1024, 3, 1, 1, 1},   {L_CONV, 1024, 13, 13, 512, 1, 1, 0, 1},
    {L_CONV, 512, 13, 13, 1024, 3, 1, 1, 1},   {L_CONV, 1024, 13, 13, 512, 1, 1, 0, 1},
    {L_CONV, 512, 13, 13, 1024, 3, 1, 1, 1},   {L_CONV, 1024, 13, 13, 1024, 3, 1, 1, 1},
    {L_CONV, 1024, 13, 13, 1024, 3, 1, 1, 1},  {L_ROUTE, 0, 0, 0, 0, 0, 0, 0, 0},
    {L_CONV, 512, 26, 26, 64, 1, 1, 0, 1},     {L_REORG, 64, 26, 26, 256, 0, 2, 0, 0},
    {L_ROUTE, 0, 0, 0, 0, 0, 0, 0, 0},         {L_CONV, 1280, 13, 13, 1024, 3, 1, 1, 1},
    {L_CONV, 1024, 13, 13, 425, 1, 1, 0, 0},   {L_REGION, 425, 13, 13, 0, 0, 0, 0, 0},
};

#define FWD_BODY(T, IS_I16)                                                                         \
    T *bufs[32];                                                                                    \
    memset(bufs, 0, sizeof(bufs));                                                                  \
    const T *cur = NULL;                                                                            \
    T *in0 = (T *)calloc((size_t)3 * 416 * 416, sizeof(T));                                         \
    T *cat = NULL; /* [1280][13][16]: reorg output placed right before conv-24 output */            \
    size_t woff = 0, boff = 0;                                                                      \
    int ord = 0;                                                                                    \
    int current_Qa = 0, route24_q = 0, pending_route_q = -1;                                        \
    (void)route24_q; (void)pending_route_q;

int orc_yolov2_forward_i16(const orc_weights_i16 *wp, const float *input,
                           int16_t *region_i16, float *region_f32, int16_t **layer_dump)
{
    if (wp->n_act_q < 1) return -1; /* yolo2_model.cpp:258-260 */
    FWD_BODY(int16_t, 1)
    orc_quantize_input(input, in0, (size_t)3 * 416 * 416, wp->act_q[0]);
    current_Qa = wp->act_q[0]; /* :290 */
    cur = in0;
    cat = (int16_t *)calloc((size_t)1280 * 13 * 16, sizeof(int16_t));
    int rc = 0;
    for (int i = 0; i < 32; ++i) {
        const orc_layer *l = &NET[i];
        const int ow = l->type == L_CONV ? (l->w - l->size + 2 * l->pad) / l->stride + 1 : l->w / 2;
        const int oh = l->type == L_CONV ? (l->h - l->size + 2 * l->pad) / l->stride + 1 : l->h / 2;
        switch (l->type) {
        case L_CONV: {
            /* yolo2_model.cpp:311-321 */
            int Qa_in = ord < wp->n_act_q ? wp->act_q[ord] : current_Qa;
            int Qa_out = ord + 1 < wp->n_act_q ? wp->act_q[ord + 1] : Qa_in;
            int Qw = ord < wp->n_weight_q ? wp->weight_q[ord] : 0;
            int Qb = ord < wp->n_bias_q ? wp->bias_q[ord] : 0;
            if (pending_route_q >= 0) Qa_in = pending_route_q;
            const int16_t *src = cur;
            if (i == 26) src = bufs[16]; /* route 25 -> layer 16 output, :94-95 */
            if (i == 29) src = cat;      /* route 28 -> concat(27, 24), :97-101 */
            int16_t *dst;
            if (i == 24) dst = cat + (size_t)256 * 13 * 16;
            else dst = (int16_t *)calloc((size_t)l->n * oh * orc_w8(ow), sizeof(int16_t));
            orc_conv_i16(src, dst, wp->weights + woff, wp->bias + boff, l->c, l->n, l->size, l->stride,
                         l->w, l->h, ow, oh, l->pad, l->leaky, Qw, Qa_in, Qa_out, Qb);
            woff += orc_yolo2_weight_len[ord];
            boff += orc_yolo2_bias_len[ord];
            current_Qa = Qa_out;                 /* :331 */
            if (i == 24) route24_q = current_Qa; /* :332-334 */
            pending_route_q = -1;
            ord++;
            bufs[i] = dst;
            cur = dst;
            break;
        }
        case L_MAX: {
            int16_t *dst = (int16_t *)calloc((size_t)l->c * oh * orc_w8(ow), sizeof(int16_t));
            orc_maxpool_i16(cur, dst, l->c, l->size, l->stride, l->w, l->h, ow, oh);
            bufs[i] = dst;
            cur = dst;
            break;
        }
        case L_REORG: {
            int shift = 0;
            if (route24_q > 0) { /* :379-399 */
                int target = route24_q < current_Qa ? route24_q : current_Qa;
                shift = current_Qa - target;
                if (shift != 0) current_Qa = target;
                pending_route_q = current_Qa;
            }
            orc_reorg_i16(cur, cat, shift);
            bufs[i] = cat;
            cur = cat;
            break;
        }
        case L_ROUTE:
            break;
        case L_REGION: {
            /* :406-421 gather 13 of 16 columns, dequantise by 2^-current_Qa */
            const float scale = ldexpf(1.0f, -current_Qa);
            for (int k = 0; k < 13 * 425; ++k)
                for (int j = 0; j < 13; ++j) {
                    int16_t v = cur[(size_t)k * 16 + j];
                    if (region_i16) region_i16[(size_t)k * 13 + j] = v;
                    if (region_f32) region_f32[(size_t)k * 13 + j] = (float)v * scale;
                }
            rc = current_Qa;
            break;
        }
        }
    }
    if (layer_dump) {
        for (int i = 0; i < 32; ++i) layer_dump[i] = NULL;
        for (int i = 0; i < 32; ++i) {
            if (!bufs[i]) continue;
            const orc_layer *l = &NET[i];
            int oh = l->type == L_REORG ? 13 : (l->type == L_CONV ? l->h : l->h / 2);
            int ow = l->type == L_REORG ? 13 : (l->type == L_CONV ? l->w : l->w / 2);
            int oc = l->type == L_MAX ? l->c : l->n;
            const int16_t *src = (i == 24) ? cat + (size_t)256 * 13 * 16 : bufs[i];
            size_t n = (size_t)oc * oh * orc_w8(ow);
            layer_dump[i] = (int16_t *)malloc(n * sizeof(int16_t));
            memcpy(layer_dump[i], src, n * sizeof(int16_t));
        }
    }
    for (int i = 0; i < 32; ++i)
        if (bufs[i] && bufs[i] != cat && i != 24) free(bufs[i]);
    free(cat);
    free(in0);
    return rc;
}

int orc_yolov2_forward_f32(const orc_weights_f32 *wp, const float *input, float *region_f32,
                           float **layer_dump)
{
    FWD_BODY(float, 0)
    (void)current_Qa;
    memcpy(in0, input, sizeof(float) * 3 * 416 * 416);
    cur = in0;
    cat = (float *)calloc((size_t)1280 * 13 * 16, sizeof(float));
    for (int i = 0; i < 32; ++i) {
        const orc_layer *l = &NET[i];
        const int ow = l->type == L_CONV ? (l->w - l->size + 2 * l->pad) / l->stride + 1 : l->w / 2;
        const int oh = l->type == L_CONV ? (l->h - l->size + 2 * l->pad) / l->stride + 1 : l->h / 2;
        switch (l->type) {
        case L_CONV: {
            const float *src = cur;
            if (i == 26) src = bufs[16];
            if (i == 29) src = cat;
            float *dst;
            if (i == 24) dst = cat + (size_t)256 * 13 * 16;
            else dst = (float *)calloc((size_t)l->n * oh * orc_w8(ow), sizeof(float));
            orc_conv_f32(src, dst, wp->weights + woff, wp->bias + boff, l->c, l->n, l->size, l->stride,
                         l->w, l->h, ow, oh, l->pad, l->leaky);
            woff += orc_yolo2_weight_len[ord];
            boff += orc_yolo2_bias_len[ord];
            ord++;
            bufs[i] = dst;
            cur = dst;
            break;
        }
        case L_MAX: {
            float *dst = (float *)calloc((size_t)l->c * oh * orc_w8(ow), sizeof(float));
            orc_maxpool_f32(cur, dst, l->c, l->size, l->stride, l->w, l->h, ow, oh);
            bufs[i] = dst;
            cur = dst;
            break;
        }
        case L_REORG:
            orc_reorg_f32(cur, cat);
            bufs[i] = cat;
            cur = cat;
            break;
        case L_ROUTE:
            break;
        case L_REGION:
            for (int k = 0; k < 13 * 425; ++k)
                for (int j = 0; j < 13; ++j) region_f32[(size_t)k * 13 + j] = cur[(size_t)k * 16 + j];
            break;
        }
    }
    if (layer_dump) {
        for (int i = 0; i < 32; ++i) layer_dump[i] = NULL;
        for (int i = 0; i < 32; ++i) {
            if (!bufs[i]) continue;
            const orc_layer *l = &NET[i];
            int oh = l->type == L_REORG ? 13 : (l->type == L_CONV ? l->h : l->h / 2);
            int ow = l->type == L_REORG ? 13 : (l->type == L_CONV ? l->w : l->w / 2);
            int oc = l->type == L_MAX ? l->c : l->n;
            const float *src = (i == 24) ? cat + (size_t)256 * 13 * 16 : bufs[i];
            size_t n = (size_t)oc * oh * orc_w8(ow);
            layer_dump[i] = (float *)malloc(n * sizeof(float));
            memcpy(layer_dump[i], src, n * sizeof(float));
        }
    }
    for (int i = 0; i < 32; ++i)
        if (bufs[i] && bufs[i] != cat && i != 24) free(bufs[i]);
    free(cat);
    free(in0);
    return 0;
}

/* -------------------------------------------------------------------- region */

/* src/core/yolo_math.cpp:19: float logistic computed in double */
static inline float logistic_f(float x) { return (float)(1. / (1. + exp(-(double)x))); }

void orc_region_forward(const float *in, float *out)
{
    const int wh = 13 * 13, classes = 80, coords = 4, n = 5;
    memcpy(out, in, sizeof(float) * ORC_REGION_ELEMS); /* yolo_region.cpp:125 */
    for (int a = 0; a < n; ++a) {
        float *p = out + (size_t)a * wh * (coords + classes + 1);
        for (int i = 0; i < 2 * wh; ++i) p[i] = logistic_f(p[i]);                          /* x,y  :129-130 */
        for (int i = 0; i < wh; ++i) p[coords * wh + i] = logistic_f(p[coords * wh + i]); /* obj  :131-132 */
    }
    /* softmax over classes, stride wh, reads the RAW input (yolo_region.cpp:136-139, yolo_math.cpp:226-241) */
    for (int a = 0; a < n; ++a)
        for (int g = 0; g < wh; ++g) {
            const float *ip = in + (size_t)a * wh * 85 + (size_t)5 * wh + g;
            float *op = out + (size_t)a * wh * 85 + (size_t)5 * wh + g;
            float sum = 0, largest = -3.402823466e+38F;
            for (int i = 0; i < classes; ++i)
                if (ip[i * wh] > largest) largest = ip[i * wh];
            for (int i = 0; i < classes; ++i) {
                float e = (float)exp((double)(ip[i * wh] / 1.0f - largest / 1.0f));
                sum += e;
                op[i * wh] = e;
            }
            for (int i = 0; i < classes; ++i) op[i * wh] /= sum;
        }
}
