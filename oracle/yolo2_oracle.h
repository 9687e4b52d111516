/*
 * yolo2_oracle.h -- CPU restatement of the reference's YOLOv2 accelerator path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (the HIP library, the host
 * CLI, the Python binding) may include, link or call this.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * /root/reference).  Parity pin: the restatement is compared bit-for-bit with the
 * reference itself compiled from its own sources (oracle/_ref, see oracle/Makefile)
 * and with the fixtures under tests/golden/ that were generated from that build.
 * The reference ships no golden vectors of its own.
 *
 * Tensor layout (both precisions) is the reference's DRAM layout:
 *   feature map  [C][H][W8]   W8 = ceil(W/8)*8   (hls/models/yolov2/yolo2_accel.cpp:89-99)
 *   weights      weights_reorg order: for m0 step 32, for n0 step 4:
 *                block[k*k][TM_MIN][TN_MIN]       (src/models/yolov2/yolov2_weight_gen.cpp:43-67)
 *   bias         dense [N]
 */
#ifndef YOLO2_ORACLE_H
#define YOLO2_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_TN 4   /* hls/core/params.hpp Tn: numerical constant of the int16 format */
#define ORC_TM 32  /* hls/core/params.hpp Tm: weight-file tiling only */

static inline int orc_w8(int w) { return (w + 7) & ~7; }

/* Worker threads for the per-output-channel loops; results are independent of it.
 * Default 1 (the reference is single-threaded and not re-entrant). */
void orc_set_threads(int n);
int orc_get_threads(void);

/* int16 convolution + bias + per-(4 channel, tap) requantise + saturate + leaky.
 * Restates compute() int16 branch (hls/core/core_compute.cpp:22-120), the write-back
 * leaky (core_compute.cpp:175-210), input halo/padding (hls/core/core_io.cpp:44-138),
 * weight stream order (core_io.cpp:140-199) and the tile loops of YOLO2_FPGA
 * (hls/models/yolov2/yolo2_accel.cpp:127-170) as a per-output definition. */
void orc_conv_i16(const int16_t *in, int16_t *out, const int16_t *w_reorg, const int16_t *bias,
                  int C, int N, int K, int stride, int W, int H, int OW, int OH, int pad,
                  int leaky, int Qw, int Qa_in, int Qa_out, int Qb);

/* fp32 twin: compute() fp32 branch (core_compute.cpp:121-172), leaky x*0.1f (:201-205). */
void orc_conv_f32(const float *in, float *out, const float *w_reorg, const float *bias,
                  int C, int N, int K, int stride, int W, int H, int OW, int OH, int pad, int leaky);

/* 2x2/stride-2 style max pool, padding forced to 0, init/pad value -32768 or -1024*1024.
 * pool_yolo2 (core_compute.cpp:266-305), pad value (core_io.cpp:96-103),
 * padding=0 (hls/core/core_scheduler.cpp:72-73). */
void orc_maxpool_i16(const int16_t *in, int16_t *out, int C, int K, int stride, int W, int H, int OW, int OH);
void orc_maxpool_f32(const float *in, float *out, int C, int K, int stride, int W, int H, int OW, int OH);

/* Darknet legacy reorg on the 64x26x26 tensor (stride 2) from [64][26][32] to [256][13][16],
 * then the int16 route-28 Q alignment shift (shift>0: arithmetic >>, shift<0: <<, sat16).
 * reorg_cpu + glue (hls/models/yolov2/yolo2_model.cpp:112-129,358-403). shift==0 -> no shift. */
void orc_reorg_i16(const int16_t *in, int16_t *out, int shift);
void orc_reorg_f32(const float *in, float *out);

/* Input quantisation (yolo2_model.cpp:257-273): sat16(llround(clamp(x*2^q))). */
void orc_quantize_input(const float *in, int16_t *out, size_t n, int q_in);

/* Integer leaky: x<0 ? x/10 (trunc toward 0) : x  (core_compute.cpp:191-197). */
int16_t orc_leaky_i16(int16_t x);

/* ------------------------------------------------------------------ network */

typedef struct {
    const int16_t *weights;   /* 50,941,792 elems, reorg order, per-layer pad already stripped */
    const int16_t *bias;      /* 10,761 elems */
    const int32_t *weight_q;  /* >= 23 */
    const int32_t *bias_q;    /* >= 23 */
    const int32_t *act_q;     /* n_act_q entries (reference expects >= 24) */
    int n_weight_q, n_bias_q, n_act_q;
} orc_weights_i16;

typedef struct {
    const float *weights;
    const float *bias;
} orc_weights_f32;

#define ORC_REGION_ELEMS (425 * 13 * 13)

/* Whole-network int16 forward for one frame (yolov2_hls_ps, yolo2_model.cpp:229-449, with
 * the config/yolov2.cfg layer table hard-wired: SURVEY.md section 8a).
 * input: float CHW 3x416x416.  region_i16: [425][13][13] raw int16 (13-of-16 gather done,
 * yolo2_model.cpp:406-414).  region_f32 (optional): dequantised floats (:416-421).
 * Returns the final activation Q (current_Qa) or <0 on error.
 * If layer_dump != NULL it receives pointers to malloc'd copies of every layer output in the
 * reference layout (caller frees) -- used to localise a mismatch. */
int orc_yolov2_forward_i16(const orc_weights_i16 *wp, const float *input,
                           int16_t *region_i16, float *region_f32, int16_t **layer_dump);

int orc_yolov2_forward_f32(const orc_weights_f32 *wp, const float *input, float *region_f32,
                           float **layer_dump);

/* Region layer post-activation (forward_region_layer, src/core/yolo_region.cpp:123-141;
 * logistic in double src/core/yolo_math.cpp:19; softmax :226-241).  in/out: [5][85][13][13]. */
void orc_region_forward(const float *in, float *out);

/* Strip the per-layer odd-length pad of weights_reorg_int16.bin / bias_int16.bin
 * (load_weights, yolo2_model.cpp:198-224).  Returns elements written or -1 if truncated. */
long orc_strip_int16_layer_pad(const int16_t *file, size_t file_elems, const int *layer_len,
                               int n_layers, int16_t *dst);

/* YOLOv2 layer table helpers (model_config.cpp:4-10) */
extern const int orc_yolo2_weight_len[23];
extern const int orc_yolo2_bias_len[23];

#ifdef __cplusplus
}
#endif
#endif
