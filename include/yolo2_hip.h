/*
 * yolo2_hip.h -- C ABI of the MI355X (gfx950) YOLOv2 accelerator library, libyolo2_hip.so.
 *
 * This is the drop-in boundary: it stands where the reference's AXI-Lite/udmabuf driver
 * stands (linux_app/include/yolo2_accel_linux.h, linux_app/include/dma_buffer_manager.h)
 * and where the host simulation calls YOLO2_FPGA (hls/models/yolov2/yolo2_accel.hpp:10-17).
 * Plain C: pointers, sizes and ints only; no C++ exceptions cross it; every entry point
 * returns one of the reference's status codes (linux_app/include/yolo2_config.h:146-151).
 *
 * "Physical address" in the reference == a device-accessible address here (uint64_t):
 * either HBM from yolo2_hip_alloc(), or mapped pinned host memory from memory_allocate_*().
 *
 * Three tiers, innermost first:
 *   1. driver + per-layer calls with the reference's exact argument lists and DRAM layouts
 *      (feature maps [C][H][W8], weights in weights_reorg order) -- lets the reference's C
 *      layer loop (linux_app/src/yolo2_inference.c:763-910) drive the GPU unchanged;
 *   2. whole-network batched entry: weights resident in HBM, 28 layer kernels per batch
 *      on one stream, one host sync per batch instead of one per layer;
 *   3. helpers the callers on either side need (weight-file quirks, region gather).
 */
#ifndef YOLO2_HIP_H
#define YOLO2_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* linux_app/include/yolo2_config.h:146-151 */
#define YOLO2_SUCCESS     0
#define YOLO2_ERROR      -1
#define YOLO2_TIMEOUT    -2
#define YOLO2_INIT_ERROR -3
#define YOLO2_MMAP_ERROR -4   /* here: allocation / mapping failure */
#define YOLO2_DMA_ERROR  -5   /* here: host<->device copy failure */

/* ------------------------------------------------------------------ tier 1: driver */

/* yolo2_accel_init / yolo2_accel_cleanup (yolo2_accel_linux.h:19-31).  Binds the calling
 * process to the HIP device selected by yolo2_hip_select_device() (default 0). */
int  yolo2_accel_init(void);
void yolo2_accel_cleanup(void);
int  yolo2_hip_select_device(int device);
int  yolo2_hip_device_count(void);
const char *yolo2_hip_last_error(void);

/* yolo2_set_q_values (yolo2_accel_linux.h:33-41): the reference latches Q values in AXI
 * GPIOs; the per-layer calls below also take them as arguments, which win. */
void yolo2_set_q_values(int32_t qw, int32_t qa_in, int32_t qa_out, int32_t qb);
int  yolo2_is_busy(void);                            /* yolo2_accel_linux.h:43-47 */
int  yolo2_is_done(void);                            /* :49-53 */
int  yolo2_wait_for_completion(uint32_t timeout_ms); /* :55-61 */

/* yolo2_get_status / yolo2_read_reg / yolo2_write_reg (yolo2_accel_linux.h:57-61, 123-134).  The
 * reference exposes the HLS IP's AXI-Lite register file (offsets: linux_app/include/yolo2_config.h:
 * 36-71).  Here it is a 4 KiB shadow register file with the same offsets: the per-layer calls latch
 * their addresses and scalars into it exactly like yolo2_accel_linux.c:446-505 writes them, AP_CTRL
 * (offset 0) reads ap_done|ap_idle|ap_ready (0x0e) while the device is idle and ap_start (0x01)
 * while a layer is in flight, and writing ap_start to AP_CTRL runs the layer the registers describe
 * (conv for LayerType 0, maxpool for 1) with the Q values of yolo2_set_q_values(), blocking.
 * All three return 0 / do nothing before yolo2_accel_init(), like the reference. */
uint32_t yolo2_get_status(void);
uint32_t yolo2_read_reg(uint32_t offset);
void     yolo2_write_reg(uint32_t offset, uint32_t value);
/* Per-layer calls (conv + maxpool) served since yolo2_accel_init(): the reference's layer loop
 * (linux_app/src/yolo2_inference.c:763-910) makes 28 per frame. */
long yolo2_hip_driver_calls(void);

/* yolo2_execute_conv_layer (yolo2_accel_linux.h:70-99).  Same 27 arguments and meaning as
 * YOLO2_FPGA with LayerType 0.  input/output are [C][H][W8] int16, weight is the layer's
 * slice of weights_reorg_int16 (blocks of TM x TN x K*K), beta the layer's int16 biases.
 * tm/tn/tr/tc/ofm_num_bound/mloops* are validated like yolo2_accel_linux.c:383-414 and
 * otherwise unused (GPU tiling is internal and does not change results).  Blocking. */
int yolo2_execute_conv_layer(uint64_t input_addr, uint64_t output_addr, uint64_t weight_addr,
                             uint64_t beta_addr, int ifm_num, int ofm_num, int ksize, int kstride,
                             int input_w, int input_h, int output_w, int output_h, int padding,
                             int is_nl, int is_bn, int tm, int tn, int tr, int tc, int ofm_num_bound,
                             int mloopsxTM, int mloops_a1xTM, int layer_type, int qw, int qa_in,
                             int qa_out, int qb, uint32_t timeout_ms);

/* yolo2_execute_maxpool_layer (yolo2_accel_linux.h:104-121). */
int yolo2_execute_maxpool_layer(uint64_t input_addr, uint64_t output_addr, int channels, int ksize,
                                int kstride, int input_w, int input_h, int output_w, int output_h,
                                int padding, int tm, int tr, int tc, int ofm_num_bound, int mloopsxTM,
                                int mloops_a1xTM, uint32_t timeout_ms);

/* Arithmetic form the most recent yolo2_execute_conv_layer ran (codes of yolo2_hip_layer_path below;
 * -1 = the generic one-thread-per-output kernel for shapes the tiled kernel does not cover).  The
 * per-layer calls prove the form for the whole layer from the weights and Q values of that call. */
int yolo2_hip_last_layer_path(void);

/* fp32 twin of the conv call (host-sim form YOLO2_FPGA without INT16_MODE,
 * hls/models/yolov2/yolo2_accel.hpp:10-17): float tensors, Q arguments absent. */
int yolo2_execute_conv_layer_f32(uint64_t input_addr, uint64_t output_addr, uint64_t weight_addr,
                                 uint64_t beta_addr, int ifm_num, int ofm_num, int ksize, int kstride,
                                 int input_w, int input_h, int output_w, int output_h, int padding,
                                 int is_nl, uint32_t timeout_ms);

/* Buffers: dma_buffer_manager.h:94-139.  ptr is CPU-visible, phys_addr GPU-visible; both
 * name the same mapped pinned pages, so flush/invalidate only have to order, not copy. */
#ifndef DMA_BUFFER_MANAGER_H   /* the reference's own header may be included first: same types */
typedef struct {
    void *ptr;
    size_t size;
    uint64_t phys_addr;
} memory_buffer_t;
/* dma_buffer_t (dma_buffer_manager.h:24-30): fd is -1 and device_name "hip-pinned" here */
typedef struct {
    void *virt_addr;
    uint64_t phys_addr;
    size_t size;
    int fd;
    char device_name[64];
} dma_buffer_t;
#endif
/* dma_buffer_init / dma_buffer_cleanup (dma_buffer_manager.h:32-42; called by the reference's main,
 * linux_app/src/main.c:572,1301): init checks that a HIP device is present (the udmabuf check of
 * dma_buffer_manager.c:137-182), cleanup frees every buffer still allocated (:184-192).
 * dma_buffer_alloc .. dma_buffer_get_phys (:44-92): the udmabuf-level interface under memory_allocate_*. */
int      dma_buffer_init(void);
void     dma_buffer_cleanup(void);
int      dma_buffer_alloc(size_t size, dma_buffer_t *buffer);
void     dma_buffer_free(dma_buffer_t *buffer);
void     dma_buffer_sync_for_device(dma_buffer_t *buffer, size_t offset, size_t size);
void     dma_buffer_sync_for_cpu(dma_buffer_t *buffer, size_t offset, size_t size);
uint64_t dma_buffer_get_phys(dma_buffer_t *buffer, size_t offset);
int      memory_allocate_ddr(size_t size, size_t alignment, memory_buffer_t *buffer);
void     memory_free_ddr(memory_buffer_t *buffer);
int      memory_allocate_weights(size_t size, memory_buffer_t *buffer);
int      memory_allocate_bias(size_t size, memory_buffer_t *buffer);
int      memory_allocate_inference_buffer(memory_buffer_t *buffer);
uint64_t memory_get_phys_addr(void *virt_addr);
void     memory_flush_cache(void *addr, size_t size);
void     memory_invalidate_cache(void *addr, size_t size);

/* HBM buffers for callers that manage residency themselves.  yolo2_hip_alloc allocates on the calling thread's CURRENT
 * device (the driver tier is a one-device interface); a caller that holds contexts on several devices uses _alloc_on. */
int  yolo2_hip_alloc(size_t bytes, uint64_t *dev_addr);
void yolo2_hip_free(uint64_t dev_addr);
int  yolo2_hip_memcpy_h2d(uint64_t dst, const void *src, size_t bytes);
int  yolo2_hip_memcpy_d2h(void *dst, uint64_t src, size_t bytes);
int  yolo2_hip_memset(uint64_t dst, int value, size_t bytes);

/* ------------------------------------------------------- tier 2: whole network, batched */

typedef struct yolo2_hip_ctx yolo2_hip_ctx; /* one per device; thread-compatible */

#define YOLO2_N_CONV       23
#define YOLO2_N_WEIGHTS    50941792   /* hls/models/yolov2/yolo2_accel.cpp:41 */
#define YOLO2_N_BIAS       10761      /* :42 */
#define YOLO2_REGION_ELEMS (425 * 13 * 13)
#define YOLO2_FRAME_ELEMS  (3 * 416 * 416)

int  yolo2_hip_create(int device, yolo2_hip_ctx **ctx);
void yolo2_hip_destroy(yolo2_hip_ctx *ctx);   /* also leaves the context's RCCL communicator, if it joined one */
int  yolo2_hip_ctx_device(yolo2_hip_ctx *ctx);
/* HBM on the context's device, whatever device the calling thread has current (and leaves that binding in force) */
int  yolo2_hip_alloc_on(yolo2_hip_ctx *ctx, size_t bytes, uint64_t *dev_addr);

/* Weights as yolov2_hls_ps holds them after load_weights (yolo2_model.cpp:158-227): the
 * weights_reorg_int16 stream with the per-layer file pad already stripped
 * (yolo2_strip_int16_layer_pad below), dense int16 biases, three int32 Q tables.
 * Uploads once, re-packs partial tiles on the device, and proves per layer which arithmetic
 * width is exact for these weights and Q values (yolo2_hip_layer_path). */
int yolo2_hip_load_weights_int16(yolo2_hip_ctx *ctx, const int16_t *weights_reorg, size_t n_weights,
                                 const int16_t *bias, size_t n_bias, const int32_t *weight_q,
                                 int n_weight_q, const int32_t *bias_q, int n_bias_q,
                                 const int32_t *act_q, int n_act_q);
/* Same, the two blobs already in HBM (e.g. received by an RCCL broadcast). */
int yolo2_hip_load_weights_int16_dev(yolo2_hip_ctx *ctx, uint64_t weights_reorg_dev, size_t n_weights,
                                     uint64_t bias_dev, size_t n_bias, const int32_t *weight_q,
                                     int n_weight_q, const int32_t *bias_q, int n_bias_q,
                                     const int32_t *act_q, int n_act_q);

/* 0 = 32-bit form A, 1 = 32-bit form B (pre-shifted accumulator), 3 = form C (packed int16
 * accumulators), 4 = form D (form C with the requantisation shift folded into pre-scaled weights),
 * 2 = 64-bit exact path, <0 = bad ordinal / not loaded.  All are bit-exact; the
 * loader picks the narrowest one it can PROVE exact for the weights and Q values at hand
 * (csrc/kernels_int16.hpp explains the forms). */
int yolo2_hip_layer_path(yolo2_hip_ctx *ctx, int conv_ordinal);
/* The form is chosen per block of 32 output channels (a few large-weight channels must not slow
 * a whole layer down): layer_path() reports the form most blocks use, this the count per form
 * (index = the codes above). */
int yolo2_hip_layer_path_counts(yolo2_hip_ctx *ctx, int conv_ordinal, int counts[5]);

/* (Re)allocates the activation tensors for exactly `batch` frames per call. */
int yolo2_hip_set_batch(yolo2_hip_ctx *ctx, int batch);

/* One pass of the accelerator path over a batch (everything yolov2_hls_ps does between the
 * input memcpy and the region gather, yolo2_model.cpp:257-421, for `batch` frames):
 *   frames_dev  float [batch][3][416][416], letterboxed, in [0,1]  (device)
 *   region_dev  int16 [batch][425][13][13] raw region tensor       (device)
 *   *final_q    activation Q of that tensor (float = int16 * 2^-final_q)
 * Enqueues on `stream` (a hipStream_t, NULL = default stream) and returns without
 * synchronising: the caller owns the one sync per batch. */
int yolo2_hip_run_batch_int16(yolo2_hip_ctx *ctx, uint64_t frames_dev, int batch,
                              uint64_t region_dev, int *final_q, void *stream);

/* Convenience for C hosts: pageable host buffers in/out, synchronous. */
int yolo2_hip_run_batch_int16_host(yolo2_hip_ctx *ctx, const float *frames, int batch,
                                   int16_t *region, int *final_q);

/* Streaming form for hosts that hold many frames in ordinary (pageable) memory: processes
 * n_frames in chunks of `batch`, with the H2D copy of chunk k+1, the kernels of chunk k and the
 * D2H copy of chunk k-1 overlapped on three HIP streams through pinned staging buffers, so the
 * PCIe-inclusive rate approaches the device-resident rate.  Results are identical to calling
 * yolo2_hip_run_batch_int16 chunk by chunk (a trailing partial chunk is padded with the last frame). */
int yolo2_hip_run_frames_int16(yolo2_hip_ctx *ctx, const float *frames, int n_frames, int batch,
                               int16_t *region, int *final_q);

/* ---- fp16 MFMA path (floating-point form of the same network: a true dense contraction).
 * Weights as yolov2_hls_ps holds them for Precision::FP32 (weights_reorg.bin / bias.bin,
 * yolo2_model.cpp:171-181); converted once to fp16 in a K-contiguous layout on the device.
 * Activations fp16, accumulation fp32 on v_mfma_f32_32x32x16_f16, bias + leaky in fp32.
 * region_dev: float [batch][425][13][13] (what yolov2_hls_ps hands to forward_region_layer in
 * fp32 mode, yolo2_model.cpp:422-424).  Validated against the fp32 reference at box tolerance. */
int yolo2_hip_load_weights_fp32(yolo2_hip_ctx *ctx, const float *weights_reorg, size_t n_weights,
                                const float *bias, size_t n_bias);
/* Same, the two blobs already in HBM (e.g. received by the RCCL broadcast below). */
int yolo2_hip_load_weights_fp32_dev(yolo2_hip_ctx *ctx, uint64_t weights_reorg_dev, size_t n_weights,
                                    uint64_t bias_dev, size_t n_bias);
int yolo2_hip_run_batch_fp16(yolo2_hip_ctx *ctx, uint64_t frames_dev, int batch, uint64_t region_dev,
                             void *stream);
int yolo2_hip_run_batch_fp16_host(yolo2_hip_ctx *ctx, const float *frames, int batch, float *region);
/* The kernel the fp16 pass runs for layer `layer_idx` at the current batch (after the first run at that batch; "" before).
 * The selection is made ONCE per (weights, batch) into a launch table; the YOLO2_F16_* A/B switches are read when the weights
 * are loaded, never on the launch path. */
const char *yolo2_hip_fp16_layer_kernel(yolo2_hip_ctx *ctx, int layer_idx);
/* The extent check every step of that table passes before it may be launched, on explicit numbers (testable without a GPU):
 * a kernel of store kind `store` (0 = stores the full-resolution tensor only and ignores a fused pool, 1 = full-resolution or
 * pooled according to `pool`, 2 = pooled only) for a B x H x W conv writing items of Cp_out halves, channels
 * [out_ch_off, out_ch_off + n_store), into a tensor of dst_B x dst_H x dst_W pixels with dst_Cp halves per item.
 * YOLO2_SUCCESS, or YOLO2_ERROR with yolo2_hip_last_error() naming the mismatch. */
int yolo2_hip_f16_store_check(int store, int pool, int B, int H, int W, int Cp_out, int out_ch_off, int n_store, int dst_B,
                              int dst_H, int dst_W, int dst_Cp);

/* Copies layer `layer_idx`'s output of frame `frame` from the last run into the reference's
 * [C][H][W8] int16 layout (pad columns zero) -- the yolov2_region_*_hw.txt style parity hook
 * (SURVEY.md section 4).  out_elems receives C*H*W8.  layer_idx -1 = the quantised network input
 * (input quantise of yolo2_model.cpp:257-273), [3][416][416]. */
int yolo2_hip_debug_layer_output(yolo2_hip_ctx *ctx, int layer_idx, int frame, int16_t *out,
                                 size_t capacity_elems, size_t *out_elems);

/* Per-layer device time: enabling records hipEvent pairs around every layer kernel of
 * subsequent runs, on the stream they are launched on (the analogue of the per-layer latency
 * report in linux_app/src/yolo2_inference.c:75-142).  layer_times_ms returns the mean over the
 * profiled runs since the last set_profiling call (at most the latest 32). */
int yolo2_hip_set_profiling(yolo2_hip_ctx *ctx, int enable);
int yolo2_hip_layer_times_ms(yolo2_hip_ctx *ctx, float *ms32 /* [32] */);

/* A batch >= 16 runs as two, a batch of 48..127 as three part-batches ("lanes", sizes within one frame of
 * each other) on internal streams forked from and joined to the caller's stream, so that every layer is
 * several concurrent launches and one's idle tail is filled by the others (+4-6 % frames/s at batch 64).
 * Returns the lane count (1 = none).  Per-layer times and launch geometry then describe lane 0's launches. */
int yolo2_hip_num_lanes(yolo2_hip_ctx *ctx);
/* The same for the fp16 path (two half-batch lanes from batch 64; known after the first run_batch_fp16). */
int yolo2_hip_num_lanes_fp16(yolo2_hip_ctx *ctx);
/* Sets that lane count (1 = no lanes; default 2, or YOLO2_F16_LANES at weight load).  Used by bench.py to time one lane's launches
 * ALONE for the per-kernel roofline object. */
int yolo2_hip_set_fp16_lanes(yolo2_hip_ctx *ctx, int lanes);

/* ------------------------------------------------------- "fp32 fast": the matrix-core path INSIDE the fp32 tolerance
 *
 * BASELINE.json: "MFMA used only on the fp16/fp32 path where conv is a true dense contraction ... detections ... within 1e-3
 * box-coord tolerance for fp32".  The plain fp16 path is outside that tolerance (5.8e-3), the exact fp32 path is bit-identical
 * but VALU-bound.  This entry computes the reference's fp32 arithmetic (hls/core/core_compute.cpp:121-172: per output
 * bias + sum of w x, leaky x < 0 ? 0.1 x : x) on v_mfma_f32_32x32x16_f16 with every value carried as two halves
 * hi = fp16(v), lo = fp16(v - hi) and every product taken as a_hi w_hi + a_lo w_hi + a_hi w_lo (fp32 accumulate; the dropped
 * lo x lo term is 2^-22 relative): ~1e-6 relative error per layer instead of fp16's 5e-4, for 3x the MFMA work of the fp16 path.
 * Same contract as yolo2_hip_run_batch_fp16: float CHW frames in HBM -> dense fp32 [batch][425][13][13] region tensor, enqueued on
 * `stream`, needs yolo2_hip_load_weights_fp32*.  Not bit-exact with the reference (another summation order); the GPU tests hold
 * every one of the 845 boxes of the fixture frames within 1e-3 in all four coordinates (measured: ~1e-5). */
int yolo2_hip_run_batch_f32tol(yolo2_hip_ctx *ctx, uint64_t frames_dev, int batch, uint64_t region_dev, void *stream);
int yolo2_hip_run_batch_f32tol_host(yolo2_hip_ctx *ctx, const float *frames, int batch, float *region);
/* kernel the split-mode launch table runs for a layer / its lane count (after the first run at this batch) */
const char *yolo2_hip_f32tol_layer_kernel(yolo2_hip_ctx *ctx, int layer_idx);
int yolo2_hip_num_lanes_f32tol(yolo2_hip_ctx *ctx);

/* 1 when conv layer `layer_idx` (0..31) runs fused with the 2x2 max pool after it (k_conv_i16_pool: the
 * full-resolution tensor is never written, except layer 16's, which also feeds the route), 0 otherwise.
 * Chosen per batch by set_batch (timed); YOLO2_NO_POOLFUSE=1 disables, YOLO2_POOLFUSE=1 forces it wherever legal. */
int yolo2_hip_layer_pool_fused(yolo2_hip_ctx *ctx, int layer_idx);

/* Launch geometry of the conv kernel family, for the roofline report.  block = 128: k_conv_i16_w16.  pixels_per_lane = -S: the
 * layer runs k_conv_i16_ks (chain split over S workgroups).  pixels_per_lane = 0 means
 * the layer runs the split-K kernel (64/S pixels x S K-splits per wavefront, partial clamp-affine
 * maps combined with wavefront shuffles), which set_batch picks for small batches when it times
 * faster; YOLO2_SPLITK=0 / 1 in the environment disables / forces it wherever its bounds hold. */
int yolo2_hip_conv_launch_info(yolo2_hip_ctx *ctx, int conv_ordinal, int *grid_x, int *grid_y,
                               int *block, int *lds_bytes, int *pixels_per_lane);

/* The launch plan of conv layer `conv_ordinal` as text, e.g. "P=1 pad=0 w16=0 hiacc=1 ks=0 splitk=0 pp=1 grp=1 fused=0 form=4 extra=0"
 * (pixels per lane, occupancy cap, 16-channels-per-wavefront kernel, form D without v_perm, K-split over workgroups / over lanes,
 * 1x1 grouping, conv + pool fusion, arithmetic form, extra launches for blocks of another form).  Plans come from the committed plan
 * table (config/plan_gfx950.txt) for the batches it knows - the same in every process - and from timing otherwise. */
int yolo2_hip_conv_plan_string(yolo2_hip_ctx *ctx, int conv_ordinal, char *buf, int cap);
/* Where the current batch's conv plan came from: 1 = the plan table shipped next to the library (config/plan_gfx950.txt: the same
 * kernels in every process), 2 = timed in this process (a batch or a model the table does not hold, or YOLO2_AUTOTUNE=1),
 * 3 = static defaults (autotune=0), 4 = forced by a test hook (force_p), 5 = the weight-side cache bound with
 * yolo2_hip_set_plan_cache (the same kernels in every process for THIS weight set), 0 = no batch planned yet. */
int yolo2_hip_plan_source(yolo2_hip_ctx *ctx);

/* ------------------------------------------------------- options: one parsed-once set per context
 *
 * Every switch that steers kernel selection, lanes or planning is a named option of the context.  The set is filled from the
 * environment (YOLO2_<NAME>, upper case) ONCE, inside yolo2_hip_create; afterwards only this call changes it (it takes effect at the
 * next weight load / yolo2_hip_set_batch).  Nothing on a planning or launch path reads the environment.  Names (README.md has the
 * list with meanings): planning - autotune, plan_file, plan_write, lanes, no_lanes, lane_split, no_plan_cache, verbose; kernel-family
 * A/B switches - splitk, poolfuse, no_poolfuse, no_hiacc, no_ks, no_w16, no_grp, no_xcd_remap, splitk_no_pack, f16_*; test hooks
 * that force one shape everywhere - force_path, force_p, force_w16, force_hiacc, force_ks, f32_p.  value NULL or "" restores the
 * default; flags are on for any value but "0".  YOLO2_ERROR for an unknown name or a value out of range.  ctx = NULL addresses the
 * process-wide set the context-less driver tier (yolo2_execute_conv_layer ...) plans with (filled from the environment at first use).
 * (The reference's own variables - YOLO2_VERBOSE, YOLO2_NO_DUMP, YOLO2_DUMP_REGION*: linux_app/include/yolo2_log.h:27-36,
 * hls/models/yolov2/yolo2_model.cpp:427-438 - keep their names.) */
int yolo2_hip_set_option(yolo2_hip_ctx *ctx, const char *name, const char *value);
/* The options that differ from their defaults as "name=value name=value" ("" if none): what bench.py discloses with its record. */
int yolo2_hip_options_string(yolo2_hip_ctx *ctx, char *buf, int cap);

/* ------------------------------------------------------- the weight-side plan cache
 *
 * The reference prepares a weight set once, offline (src/models/yolov2/yolov2_weight_gen.cpp:34-68 writes weights_reorg*.bin;
 * the loader hls/models/yolov2/yolo2_model.cpp:158-227 only reads).  The analogue here: a small text file beside the weight set,
 * `<weights_reorg_int16.bin>.y2plan`, that this library writes the first time it has timed a batch for that weight set and reads at
 * every later load: a hash of the blobs and Q tables, the per-block bounds behind the arithmetic-form proofs (k_weight_bound* is
 * then skipped), the forms / scale shifts derived from them (re-derived and compared), and the conv plan of every timed batch as
 * plan_gfx950.txt lines (the autotune is then skipped and every process runs the same kernels).  A missing, stale (other hash),
 * damaged (checksum) or unwritable file costs time, never correctness.  weights_reorg_int16.bin stays the interchange format.
 * set_plan_cache binds the file to the context (NULL / "" unbinds) and takes effect at the next yolo2_hip_load_weights_int16*. */
int yolo2_hip_set_plan_cache(yolo2_hip_ctx *ctx, const char *path);
/* hash of the loaded weight set, whether the file's bounds were used at the last load, plan lines / batches held */
int yolo2_hip_plan_cache_info(yolo2_hip_ctx *ctx, uint64_t *weights_hash, int *bounds_from_file, int *n_lines, int *n_batches);
/* Test hook (no GPU): would the file be accepted for a weight set with this hash?  YOLO2_ERROR + the reason otherwise. */
int yolo2_hip_plan_cache_check(const char *path, uint64_t weights_hash, int *n_lines);

/* The K-split-over-workgroups kernel (k_conv_i16_ks, single frames) stores `splits` clamp-affine triples of 24 bytes per output item
 * (4 channels) and pixel into the context's scratch.  The rule every such plan passes in plan_conv and again at launch, on plain
 * numbers (no GPU): splits in {2,4,8,16} and splits x cg_out x npix x 24 <= cap_bytes; cap_bytes = 0 (a context without scratch:
 * batch > 4) refuses every split.  Round 3's GPU memory fault was this rule violated. */
int yolo2_hip_i16_plan_check(int splits, int cg_out, int npix, size_t cap_bytes);
/* Bytes of that scratch the context holds for its current batch: the largest splits x items x pixels x 24 among the plans that were
 * accepted - 0 when no layer runs the K-split kernel (round 3 held 66 MB per frame for every context of <= 4 frames). */
size_t yolo2_hip_ks_scratch_bytes(yolo2_hip_ctx *ctx);

/* fp32 whole network in the reference's own arithmetic (what yolov2_hls_ps does at Precision::FP32,
 * hls/models/yolov2/yolo2_model.cpp:229-449: compute() fp32 branch core_compute.cpp:121-172 in its
 * operation order without FMA contraction, pool_yolo2, the legacy reorg): one frame
 * float [3][416][416] on the host -> float [425][13][13] on the host, bit-identical to the
 * reference's region tensor, hence identical boxes (BASELINE.json asks for 1e-3).  A correctness
 * path built from the one-thread-per-output kernels; the fast floating-point path is
 * yolo2_hip_run_batch_fp16.  Needs yolo2_hip_load_weights_fp32. */
int yolo2_hip_run_frame_fp32_host(yolo2_hip_ctx *ctx, const float *frame, float *region);

/* The same exact fp32 arithmetic, batched and TILED (csrc/kernels_f32.hpp: LDS-staged input tiles, scalar weight
 * loads, P pixels x 8 channels of fp32 accumulators per lane, the reference's operation order kept product by product):
 * bit-identical to the reference's fp32 region tensor at a few hundred times the one-thread-per-output pass.
 *   frames_dev  float [batch][3][416][416]      region_dev  float [batch][425][13][13]      (device)
 * Asynchronous on `stream` like the other batched entries; the _host form is synchronous. */
int yolo2_hip_run_batch_fp32(yolo2_hip_ctx *ctx, uint64_t frames_dev, int batch, uint64_t region_dev, void *stream);
int yolo2_hip_run_batch_fp32_host(yolo2_hip_ctx *ctx, const float *frames, int batch, float *region);

/* GPU pre-processing (the step before the path): the reference host's
 * load_image_stb (bytes / 255.f, src/core/yolo_image.cpp:29-63) + letterbox_image (two-pass bilinear
 * resize_image onto a 0.5 canvas, src/core/yolo_image.cpp:84-165) as one kernel, float-for-float in
 * the reference's operation order, so the frame is bit-identical to the host code's.
 * image_dev: w x h x channels interleaved bytes (channels 1 or 3; 1 is replicated like stb's grey
 * load); frame_dev: float [3][net_h][net_w].  Asynchronous on `stream`. */
int yolo2_hip_letterbox_u8(uint64_t image_dev, int w, int h, int channels, uint64_t frame_dev,
                           int net_w, int net_h, void *stream);

/* Camera-style whole-network entry: n images of arbitrary sizes as HOST bytes -> int16 region
 * tensors [n][425][13][13] on the host, in chunks of `batch` images.  The bytes (not the 4x larger
 * float frames) cross PCIe; letterboxing runs on the GPU; upload, kernels and download of
 * consecutive chunks overlap on three HIP streams.  Synchronous. */
int yolo2_hip_run_images_u8_host(yolo2_hip_ctx *ctx, const uint8_t *const *images, const int *widths,
                                 const int *heights, int channels, int n, int batch,
                                 int16_t *region_host, int *final_q);

/* ------------------------------------------------------- the step after the path, on the GPU
 *
 * forward_region_layer + get_network_boxes + do_nms_sort (src/core/yolo_region.cpp:123-236, src/core/yolo_post.cpp:54-85)
 * for a whole batch, one workgroup per frame, from a region tensor that is already in HBM (the output of
 * yolo2_hip_run_batch_*).  For the int16 tensor the result is IDENTICAL to the reference host code's: every exp the
 * reference applies to one of the 65,536 possible values int16 x 2^-Q is a table the host's own libm filled, all other
 * arithmetic is IEEE +,-,*,/ in the reference's order, the per-class sort is stable like glibc's qsort.  The float entry
 * (fp16 / fp32 tensors) evaluates exp on the device: within 1 ulp of the host's.
 *   im_w / im_h [batch]   original image sizes (letterbox correction, relative = 1 like yolov2_main.cpp:310-312)
 *   dets / counts         optional: (frame, det, class, prob, box) records for every prob > 0 after NMS, in the order
 *                         the reference prints them (dets in array order, classes inner), cap_per_frame records per frame;
 *                         counts[f] = records frame f has (may exceed the cap: the surplus is dropped)
 *   rows / totals         optional: the reference's dets[] array after do_nms_sort, [batch][845][85] =
 *                         x, y, w, h, objectness, prob[80], and the number of slots in front that hold a candidate
 *   proc                  optional: l.output of forward_region_layer, [batch][425][13][13]
 * Synchronous (host buffers out); enqueues on `stream` (NULL = default). */
typedef struct {
    int frame, det, cls;
    float prob, x, y, w, h;
} yolo2_hip_det;
int yolo2_hip_postprocess_int16(yolo2_hip_ctx *ctx, uint64_t region_dev, int batch, int final_q, const int *im_w,
                                const int *im_h, float thresh, float nms, yolo2_hip_det *dets, int cap_per_frame,
                                int *counts, float *rows, int *totals, float *proc, void *stream);
int yolo2_hip_postprocess_f32(yolo2_hip_ctx *ctx, uint64_t region_dev, int batch, const int *im_w, const int *im_h,
                              float thresh, float nms, yolo2_hip_det *dets, int cap_per_frame, int *counts,
                              float *rows, int *totals, float *proc, void *stream);

/* The whole chain for camera-style input at the path's rate: n images of arbitrary sizes as HOST bytes -> detection records, in
 * chunks of `batch` images.  Bytes cross PCIe in, letterboxing and the network run on the GPU, the region tensor STAYS in HBM and
 * the tail (region + boxes + NMS + record compaction) runs on the same device right behind the network; only the records (32 bytes
 * each) and the per-frame counts come back.  Upload, kernels and download of consecutive chunks overlap on three HIP streams.
 * What the reference's streaming loop does frame by frame on the CPU (linux_app/src/main.c:878-1288: yolo2_run_inference +
 * yolo2_forward_region_layer / get_region_detections / do_nms_sort per frame).
 *   flags   YOLO2_DETS_BEST_CLASS: one record per detection - its best class (linux_app/src/main.c:1040-1052); a frame then has at
 *           most 845 records, so cap_per_frame = 845 never truncates.  0: a record per (detection, class) with prob > 0.
 *   dets    [n][cap_per_frame]; counts[f] = records frame f produced (if > cap_per_frame the surplus was dropped); dets[].frame is the
 *           image's index in `images`.  Records identical to yolo2_hip_postprocess_int16's on the same tensors.  Synchronous. */
#define YOLO2_DETS_BEST_CLASS 1
int yolo2_hip_run_images_u8_dets(yolo2_hip_ctx *ctx, const uint8_t *const *images, const int *widths, const int *heights,
                                 int channels, int n, int batch, float thresh, float nms, int flags, yolo2_hip_det *dets,
                                 int cap_per_frame, int *counts, int *final_q);

/* ------------------------------------------------------- multi-GPU: frame sharding, one weight broadcast
 *
 * SURVEY.md 8(b) "init(device_list) ... load_weights (H2D on rank 0, RCCL broadcast to the rest)", 8(e): frames are
 * independent, so every GPU holds the whole weight set, frames split into contiguous ranges, and the only collective is
 * ONE broadcast of the weight blobs at init (ncclBroadcast from librccl.so, RCCL over xGMI; the library is loaded at the
 * first call of this section, single-GPU users never load it).  Both launch models share that one broadcast routine. */

/* contiguous range [lo, hi) of `rank` among `world` shards of `total` frames; sizes differ by at most one */
int yolo2_hip_shard_range(int total, int rank, int world, int *lo, int *hi);

/* (a) one process, n devices - the C host's model (yolov2_detect --devices 0,1,...): one context per device, a RCCL
 * communicator over the list (ncclCommInitAll).  A list that names one device twice rehearses the sharding on a single
 * GPU: RCCL refuses duplicate devices, so such a list copies device-to-device instead (yolo2_hip_multi_uses_rccl = 0). */
typedef struct yolo2_hip_multi yolo2_hip_multi;
int  yolo2_hip_multi_create(const int *devices, int n_devices, yolo2_hip_multi **m);
void yolo2_hip_multi_destroy(yolo2_hip_multi *m);
int  yolo2_hip_multi_num_devices(yolo2_hip_multi *m);
int  yolo2_hip_multi_uses_rccl(yolo2_hip_multi *m);
yolo2_hip_ctx *yolo2_hip_multi_ctx(yolo2_hip_multi *m, int i);      /* borrowed: for set_batch / profiling / layer_path */
/* blobs: host -> devices[0] -> ncclBroadcast to the rest -> every context loads its device copy */
int  yolo2_hip_multi_load_weights_int16(yolo2_hip_multi *m, const int16_t *weights_reorg, size_t n_weights,
                                        const int16_t *bias, size_t n_bias, const int32_t *weight_q, int n_weight_q,
                                        const int32_t *bias_q, int n_bias_q, const int32_t *act_q, int n_act_q);
int  yolo2_hip_multi_load_weights_fp32(yolo2_hip_multi *m, const float *weights_reorg, size_t n_weights,
                                       const float *bias, size_t n_bias);
/* n_frames host frames -> region tensors, shard i streamed by device i (yolo2_hip_run_frames_int16 / _run_images_u8_host
 * on its own host thread, chunks of batch_per_device); no data-path collective.  Results do not depend on the device count. */
int  yolo2_hip_multi_run_frames_int16(yolo2_hip_multi *m, const float *frames, int n_frames, int batch_per_device,
                                      int16_t *region, int *final_q);
int  yolo2_hip_multi_run_images_u8_host(yolo2_hip_multi *m, const uint8_t *const *images, const int *widths,
                                        const int *heights, int channels, int n, int batch_per_device,
                                        int16_t *region_host, int *final_q);

/* images -> detection records, shard i on device i: network AND tail run on the device that owns the shard (no region tensor
 * travels to the host or to device 0); see yolo2_hip_run_images_u8_dets. */
int  yolo2_hip_multi_run_images_u8_dets(yolo2_hip_multi *m, const uint8_t *const *images, const int *widths, const int *heights,
                                        int channels, int n, int batch_per_device, float thresh, float nms, int flags,
                                        yolo2_hip_det *dets, int cap_per_frame, int *counts, int *final_q);

/* (b) one process per device (torchrun / MPI style; what bench.py --gpus N runs): rank 0 makes the 128-byte id
 * (ncclGetUniqueId), the launcher hands it to every rank, each rank joins with its context (ncclCommInitRank), then
 * ALL ranks call the _bcast loader: the root passes the host blobs and Q tables, the others NULL / 0. */
int  yolo2_hip_rccl_unique_id(void *id128);
int  yolo2_hip_rccl_init_rank(yolo2_hip_ctx *ctx, int device, const void *id128, int nranks, int rank);
void yolo2_hip_rccl_finalize(yolo2_hip_ctx *ctx);
int  yolo2_hip_load_weights_int16_bcast(yolo2_hip_ctx *ctx, const int16_t *weights_reorg, size_t n_weights,
                                        const int16_t *bias, size_t n_bias, const int32_t *weight_q, int n_weight_q,
                                        const int32_t *bias_q, int n_bias_q, const int32_t *act_q, int n_act_q, int root);
int  yolo2_hip_load_weights_fp32_bcast(yolo2_hip_ctx *ctx, const float *weights_reorg, size_t n_weights,
                                       const float *bias, size_t n_bias, int root);
/* Failure of a _bcast loader is COLLECTIVE: every rank first does its local part (the root: argument checks, allocation, H2D;
 * the others: allocation), the ranks agree on a status word (one ncclAllReduce(min) of an int), and either all of them run the
 * broadcast or all of them return an error; a second agreement follows the per-rank load.  A root-side error therefore never
 * leaves the other ranks blocked inside ncclBroadcast.  A rank whose process dies is the launcher's business. */

/* What the communicator itself reports (ncclCommCount / ncclCommUserRank / ncclCommCuDevice / ncclGetVersion), the file the
 * RCCL entry points were resolved from (dladdr: a process that already mapped a librccl with the same soname - torch ships one -
 * gets that copy from dlopen), and the most recent weight broadcast through it.  "Did RCCL see N ranks" is answered by nranks. */
typedef struct {
    int nranks, rank, device;
    int version;                 /* ncclGetVersion: e.g. 22606 */
    int bcasts;                  /* weight broadcasts run through this communicator */
    double last_bcast_ms;        /* host wall time of the last one (enqueue to stream-synchronised), this rank */
    uint64_t last_bcast_bytes;   /* bytes each rank received / the root sent per peer */
    char lib_path[256];
} yolo2_hip_rccl_info_t;
int  yolo2_hip_rccl_info(yolo2_hip_ctx *ctx, yolo2_hip_rccl_info_t *out);
int  yolo2_hip_multi_rccl_info(yolo2_hip_multi *m, yolo2_hip_rccl_info_t *out);   /* model (a): rank/device of member 0 */

/* ------------------------------------------------------------------- tier 3: helpers */

/* The layer table the batched entry implements (config/yolov2.cfg as parsed by
 * src/core/yolo_net.cpp:218-291).  desc = {type, c, h, w, n, size, stride, pad, leaky} with
 * type 0 conv, 1 maxpool, 2 reorg, 3 route, 4 region (linux_app/include/yolo2_config.h:118-122).
 * Hosts that parse a .cfg compare it with this before using yolo2_hip_run_batch_*. */
int yolo2_hip_num_layers(void);
int yolo2_hip_layer_desc(int layer_idx, int desc[9]);

/* weights_reorg_int16.bin / bias_int16.bin carry one pad element after every odd-length
 * layer (yolo2_model.cpp:198-224).  Returns elements written, or -1 if the file is short. */
long yolo2_strip_int16_layer_pad(const int16_t *file, size_t file_elems, const int *layer_len,
                                 int n_layers, int16_t *dst);
extern const int yolo2_weight_len[YOLO2_N_CONV]; /* model_config.cpp:4-7  */
extern const int yolo2_bias_len[YOLO2_N_CONV];   /* model_config.cpp:9-10 */

#ifdef __cplusplus
}
#endif
#endif /* YOLO2_HIP_H */
