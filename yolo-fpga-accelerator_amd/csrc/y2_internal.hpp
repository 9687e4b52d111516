// y2_internal.hpp -- what the translation units of libyolo2_hip.so share (nothing here is part of the C ABI).
//
//   yolo2_hip.hip     errors, model tables, context lifecycle, profiling, pre-processing, host-buffer streaming entries
//   yolo2_driver.hip  tier 1: the reference's accelerator-driver interface (linux_app/include/yolo2_accel_linux.h,
//                     dma_buffer_manager.h): device state, buffers, register file, per-layer calls
//   yolo2_int16.hip   the int16 path: weight loading / arithmetic-form proofs, launch planning, autotune, run_batch_int16
//   yolo2_fp16.hip    the fp16 MFMA path: weight packing, per-context launch table, run_batch_fp16
//   yolo2_fp32.hip    the exact fp32 path (tiled and one-thread-per-output)
//   yolo2_post.hip    region layer + boxes + NMS;   yolo2_multi.hip  frame sharding + the RCCL weight broadcast
// Every non-template kernel is launched from exactly one of them (its kernels_*.hpp is included there only).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/yolo2_hip.h"
#include "conv_common.hpp"
#include "layout.hpp"

// ---------------------------------------------------------------------------- errors (yolo2_hip.hip)

// stores the message for yolo2_hip_last_error() (thread-local) and returns `code`
int y2_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
#define fail y2_fail

#define HIP_TRY(expr, code)                                                                      \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) return fail(code, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)


// ---------------------------------------------------------------------------- options (yolo2_plan.hip)
//
// Every switch that steers kernel selection, lanes or planning lives here, ONE object per context: filled from the environment
// (YOLO2_<NAME>, upper case) once, when the context is created, and changed on a live context only through
// yolo2_hip_set_option(ctx, "<name>", "<value>").  Nothing on a planning or launch path calls getenv (round 3 read 26 variables at
// plan time, several of them per plan_conv call).  A lane inherits its parent's object.  README.md lists the names.
// Groups: (a) planning knobs a deployment may set (autotune, plan_file, plan_write, lanes, no_lanes, lane_priority, verbose);
// (b) A/B switches of kernel families (no_* / f16_*); (c) test hooks that force one kernel shape everywhere (force_*, f32_p).
struct Y2Options {
    // (a)
    int autotune = -1;            // -1: weight cache, then plan table, then timing; 0: static defaults, nothing timed; 1: always time
    std::string plan_file;        // replaces config/plan_gfx950.txt next to the library
    std::string plan_write;       // append every timed plan to this file (tools/make_plan.py)
    int lanes = 0;                // int16 lanes (0 = the default rule: 3 for batches 48..127, 2 from batch 16)
    bool no_lanes = false;
    std::string lane_split;       // diagnostic: "20,22,22"
    int lane_priority = 1;        // 0: lanes at default stream priority, lane 0 on the caller's stream (round-3 interim design)
    bool verbose = false;
    bool no_plan_cache = false;   // ignore a bound weight-side plan cache
    // (b)
    int splitk = -1;              // -1 tuned; 0 no K-split of either kind; 1 the lane-split kernel wherever legal
    int poolfuse = -1;            // 1: conv + pool fused wherever legal
    bool no_poolfuse = false, no_hiacc = false, no_ks = false, no_w16 = false, no_grp = false, grp16 = false, no_xcd_remap = false, splitk_no_pack = false;
    int f16_lanes = 2;
    bool f16_no_lanes = false, f16_no_mfma0 = false, f16_no_glds = false, f16_no_poolfuse = false, f16_no_halo = false, f16_no_persist = false,
         f16_persist_all = false, f16_ring_all = false, f16_no_ring = false, f16_no_c32 = false, f16_m16 = false, f16_w8 = false, f16_no_wide = false,
         f16_no_fuse1x1 = false, f16_no_rw = false, f16_no_rwb = false, f16_no_rwc = false, f16_ring256 = false, f16_ring_sq = false;
    int stamp_layer = -1;
    int f16_skip = 0;             // DIAGNOSTIC (results wrong by construction): bit mask of fp16 launches left out of the table, to measure
                                  // what a launch costs INSIDE the two-lane step: 1 = 1x1 layers 5 and 9, 2 = layer 0, 4 = layers 2/4/6,
                                  // 8 = pools + reorg, 16 = the other 1x1 layers, 32 = the 13x13 3x3 layers
    // (c)
    int force_path = -1, force_p = 0, force_ks = 0, f32_p = 0;
    bool force_w16 = false, force_hiacc = false;

    static Y2Options from_env();
    // 0 = set, -1 = unknown name / bad value.  value NULL or "" restores the default.
    int set(const char *name, const char *value);
    // the settings that differ from the defaults, "name=value name=value" ("" if none): what bench.py discloses
    std::string describe() const;
    // switches that ask for a plan the committed table / the weight cache do not hold: those are bypassed
    bool steered() const { return splitk >= 0 || poolfuse >= 0 || no_poolfuse || no_w16 || no_hiacc || no_ks || no_grp || no_xcd_remap; }   // (grp16 keeps the plan sources: it changes no plan field)
};
const Y2Options &y2_process_options();   // from the environment, parsed once: the context-less driver tier and process-wide latches

// ---------------------------------------------------------------------------- launch plans as data (yolo2_plan.hip)

// One launch of a conv layer as the plan table / the weight cache store it:  B L S path P pad splitk pp w16 fuse hiacc ks
struct Y2PlanLine { int path, P, pad, splitk, pp, w16, fuse, hiacc, ks; };
typedef std::pair<int, std::pair<int, int>> Y2PlanKey;   // (B, (L, S))
// parses and range-checks one line (a line outside what the planner can produce is rejected); false: not a plan line
bool y2_plan_line_parse(const char *text, Y2PlanKey *key, Y2PlanLine *pl);
// the committed table (config/plan_gfx950.txt next to the library, or opt.plan_file): loaded once per file name
bool y2_plan_table_has_batch(const Y2Options &opt, int B);
bool y2_plan_table_lookup(const Y2Options &opt, int B, int L, int S, Y2PlanLine *out);

// The K-split-across-workgroups kernel's scratch rule on plain numbers (also exported: yolo2_hip_i16_plan_check):
// `splits` triples of 24 bytes per output item; cap = 0 (a context without scratch) refuses every split.
bool y2_ks_fits(int splits, int cg_out, int npix, size_t cap_bytes);
size_t y2_ks_bytes(int splits, int cg_out, int npix);

// The weight-side cache: one small text file beside a weight set, keyed by a hash of the blobs and Q tables,
// holding what loading and planning would otherwise recompute - the per-block bounds behind the arithmetic-form proofs (so
// k_weight_bound* is skipped), the forms and scale shifts derived from them (re-derived and compared at load: a mismatch drops the
// file) and the conv plan of every batch this weight set has been timed for (so the autotune is skipped and every process runs
// the same kernels).  A missing, stale or damaged file costs time, never correctness: hash and body checksum must both match.
struct Y2PlanCache {
    std::mutex mu;
    std::string path;
    uint64_t hash = 0;            // of the weight set currently loaded (0 = none yet)
    bool bounds_valid = false;    // the file matched this weight set and its bounds were used
    struct Bounds { int maxsum = 0, maxbias = 0; std::vector<int> sum_mb, bias_mb, abs_mb, form, scale; };
    Bounds bounds[YOLO2_N_CONV];
    std::map<Y2PlanKey, Y2PlanLine> lines;
    std::map<int, int> per_batch;
    bool dirty = false;
    // reads `path`; true iff header, hash and checksum match `want_hash` (then bounds / lines are filled)
    bool load(uint64_t want_hash, std::string *why);
    bool save();                   // temp file + rename; false (and nothing else) if the directory is not writable
};
uint64_t y2_hash_bytes(uint64_t seed, const void *data, size_t n);   // host side: Q tables, file body

// ---------------------------------------------------------------------------- model table (yolo2_hip.hip)

enum LType { L_CONV, L_MAX, L_ROUTE, L_REORG, L_REGION };
struct LayerDesc {
    LType type;
    int c, h, w, n, size, leaky;
};
// config/yolov2.cfg as parsed by the reference; the C host re-derives the same
// table from the .cfg file and checks it against this one before using the batched entry.
extern const LayerDesc kNet[32];

static inline unsigned blocks_for(long n, int bs) { return (unsigned)((n + bs - 1) / bs); }
static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
static inline long packed_weight_elems(int C, int N, int K)
{
    return (long)((N + y2::kTm - 1) / y2::kTm) * ((C + y2::kTn - 1) / y2::kTn) * K * K * 128;
}

// ---------------------------------------------------------------------------- launch plan of one conv launch (int16 / fp32 tiled)

constexpr int kMaxTileItems = 2048;  // 8 staging registers x 256 threads (k_conv_i16 NST <= 8)

struct ConvPlan {
    int C = 0, N = 0, K = 0, H = 0, W = 0, leaky = 0;
    int Qw = 0, Qa_in = 0, Qa_out = 0, Qb = 0;
    int path = 0;  // 0 = form A, 1 = form B (pre-shifted accumulator), 3 = form C (packed int16), 4 = form D (C, shift-free), 2 = 64-bit
    int P = 8;
    int mb_count = 0;          // output-channel blocks this launch covers (0 = all of the layer)
    int splitk_ok = 0;         // the loader proved the split-K bounds for these blocks (|t| < 2^29, sums < 2^30)
    int splitk = 0;            // small batches: S K-splits x 64/S pixels per wavefront, shuffle-combined (k_conv_i16_splitk); 0 or S
    int splitk_pp = 1;         // pixels per lane of the split-K kernel (2: two pixel tiles share the staged weight slices)
    int splitk_pack = 0;       // the lane-split kernel carries packed int16 triples (form D launches)
    size_t ks_cap = 0;         // bytes of the context's triple scratch: plan_conv refuses a split whose triples would not fit
    int ks = 0;                // single frames: K-split across workgroups (k_conv_i16_ks + k_ks_finalize); 0 or the number of splits
    int hiacc = 0;             // form D launches only: 1 = the kernel keeps one accumulator register per channel (no v_perm: MODE 5)
    int w16 = 0;               // 1: k_conv_i16_w16 - two wavefronts of 16 output channels each per workgroup instead of four of 8
    int grp = 1;               // 1x1 convs: channel groups per barrier (8 when it divides CGin and fits LDS staging)
    int pool_fused = 0;        // conv + leaky + 2x2 pool in one kernel (k_conv_i16_pool): 1 = pooled tensor only, 2 = + full tensor
    int lds_pad = 0;           // extra dynamic LDS requested only to cap workgroups per CU (autotuned):
                               // fewer co-resident workgroups finish sooner each, which shortens the
                               // idle tail of layers that are only a few workgroup-generations long
    dim3 grid;
    int lds_bytes = 0;
    y2::ConvArgs args;
};

// ---------------------------------------------------------------------------- the context

struct Tensor {
    y2::ActGeom g;
    int2 *d = nullptr;
};

// Device buffers of one post-processing call in flight (yolo2_post.hip)
struct Y2PostBufs {
    int cap_frames = 0;
    size_t cap_dets = 0;
    float *rows = nullptr, *rows2 = nullptr;
    int *totals = nullptr, *counts = nullptr;
    void *geom = nullptr;            // [cap_frames] letterbox-correction records (y2_post_geom_bytes() each)
    yolo2_hip_det *dets = nullptr;   // [cap_frames][cap]
};

// Staging for the host-buffer entries (run_frames / run_images): two buffer sets and three streams
// (upload, kernels, download), kept with the context and grown on demand so that a caller streaming
// chunk after chunk does not pay pinned-memory allocation per call.
struct PipeBufs {
    size_t host_in = 0, dev_in = 0;   // capacities in bytes (dev_in: raw image bytes, 0 for float frames)
    int batch = 0;
    uint8_t *hin[2] = {nullptr, nullptr}, *dbytes[2] = {nullptr, nullptr};
    float *din[2] = {nullptr, nullptr};
    int16_t *hout[2] = {nullptr, nullptr}, *dout[2] = {nullptr, nullptr};
    hipStream_t s_in = nullptr, s_run = nullptr, s_out = nullptr;
    hipEvent_t e_in[2] = {nullptr, nullptr}, e_run[2] = {nullptr, nullptr}, e_out[2] = {nullptr, nullptr};
    hipEvent_t e_lane[2][8] = {};      // images -> detections entry: lane i has finished its part of the chunk in buffer set b (created on demand)
    // the tail as a pipeline stage (yolo2_hip_run_images_u8_dets): per buffer set its device buffers and pinned host mirrors
    int post_batch = 0, post_cap = 0;
    Y2PostBufs post[2];
    uint8_t *hgeom[2] = {nullptr, nullptr};
    yolo2_hip_det *hdets[2] = {nullptr, nullptr};
    int *hcounts[2] = {nullptr, nullptr};
};

struct F16Plan;   // yolo2_fp16.hip: the per-context launch table of the fp16 path

struct yolo2_hip_ctx {
    PipeBufs pipe;
    int device = 0;
    Y2Options opt;                     // parsed once at creation (environment), changed only by yolo2_hip_set_option; lanes copy it
    std::shared_ptr<Y2PlanCache> plan_cache;   // weight-side cache bound by yolo2_hip_set_plan_cache (lanes share the parent's)
    bool weights_loaded = false;
    short *wpk = nullptr;      // all layers, packed
    short *bias_pk = nullptr;  // all layers, padded to 32
    long wpk_off[YOLO2_N_CONV], bias_off[YOLO2_N_CONV];
    int maxsum[YOLO2_N_CONV], maxbias[YOLO2_N_CONV];
    std::vector<int> weight_q, bias_q, act_q;
    ConvPlan plan[32];                 // per conv layer: the launch covering most output-channel blocks
    std::vector<ConvPlan> extra[32];   // further launches for blocks that need another arithmetic form
    std::vector<int> maxsum_mb[YOLO2_N_CONV], maxbias_mb[YOLO2_N_CONV], maxabs_mb[YOLO2_N_CONV];
    std::vector<signed char> wscale_mb[YOLO2_N_CONV];   // log2 of the factor each block's packed weights currently carry (form D)
    std::vector<signed char> form_mb[YOLO2_N_CONV];     // arithmetic form resolve_q chose per block of 32 output channels
    int *mb_lists = nullptr;           // device: block index lists of all split layers
    int plan_source = 0;               // how set_batch planned the conv launches: 1 plan table, 2 timed (autotune), 3 static defaults, 4 forced, 5 weight cache
    int *ks_trip = nullptr;            // device scratch of the K-split-across-workgroups kernel (triples of every split): sized from the accepted plans
    size_t ks_trip_bytes = 0;
    // Lanes: a batch is run as part-batches on internal streams (forked from / joined to the
    // caller's stream with events).  Every layer is then several concurrent launches, and the idle tail of
    // one (a layer is only a few workgroup-generations long at batch 64) is filled by the others.
    // A lane is a child context that shares the parent's weights.
    std::vector<yolo2_hip_ctx *> lanes;
    std::vector<int> lane_first;       // first frame of each lane within the batch
    std::vector<yolo2_hip_ctx *> f16_lanes;   // fp16 path: two half-batch lanes (share wh / biasf / w0f)
    bool is_lane = false, laned = false;
    hipStream_t lane_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool fuse_pool[32] = {false};      // conv layer i stores the pooled tensor of layer i+1 itself (k_conv_i16_pool)
    ConvPlan fplan[32];                // the fused launches of those layers (plan / extra keep the unfused ones)
    std::vector<ConvPlan> fextra[32];
    int path_counts[YOLO2_N_CONV][5];
    int reorg_shift = 0, final_q = 0;
    int batch = 0;
    Tensor t_in, t_out[32], t_cat;
    // ---- fp16 MFMA path
    struct HalfTensor {
        int C = 0, Cp = 0, H = 0, W = 0, Wp = 0, PL = 0, B = 0;
        size_t items = 0;
        _Float16 *d = nullptr;
    };
    bool f16_loaded = false;
    // split-fp16 ("fp32tol") mode: a twin context that runs the fp16 launch table on items of three parts [hi | lo | hi] with
    // weights packed [w_hi | w_hi | w_lo] (kernels_f16.hpp, SPLIT instantiations).  The twin owns its packed weights (wh / biasf), its
    // tensors, table and lanes; it borrows the fp32 blobs and w0f from this context.
    bool split = false;                // this context IS such a twin (or a lane of one)
    bool borrows_f32 = false;          // w0f / wf32 / bf32 belong to the parent
    yolo2_hip_ctx *tol = nullptr;      // the parent's twin, made at the first yolo2_hip_run_batch_f32tol
    _Float16 *wh = nullptr;
    float *biasf = nullptr;
    float *w0f = nullptr;  // layer 0: [27][32] fp32 weights + [32] bias for the fused conv0+pool kernel
    float *wf32 = nullptr, *bf32 = nullptr;   // the fp32 blobs as loaded (reference stream order), for the exact fp32 pass
    F16Plan *f16_plan = nullptr;              // launch table of the fp16 pass (built once per (weights, batch); owned)
    // ---- tiled exact fp32 path (kernels_f32.hpp): packed weights [mb][cg][tap][32][4] floats, items of 4 floats
    float *wpkf = nullptr, *biasf32_pk = nullptr;
    long wpkf_off[YOLO2_N_CONV], biasf32_off[YOLO2_N_CONV];
    struct FTensor {
        y2::ActGeom g;
        float4 *d = nullptr;
    };
    int f32_batch = 0;
    FTensor f_in, f_out[32], f_cat;
    ConvPlan fp32_plan[32];
    long wh_off[YOLO2_N_CONV], biasf_off[YOLO2_N_CONV];
    int f16_batch = 0;
    HalfTensor h_in, h_out[32], h_cat;
    // per-layer device timing: a ring of event sets, one per profiled run (hipEvents on the
    // stream the kernels are launched on); the analogue of yolo2_inference.c:75-142
    static constexpr int kProfSlots = 32;
    bool prof = false;
    hipEvent_t ev[kProfSlots][33];
    bool ev_made = false;
    long prof_runs = 0;
};

// ---------------------------------------------------------------------------- helpers that cross translation units

// yolo2_hip.hip
void y2_free_activations(yolo2_hip_ctx *c);
void y2_free_f16_activations(yolo2_hip_ctx *c);
void y2_free_f32_activations(yolo2_hip_ctx *c);
void y2_destroy_lanes(yolo2_hip_ctx *c);
// Streams of the lanes (yolo2_hip.hip): created at the device's HIGHEST stream priority.  HIP multiplexes the streams of a process
// onto a few hardware queues PER PRIORITY LEVEL (4 by default); at the default priority the lanes share that pool with every other
// stream of the process - the streaming entries' copy streams, torch's and RCCL's internal streams - and two lanes that land on one
// queue run one after the other (measured: -5 % under torch.distributed, -10 % in the streaming entry).  At their own level the
// lanes have a pool to themselves.  own_stream_for_lane0 = false (YOLO2_LANE_PRIORITY=0, the round-3 interim design): default
// priority, lane 0 runs on the caller's stream.
int y2_lane_stream_create(hipStream_t *s);
bool y2_lane0_own_stream();
int y2_ensure_prof_events(yolo2_hip_ctx *c);
int y2_ensure(void **p, size_t *cap, size_t need);   // grow-only device scratch

// yolo2_fp16.hip
void y2_f16_plan_free(yolo2_hip_ctx *c);

// The per-layer driver calls' device work (yolo2_driver.hip validates, latches the register file, takes the lock, binds the
// device, and synchronises with the reference's timeout semantics afterwards; these only enqueue on the null stream).
// yolo2_int16.hip:
int y2_drv_conv_i16(const short *in, short *out, const short *w, const short *beta, int ifm, int ofm, int ksize, int kstride,
                    int iw, int ih, int ow, int oh, int pad, int is_nl, int qw, int qa_in, int qa_out, int qb, int *path_out);
void y2_drv_pool_i16(const short *in, short *out, int channels, int ksize, int kstride, int iw, int ih, int ow, int oh);
void y2_drv_release_i16(void);   // frees the grow-only scratch of y2_drv_conv_i16 (yolo2_accel_cleanup)
// yolo2_fp32.hip:
void y2_drv_conv_f32(const float *in, float *out, const float *w, const float *beta, int ifm, int ofm, int ksize, int kstride, int iw,
                     int ih, int ow, int oh, int pad, int is_nl);

// yolo2_post.hip: the tail (region + boxes + NMS + record compaction) as a stage of a pipeline - caller-owned buffers, enqueue only.
int y2_post_alloc(int device, int batch, int cap, Y2PostBufs *b);
void y2_post_free(Y2PostBufs *b);
size_t y2_post_geom_bytes(void);
int y2_post_fill_geom(void *geom_host, const int *im_w, const int *im_h, int n);   // host side: correct_region_boxes' per-frame constants
// region tensor [batch][425][13][13] int16 on `device` -> b->dets / b->counts (cap records per frame; best_only: one per detection);
// b->geom must hold the frames' records (copied on `st` in front of this call).  Enqueues on st, returns without synchronising.
int y2_post_enqueue_int16(int device, const int16_t *region_dev, int batch, int final_q, float thresh, float nms, int cap, int best_only,
                          Y2PostBufs *b, hipStream_t st);

