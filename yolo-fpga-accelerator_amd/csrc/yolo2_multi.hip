// yolo2_multi.hip -- more than one MI355X behind the C ABI (include/yolo2_hip.h, "multi-GPU" section).
//
// The path shards by frame (frames are independent, no cross-frame state): every GPU holds the
// whole weight set, frames are split into contiguous ranges, and the ONLY collective is the broadcast of the
// weight blobs at init - ncclBroadcast from librccl.so, i.e. RCCL over xGMI.  Nothing here launches a kernel:
// it is built on the single-device entries of yolo2_hip.hip and on ONE broadcast routine (bcast_blobs) that
// both launch models share:
//   * one process, n devices  (the C host: yolov2_detect --devices 0,1,...): ncclCommInitAll, one host thread
//     per device for the frame shards;
//   * one process per device  (torchrun / MPI style, what bench.py --gpus N runs): rank 0 creates a
//     ncclUniqueId, the launcher distributes its 128 bytes, every rank calls ncclCommInitRank.
// librccl.so is dlopen()ed at the first multi-GPU call, so single-GPU users never load it.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/yolo2_hip.h"

extern "C" int yolo2_hip_set_error(int code, const char *msg);   // yolo2_hip.hip: stores the message for yolo2_hip_last_error()

namespace {

int mfail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    return yolo2_hip_set_error(code, buf);
}

// ---- librccl.so, loaded on first use
struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommCuDevice)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string path;   // the file ncclBroadcast was resolved from (dladdr): a process that has already mapped a librccl
                        // with the same soname (torch ships its own) gets THAT copy from dlopen, and the record says so
};
Rccl g_rccl;
std::once_flag g_rccl_once;
std::string g_rccl_err;

const Rccl *rccl()
{
    std::call_once(g_rccl_once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names)
            if ((g_rccl.h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
        if (!g_rccl.h) { g_rccl_err = std::string("cannot load librccl.so: ") + dlerror(); return; }
#define Y2_SYM(field, name)                                                         \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(g_rccl.h, name)); \
    if (!g_rccl.field) { g_rccl_err = std::string("librccl.so lacks ") + name; dlclose(g_rccl.h); g_rccl.h = nullptr; return; }
        Y2_SYM(GetUniqueId, "ncclGetUniqueId")
        Y2_SYM(CommInitRank, "ncclCommInitRank")
        Y2_SYM(CommInitAll, "ncclCommInitAll")
        Y2_SYM(CommDestroy, "ncclCommDestroy")
        Y2_SYM(Broadcast, "ncclBroadcast")
        Y2_SYM(AllReduce, "ncclAllReduce")
        Y2_SYM(GroupStart, "ncclGroupStart")
        Y2_SYM(GroupEnd, "ncclGroupEnd")
        Y2_SYM(CommCount, "ncclCommCount")
        Y2_SYM(CommUserRank, "ncclCommUserRank")
        Y2_SYM(CommCuDevice, "ncclCommCuDevice")
        Y2_SYM(GetVersion, "ncclGetVersion")
        Y2_SYM(GetErrorString, "ncclGetErrorString")
#undef Y2_SYM
        Dl_info info;
        if (dladdr(reinterpret_cast<void *>(g_rccl.Broadcast), &info) && info.dli_fname) g_rccl.path = info.dli_fname;
    });
    return g_rccl.h ? &g_rccl : nullptr;
}

#define RCCL_TRY(expr)                                                                                         \
    do {                                                                                                       \
        ncclResult_t r_ = (expr);                                                                              \
        if (r_ != ncclSuccess) return mfail(YOLO2_DMA_ERROR, "%s failed: %s", #expr, rccl()->GetErrorString(r_)); \
    } while (0)
#define HIPM_TRY(expr, code)                                                                              \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) return mfail(code, "%s failed: %s", #expr, hipGetErrorString(e_));          \
    } while (0)

// One member of a broadcast: a device, its communicator handle and its copies of the blobs.
struct Member {
    int device = 0;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    std::vector<void *> bufs;
};

// THE broadcast (both launch models): every blob travels as bytes from `root`'s buffer to every other member's.
// `members` are this PROCESS's members (all of them with ncclCommInitAll, exactly one with ncclCommInitRank);
// the calls of one process are grouped, as RCCL requires when one thread drives several devices.  A failed call inside
// a group does not skip ncclGroupEnd (an open group would swallow every later RCCL call of this thread).
int bcast_blobs(std::vector<Member> &members, const std::vector<size_t> &bytes, int root, double *ms_out = nullptr)
{
    const Rccl *R = rccl();
    if (!R) return mfail(YOLO2_INIT_ERROR, "%s", g_rccl_err.c_str());
    const auto t0 = std::chrono::steady_clock::now();
    int rc = YOLO2_SUCCESS;
    for (size_t k = 0; k < bytes.size() && rc == YOLO2_SUCCESS; ++k) {
        RCCL_TRY(R->GroupStart());
        for (Member &m : members) {
            if (hipSetDevice(m.device) != hipSuccess) { rc = mfail(YOLO2_INIT_ERROR, "hipSetDevice(%d) failed", m.device); break; }
            const ncclResult_t r = R->Broadcast(m.bufs[k], m.bufs[k], bytes[k], ncclInt8, root, m.comm, m.stream);
            if (r != ncclSuccess) { rc = mfail(YOLO2_DMA_ERROR, "ncclBroadcast failed: %s", R->GetErrorString(r)); break; }
        }
        const ncclResult_t ge = R->GroupEnd();
        if (ge != ncclSuccess && rc == YOLO2_SUCCESS) rc = mfail(YOLO2_DMA_ERROR, "ncclGroupEnd failed: %s", R->GetErrorString(ge));
    }
    for (Member &m : members) {   // drain what was enqueued, also on failure
        if ((hipSetDevice(m.device) != hipSuccess || hipStreamSynchronize(m.stream) != hipSuccess) && rc == YOLO2_SUCCESS)
            rc = mfail(YOLO2_DMA_ERROR, "synchronising the broadcast stream of device %d failed", m.device);
    }
    if (ms_out) *ms_out = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

// Agreement on a status word across the communicator (one-process-per-device model): every rank contributes its local
// status (0 or a negative YOLO2_* code), every rank receives the minimum.  This is what makes a failure COLLECTIVE: a root
// that cannot read its blobs, or a rank that cannot allocate, does not leave the others blocked inside the broadcast.
int agree_status(const Rccl *R, int device, ncclComm_t comm, hipStream_t stream, int *dev_word, int local, int *agreed)
{
    HIPM_TRY(hipSetDevice(device), YOLO2_INIT_ERROR);
    HIPM_TRY(hipMemcpyAsync(dev_word, &local, sizeof(int), hipMemcpyHostToDevice, stream), YOLO2_DMA_ERROR);
    RCCL_TRY(R->AllReduce(dev_word, dev_word, 1, ncclInt32, ncclMin, comm, stream));
    HIPM_TRY(hipMemcpyAsync(agreed, dev_word, sizeof(int), hipMemcpyDeviceToHost, stream), YOLO2_DMA_ERROR);
    HIPM_TRY(hipStreamSynchronize(stream), YOLO2_DMA_ERROR);
    return YOLO2_SUCCESS;
}

// Q tables travel in one fixed-size int32 record: [n_wq, n_bq, n_aq, 3 x 64 values]
constexpr int kQRec = 3 + 3 * 64;
int pack_q(int32_t *rec, const int32_t *wq, int nw, const int32_t *bq, int nb, const int32_t *aq, int na)
{
    if (nw < 0 || nb < 0 || na < 0 || nw > 64 || nb > 64 || na > 64) return mfail(YOLO2_ERROR, "Q tables longer than 64 entries");
    memset(rec, 0, kQRec * sizeof(int32_t));
    rec[0] = nw; rec[1] = nb; rec[2] = na;
    memcpy(rec + 3, wq, nw * sizeof(int32_t));
    memcpy(rec + 3 + 64, bq, nb * sizeof(int32_t));
    memcpy(rec + 3 + 128, aq, na * sizeof(int32_t));
    return YOLO2_SUCCESS;
}

}  // namespace

// ---------------------------------------------------------------------------- shard arithmetic

extern "C" int yolo2_hip_shard_range(int total, int rank, int world, int *lo, int *hi)
{
    if (total < 0 || world <= 0 || rank < 0 || rank >= world || !lo || !hi) return mfail(YOLO2_ERROR, "bad shard arguments");
    const int base = total / world, rem = total % world;
    *lo = rank * base + std::min(rank, rem);
    *hi = *lo + base + (rank < rem ? 1 : 0);
    return YOLO2_SUCCESS;
}

// ---------------------------------------------------------------------------- one process per device

struct RankState {
    ncclComm_t comm = nullptr;
    int nranks = 1, rank = 0, device = 0;
    hipStream_t stream = nullptr;
    std::mutex use;              // held while a collective of this rank is in flight (finalize waits for it)
    double last_bcast_ms = 0;    // the most recent weight broadcast through this communicator
    size_t last_bcast_bytes = 0;
    int bcasts = 0;
};
// Contexts that joined a communicator.  The states are shared_ptr: a caller takes its own reference under g_rank_mu and keeps
// the state alive while it uses it, whatever a concurrent init_rank / finalize of ANOTHER context does to the vector.
static std::mutex g_rank_mu;
static std::vector<std::pair<yolo2_hip_ctx *, std::shared_ptr<RankState>>> g_ranks;

static std::shared_ptr<RankState> rank_state(yolo2_hip_ctx *c)   // g_rank_mu must be held
{
    for (auto &p : g_ranks)
        if (p.first == c) return p.second;
    return nullptr;
}
static std::shared_ptr<RankState> rank_state_locked(yolo2_hip_ctx *c)
{
    std::lock_guard<std::mutex> lk(g_rank_mu);
    return rank_state(c);
}

extern "C" int yolo2_hip_rccl_unique_id(void *id128)
{
    if (!id128) return mfail(YOLO2_ERROR, "null id buffer");
    const Rccl *R = rccl();
    if (!R) return mfail(YOLO2_INIT_ERROR, "%s", g_rccl_err.c_str());
    ncclUniqueId id;
    RCCL_TRY(R->GetUniqueId(&id));
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    memcpy(id128, &id, sizeof(id));
    return YOLO2_SUCCESS;
}

extern "C" int yolo2_hip_rccl_init_rank(yolo2_hip_ctx *ctx, int device, const void *id128, int nranks, int rank)
{
    if (!ctx || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return mfail(YOLO2_ERROR, "bad communicator arguments");
    const Rccl *R = rccl();
    if (!R) return mfail(YOLO2_INIT_ERROR, "%s", g_rccl_err.c_str());
    std::lock_guard<std::mutex> lk(g_rank_mu);
    if (rank_state(ctx)) return mfail(YOLO2_ERROR, "this context already belongs to a communicator");
    HIPM_TRY(hipSetDevice(device), YOLO2_INIT_ERROR);
    auto st = std::make_shared<RankState>();
    st->nranks = nranks; st->rank = rank; st->device = device;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    RCCL_TRY(R->CommInitRank(&st->comm, nranks, id, rank));
    if (hipStreamCreateWithFlags(&st->stream, hipStreamNonBlocking) != hipSuccess) {
        (void)R->CommDestroy(st->comm);
        return mfail(YOLO2_ERROR, "stream creation for the communicator failed");
    }
    g_ranks.emplace_back(ctx, st);
    return YOLO2_SUCCESS;
}

extern "C" void yolo2_hip_rccl_finalize(yolo2_hip_ctx *ctx)
{
    std::shared_ptr<RankState> st;
    {
        std::lock_guard<std::mutex> lk(g_rank_mu);
        for (size_t i = 0; i < g_ranks.size(); ++i)
            if (g_ranks[i].first == ctx) {
                st = g_ranks[i].second;
                g_ranks.erase(g_ranks.begin() + (long)i);
                break;
            }
    }
    if (!st) return;
    std::lock_guard<std::mutex> use(st->use);   // a collective still in flight on this rank finishes first
    (void)hipSetDevice(st->device);
    if (st->comm && rccl()) (void)rccl()->CommDestroy(st->comm);
    if (st->stream) (void)hipStreamDestroy(st->stream);
    st->comm = nullptr;
    st->stream = nullptr;
}

// What the communicator itself says (ncclCommCount / ncclCommUserRank / ncclCommCuDevice), the library it was resolved from
// and the last weight broadcast through it: lets a launcher prove that RCCL saw N ranks (bench.py's "rccl" object).
extern "C" int yolo2_hip_rccl_info(yolo2_hip_ctx *ctx, yolo2_hip_rccl_info_t *out)
{
    if (!ctx || !out) return mfail(YOLO2_ERROR, "null argument");
    const std::shared_ptr<RankState> st = rank_state_locked(ctx);
    if (!st) return mfail(YOLO2_ERROR, "yolo2_hip_rccl_init_rank() has not been called for this context");
    const Rccl *R = rccl();
    if (!R) return mfail(YOLO2_INIT_ERROR, "%s", g_rccl_err.c_str());
    std::lock_guard<std::mutex> use(st->use);
    if (!st->comm) return mfail(YOLO2_ERROR, "the communicator has been finalized");
    memset(out, 0, sizeof(*out));
    RCCL_TRY(R->CommCount(st->comm, &out->nranks));
    RCCL_TRY(R->CommUserRank(st->comm, &out->rank));
    RCCL_TRY(R->CommCuDevice(st->comm, &out->device));
    RCCL_TRY(R->GetVersion(&out->version));
    out->bcasts = st->bcasts;
    out->last_bcast_ms = st->last_bcast_ms;
    out->last_bcast_bytes = (uint64_t)st->last_bcast_bytes;
    snprintf(out->lib_path, sizeof(out->lib_path), "%s", R->path.c_str());
    return YOLO2_SUCCESS;
}

// Root passes the host blobs, every other rank nullptr; all ranks receive identical device copies and load them.
// Failure is collective: every rank does its local part first (root: argument checks, allocation, H2D; others: allocation),
// the ranks then AGREE on a status word, and either all of them run the broadcast or all of them return an error - no rank is
// left waiting inside ncclBroadcast for a root that has already given up.  The same agreement follows the per-rank load.
// (A rank that cannot even reach the agreement - its process died - is the launcher's business: torchrun / mpirun end the job.)
template <typename T, typename LoadFn>
static int load_bcast(yolo2_hip_ctx *ctx, const T *weights, size_t n_weights, const T *bias, size_t n_bias, int32_t *qrec, int local_rc,
                      int root, LoadFn load)
{
    const std::shared_ptr<RankState> st = rank_state_locked(ctx);
    if (!st) return mfail(YOLO2_ERROR, "yolo2_hip_rccl_init_rank() has not been called for this context");
    const Rccl *R = rccl();
    if (!R) return mfail(YOLO2_INIT_ERROR, "%s", g_rccl_err.c_str());
    std::lock_guard<std::mutex> use(st->use);
    if (!st->comm) return mfail(YOLO2_ERROR, "the communicator has been finalized");
    // `root` must be the same number on every rank (it is an argument of the collective); a bad value is bad everywhere
    if (root < 0 || root >= st->nranks) return mfail(YOLO2_ERROR, "bad root rank %d", root);
    const bool is_root = st->rank == root;
    HIPM_TRY(hipSetDevice(st->device), YOLO2_INIT_ERROR);
    void *wd = nullptr, *bd = nullptr, *qd = nullptr;
    int *word = nullptr;
    const size_t wbytes = (size_t)YOLO2_N_WEIGHTS * sizeof(T), bbytes = (size_t)YOLO2_N_BIAS * sizeof(T), qbytes = kQRec * sizeof(int32_t);
    auto release = [&]() { (void)hipFree(wd); (void)hipFree(bd); (void)hipFree(qd); (void)hipFree(word); };
    // the status word first: without it this rank cannot take part in the agreement at all
    if (hipMalloc((void **)&word, sizeof(int)) != hipSuccess) return mfail(YOLO2_MMAP_ERROR, "device word for the status agreement could not be allocated");
    // ---- local part; errors are recorded, not returned
    int local = local_rc;
    std::string local_msg = local ? yolo2_hip_last_error() : "";
    auto note = [&](int code, const char *msg) { if (local == YOLO2_SUCCESS) { local = code; local_msg = msg; } };
    if (is_root && (!weights || !bias)) note(YOLO2_ERROR, "the root rank must pass the weight blobs");
    else if (is_root && (n_weights < YOLO2_N_WEIGHTS || n_bias < YOLO2_N_BIAS)) note(YOLO2_ERROR, "weight blobs too small");
    if (local == YOLO2_SUCCESS &&
        (hipMalloc(&wd, wbytes) != hipSuccess || hipMalloc(&bd, bbytes) != hipSuccess || hipMalloc(&qd, qbytes) != hipSuccess))
        note(YOLO2_MMAP_ERROR, "device buffers for the weight broadcast could not be allocated");
    if (local == YOLO2_SUCCESS && is_root &&
        (hipMemcpy(wd, weights, wbytes, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(bd, bias, bbytes, hipMemcpyHostToDevice) != hipSuccess ||
         hipMemcpy(qd, qrec, qbytes, hipMemcpyHostToDevice) != hipSuccess))
        note(YOLO2_DMA_ERROR, "H2D of the weight blobs failed");
    (void)hipGetLastError();
    // ---- agreement 1: does every rank have what the broadcast needs?
    int agreed = 0;
    int rc = agree_status(R, st->device, st->comm, st->stream, word, local, &agreed);
    if (rc == YOLO2_SUCCESS && agreed != YOLO2_SUCCESS)
        rc = local ? mfail(local, "%s", local_msg.c_str())
                   : mfail(agreed, "another rank could not prepare the weight broadcast (status %d): no rank loads", agreed);
    if (rc) { release(); return rc; }
    // ---- the broadcast
    std::vector<Member> me(1);
    me[0].device = st->device; me[0].comm = st->comm; me[0].stream = st->stream;
    me[0].bufs = {wd, bd, qd};
    double ms = 0;
    local = bcast_blobs(me, {wbytes, bbytes, qbytes}, root, &ms);
    if (local) local_msg = yolo2_hip_last_error();
    st->last_bcast_ms = ms; st->last_bcast_bytes = wbytes + bbytes + qbytes; st->bcasts++;
    if (local == YOLO2_SUCCESS && hipMemcpy(qrec, qd, qbytes, hipMemcpyDeviceToHost) != hipSuccess) note(YOLO2_DMA_ERROR, "D2H of the Q tables failed");
    if (local == YOLO2_SUCCESS && (local = load(wd, bd)) != YOLO2_SUCCESS) local_msg = yolo2_hip_last_error();
    (void)hipDeviceSynchronize();
    // ---- agreement 2: did every rank load?
    rc = agree_status(R, st->device, st->comm, st->stream, word, local, &agreed);
    if (rc == YOLO2_SUCCESS && agreed != YOLO2_SUCCESS)
        rc = local ? mfail(local, "%s", local_msg.c_str()) : mfail(agreed, "another rank failed to load the broadcast weights (status %d)", agreed);
    release();
    return rc;
}

extern "C" int yolo2_hip_load_weights_int16_bcast(yolo2_hip_ctx *ctx, const int16_t *weights_reorg, size_t n_weights, const int16_t *bias,
                                                  size_t n_bias, const int32_t *weight_q, int n_weight_q, const int32_t *bias_q,
                                                  int n_bias_q, const int32_t *act_q, int n_act_q, int root)
{
    int32_t rec[kQRec] = {0};
    int local = YOLO2_SUCCESS;   // a root-side argument error is carried INTO the collective (load_bcast), not returned in front of it
    if (weights_reorg) {   // root
        if (!weight_q || !bias_q || !act_q) local = mfail(YOLO2_ERROR, "null Q table");
        else local = pack_q(rec, weight_q, n_weight_q, bias_q, n_bias_q, act_q, n_act_q);
    }
    return load_bcast<int16_t>(ctx, weights_reorg, n_weights, bias, n_bias, rec, local, root, [&](void *wd, void *bd) {
        return yolo2_hip_load_weights_int16_dev(ctx, (uint64_t)(uintptr_t)wd, YOLO2_N_WEIGHTS, (uint64_t)(uintptr_t)bd, YOLO2_N_BIAS, rec + 3,
                                                rec[0], rec + 3 + 64, rec[1], rec + 3 + 128, rec[2]);
    });
}

extern "C" int yolo2_hip_load_weights_fp32_bcast(yolo2_hip_ctx *ctx, const float *weights_reorg, size_t n_weights, const float *bias,
                                                 size_t n_bias, int root)
{
    int32_t rec[kQRec] = {0};
    return load_bcast<float>(ctx, weights_reorg, n_weights, bias, n_bias, rec, YOLO2_SUCCESS, root, [&](void *wd, void *bd) {
        return yolo2_hip_load_weights_fp32_dev(ctx, (uint64_t)(uintptr_t)wd, YOLO2_N_WEIGHTS, (uint64_t)(uintptr_t)bd, YOLO2_N_BIAS);
    });
}

// ---------------------------------------------------------------------------- one process, n devices

struct yolo2_hip_multi {
    std::vector<int> devices;
    std::vector<yolo2_hip_ctx *> ctx;
    std::vector<ncclComm_t> comms;     // empty when there is a single device, or when a device is listed twice (no RCCL then)
    std::vector<hipStream_t> streams;
    bool duplicate = false;
    double last_bcast_ms = 0;
    size_t last_bcast_bytes = 0;
    int bcasts = 0;
};

extern "C" int yolo2_hip_multi_create(const int *devices, int n_devices, yolo2_hip_multi **out)
{
    if (!devices || n_devices < 1 || n_devices > 64 || !out) return mfail(YOLO2_ERROR, "bad device list");
    const int ndev = yolo2_hip_device_count();
    yolo2_hip_multi *m = new (std::nothrow) yolo2_hip_multi();
    if (!m) return mfail(YOLO2_ERROR, "out of host memory");
    m->devices.assign(devices, devices + n_devices);
    for (int i = 0; i < n_devices; ++i) {
        if (devices[i] < 0 || devices[i] >= ndev) { delete m; return mfail(YOLO2_INIT_ERROR, "no HIP device %d (the GPU path has no CPU fallback)", devices[i]); }
        for (int j = 0; j < i; ++j) m->duplicate |= devices[j] == devices[i];
    }
    int rc = YOLO2_SUCCESS;
    for (int i = 0; i < n_devices && rc == YOLO2_SUCCESS; ++i) {
        yolo2_hip_ctx *c = nullptr;
        rc = yolo2_hip_create(devices[i], &c);
        if (rc == YOLO2_SUCCESS) m->ctx.push_back(c);
    }
    for (int i = 0; i < n_devices && rc == YOLO2_SUCCESS; ++i) {
        hipStream_t s = nullptr;
        if (hipSetDevice(devices[i]) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess)
            rc = mfail(YOLO2_ERROR, "stream creation on device %d failed", devices[i]);
        else m->streams.push_back(s);
    }
    // A communicator over the listed devices.  RCCL refuses one device twice, so a list with duplicates (useful to
    // rehearse the sharding on a single GPU) has no communicator and falls back to device-to-device copies.
    if (rc == YOLO2_SUCCESS && n_devices > 1 && !m->duplicate) {
        const Rccl *R = rccl();
        if (!R) rc = mfail(YOLO2_INIT_ERROR, "%s", g_rccl_err.c_str());
        else {
            m->comms.resize((size_t)n_devices);
            const ncclResult_t r = R->CommInitAll(m->comms.data(), n_devices, devices);
            if (r != ncclSuccess) { m->comms.clear(); rc = mfail(YOLO2_INIT_ERROR, "ncclCommInitAll failed: %s", R->GetErrorString(r)); }
        }
    }
    if (rc) { yolo2_hip_multi_destroy(m); return rc; }
    *out = m;
    return YOLO2_SUCCESS;
}

extern "C" void yolo2_hip_multi_destroy(yolo2_hip_multi *m)
{
    if (!m) return;
    for (size_t i = 0; i < m->comms.size(); ++i)
        if (m->comms[i] && rccl()) { (void)hipSetDevice(m->devices[i]); (void)rccl()->CommDestroy(m->comms[i]); }
    for (size_t i = 0; i < m->streams.size(); ++i) { (void)hipSetDevice(m->devices[i]); (void)hipStreamDestroy(m->streams[i]); }
    for (yolo2_hip_ctx *c : m->ctx) yolo2_hip_destroy(c);
    delete m;
}

extern "C" int yolo2_hip_multi_num_devices(yolo2_hip_multi *m) { return m ? (int)m->ctx.size() : 0; }
extern "C" yolo2_hip_ctx *yolo2_hip_multi_ctx(yolo2_hip_multi *m, int i) { return m && i >= 0 && i < (int)m->ctx.size() ? m->ctx[(size_t)i] : nullptr; }
extern "C" int yolo2_hip_multi_uses_rccl(yolo2_hip_multi *m) { return m && !m->comms.empty() ? 1 : 0; }

extern "C" int yolo2_hip_multi_rccl_info(yolo2_hip_multi *m, yolo2_hip_rccl_info_t *out)
{
    if (!m || !out) return mfail(YOLO2_ERROR, "null argument");
    if (m->comms.empty()) return mfail(YOLO2_ERROR, "this device list has no RCCL communicator (one device, or a device listed twice)");
    const Rccl *R = rccl();
    if (!R) return mfail(YOLO2_INIT_ERROR, "%s", g_rccl_err.c_str());
    memset(out, 0, sizeof(*out));
    RCCL_TRY(R->CommCount(m->comms[0], &out->nranks));
    RCCL_TRY(R->CommUserRank(m->comms[0], &out->rank));
    RCCL_TRY(R->CommCuDevice(m->comms[0], &out->device));
    RCCL_TRY(R->GetVersion(&out->version));
    out->bcasts = m->bcasts;
    out->last_bcast_ms = m->last_bcast_ms;
    out->last_bcast_bytes = (uint64_t)m->last_bcast_bytes;
    snprintf(out->lib_path, sizeof(out->lib_path), "%s", R->path.c_str());
    return YOLO2_SUCCESS;
}

// Blobs go host -> device 0, from there to every other device (ncclBroadcast; device-to-device copies when the list
// has no communicator), then every context loads its own device copy.
template <typename T, typename LoadFn>
static int multi_load(yolo2_hip_multi *m, const T *weights, size_t n_weights, const T *bias, size_t n_bias, LoadFn load)
{
    if (!m || !weights || !bias) return mfail(YOLO2_ERROR, "null argument");
    if (n_weights < YOLO2_N_WEIGHTS || n_bias < YOLO2_N_BIAS) return mfail(YOLO2_ERROR, "weight blobs too small");
    const size_t wbytes = (size_t)YOLO2_N_WEIGHTS * sizeof(T), bbytes = (size_t)YOLO2_N_BIAS * sizeof(T);
    const int n = (int)m->ctx.size();
    std::vector<Member> mem((size_t)n);
    int rc = YOLO2_SUCCESS;
    for (int i = 0; i < n && rc == YOLO2_SUCCESS; ++i) {
        mem[(size_t)i].device = m->devices[(size_t)i];
        mem[(size_t)i].stream = m->streams[(size_t)i];
        mem[(size_t)i].comm = m->comms.empty() ? nullptr : m->comms[(size_t)i];
        void *wd = nullptr, *bd = nullptr;
        if (hipSetDevice(m->devices[(size_t)i]) != hipSuccess || hipMalloc(&wd, wbytes) != hipSuccess || hipMalloc(&bd, bbytes) != hipSuccess)
            rc = mfail(YOLO2_MMAP_ERROR, "device buffers for the weight blobs could not be allocated on device %d", m->devices[(size_t)i]);
        mem[(size_t)i].bufs = {wd, bd};
    }
    if (rc == YOLO2_SUCCESS) {
        (void)hipSetDevice(m->devices[0]);
        if (hipMemcpy(mem[0].bufs[0], weights, wbytes, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(mem[0].bufs[1], bias, bbytes, hipMemcpyHostToDevice) != hipSuccess)
            rc = mfail(YOLO2_DMA_ERROR, "H2D of the weight blobs failed");
    }
    if (rc == YOLO2_SUCCESS && n > 1) {
        if (!m->comms.empty()) {
            rc = bcast_blobs(mem, {wbytes, bbytes}, 0, &m->last_bcast_ms);
            m->last_bcast_bytes = wbytes + bbytes;
            m->bcasts++;
        }
        else
            for (int i = 1; i < n && rc == YOLO2_SUCCESS; ++i)
                for (int k = 0; k < 2; ++k)
                    if (hipMemcpyPeer(mem[(size_t)i].bufs[(size_t)k], mem[(size_t)i].device, mem[0].bufs[(size_t)k], mem[0].device, k ? bbytes : wbytes) != hipSuccess)
                        rc = mfail(YOLO2_DMA_ERROR, "device-to-device copy of the weight blobs failed");
    }
    for (int i = 0; i < n && rc == YOLO2_SUCCESS; ++i) rc = load(m->ctx[(size_t)i], mem[(size_t)i].bufs[0], mem[(size_t)i].bufs[1], m->devices[(size_t)i]);
    for (int i = 0; i < n; ++i) {
        (void)hipSetDevice(m->devices[(size_t)i]);
        (void)hipDeviceSynchronize();
        for (void *p : mem[(size_t)i].bufs) (void)hipFree(p);
    }
    return rc;
}

extern "C" int yolo2_hip_multi_load_weights_int16(yolo2_hip_multi *m, const int16_t *weights_reorg, size_t n_weights, const int16_t *bias,
                                                  size_t n_bias, const int32_t *weight_q, int n_weight_q, const int32_t *bias_q,
                                                  int n_bias_q, const int32_t *act_q, int n_act_q)
{
    if (!weight_q || !bias_q || !act_q) return mfail(YOLO2_ERROR, "null Q table");
    return multi_load<int16_t>(m, weights_reorg, n_weights, bias, n_bias, [&](yolo2_hip_ctx *c, void *wd, void *bd, int) {
        return yolo2_hip_load_weights_int16_dev(c, (uint64_t)(uintptr_t)wd, YOLO2_N_WEIGHTS, (uint64_t)(uintptr_t)bd, YOLO2_N_BIAS, weight_q,
                                                n_weight_q, bias_q, n_bias_q, act_q, n_act_q);
    });
}

extern "C" int yolo2_hip_multi_load_weights_fp32(yolo2_hip_multi *m, const float *weights_reorg, size_t n_weights, const float *bias,
                                                 size_t n_bias)
{
    return multi_load<float>(m, weights_reorg, n_weights, bias, n_bias, [&](yolo2_hip_ctx *c, void *wd, void *bd, int) {
        return yolo2_hip_load_weights_fp32_dev(c, (uint64_t)(uintptr_t)wd, YOLO2_N_WEIGHTS, (uint64_t)(uintptr_t)bd, YOLO2_N_BIAS);
    });
}

// Frames shard by contiguous ranges (yolo2_hip_shard_range); every device streams its range through
// yolo2_hip_run_frames_int16 / _run_images_u8_host on its own host thread; there is no data-path collective.
template <typename RunFn>
static int multi_run(yolo2_hip_multi *m, int n_frames, RunFn run)
{
    const int n = (int)m->ctx.size();
    std::vector<int> rcs((size_t)n, YOLO2_SUCCESS);
    std::vector<std::string> errs((size_t)n);
    std::vector<std::thread> th;
    for (int i = 0; i < n; ++i) {
        int lo = 0, hi = 0;
        (void)yolo2_hip_shard_range(n_frames, i, n, &lo, &hi);
        if (hi == lo) continue;
        th.emplace_back([&, i, lo, hi] {
            rcs[(size_t)i] = run(m->ctx[(size_t)i], lo, hi);
            if (rcs[(size_t)i]) errs[(size_t)i] = yolo2_hip_last_error();   // (the message is thread-local: carry it over)
        });
    }
    for (auto &t : th) t.join();
    for (int i = 0; i < n; ++i)
        if (rcs[(size_t)i]) return mfail(rcs[(size_t)i], "device %d: %s", m->devices[(size_t)i], errs[(size_t)i].c_str());
    return YOLO2_SUCCESS;
}

extern "C" int yolo2_hip_multi_run_frames_int16(yolo2_hip_multi *m, const float *frames, int n_frames, int batch_per_device,
                                                int16_t *region, int *final_q)
{
    if (!m || !frames || !region) return mfail(YOLO2_ERROR, "null argument");
    if (n_frames <= 0 || batch_per_device <= 0) return mfail(YOLO2_ERROR, "bad frame count / batch");
    std::vector<int> q(m->ctx.size(), 0);
    const int rc = multi_run(m, n_frames, [&](yolo2_hip_ctx *c, int lo, int hi) {
        int qq = 0;
        const int r = yolo2_hip_run_frames_int16(c, frames + (size_t)lo * YOLO2_FRAME_ELEMS, hi - lo, std::min(batch_per_device, hi - lo),
                                                 region + (size_t)lo * YOLO2_REGION_ELEMS, &qq);
        for (size_t i = 0; i < m->ctx.size(); ++i)
            if (m->ctx[i] == c) q[i] = qq;
        return r;
    });
    if (rc == YOLO2_SUCCESS && final_q) *final_q = q[0];
    return rc;
}

extern "C" int yolo2_hip_multi_run_images_u8_host(yolo2_hip_multi *m, const uint8_t *const *images, const int *widths, const int *heights,
                                                  int channels, int n, int batch_per_device, int16_t *region, int *final_q)
{
    if (!m || !images || !widths || !heights || !region) return mfail(YOLO2_ERROR, "null argument");
    if (n <= 0 || batch_per_device <= 0) return mfail(YOLO2_ERROR, "bad image count / batch");
    std::vector<int> q(m->ctx.size(), 0);
    const int rc = multi_run(m, n, [&](yolo2_hip_ctx *c, int lo, int hi) {
        int qq = 0;
        const int r = yolo2_hip_run_images_u8_host(c, images + lo, widths + lo, heights + lo, channels, hi - lo, std::min(batch_per_device, hi - lo),
                                                   region + (size_t)lo * YOLO2_REGION_ELEMS, &qq);
        for (size_t i = 0; i < m->ctx.size(); ++i)
            if (m->ctx[i] == c) q[i] = qq;
        return r;
    });
    if (rc == YOLO2_SUCCESS && final_q) *final_q = q[0];
    return rc;
}

extern "C" int yolo2_hip_multi_run_images_u8_dets(yolo2_hip_multi *m, const uint8_t *const *images, const int *widths, const int *heights,
                                                  int channels, int n, int batch_per_device, float thresh, float nms, int flags,
                                                  yolo2_hip_det *dets, int cap_per_frame, int *counts, int *final_q)
{
    if (!m || !images || !widths || !heights || !dets || !counts) return mfail(YOLO2_ERROR, "null argument");
    if (n <= 0 || batch_per_device <= 0 || cap_per_frame <= 0) return mfail(YOLO2_ERROR, "bad image count / batch / capacity");
    std::vector<int> q(m->ctx.size(), 0);
    const int rc = multi_run(m, n, [&](yolo2_hip_ctx *c, int lo, int hi) {
        int qq = 0;
        // shard [lo, hi): its own device runs the network AND the tail; records are renumbered to global frame indices below
        const int r = yolo2_hip_run_images_u8_dets(c, images + lo, widths + lo, heights + lo, channels, hi - lo, std::min(batch_per_device, hi - lo),
                                                   thresh, nms, flags, dets + (size_t)lo * cap_per_frame, cap_per_frame, counts + lo, &qq);
        if (r == YOLO2_SUCCESS)
            for (int f = lo; f < hi; ++f)
                for (int k = 0, cnt = std::min(counts[f], cap_per_frame); k < cnt; ++k) dets[(size_t)f * cap_per_frame + k].frame = f;
        for (size_t i = 0; i < m->ctx.size(); ++i)
            if (m->ctx[i] == c) q[i] = qq;
        return r;
    });
    if (rc == YOLO2_SUCCESS && final_q) *final_q = q[0];
    return rc;
}

