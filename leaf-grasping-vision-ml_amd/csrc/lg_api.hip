// C-ABI of liblgrasp.so (see include/leafgrasp.h).  Host orchestration only: workspaces, streams,
// the side-stream work beside the sweeps (orientation, border maxima, stem bits, bit-row export), per-kernel event timing.  No exceptions cross the boundary.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <functional>
#include <string>
#include <atomic>
#include <memory>
#include <sched.h>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "lg_cnn.h"
#include "lg_leaf.h"
#include "lg_internal.h"
#include "lg_orient.h"
#include "lg_pool.h"

#define LG_VERSION_STR "leafgrasp-gfx950 0.4"


struct LgProfSlot {
    std::string name;
    std::vector<hipEvent_t> ev;  // pairs
    size_t used = 0;
    int launches = 0;
    double total_ms = 0.0;
};

struct lg_ctx {
    int device = 0;
    std::string err;
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_prep = nullptr, ev_copy = nullptr;
    // sub-batch pipeline of lg_select_grasp: latency-bound stages run beside the bandwidth / MFMA bound ones
    hipStream_t s_dt[2] = {nullptr, nullptr}, s_main = nullptr, s_topk = nullptr;
    std::vector<hipEvent_t> ev_pool;  // untimed events, 6 per sub-batch
    hipEvent_t ev_begin = nullptr;
    // workspace, sized for (capB, capH, capW)
    int capB = 0, capH = 0, capW = 0, capK = 0;
    uint32_t* tmp = nullptr;
    unsigned long long *bits = nullptr, *stem = nullptr, *tilekeys = nullptr;
    uint32_t* maxfix = nullptr;
    LgDtBatch* dt_batch = nullptr;   // search-or-sweeps sums of a batch (lg_bbox_kernel)
    uint8_t* mask_ws = nullptr;      // lg_select_grasp_labels: the 0 / 1 mask it derives from the labels (grown on demand)
    size_t mask_ws_cap = 0;
    int32_t* ids_dev = nullptr;      //   and the frames' leaf ids (device / pinned host staging)
    int32_t* ids_host = nullptr;
    int ids_cap = 0;
    LgWin* win = nullptr;           // [B] sweep windows (lg_bbox_kernel)
    LgFrameParams* fp_dev = nullptr;
    LgFrameParams* fp_host = nullptr;        // pinned
    unsigned long long* bits_host = nullptr;  // pinned
    unsigned long long* bits_host_dev = nullptr;  // device-side alias of bits_host (zero-copy export)
    LgWin* win_host = nullptr;                 // pinned copy of the sweep windows / bounding boxes
    float* ws_maps[LG_NUM_MAPS] = {nullptr};
    float* ws_maps_base[LG_NUM_MAPS] = {nullptr};  // allocation bases (ws_maps[i] = base + i * skew)
    uint8_t* ws_valid = nullptr;
    int32_t *cand_xy = nullptr, *cand_n = nullptr;
    float *cand_info = nullptr, *patches = nullptr, *logits = nullptr;
    lg_grasp_result* res_dev = nullptr;       // [B] result rows written by lg_finish_kernel
    lg_grasp_result* res_host = nullptr;      // pinned copy
    LgCnn cnn;
    LgLeafWs* leaf = nullptr;
    LgLeafProf* leaf_prof = nullptr;   // per-kernel times of the leaf stage (lg_profile_enable)
    LgOrientWs* orient = nullptr;   // device-side orientation scratch (lg_orient.hip)
    hipEvent_t ev_orient = nullptr, ev_side = nullptr, ev_search = nullptr;
    int opt_dt_algo = 0;            // LG_DT_SEARCH_ALGO=1: one-level row search at every batch size; 2: anchors + bands (3 / 4: with four / one
                                    // anchor rows per lane); 0: by batch size
    int opt_dt_search = 2;          // LG_DT_SEARCH=0: d_in by the two sweeps for every frame; 1: by the row search wherever it applies;
                                    // 2 (default): per frame, the search while the batch's estimated search work stays below the sweeps' latency
    int prof_on = 0;  // 0 off, 1 every kernel (event pairs on the stream), 2 only launches that stamp their own events
    std::vector<LgProfSlot> prof;
    int host_threads = 8;
    LgPool* pool = nullptr;
    // experiment switches, read ONCE at lg_create (never on the per-call path)
    int opt_subbatch = 0;        // LG_SUBBATCH=n: sub-batch multi-stream pipeline inside lg_select_grasp (0 = off)
    bool opt_trace = false;      // LG_TRACE: per-call timeline on stderr
    int opt_no_skip = 0;         // LG_NO_SKIP=1: lg_final_kernel without the constant-tile fast path (dense-path roofline);
                                 //            =3: also without the wave-level off-leaf shortcut
    bool opt_nt_stores = false;  // LG_NT_STORES: non-temporal plane stores (measured slower)
    int opt_side_tail = 1;         // LG_SIDE_TAIL=0: frame-border maxima + stem bits after the sweeps on the caller's stream (round 1); 1: on the
                                   // side stream behind the orientation kernel; 2: on a third stream
    bool export_pending = false;   // this call's bit-row export has not been queued yet (enq_export)
    std::string orient_note;       // why the device-side orientation scratch could not be set up (host analysis is used then)
    std::atomic<bool> busy{false}; // one call in flight per handle (SURVEY 8b "Threading"): a concurrent second call gets LG_ERR_BUSY
    hipStream_t s_cnn = nullptr;   // LG_CNN_CUS=n: the CNN runs on a stream of its own restricted to n CUs (experiment: room for a second
    hipEvent_t ev_cnn0 = nullptr, ev_cnn1 = nullptr;   // batch's memory-bound kernels beside it, bench.py --inflight)
    bool opt_host_orient = false;  // LG_HOST_ORIENT: contour analysis of every frame on the host threads (the round-1 path)
};

namespace {

int fail(lg_handle h, int code, const char* what, hipError_t e = hipSuccess) {
    if (h) {
        char buf[512];
        if (e != hipSuccess)
            snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
        else
            snprintf(buf, sizeof(buf), "%s", what);
        h->err = buf;
    }
    return code;
}

#define LG_HIP(h, call)                                                    \
    do {                                                                   \
        hipError_t e_ = (call);                                            \
        if (e_ != hipSuccess) return fail(h, LG_ERR_HIP, #call, e_);       \
    } while (0)

// One call in flight per handle: the workspace, the profiling slots and the error string belong to the call that holds the
// flag.  rospy runs every subscriber callback on its own thread and the reference node guards itself with a plain bool
// (leaf_grasp_node_v3.py:104-107); here a concurrent second call on the SAME handle returns LG_ERR_BUSY without touching
// anything (handles are independent: different handles run side by side).
struct BusyGuard {
    lg_ctx* h;
    bool ok;
    explicit BusyGuard(lg_ctx* h_) : h(h_), ok(!h_->busy.exchange(true, std::memory_order_acquire)) {}
    ~BusyGuard() { if (ok) h->busy.store(false, std::memory_order_release); }
};
#define LG_ENTER(h)        \
    BusyGuard busy_guard_(h); \
    if (!busy_guard_.ok) return LG_ERR_BUSY

struct ProfScope {  // records an event pair around a launch when profiling is on
    lg_ctx* h;
    hipStream_t s;
    LgProfSlot* slot = nullptr;
    hipEvent_t e1 = nullptr;
    hipEvent_t e0 = nullptr;
    bool ext = false;  // true: the launch itself stamps e0/e1 (hipExtLaunchKernelGGL), nothing is recorded here
    ProfScope(lg_ctx* h_, const char* name, hipStream_t s_, bool ext_ = false) : h(h_), s(s_), ext(ext_) {
        if (!h->prof_on || (h->prof_on == 2 && !ext)) return;
        for (auto& p : h->prof)
            if (p.name == name) slot = &p;
        if (!slot) {
            h->prof.push_back(LgProfSlot());
            slot = &h->prof.back();
            slot->name = name;
        }
        if (slot->used + 2 > slot->ev.size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { slot = nullptr; return; }
            slot->ev.push_back(a);
            slot->ev.push_back(b);
        }
        e0 = slot->ev[slot->used];
        if (!ext) hipEventRecord(e0, s);
        e1 = slot->ev[slot->used + 1];
        slot->used += 2;
    }
    ~ProfScope() {
        if (slot && e1 && !ext) hipEventRecord(e1, s);
    }
};

void prof_flush(lg_ctx* h) {  // resolve recorded pairs into totals (call after a synchronise)
    for (auto& p : h->prof) {
        for (size_t i = 0; i + 1 < p.used; i += 2) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, p.ev[i], p.ev[i + 1]) == hipSuccess) {
                p.total_ms += ms;
                p.launches++;
            }
        }
        p.used = 0;
    }
}

template <typename T>
hipError_t dev_alloc(T** p, size_t n) {
    return hipMalloc((void**)p, n * sizeof(T));
}

void free_ws(lg_ctx* h) {
    auto F = [](void* p) { if (p) hipFree(p); };
    F(h->tmp); F(h->bits); F(h->stem); F(h->tilekeys); F(h->maxfix); F(h->dt_batch); F(h->win); F(h->fp_dev);
    for (int i = 0; i < LG_NUM_MAPS; i++) { F(h->ws_maps_base[i]); h->ws_maps_base[i] = h->ws_maps[i] = nullptr; }
    F(h->ws_valid); F(h->cand_xy); F(h->cand_n); F(h->cand_info); F(h->patches); F(h->logits);
    auto HF = [](void* p) { if (p) hipHostFree(p); };
    HF(h->fp_host); HF(h->bits_host); HF(h->win_host); HF(h->res_host);
    F(h->res_dev);
    h->tmp = nullptr; h->bits = h->stem = h->tilekeys = nullptr; h->maxfix = nullptr; h->dt_batch = nullptr; h->win = nullptr; h->fp_dev = nullptr;
    h->ws_valid = nullptr; h->cand_xy = h->cand_n = nullptr; h->cand_info = h->patches = h->logits = nullptr;
    h->fp_host = nullptr; h->bits_host = nullptr; h->win_host = nullptr; h->bits_host_dev = nullptr; h->res_dev = h->res_host = nullptr;
    h->capB = h->capH = h->capW = h->capK = 0;
}

int ensure_ws(lg_ctx* h, int B, int H, int W, int K) {
    if (B <= h->capB && H == h->capH && W == h->capW && K <= h->capK) return LG_OK;
    int nB = std::max(B, h->capB), nK = std::max(K, std::max(h->capK, 20));
    hipDeviceSynchronize();
    free_ws(h);
    const size_t px = (size_t)nB * H * W;
    const int WW = (W + 63) / 64;
    const size_t words = (size_t)nB * H * WW;
    const int tiles = ((W + LG_TW - 1) / LG_TW) * ((H + LG_TH - 1) / LG_TH);
    LG_HIP(h, dev_alloc(&h->tmp, px * 2));
    LG_HIP(h, dev_alloc(&h->bits, words));
    LG_HIP(h, dev_alloc(&h->stem, words));
    LG_HIP(h, dev_alloc(&h->tilekeys, (size_t)nB * tiles));
    LG_HIP(h, dev_alloc(&h->maxfix, (size_t)nB * 2));
    LG_HIP(h, dev_alloc(&h->dt_batch, (size_t)1));
    LG_HIP(h, hipMemset(h->dt_batch, 0, sizeof(LgDtBatch)));
    LG_HIP(h, dev_alloc(&h->win, (size_t)nB));
    LG_HIP(h, dev_alloc(&h->fp_dev, (size_t)nB));
    LG_HIP(h, hipHostMalloc((void**)&h->fp_host, sizeof(LgFrameParams) * nB));
    LG_HIP(h, hipHostMalloc((void**)&h->win_host, sizeof(LgWin) * nB));
    LG_HIP(h, hipHostMalloc((void**)&h->bits_host, sizeof(unsigned long long) * words));
    // device-side alias of the pinned image: lg_export_rows_kernel posts only the bounding-box bits into it.
    // LG_EXPORT_MEMCPY=1 (A/B) or an unmapped allocation: whole-batch hipMemcpyAsync instead.
    if (getenv("LG_EXPORT_MEMCPY") || hipHostGetDevicePointer((void**)&h->bits_host_dev, h->bits_host, 0) != hipSuccess)
        h->bits_host_dev = nullptr;   // fall back to a full-batch hipMemcpyAsync
    LG_HIP(h, dev_alloc(&h->ws_valid, px));   // validity plane for callers that do not want it back
    LG_HIP(h, dev_alloc(&h->cand_xy, (size_t)nB * nK * 2));
    LG_HIP(h, dev_alloc(&h->cand_n, (size_t)nB));
    LG_HIP(h, dev_alloc(&h->cand_info, (size_t)nB * nK * 2));
    // candidate patches as haloed planes (the CNN's staging layout); the halo is zeroed here, once
    LG_HIP(h, dev_alloc(&h->patches, (size_t)nB * nK * lg_cnn_halo_patch_floats()));
    LG_HIP(h, hipMemset(h->patches, 0, (size_t)nB * nK * lg_cnn_halo_patch_floats() * sizeof(float)));
    LG_HIP(h, dev_alloc(&h->logits, (size_t)nB * nK));
    LG_HIP(h, dev_alloc(&h->res_dev, (size_t)nB));
    LG_HIP(h, hipHostMalloc((void**)&h->res_host, sizeof(lg_grasp_result) * nB));
    if (!h->opt_host_orient && H <= 16384 && W <= 8192) {
        // no device-side scratch (allocation, or the LDS request on a part with less of it): the host contour analysis of
        // every frame is a complete path of its own -- scoring goes on, the reason stays readable in orient_note
        std::string err;
        // (LG_ORIENT_FAIL, read here once per workspace: makes the set-up fail -- the test of this hand-over)
        if (getenv("LG_ORIENT_FAIL") || lg_orient_ensure(h->orient, nB, H, &err)) {
            h->orient_note = err.empty() ? "orientation scratch: LG_ORIENT_FAIL is set" : err;
            lg_orient_free(h->orient);
            (void)hipGetLastError();   // the failed allocation's error is sticky until read: it must not fail the call that goes on without it
        }
    } else {
        lg_orient_free(h->orient);   // (a scratch sized for another image height must not outlive it: host analysis from here on)
    }
    h->capB = nB; h->capH = H; h->capW = W; h->capK = nK;
    return LG_OK;
}

int ensure_ws_map(lg_ctx* h, int i) {  // internal plane when the caller does not want map i
    if (h->ws_maps[i]) return LG_OK;
    // LG_PLANE_SKEW=<floats>: plane i starts i * skew floats into its allocation, so that the eight planes' equal pixel
    // offsets do not land on the same HBM channel (experiment; 0 = off)
    static const size_t skew = getenv("LG_PLANE_SKEW") ? (size_t)atoi(getenv("LG_PLANE_SKEW")) / 4 * 4 : 0;
    LG_HIP(h, dev_alloc(&h->ws_maps_base[i], (size_t)h->capB * h->capH * h->capW + LG_NUM_MAPS * skew));
    h->ws_maps[i] = h->ws_maps_base[i] + (size_t)i * skew;
    return LG_OK;
}

// ImageProcessor._create_gaussian_kernel (image_processor.py:25-32): the 2-D kernel exp(-(dx^2 + dy^2) / 2 sigma^2) / sum with
// sigma = size / 6 is the outer product of this 1-D factor with itself
void gaussian1d(int size, float* k1) {
    const double sigma = size / 6.0;
    const int c = size / 2;
    double e[LG_MAX_GAUSS], s = 0;
    for (int i = 0; i < size; i++) { e[i] = exp(-((i - c) * (i - c)) / (2 * sigma * sigma)); s += e[i]; }
    for (int i = 0; i < size; i++) k1[i] = (float)(e[i] / s);
}

void parallel_for(lg_ctx* h, int n, const std::function<void(int)>& fn) {
    if (h->pool && n > 1) h->pool->run(n, fn);
    else
        for (int i = 0; i < n; i++) fn(i);
}

}  // namespace

extern "C" {

const char* lg_version(void) { return LG_VERSION_STR; }

void lg_default_params(lg_params* p) {
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->cx = 707.0; p->cy = 494.0; p->f = 0.0;  // grasp_point_selector.py:29-31 (f_norm unset)
    p->w_approach = 0.4f; p->w_sdf = 0.3f; p->w_flat = 0.2f; p->w_access = 0.1f;
    p->sdf_w_interior = 0.4f; p->sdf_w_align = 0.4f; p->sdf_w_sdf = 0.2f;
    p->optimal_distance = 20.f;
    p->access_w_dist = 0.7f; p->access_w_dir = 0.3f;
    p->flat_scale = 5.f;
    p->iso_w_close = 0.7f; p->iso_w_wide = 0.3f;
    p->iso_ramp_top = 1.0f; p->iso_ramp_bottom = 0.2f;
    p->min_edge_distance = 20.f; p->stem_valid_thresh = 0.8f;
    p->stem_se = 30; p->stem_bottom_div = 3;
    p->top_k = 20; p->nms_min_distance = 10;
    p->pregrasp_clearance = 15;
    p->mask_is_bool = 1;
    p->gaussian_size = 5;   // the node's ImageProcessor(..., gaussian_kernel_size=5) (leaf_grasp_node_v3.py:37,66-67)
    p->chamfer_init_dist0 = (int32_t)LG_INIT0;   // INT_MAX >> 2
}

static thread_local std::string g_create_err = "null handle";

int lg_create(int device, lg_handle* out) {
    if (!out) return LG_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    {
        const hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess || device < 0 || device >= n) {
            g_create_err = std::string("lg_create: hipGetDeviceCount -> ") + hipGetErrorString(e) + ", " + std::to_string(n) +
                           " device(s), asked for " + std::to_string(device);
            return LG_ERR_HIP;
        }
    }
    lg_ctx* h = new (std::nothrow) lg_ctx();
    if (!h) return LG_ERR_NOMEM;
    h->device = device;
    // The side stream is created with the highest priority: ROCm multiplexes normal-priority streams onto a few
    // in-order hardware queues, and when the side stream lands on the caller's queue the 0.65 ms bit-row export
    // sits between pack_bits and the sweeps (measured: sweeps finish at 2.4 ms instead of 1.8 ms at B=128).
    // Priority streams get hardware queues of their own.
    int prio_least = 0, prio_greatest = 0;
    if (hipSetDevice(device) == hipSuccess) hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    if (hipSetDevice(device) != hipSuccess ||
        hipStreamCreateWithPriority(&h->copy_stream, hipStreamNonBlocking, prio_greatest) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_prep, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_copy, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_orient, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_side, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_search, hipEventDisableTiming) != hipSuccess) {
        g_create_err = std::string("lg_create: stream / event creation -> ") + hipGetErrorString(hipGetLastError());
        delete h;
        return LG_ERR_HIP;
    }
    if (hipStreamCreateWithFlags(&h->s_dt[0], hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&h->s_dt[1], hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&h->s_main, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&h->s_topk, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_begin, hipEventDisableTiming) != hipSuccess) {
        g_create_err = std::string("lg_create: stream / event creation -> ") + hipGetErrorString(hipGetLastError());
        delete h;
        return LG_ERR_HIP;
    }
    // host workers: this process's share of the CPUs it may run on (one process per GPU: LOCAL_WORLD_SIZE ranks split the
    // node), at most 16
    unsigned hw = std::thread::hardware_concurrency();
    {
        cpu_set_t set;
        CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0) hw = (unsigned)CPU_COUNT(&set);
    }
    unsigned ranks = 1;
    if (const char* e = getenv("LOCAL_WORLD_SIZE")) ranks = (unsigned)std::max(1, atoi(e));
    h->host_threads = (int)std::max(1u, std::min((hw ? hw : 1u) / ranks, 16u));
    if (const char* e = getenv("LG_HOST_THREADS")) h->host_threads = std::max(1, atoi(e));
    if (const char* e = getenv("LG_SUBBATCH")) h->opt_subbatch = std::max(1, atoi(e));
    h->opt_trace = getenv("LG_TRACE") != nullptr;
    if (const char* e = getenv("LG_NO_SKIP")) h->opt_no_skip = std::max(1, atoi(e)) & 3;   // (both bits leave the results unchanged)
    h->opt_nt_stores = getenv("LG_NT_STORES") != nullptr;
    h->opt_host_orient = getenv("LG_HOST_ORIENT") != nullptr;
    if (const char* e = getenv("LG_DT_SEARCH")) h->opt_dt_search = std::max(0, std::min(2, atoi(e)));
    if (const char* e = getenv("LG_DT_SEARCH_ALGO")) h->opt_dt_algo = std::max(0, std::min(4, atoi(e)));
    if (const char* e = getenv("LG_SIDE_TAIL")) h->opt_side_tail = std::max(0, std::min(2, atoi(e)));
    if (const char* e = getenv("LG_CNN_CUS")) {
        const int n = atoi(e);
        int total = 0;
        hipDeviceGetAttribute(&total, hipDeviceAttributeMultiprocessorCount, device);
        if (n >= 8 && n < total) {
            // the CU mask's bits are dealt round-robin to the XCDs: the first n bits = n / 8 CUs on each of the 8 XCDs
            uint32_t mask[16] = {0};
            for (int i = 0; i < n / 8 * 8; i++) mask[i >> 5] |= 1u << (i & 31);
            if (hipExtStreamCreateWithCUMask(&h->s_cnn, (total + 31) / 32, mask) == hipSuccess &&
                hipEventCreateWithFlags(&h->ev_cnn0, hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&h->ev_cnn1, hipEventDisableTiming) == hipSuccess)
                h->cnn.max_cus = n / 8 * 8;
            else
                h->s_cnn = nullptr;
        }
    }
    h->pool = new (std::nothrow) LgPool(h->host_threads - 1);  // the calling thread is the last worker
    h->leaf_prof = lg_leaf_prof_new();
    *out = h;
    return LG_OK;
}

int lg_destroy(lg_handle h) {
    if (!h) return LG_ERR_INVALID;
    if (h->busy.exchange(true, std::memory_order_acquire)) return LG_ERR_BUSY;   // a call is running on another thread
    hipSetDevice(h->device);
    hipDeviceSynchronize();
    delete h->pool;
    h->pool = nullptr;
    free_ws(h);
    if (h->mask_ws) hipFree(h->mask_ws);
    if (h->ids_dev) hipFree(h->ids_dev);
    if (h->ids_host) hipHostFree(h->ids_host);
    lg_cnn_free(&h->cnn);
    lg_leaf_free(h->leaf);
    lg_leaf_prof_free(h->leaf_prof);
    lg_orient_free(h->orient);
    if (h->ev_orient) hipEventDestroy(h->ev_orient);
    if (h->ev_side) hipEventDestroy(h->ev_side);
    if (h->ev_search) hipEventDestroy(h->ev_search);
    for (auto& p : h->prof)
        for (auto e : p.ev) hipEventDestroy(e);
    for (auto e : h->ev_pool) hipEventDestroy(e);
    if (h->ev_begin) hipEventDestroy(h->ev_begin);
    for (hipStream_t q : {h->s_dt[0], h->s_dt[1], h->s_main, h->s_topk, h->s_cnn})
        if (q) hipStreamDestroy(q);
    if (h->ev_cnn0) hipEventDestroy(h->ev_cnn0);
    if (h->ev_cnn1) hipEventDestroy(h->ev_cnn1);
    if (h->ev_prep) hipEventDestroy(h->ev_prep);
    if (h->ev_copy) hipEventDestroy(h->ev_copy);
    if (h->copy_stream) hipStreamDestroy(h->copy_stream);
    delete h;
    return LG_OK;
}

const char* lg_last_error(lg_handle h) { return h ? h->err.c_str() : g_create_err.c_str(); }   // NULL: why lg_create failed

const char* lg_orientation_note(lg_handle h) { return h ? h->orient_note.c_str() : ""; }

int lg_profile_enable(lg_handle h, int on) {
    if (!h) return LG_ERR_INVALID;
    LG_ENTER(h);
    for (auto& p : h->prof) { p.used = 0; p.launches = 0; p.total_ms = 0.0; }
    h->prof_on = on < 0 ? 0 : on;
    lg_leaf_prof_enable(h->leaf_prof, on == 1);
    return LG_OK;
}

int lg_profile_read(lg_handle h, const char* name, int* launches, double* total_ms) {
    if (!h || !name) return LG_ERR_INVALID;
    LG_ENTER(h);
    hipSetDevice(h->device);
    hipDeviceSynchronize();
    prof_flush(h);
    if (launches) *launches = 0;
    if (total_ms) *total_ms = 0.0;
    if (lg_leaf_prof_read(h->leaf_prof, name, launches, total_ms)) return LG_OK;
    for (auto& p : h->prof)
        if (p.name == name) {
            if (launches) *launches = p.launches;
            if (total_ms) *total_ms = p.total_ms;
            return LG_OK;
        }
    return LG_OK;
}

int lg_debug_dt_max(lg_handle h, int frame, uint32_t out[2], int32_t win[4]) {
    if (!h || !out || frame < 0 || frame >= h->capB || !h->maxfix) return LG_ERR_INVALID;
    LG_ENTER(h);
    hipSetDevice(h->device);
    hipDeviceSynchronize();
    if (hipMemcpy(out, h->maxfix + 2 * (size_t)frame, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess)
        return fail(h, LG_ERR_HIP, "lg_debug_dt_max: copy failed");
    if (win) {
        LgWin w;
        if (hipMemcpy(&w, h->win + frame, sizeof(w), hipMemcpyDeviceToHost) != hipSuccess)
            return fail(h, LG_ERR_HIP, "lg_debug_dt_max: copy failed");
        win[0] = w.wx0; win[1] = w.wx0 + w.nw * lg_dt_geometry(h->capW, nullptr); win[2] = w.wy0; win[3] = w.wy1;
        if (win[1] > h->capW) win[1] = h->capW;
    }
    return LG_OK;
}

int lg_debug_dt_form(lg_handle h, int frame, int32_t form[2]) {
    if (!h || !form || frame < 0 || frame >= h->capB || !h->win) return LG_ERR_INVALID;
    LG_ENTER(h);
    hipSetDevice(h->device);
    hipDeviceSynchronize();
    LgWin w;
    if (hipMemcpy(&w, h->win + frame, sizeof(w), hipMemcpyDeviceToHost) != hipSuccess)
        return fail(h, LG_ERR_HIP, "lg_debug_dt_form: copy failed");
    form[0] = w.search_in ? 1 : 0;
    form[1] = w.skip_out ? 1 : 0;
    return LG_OK;
}

}  // extern "C"

namespace {

struct Plan {  // one call's geometry, parameters and plane pointers (absolute, frame 0)
    int B, H, W, WW, tiles_x, tiles_y;
    lg_params P;
    const float* depth;
    const uint8_t* mask;
    const int16_t* labels = nullptr;   // lg_select_grasp_labels: mask (the handle's workspace) is written from these by the bit-row pass
    float* maps[LG_NUM_MAPS];
    uint8_t* valid;
};

// which form of the row search a batch of n frames takes (LG_DT_SEARCH_ALGO forces one): one level up to 64 frames of 1080p (one
// launch, the device is not full: 32 of 1080p 0.14 vs 0.18 ms), anchors + bands above (fewer evaluations: 64 of 4K 1.08 vs 1.54 ms)
int dt_algo(const lg_ctx* h, int n, int H, int W) {
    if (h->opt_dt_algo) return h->opt_dt_algo;
    return (long long)n * H * W <= 64ll * 1080 * 1920 ? 1 : 2;
}

// frame-border maxima of d_out + stem bits: both read only the bit rows, neither is needed before the plane kernel
int enq_tail(lg_ctx* h, const Plan& pl, int off, int n, hipStream_t s) {
    const size_t words = (size_t)pl.H * pl.WW;
    {
        ProfScope ps(h, "dt_border", s);
        lg_launch_dout_border(h->bits + off * words, h->win + off, h->maxfix + 2 * (size_t)off, n, pl.H, pl.W, pl.WW, s);
    }
    // (on the side stream both run before the bit-row export, which slows every concurrent memory-bound kernel 4x)
    {
        LgSeSpans se;
        lg_make_se_spans(pl.P.stem_se, &se);
        ProfScope ps(h, "stem", s);
        lg_launch_stem_bits(h->bits + off * words, h->stem + off * words, n, pl.H, pl.W, pl.WW,
                            pl.H - pl.H / pl.P.stem_bottom_div, se, s);
    }
    return LG_OK;
}

// pack bits + D2H of the bit rows on the copy stream
int enq_prep(lg_ctx* h, const Plan& pl, int off, int n, hipStream_t s, hipEvent_t ev_prep) {
    h->export_pending = true;
    const size_t px = (size_t)pl.H * pl.W, words = (size_t)pl.H * pl.WW;
    LG_HIP(h, hipMemsetAsync(h->maxfix + 2 * (size_t)off, 0, sizeof(uint32_t) * 2 * n, s));
    {
        ProfScope ps(h, "prep", s);
        if (pl.labels)
            lg_launch_pack_labels(pl.labels + off * px, h->ids_dev + off, h->mask_ws + off * px, h->bits + off * words, n, pl.H, pl.W,
                                  pl.WW, s);
        else
            lg_launch_pack_bits(pl.mask + off * px, h->bits + off * words, n, pl.H, pl.W, pl.WW, s);
    }
    {
        ProfScope ps(h, "bbox", s);
        lg_launch_bbox(h->bits + off * words, h->win + off, n, pl.H, pl.W, pl.WW, h->opt_dt_search, h->dt_batch, s);
    }
    LG_HIP(h, hipEventRecord(ev_prep, s));
    LG_HIP(h, hipStreamWaitEvent(h->copy_stream, ev_prep, 0));
    if (h->orient) {   // contour analysis on the device, beside the sweeps; the host reads theta / status back
        {
            ProfScope ps(h, "orient", h->copy_stream);
            lg_launch_orient(h->orient, h->bits + off * words, h->win + off, h->fp_dev + off, off, n, pl.H, pl.W, pl.WW, h->copy_stream);
        }
        LG_HIP(h, hipMemcpyAsync(h->fp_host + off, h->fp_dev + off, sizeof(LgFrameParams) * n, hipMemcpyDeviceToHost, h->copy_stream));
        LG_HIP(h, hipMemcpyAsync(h->orient->h_status + off, h->orient->status + off, sizeof(int) * n, hipMemcpyDeviceToHost,
                                 h->copy_stream));
        LG_HIP(h, hipEventRecord(h->ev_orient, h->copy_stream));
    }
    if (h->opt_side_tail) {   // beside the sweeps (latency bound, two workgroups per CU), not behind them: 0.3 ms per 256 frames;
        // LG_SIDE_TAIL=2 puts them on a third stream whatever the batch, so that orientation -> border -> stem is not one chain as
        // long as the sweeps themselves -- measured slower at 256 frames (9.76-9.94 vs 9.56-9.77 ms per step, three alternating runs)
        // Small batches are a chain of latencies, not of throughput: orientation (0.13 ms for one frame), border maxima (0.08) and
        // stem bits (0.01) one behind the other on the side stream were the longest chain between the bit rows and the plane
        // kernel of a single-frame call (0.22 ms; the row search beside them takes 0.08).  Up to 32 frames the border maxima and
        // the stem bits go to a stream of their own (s_dt[0]; the search uses s_dt[1]).
        const bool own_tail = h->opt_subbatch == 0 && (h->opt_side_tail == 2 || n <= 32);
        hipStream_t ts = own_tail ? h->s_dt[0] : h->copy_stream;
        if (ts != h->copy_stream) LG_HIP(h, hipStreamWaitEvent(ts, ev_prep, 0));
        const int rc = enq_tail(h, pl, off, n, ts);
        if (rc) return rc;
        LG_HIP(h, hipEventRecord(h->ev_side, ts));
    }
    return LG_OK;
}

// Bit rows of the bounding boxes + the windows to the host: only for the host contour analysis (LG_HOST_ORIENT, or frames the
// device orientation kernel hands back) -- since round 3 the pre-grasp probes run on the device (lg_finish_kernel) and a normal
// call exports nothing.  The export writes pinned host memory over PCIe and slows every memory-bound kernel beside it about 4x,
// which is why it is only queued when somebody is about to wait for it.  `after`: an event to wait for first, or nullptr.
int enq_export(lg_ctx* h, const Plan& pl, int off, int n, hipEvent_t after) {
    const size_t words = (size_t)pl.H * pl.WW;
    if (after) LG_HIP(h, hipStreamWaitEvent(h->copy_stream, after, 0));
    if (h->bits_host_dev)   // rows of the bounding boxes only, posted writes by a small grid on the priority stream
        lg_launch_export_rows(h->bits + off * words, h->win + off, h->bits_host_dev + off * words, n, pl.H, pl.WW, h->copy_stream);
    else
        LG_HIP(h, hipMemcpyAsync(h->bits_host + off * words, h->bits + off * words, sizeof(unsigned long long) * n * words,
                                 hipMemcpyDeviceToHost, h->copy_stream));
    LG_HIP(h, hipMemcpyAsync(h->win_host + off, h->win + off, sizeof(LgWin) * n, hipMemcpyDeviceToHost, h->copy_stream));
    LG_HIP(h, hipEventRecord(h->ev_copy, h->copy_stream));
    h->export_pending = false;
    return LG_OK;
}

// forward + backward distance sweeps (+ frame-border maxima and stem bits when they do not run on the side stream)
int enq_dt(lg_ctx* h, const Plan& pl, int off, int n, hipStream_t s) {
    const size_t px = (size_t)pl.H * pl.W, words = (size_t)pl.H * pl.WW;
    // d_in by the row search for the frames lg_bbox_kernel picked (LgWin::search_in), on a stream of its own beside the sweeps of
    // the other frames (and the d_out sweeps of the few frames that need them): throughput-bound work on every CU next to
    // latency-bound work on one workgroup per frame.  Inside the sub-batch pipeline (whose stages own the side streams): in line.
    hipStream_t ss = (s == h->s_dt[0] || s == h->s_dt[1]) ? s : h->s_dt[1];
    if (h->opt_dt_search) {
        if (ss != s) LG_HIP(h, hipStreamWaitEvent(ss, h->ev_prep, 0));
        {
            ProfScope ps(h, "dt_hrun", ss);
            lg_launch_hrun(h->bits + off * words, h->tmp + 2 * off * px, h->win + off, n, pl.H, pl.W, pl.WW, ss);
        }
        const int algo = dt_algo(h, n, pl.H, pl.W);
        auto launch = [&](int phase) {
            return lg_launch_dtsearch(phase, algo, h->bits + off * words, h->tmp + 2 * off * px, pl.maps[LG_MAP_DISTANCE] + off * px,
                                      h->maxfix + 2 * (size_t)off, h->win + off, n, pl.H, pl.W, pl.WW, ss);
        };
        {
            ProfScope ps(h, "dt_search", ss);   // the one-level search, or the anchor rows
            launch(0);
        }
        if (algo != 1) {
            ProfScope ps(h, "dt_band", ss);     // the rows between the anchors
            for (int phase = 1; launch(phase) > 0; phase++) {}
        }
        if (ss != s) LG_HIP(h, hipEventRecord(h->ev_search, ss));
    }
    {
        ProfScope ps(h, "dt_fwd", s);
        if (lg_launch_dt(false, pl.mask + off * px, h->tmp + 2 * off * px, nullptr, h->maxfix + 2 * (size_t)off, h->win + off, n,
                         pl.H, pl.W, (uint32_t)pl.P.chamfer_init_dist0, s))
            return fail(h, LG_ERR_UNSUPPORTED, "dt: width");
    }
    {
        ProfScope ps(h, "dt_bwd", s);
        lg_launch_dt(true, pl.mask + off * px, h->tmp + 2 * off * px, pl.maps[LG_MAP_DISTANCE] + off * px,
                     h->maxfix + 2 * (size_t)off, h->win + off, n, pl.H, pl.W, (uint32_t)pl.P.chamfer_init_dist0, s);
    }
    if (h->opt_dt_search && ss != s) LG_HIP(h, hipStreamWaitEvent(s, h->ev_search, 0));
    if (!h->opt_side_tail) return enq_tail(h, pl, off, n, s);
    return LG_OK;
}

// host contour analysis of one frame (its bit rows must have landed: ev_copy synchronised)
void host_orient_frame(lg_ctx* h, const Plan& pl, int b) {
    const int H = pl.H, W = pl.W, WW = pl.WW;
    double o[5];
    // only the rows / words of the bounding box are on the host: analyse them as a band (the rest is all zero)
    const LgWin& w = h->win_host[b];
    const int hy = w.by1 - w.by0 + 1;
    int ok = hy > 0 ? lg_host_orientation_band(h->bits_host + ((size_t)b * H + w.by0) * WW, hy, W, WW, w.by0, w.bx0 >> 6, w.bx1 >> 6, o) : 0;
    LgFrameParams f;
    f.has_angle = ok;
    f.theta = ok ? (float)o[0] : NAN;
    f.sin_t = ok ? (float)sin(o[0]) : 0.f;
    f.cos_t = ok ? (float)cos(o[0]) : 0.f;
    h->fp_host[b] = f;
}

// Orientation of frames [off, off+n) into fp_host (and fp_dev).  Device path: wait for lg_orient_kernel's results (it runs
// beside the sweeps on the side stream); frames it handed back (status 1: more runs than its scratch holds) are analysed on
// the host threads.  LG_HOST_ORIENT: every frame on the host.  *upload = fp_host has entries the device does not have yet.
int finish_orient(lg_ctx* h, const Plan& pl, int off, int n, bool* upload) {
    *upload = true;
    if (!h->orient) {
        if (h->export_pending) { const int rc = enq_export(h, pl, 0, pl.B, nullptr); if (rc) return rc; }
        LG_HIP(h, hipEventSynchronize(h->ev_copy));   // bit rows are on the host; the sweeps are running
        parallel_for(h, n, [=, &pl](int i) { host_orient_frame(h, pl, off + i); });
        return LG_OK;
    }
    LG_HIP(h, hipEventSynchronize(h->ev_orient));
    std::vector<int> redo;
    for (int i = 0; i < n; i++)
        if (h->orient->h_status[off + i]) redo.push_back(off + i);
    *upload = !redo.empty();
    if (redo.empty()) return LG_OK;
    if (h->export_pending) { const int rc = enq_export(h, pl, 0, pl.B, nullptr); if (rc) return rc; }
    LG_HIP(h, hipEventSynchronize(h->ev_copy));
    const int* rp = redo.data();
    parallel_for(h, (int)redo.size(), [=, &pl](int i) { host_orient_frame(h, pl, rp[i]); });
    return LG_OK;
}

// frame scalars H2D + the fused score-plane kernel
int enq_final(lg_ctx* h, const Plan& pl, int off, int n, hipStream_t s, bool upload_fp) {
    const size_t px = (size_t)pl.H * pl.W, words = (size_t)pl.H * pl.WW;
    const int H = pl.H, W = pl.W;
    const lg_params& P = pl.P;
    if (h->orient) LG_HIP(h, hipStreamWaitEvent(s, h->ev_orient, 0));   // fp_dev was written on the side stream
    if (h->opt_side_tail) LG_HIP(h, hipStreamWaitEvent(s, h->ev_side, 0));  // maxfix (d_out border) and the stem bits
    if (upload_fp) LG_HIP(h, hipMemcpyAsync(h->fp_dev + off, h->fp_host + off, sizeof(LgFrameParams) * n, hipMemcpyHostToDevice, s));
    LgFinalArgs a;
    memset(&a, 0, sizeof(a));
    a.depth = pl.depth + off * px; a.bits = h->bits + off * words; a.stem_bits = h->stem + off * words;
    a.maxfix = h->maxfix + 2 * (size_t)off; a.fp = h->fp_dev + off;
    a.win = h->win + off; a.win_wc = lg_dt_geometry(W, nullptr);
    for (int i = 0; i < LG_NUM_MAPS; i++) a.maps[i] = pl.maps[i] ? pl.maps[i] + off * px : nullptr;
    a.valid = pl.valid ? pl.valid + off * px : nullptr;
    a.tilekeys = h->tilekeys + (size_t)off * pl.tiles_x * pl.tiles_y;
    a.B = n; a.H = H; a.W = W; a.WW = pl.WW; a.tiles_x = pl.tiles_x; a.tiles_y = pl.tiles_y;
    a.cxi = (int)floor(P.cx); a.cyi = (int)floor(P.cy);
    a.cxf = (float)(P.cx - floor(P.cx)); a.cyf = (float)(P.cy - floor(P.cy)); a.f = (float)P.f;
    a.w_approach = P.w_approach; a.w_sdf = P.w_sdf; a.w_flat = P.w_flat; a.w_access = P.w_access;
    a.sdf_w_interior = P.sdf_w_interior; a.sdf_w_align = P.sdf_w_align; a.sdf_w_sdf = P.sdf_w_sdf;
    a.optimal_distance = P.optimal_distance;
    a.access_w_dist = P.access_w_dist; a.access_w_dir = P.access_w_dir; a.flat_scale = P.flat_scale;
    a.iso_w_close = P.iso_w_close; a.iso_w_wide = P.iso_w_wide;
    a.iso_ramp_top = P.iso_ramp_top; a.iso_ramp_bottom = P.iso_ramp_bottom;
    {
        // max of the chamfer-3 transform of an all-ones image: INIT + ceil(min(H,W)/2) * 0.955
        uint32_t dmax = (uint32_t)((std::min(H, W) + 1) / 2);
        float mx = (float)((uint32_t)P.chamfer_init_dist0 + dmax * LG_A3) * (1.0f / 65536.0f);
        a.init0 = (uint32_t)P.chamfer_init_dist0;
        a.iso_inv_max = 1.0f / mx;  // reference divides by (max + 1e-6), which rounds to max in float32
    }
    a.min_edge_distance = P.min_edge_distance; a.stem_valid_thresh = P.stem_valid_thresh;
    a.inv_maxd = (float)(1.0 / sqrt((double)W * W + (double)H * H));
    a.inv_2s2 = 1.0f / (2.0f * P.optimal_distance * P.optimal_distance);   // (correctly rounded, like the device's __frcp_rn)
    a.iso_ramp_step = (H > 1) ? (P.iso_ramp_bottom - P.iso_ramp_top) / (float)(H - 1) : 0.0f;
    gaussian1d(P.gaussian_size, a.k1);
    a.gauss_r = P.gaussian_size / 2;
    a.no_skip = h->opt_no_skip;
    a.persist = 0;   // (lg_launch_final: the tile walk stays an experiment switch)
    a.nt_stores = h->opt_nt_stores ? 1 : 0;  // measured: non-temporal plane stores are slower here (0.57 vs 0.50 ms)
    {
        ProfScope ps(h, "final", s, true);
        lg_launch_final(a, s, ps.slot ? ps.e0 : nullptr, ps.slot ? ps.e1 : nullptr);
    }
    return LG_OK;
}

int make_plan(lg_ctx* h, Plan& pl, const float* depth, const uint8_t* mask, int B, int H, int W, const lg_params* pin,
              float* const out_maps[LG_NUM_MAPS], uint8_t* out_valid, const char* who) {
    if (!depth || !mask || B <= 0 || H < 8 || W < 8 || W > 8192 || H > 16384)
        return fail(h, LG_ERR_INVALID, "bad pointer or shape (need H,W >= 8, W <= 8192)");
    if (pin) pl.P = *pin; else lg_default_params(&pl.P);
    if (pl.P.stem_se < 1 || pl.P.stem_se > 64 || pl.P.stem_bottom_div < 1)
        return fail(h, LG_ERR_INVALID, "stem_se must be in [1,64], stem_bottom_div >= 1");
    if (pl.P.nms_min_distance < 0 || pl.P.pregrasp_clearance < 0 || pl.P.pregrasp_clearance > 31)
        return fail(h, LG_ERR_INVALID, "nms_min_distance must be >= 0, pregrasp_clearance in [0,31]");
    // _calculate_flatness_map smooths with the caller's ImageProcessor (grasp_point_selector.py:635-657): an even size gives
    // an (H+1) x (W+1) plane there and the fusion raises -> None triple; odd sizes above 7 exceed this kernel's halo
    if (pl.P.gaussian_size < 1 || pl.P.gaussian_size > 7 || (pl.P.gaussian_size & 1) == 0)
        return fail(h, LG_ERR_UNSUPPORTED, "gaussian_size must be 1, 3, 5 or 7 (an even size fails in the reference too: shape mismatch in the fusion)");
    if (pl.P.gaussian_size / 2 >= std::min(H, W))   // torch's reflect padding refuses it: the reference ends in its None triple
        return fail(h, LG_ERR_INVALID, "gaussian_size / 2 must be smaller than H and W (reflect padding)");
    if (pl.P.chamfer_init_dist0 < (int32_t)LG_INIT0)   // (a zeroed struct, or a value a real distance could reach)
        return fail(h, LG_ERR_INVALID, "chamfer_init_dist0 must be in [INT_MAX >> 2, INT_MAX] (lg_default_params: INT_MAX >> 2)");
    pl.B = B; pl.H = H; pl.W = W; pl.WW = (W + 63) / 64;
    pl.tiles_x = (W + LG_TW - 1) / LG_TW; pl.tiles_y = (H + LG_TH - 1) / LG_TH;
    if (pl.tiles_x * pl.tiles_y > 8192) return fail(h, LG_ERR_UNSUPPORTED, "image too large for the top-k tile table");
    pl.depth = depth; pl.mask = mask; pl.valid = out_valid;
    for (int i = 0; i < LG_NUM_MAPS; i++) pl.maps[i] = out_maps ? out_maps[i] : nullptr;
    (void)who;
    return LG_OK;
}

}  // namespace

extern "C" {

int lg_score_maps(lg_handle h, const float* depth, const uint8_t* mask, int B, int H, int W, const lg_params* pin,
                  float* const out_maps[LG_NUM_MAPS], uint8_t* out_valid, float* theta_host, void* stream_) {
    if (!h) return LG_ERR_INVALID;
    LG_ENTER(h);
    Plan pl;
    int rc = make_plan(h, pl, depth, mask, B, H, W, pin, out_maps, out_valid, "lg_score_maps");
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream_;
    LG_HIP(h, hipSetDevice(h->device));
    rc = ensure_ws(h, B, H, W, pl.P.top_k);
    if (rc) return rc;
    for (int i : {LG_MAP_DISTANCE, LG_MAP_TRADITIONAL})
        if (!pl.maps[i]) {
            rc = ensure_ws_map(h, i);
            if (rc) return rc;
            pl.maps[i] = h->ws_maps[i];
        }
    if (!pl.valid) pl.valid = h->ws_valid;
    rc = enq_prep(h, pl, 0, B, s, h->ev_prep);
    if (rc) return rc;
    rc = enq_dt(h, pl, 0, B, s);
    if (rc) return rc;
    // ---- orientation: lg_orient_kernel beside the distance-transform sweeps (host threads for frames it hands back)
    bool upload_fp = true;
    rc = finish_orient(h, pl, 0, B, &upload_fp);
    if (rc) return rc;
    if (theta_host)
        for (int b = 0; b < B; b++) theta_host[b] = h->fp_host[b].theta;
    rc = enq_final(h, pl, 0, B, s, upload_fp);
    if (rc) return rc;
    LG_HIP(h, hipGetLastError());
    return LG_OK;
}

int lg_smooth_depth(lg_handle h, const float* depth, int B, int H, int W, int gaussian_size, float* out, void* stream_) {
    // Stateless: only the handle's device is read, nothing of the handle is written -- the Python ImageProcessor objects of a
    // process share one handle per device across threads, so this entry point takes no part in the one-call-in-flight rule and
    // leaves the handle's error string alone (the reason of a failure: lg_last_error(NULL), per thread).
    if (!h) return LG_ERR_INVALID;
    auto bad = [](int code, const char* what) { g_create_err = what; return code; };
    if (!depth || !out || B < 1 || H < 1 || W < 1) return bad(LG_ERR_INVALID, "lg_smooth_depth: bad argument");
    if (gaussian_size < 1 || gaussian_size > LG_MAX_GAUSS) return bad(LG_ERR_UNSUPPORTED, "lg_smooth_depth: gaussian_size must be in [1,15]");
    if (gaussian_size / 2 >= std::min(H, W))   // torch: "Padding size should be less than the corresponding input dimension"
        return bad(LG_ERR_INVALID, "lg_smooth_depth: reflect padding (gaussian_size / 2) must be smaller than H and W");
    if (hipSetDevice(h->device) != hipSuccess) return bad(LG_ERR_HIP, "lg_smooth_depth: hipSetDevice failed");
    LgGaussTaps taps;
    memset(&taps, 0, sizeof(taps));
    gaussian1d(gaussian_size, taps.k);
    lg_launch_smooth(depth, out, B, H, W, gaussian_size, taps, (hipStream_t)stream_);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_create_err = std::string("lg_smooth_depth: ") + hipGetErrorString(e); return LG_ERR_HIP; }
    return LG_OK;
}

int lg_gaussian_taps(int gaussian_size, float* taps) {
    if (!taps || gaussian_size < 1 || gaussian_size > LG_MAX_GAUSS) return LG_ERR_INVALID;
    gaussian1d(gaussian_size, taps);
    return LG_OK;
}

int lg_topk_nms(lg_handle h, const float* trad, const uint8_t* valid, int B, int H, int W, int k, int min_dist,
                int32_t* out_xy, int32_t* out_n, void* stream_) {
    if (!h) return LG_ERR_INVALID;
    LG_ENTER(h);
    if (!trad || !valid || !out_xy || !out_n || B <= 0 || H < 1 || W < 1 || k < 1 || k > 64 || min_dist < 0)
        return fail(h, LG_ERR_INVALID, "lg_topk_nms: bad argument (1 <= k <= 64)");
    hipStream_t s = (hipStream_t)stream_;
    LG_HIP(h, hipSetDevice(h->device));
    const int tiles = ((W + LG_TW - 1) / LG_TW) * ((H + LG_TH - 1) / LG_TH);
    if (tiles > 8192) return fail(h, LG_ERR_UNSUPPORTED, "lg_topk_nms: image too large");
    int rc = ensure_ws(h, B, H, W, k);
    if (rc) return rc;
    ProfScope ps(h, "topk", s);
    lg_launch_topk(trad, valid, nullptr, h->tilekeys, false, B, H, W, k, min_dist, out_xy, out_n, nullptr, s);
    LG_HIP(h, hipGetLastError());
    return LG_OK;
}

int lg_gather_patches(lg_handle h, const float* depth, const uint8_t* mask, const float* const maps[LG_NUM_MAPS], int B,
                      int H, int W, int k, const int32_t* xy, const int32_t* n, float* patches, void* stream_) {
    if (!h) return LG_ERR_INVALID;
    LG_ENTER(h);
    if (!depth || !mask || !maps || !xy || !n || !patches || B <= 0 || k < 1)
        return fail(h, LG_ERR_INVALID, "lg_gather_patches: bad argument");
    for (int i = 0; i < 7; i++)
        if (!maps[i]) return fail(h, LG_ERR_INVALID, "lg_gather_patches: maps[0..6] are required");
    hipStream_t s = (hipStream_t)stream_;
    LG_HIP(h, hipSetDevice(h->device));
    ProfScope ps(h, "gather", s);
    lg_launch_gather(depth, mask, maps, B, H, W, k, xy, n, patches, false, s);
    LG_HIP(h, hipGetLastError());
    return LG_OK;
}

int lg_harvest_patches(lg_handle h, const float* depth, const uint8_t* mask, const float* const maps[LG_NUM_MAPS], int H,
                       int W, int n, const int32_t* xy, const int32_t* rot, float* out_depth, float* out_mask,
                       float* out_scores, int32_t* flags, void* stream_) {
    if (!h) return LG_ERR_INVALID;
    LG_ENTER(h);
    if (!depth || !mask || !maps || !xy || !out_depth || !out_mask || !out_scores || !flags || n < 1 || H < 32 || W < 32)
        return fail(h, LG_ERR_INVALID, "lg_harvest_patches: bad argument");
    for (int i = 0; i < 7; i++)
        if (!maps[i]) return fail(h, LG_ERR_INVALID, "lg_harvest_patches: maps[0..6] are required");
    hipStream_t s = (hipStream_t)stream_;
    LG_HIP(h, hipSetDevice(h->device));
    lg_launch_harvest(depth, mask, maps, H, W, n, xy, rot, out_depth, out_mask, out_scores, flags, s);
    LG_HIP(h, hipGetLastError());
    return LG_OK;
}

int lg_negative_masks(lg_handle h, const float* distance_map, const uint8_t* mask, int H, int W, uint8_t* out_tip,
                      uint8_t* out_stem, void* stream_) {
    if (!h) return LG_ERR_INVALID;
    LG_ENTER(h);
    if (!distance_map || !mask || !out_tip || !out_stem || H < 1 || W < 1)
        return fail(h, LG_ERR_INVALID, "lg_negative_masks: bad argument");
    hipStream_t s = (hipStream_t)stream_;
    LG_HIP(h, hipSetDevice(h->device));
    int rc = ensure_ws(h, 1, H, W, 20);
    if (rc) return rc;
    // scratch for the first erosion: the u8 view of the forward-sweep workspace (H*W bytes needed, 8*H*W available)
    lg_launch_negative_masks(distance_map, mask, out_tip, out_stem, reinterpret_cast<uint8_t*>(h->tmp), H, W, s);
    LG_HIP(h, hipGetLastError());
    return LG_OK;
}

int lg_leaf_contour(lg_handle h, const uint8_t* mask, int H, int W, int32_t* out_xy, int cap, int* n_out, void* stream_) {
    if (!h) return LG_ERR_INVALID;
    LG_ENTER(h);
    if (!mask || !n_out || (cap > 0 && !out_xy) || cap < 0 || H < 1 || W < 1)
        return fail(h, LG_ERR_INVALID, "lg_leaf_contour: bad argument");
    hipStream_t s = (hipStream_t)stream_;
    LG_HIP(h, hipSetDevice(h->device));
    int rc = ensure_ws(h, 1, H, W, 20);
    if (rc) return rc;
    const int WW = (W + 63) / 64;
    lg_launch_pack_bits(mask, h->bits, 1, H, W, WW, s);
    LG_HIP(h, hipMemcpyAsync(h->bits_host, h->bits, sizeof(unsigned long long) * (size_t)H * WW, hipMemcpyDeviceToHost, s));
    LG_HIP(h, hipStreamSynchronize(s));
    std::vector<int> pts;
    const int n = lg_host_contour_points(h->bits_host, H, W, WW, pts);
    *n_out = n;
    for (int i = 0; i < n && i < cap; i++) { out_xy[2 * i] = pts[2 * i]; out_xy[2 * i + 1] = pts[2 * i + 1]; }
    return LG_OK;
}

int lg_leaf_orientation(lg_handle h, const uint8_t* mask, int H, int W, float* out, int* found, void* stream_) {
    if (!h) return LG_ERR_INVALID;
    LG_ENTER(h);
    if (!mask || !out || !found || H < 1 || W < 1) return fail(h, LG_ERR_INVALID, "lg_leaf_orientation: bad argument");
    hipStream_t s = (hipStream_t)stream_;
    LG_HIP(h, hipSetDevice(h->device));
    int rc = ensure_ws(h, 1, H, W, 20);
    if (rc) return rc;
    const int WW = (W + 63) / 64;
    lg_launch_pack_bits(mask, h->bits, 1, H, W, WW, s);
    double o[5];
    if (h->orient) {   // the device analysis; a mask with more runs than its scratch holds falls through to the host code
        lg_launch_bbox(h->bits, h->win, 1, H, W, WW, 0, nullptr, s);
        lg_launch_orient(h->orient, h->bits, h->win, h->fp_dev, 0, 1, H, W, WW, s);
        LG_HIP(h, hipMemcpyAsync(h->orient->h_out, h->orient->out, sizeof(double) * 5, hipMemcpyDeviceToHost, s));
        LG_HIP(h, hipMemcpyAsync(h->orient->h_status, h->orient->status, sizeof(int), hipMemcpyDeviceToHost, s));
        LG_HIP(h, hipStreamSynchronize(s));
        if (!h->orient->h_status[0]) {
            *found = !std::isnan(h->orient->h_out[0]);
            for (int i = 0; i < 5; i++) out[i] = *found ? (float)h->orient->h_out[i] : NAN;
            return LG_OK;
        }
    }
    LG_HIP(h, hipMemcpyAsync(h->bits_host, h->bits, sizeof(unsigned long long) * (size_t)H * WW, hipMemcpyDeviceToHost, s));
    LG_HIP(h, hipStreamSynchronize(s));
    *found = lg_host_orientation(h->bits_host, H, W, WW, o);
    for (int i = 0; i < 5; i++) out[i] = *found ? (float)o[i] : NAN;
    return LG_OK;
}

int lg_leaf_stats(lg_handle h, const int16_t* labels, const float* depth, int H, int W, float cx, float cy, float f,
                  lg_leaf_stat* stats, int max_leaves, int* n_leaves, int32_t* extrema, void* stream_) {
    if (!h) return LG_ERR_INVALID;
    LG_ENTER(h);
    if (!labels || !depth || !stats || !n_leaves || !extrema || H < 1 || W < 1 || max_leaves < 1)
        return fail(h, LG_ERR_INVALID, "lg_leaf_stats: bad argument");
    hipStream_t s = (hipStream_t)stream_;
    LG_HIP(h, hipSetDevice(h->device));
    std::string err;
    ProfScope ps(h, "leaf", s);
    int rc = lg_leaf_run(h->leaf, labels, depth, H, W, cx, cy, f, stats, max_leaves, n_leaves, extrema, s, h->s_dt[0], &err, h->leaf_prof);
    if (rc) return fail(h, rc, err.c_str());
    return LG_OK;
}

int lg_leaf_stats_batch(lg_handle h, const int16_t* labels, const float* depth, int B, int H, int W, float cx, float cy,
                        float f, lg_leaf_stat* stats, int max_leaves, int* n_leaves, int32_t* extrema, int* status,
                        void* stream_) {
    if (!h) return LG_ERR_INVALID;
    LG_ENTER(h);
    if (!labels || !depth || !stats || !n_leaves || !extrema || !status || B < 1 || H < 1 || W < 1 || max_leaves < 1)
        return fail(h, LG_ERR_INVALID, "lg_leaf_stats_batch: bad argument");
    hipStream_t s = (hipStream_t)stream_;
    LG_HIP(h, hipSetDevice(h->device));
    std::string err;
    ProfScope ps(h, "leaf", s);
    int rc = lg_leaf_run_batch(h->leaf, labels, depth, B, H, W, cx, cy, f, stats, max_leaves, n_leaves, extrema, status, s,
                               h->s_dt[0], &err, h->leaf_prof);
    if (rc) return fail(h, rc, err.c_str());
    return LG_OK;
}

int lg_leaf_select_batch(lg_handle h, const int16_t* labels, const float* depth, int B, int H, int W, double cx, double cy,
                         double f, int32_t* ids, int32_t* n_tall, int32_t* tall, int tall_cap, void* stream_) {
    if (!h) return LG_ERR_INVALID;
    LG_ENTER(h);
    if (!labels || !depth || !ids || !n_tall || !tall || B < 1 || H < 1 || W < 1 || tall_cap < 1)
        return fail(h, LG_ERR_INVALID, "lg_leaf_select_batch: bad argument");
    hipStream_t s = (hipStream_t)stream_;
    LG_HIP(h, hipSetDevice(h->device));
    std::string err;
    ProfScope ps(h, "leaf", s);
    int rc;
    try {   // (the host half allocates: nothing may throw across the C boundary)
        rc = lg_leaf_select_batch_run(h->leaf, labels, depth, B, H, W, cx, cy, f, ids, n_tall, tall, tall_cap, s, h->s_dt[0], &err, h->leaf_prof);
    } catch (const std::bad_alloc&) {
        return fail(h, LG_ERR_NOMEM, "lg_leaf_select_batch: out of host memory");
    } catch (...) {
        return fail(h, LG_ERR_INVALID, "lg_leaf_select_batch: unexpected exception");
    }
    if (rc) return fail(h, rc, err.c_str());
    return LG_OK;
}

int lg_leaf_select_from_stats(const lg_leaf_stat* stats, int n, const int32_t* extrema, int H, int W, double cx, double cy, double f,
                              int32_t* id, int32_t* tall, int tall_cap, int32_t* n_tall) {
    if (!stats || !extrema || !id || !tall || !n_tall || n < 0 || H < 1 || W < 1 || tall_cap < 1) return LG_ERR_INVALID;
    int nt = 0;
    try {
        *id = lg_leaf_select_host(stats, n, extrema, H, W, cx, cy, f, tall, tall_cap, &nt);
    } catch (const std::bad_alloc&) {
        return LG_ERR_NOMEM;
    } catch (...) {
        return LG_ERR_INVALID;
    }
    *n_tall = nt;
    return LG_OK;
}

int lg_cnn_load(lg_handle h, const lg_cnn_weights* w) {
    if (!h || !w) return LG_ERR_INVALID;
    LG_ENTER(h);
    LG_HIP(h, hipSetDevice(h->device));
    std::string err;
    int rc = lg_cnn_upload(&h->cnn, w, &err);
    if (rc) return fail(h, rc, err.c_str());
    return LG_OK;
}

int lg_cnn_unload(lg_handle h) {
    if (!h) return LG_ERR_INVALID;
    LG_ENTER(h);
    hipSetDevice(h->device);
    hipDeviceSynchronize();
    lg_cnn_free(&h->cnn);
    return LG_OK;
}

int lg_cnn_forward(lg_handle h, const float* patches, int N, float* logits, void* stream_) {
    if (!h) return LG_ERR_INVALID;
    LG_ENTER(h);
    if (!patches || !logits || N <= 0) return fail(h, LG_ERR_INVALID, "lg_cnn_forward: bad argument");
    if (!h->cnn.loaded) return fail(h, LG_ERR_NO_MODEL, "lg_cnn_forward: no model loaded (reference: ml_predictor is None)");
    hipStream_t s = (hipStream_t)stream_;
    LG_HIP(h, hipSetDevice(h->device));
    std::string err;
    if (lg_cnn_take_error(&h->cnn)) return fail(h, LG_ERR_HIP, "lg_cnn_forward: a split item of the previous forward did not receive its parts");
    ProfScope ps(h, "cnn", s);
    int rc = lg_cnn_run(&h->cnn, patches, false, N, logits, s, &err);
    if (rc) return fail(h, rc, err.c_str());
    LG_HIP(h, hipGetLastError());
    return LG_OK;
}

static int lg_select_grasp_impl(lg_handle h, const float* depth, const uint8_t* mask, const int16_t* labels, int B, int H, int W,
                                const lg_params* pin, float* const out_maps[LG_NUM_MAPS], uint8_t* out_valid,
                                lg_grasp_result* results, void* stream_);

int lg_select_grasp(lg_handle h, const float* depth, const uint8_t* mask, int B, int H, int W, const lg_params* pin,
                    float* const out_maps[LG_NUM_MAPS], uint8_t* out_valid, lg_grasp_result* results, void* stream_) {
    if (!h) return LG_ERR_INVALID;
    LG_ENTER(h);
    return lg_select_grasp_impl(h, depth, mask, nullptr, B, H, W, pin, out_maps, out_valid, results, stream_);
}

// lg_select_grasp on mask[b] = (labels[b] == leaf_ids[b]): the node's `optimal_mask = mask_tensor == optimal_leaf_id` followed by
// select_grasp_point (leaf_grasp_node_v3.py:118-125), the comparison folded into the bit-row pass.
int lg_select_grasp_labels(lg_handle h, const float* depth, const int16_t* labels, const int32_t* leaf_ids, int B, int H, int W,
                           const lg_params* pin, float* const out_maps[LG_NUM_MAPS], uint8_t* out_valid, lg_grasp_result* results,
                           void* stream_) {
    if (!h) return LG_ERR_INVALID;
    LG_ENTER(h);
    if (!labels || !leaf_ids || B < 1 || H < 1 || W < 1) return fail(h, LG_ERR_INVALID, "lg_select_grasp_labels: null or empty input");
    LG_HIP(h, hipSetDevice(h->device));
    const size_t need = (size_t)B * H * W;
    if (need > h->mask_ws_cap) {
        if (h->mask_ws) { hipDeviceSynchronize(); hipFree(h->mask_ws); h->mask_ws = nullptr; h->mask_ws_cap = 0; }
        if (hipMalloc((void**)&h->mask_ws, need) != hipSuccess) { (void)hipGetLastError(); return fail(h, LG_ERR_NOMEM, "lg_select_grasp_labels: mask workspace"); }
        h->mask_ws_cap = need;
    }
    if (B > h->ids_cap) {
        if (h->ids_dev) { hipDeviceSynchronize(); hipFree(h->ids_dev); h->ids_dev = nullptr; }
        if (h->ids_host) { hipHostFree(h->ids_host); h->ids_host = nullptr; }
        h->ids_cap = 0;
        if (hipMalloc((void**)&h->ids_dev, sizeof(int32_t) * B) != hipSuccess ||
            hipHostMalloc((void**)&h->ids_host, sizeof(int32_t) * B) != hipSuccess) {
            (void)hipGetLastError();
            return fail(h, LG_ERR_NOMEM, "lg_select_grasp_labels: id buffers");
        }
        h->ids_cap = B;
    }
    memcpy(h->ids_host, leaf_ids, sizeof(int32_t) * B);   // (the previous call on this handle has been synchronised: one call in flight)
    LG_HIP(h, hipMemcpyAsync(h->ids_dev, h->ids_host, sizeof(int32_t) * B, hipMemcpyHostToDevice, (hipStream_t)stream_));
    return lg_select_grasp_impl(h, depth, h->mask_ws, labels, B, H, W, pin, out_maps, out_valid, results, stream_);
}

static int lg_select_grasp_impl(lg_handle h, const float* depth, const uint8_t* mask, const int16_t* labels, int B, int H, int W,
                                const lg_params* pin, float* const out_maps[LG_NUM_MAPS], uint8_t* out_valid,
                                lg_grasp_result* results, void* stream_) {
    if (!results) return fail(h, LG_ERR_INVALID, "lg_select_grasp: results is null");
    Plan pl;
    int rc = make_plan(h, pl, depth, mask, B, H, W, pin, out_maps, out_valid, "lg_select_grasp");
    if (rc) return rc;
    pl.labels = labels;
    const lg_params& P = pl.P;
    if (P.top_k < 1 || P.top_k > 64) return fail(h, LG_ERR_INVALID, "lg_select_grasp: top_k must be in [1,64]");
    hipStream_t s = (hipStream_t)stream_;
    LG_HIP(h, hipSetDevice(h->device));
    rc = ensure_ws(h, B, H, W, P.top_k);
    if (rc) return rc;
    const bool use_cnn = h->cnn.loaded;
    // all eight planes are needed when the CNN rescoring runs; otherwise only distance + traditional
    for (int i = 0; i < LG_NUM_MAPS; i++)
        if (!pl.maps[i] && (use_cnn || i == LG_MAP_DISTANCE || i == LG_MAP_TRADITIONAL)) {
            rc = ensure_ws_map(h, i);
            if (rc) return rc;
            pl.maps[i] = h->ws_maps[i];
        }
    if (!pl.valid) pl.valid = h->ws_valid;
    const int K = P.top_k;
    const size_t px = (size_t)H * W;
    const int tiles = pl.tiles_x * pl.tiles_y;

    // ---- sub-batch pipeline.  Stages per sub-batch k of SB frames:
    //   D(k): bit rows, stem bits, distance sweeps          -- latency bound, 2*SB workgroups  -> s_dt[k&1]
    //   F(k): fused score planes                             -- HBM bound                       -> s_main
    //   T(k): greedy spaced top-k                            -- latency bound, SB workgroups    -> s_topk
    //   G(k): patch gather + CNN                             -- MFMA bound                      -> s_main
    // s_main runs F(0) F(1) G(0) F(2) G(1) ... so T(k) hides behind F(k+1) and the D chain behind everything.
    int SB = B;
    // Measured on MI355X (B=128, 1080p): SB=32 is 14.8 ms/step vs 11.5 ms unpiped -- the chain
    // D(0)->F(0)->T(0) must drain before the first CNN launch, and D's latency does not shrink with SB.
    // Kept for experiments (LG_SUBBATCH=n in the environment when the handle is created); the default is one sub-batch.
    if (h->opt_subbatch > 0) SB = h->opt_subbatch;
    const int nsub = (B + SB - 1) / SB;
    const bool piped = nsub > 1;
    hipStream_t sD[2] = {piped ? h->s_dt[0] : s, piped ? h->s_dt[1] : s};
    hipStream_t sM = piped ? h->s_main : s, sT = piped ? h->s_topk : s;
    while ((int)h->ev_pool.size() < 6 * nsub) {
        hipEvent_t e;
        LG_HIP(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        h->ev_pool.push_back(e);
    }
    auto EV = [&](int k, int which) { return h->ev_pool[6 * k + which]; };  // 0 prep 1 copy 2 dt 3 final 4 topk 5 done
    if (piped) {  // internal streams start after whatever the caller queued on `s`
        LG_HIP(h, hipEventRecord(h->ev_begin, s));
        for (hipStream_t q : {sD[0], sD[1], sM, sT}) LG_HIP(h, hipStreamWaitEvent(q, h->ev_begin, 0));
    }
    auto enq_G = [&](int k) -> int {
        const int off = k * SB, n = std::min(SB, B - off);
        if (piped) LG_HIP(h, hipStreamWaitEvent(sM, EV(k, 4), 0));
        if (use_cnn) {
            {
                ProfScope ps(h, "gather", sM);
                const float* mp[LG_NUM_MAPS];
                for (int i = 0; i < LG_NUM_MAPS; i++) mp[i] = pl.maps[i] ? pl.maps[i] + off * px : nullptr;
                lg_launch_gather(depth + off * px, mask + off * px, mp, n, H, W, K, h->cand_xy + (size_t)off * K * 2,
                                 h->cand_n + off, h->patches + (size_t)off * K * lg_cnn_halo_patch_floats(), true, sM);
            }
            std::string err;
            hipStream_t sC = sM;
            if (h->s_cnn) {   // experiment: the CNN on its CU-masked stream, ordered after the gather and before the copy-back
                sC = h->s_cnn;
                LG_HIP(h, hipEventRecord(h->ev_cnn0, sM));
                LG_HIP(h, hipStreamWaitEvent(sC, h->ev_cnn0, 0));
            }
            {
                ProfScope ps(h, "cnn", sC);
                int r2 = lg_cnn_run(&h->cnn, h->patches + (size_t)off * K * lg_cnn_halo_patch_floats(), true, n * K, h->logits + (size_t)off * K, sC, &err);
                if (r2) return fail(h, r2, err.c_str());
            }
            if (h->s_cnn) {
                LG_HIP(h, hipEventRecord(h->ev_cnn1, sC));
                LG_HIP(h, hipStreamWaitEvent(sM, h->ev_cnn1, 0));
            }
        }
        return LG_OK;
    };
    const bool trace = h->opt_trace;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_start = now();
    hipEvent_t tev[6] = {nullptr};
    if (trace) {
        for (auto& e : tev) hipEventCreate(&e);
        hipEventRecord(tev[0], s);
    }
    // bit rows of the whole batch first (one short kernel) so the host never waits behind a distance sweep
    rc = enq_prep(h, pl, 0, B, sD[0], h->ev_prep);
    if (rc) return rc;
    if (piped) LG_HIP(h, hipStreamWaitEvent(sD[1], h->ev_prep, 0));
    for (int k = 0; k < nsub; k++) {
        const int off = k * SB, n = std::min(SB, B - off);
        rc = enq_dt(h, pl, off, n, sD[k & 1]);
        if (rc) return rc;
        if (piped) LG_HIP(h, hipEventRecord(EV(k, 2), sD[k & 1]));
    }
    if (trace && !piped) hipEventRecord(tev[1], s);   // after the sweeps
    const double t_enq1 = now();
    const double t_copy = now();
    bool upload_fp = true;
    rc = finish_orient(h, pl, 0, B, &upload_fp);   // theta per frame: device results (or host analysis) while the sweeps run
    if (rc) return rc;
    const double t_orient = now();
    for (int k = 0; k < nsub; k++) {
        const int off = k * SB, n = std::min(SB, B - off);
        if (piped) LG_HIP(h, hipStreamWaitEvent(sM, EV(k, 2), 0));
        rc = enq_final(h, pl, off, n, sM, upload_fp);
        if (rc) return rc;
        if (trace && !piped) hipEventRecord(tev[2], s);   // after the fused planes
        if (piped) {
            LG_HIP(h, hipEventRecord(EV(k, 3), sM));
            LG_HIP(h, hipStreamWaitEvent(sT, EV(k, 3), 0));
        }
        {
            ProfScope ps(h, "topk", sT);
            lg_launch_topk(pl.maps[LG_MAP_TRADITIONAL] + off * px, pl.valid + off * px, depth + off * px,
                           h->tilekeys + (size_t)off * tiles, true, n, H, W, K, P.nms_min_distance,
                           h->cand_xy + (size_t)off * K * 2, h->cand_n + off, h->cand_info + (size_t)off * K * 2, sT);
        }
        if (trace && !piped) hipEventRecord(tev[3], s);   // after top-k
        if (piped) LG_HIP(h, hipEventRecord(EV(k, 4), sT));
        if (k >= 1) { rc = enq_G(k - 1); if (rc) return rc; }
    }
    rc = enq_G(nsub - 1);
    if (rc) return rc;
    if (trace && !piped) hipEventRecord(tev[4], s);       // after gather + CNN
    if (piped) {  // join: the caller's stream continues after every internal stream
        for (hipStream_t q : {sD[0], sD[1], sM, sT}) {
            LG_HIP(h, hipEventRecord(EV(0, 5), q));
            LG_HIP(h, hipStreamWaitEvent(s, EV(0, 5), 0));
            LG_HIP(h, hipStreamSynchronize(q));
        }
    }
    // ---- the rest of select_grasp_point per frame -- rescoring, 3-D point, pre-grasp probes -- on the device (lg_finish_kernel);
    //      one copy of B result rows comes back
    {
        LgFinishArgs fa;
        memset(&fa, 0, sizeof(fa));
        fa.cand_n = h->cand_n; fa.cand_xy = h->cand_xy; fa.cand_info = h->cand_info; fa.logits = h->logits;
        fa.fp = h->fp_dev; fa.bits = h->bits; fa.out = h->res_dev;
        fa.B = B; fa.H = H; fa.W = W; fa.WW = pl.WW; fa.K = K; fa.use_cnn = use_cnn ? 1 : 0; fa.mask_is_bool = P.mask_is_bool;
        fa.cx = P.cx; fa.cy = P.cy; fa.f = P.f;
        lg_make_se_spans(2 * P.pregrasp_clearance + 1, &fa.se);
        ProfScope ps(h, "finish", s);
        lg_launch_finish(fa, s);
    }
    LG_HIP(h, hipMemcpyAsync(h->res_host, h->res_dev, sizeof(lg_grasp_result) * B, hipMemcpyDeviceToHost, s));
    const double t_enq2 = now();
    LG_HIP(h, hipStreamSynchronize(s));
    const double t_sync = now();
    LG_HIP(h, hipGetLastError());
    if (use_cnn && lg_cnn_take_error(&h->cnn)) return fail(h, LG_ERR_HIP, "lg_select_grasp: a split CNN item did not receive its parts");
    memcpy(results, h->res_host, sizeof(lg_grasp_result) * B);
    if (trace && !piped) {
        float a[5] = {0};
        for (int i = 1; i <= 4; i++) hipEventElapsedTime(&a[i], tev[0], tev[i]);
        fprintf(stderr, "[lg] gpu timeline (ms since call start): sweeps done %.3f | planes %.3f | topk %.3f | cnn %.3f\n", a[1], a[2],
                a[3], a[4]);
    }
    if (trace) for (auto& e : tev) if (e) hipEventDestroy(e);
    if (trace)
        fprintf(stderr, "[lg] B=%d enq1 %.3f | wait bits %.3f | orient %.3f | enq2 %.3f | gpu wait %.3f | post %.3f ms\n", B,
                t_enq1 - t_start, t_copy - t_enq1, t_orient - t_copy, t_enq2 - t_orient, t_sync - t_enq2, now() - t_sync);
    return LG_OK;
}

}  // extern "C"
