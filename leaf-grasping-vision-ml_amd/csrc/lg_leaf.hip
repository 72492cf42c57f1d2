// Leaf-selection statistics on gfx950: the per-leaf full-frame passes of
// OptimalLeafSelector.select_optimal_leaf (scripts/utils/leaf_scorer.py:25-203) as a few streaming passes, B frames per launch:
//   k_presence_bits  labels once (2 B/px): label presence bitmap -> compact slot per id (torch.unique, :32), the bit mask
//                    "pixel belongs to some leaf" (1 bit/px) and the first leaf pixel (the field's arg-min, :70)
//   k_accumulate     labels + depth of the leaf runs: per-slot area, sum x, sum y (int64, exact), sum depth, sum ray length
//                    (f64), border flag; every leaf pixel is appended to an unordered (slot, depth key) list (8 B per leaf pixel)
//   k_hist x3..4     exact median depth (np.median, :41-47) by 8-bit radix select over THAT list (leaves cover 15-30 % of a
//                    frame: each pass reads ~1.5 B/px instead of 6), on keys relative to the leaf's smallest key (three passes for
//                    a leaf whose depths span less than 2^24 float steps); the last pass also yields the successor for even
//                    counts; LDS histograms hold 16 slots, frames with more labels (up to 1024) run further slot groups
//   k_edt_bb         global clutter extrema (:66-71): arg-max of the exact Euclidean distance to the nearest leaf pixel, first
//                    occurrence, by branch and bound on the bit mask -- the distance field is 1-Lipschitz, so a square cell whose
//                    centre value + half diagonal stays below the best value found so far cannot hold the maximum.  32-px cells,
//                    then 16, 8, 4, 2, 1: a few thousand exact point evaluations per frame instead of a full transform (the
//                    column scan + per-row lower-envelope pass it replaces took 4.2 of the stage's 8.3 ms per 128 frames and
//                    stays as the fallback for a frame whose survivor list overflows).
#include "lg_leaf.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <string>
#include <vector>

#define LGL_GROUP 64         // slots per LDS histogram group
#define LGL_MAXL 1024        // distinct labels per frame supported (16 groups)
#define LGL_ACC_LDS 256      // slots whose sums are aggregated in LDS (beyond: global atomics)
#define LGL_RUN 16           // consecutive pixels per thread (run-length aggregation before atomics)
#define LGL_QCAP 65536       // survivor list capacity of the branch-and-bound pass (entries per frame and list)

namespace {

__device__ __forceinline__ uint32_t f2key(float f) {  // order-preserving float -> uint32
    uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) {
    uint32_t b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __uint_as_float(b);
}

// ---------------------------------------------------------------- presence bitmap of ids 1..32767 + leaf bit mask
// The LGL_RUN labels of a run in registers: two 16-byte loads when the row pitch and the base allow it (a wave then reads
// 2 KB of consecutive labels), element loads otherwise.  Returns false for a run without a positive label -- most runs of a
// frame are background, and they end here after one 32-byte read.
__device__ inline bool load_run(const int16_t* __restrict__ lab, int W, int y, int x0, int x1, bool vec, int16_t (&ids)[LGL_RUN]) {
    static_assert(LGL_RUN == 16, "two uint4 loads per run");
    const int16_t* p = lab + (size_t)y * W + x0;
    if (vec && x1 - x0 == LGL_RUN) {
        const uint4 a = reinterpret_cast<const uint4*>(p)[0], b = reinterpret_cast<const uint4*>(p)[1];
        if ((a.x | a.y | a.z | a.w | b.x | b.y | b.z | b.w) == 0u) return false;
        const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int k = 0; k < 8; k++) {
            ids[2 * k] = (int16_t)(w[k] & 0xFFFFu);
            ids[2 * k + 1] = (int16_t)(w[k] >> 16);
        }
    } else {
#pragma unroll
        for (int k = 0; k < LGL_RUN; k++) ids[k] = (x0 + k < x1) ? p[k] : (int16_t)0;
    }
    bool any = false;
#pragma unroll
    for (int k = 0; k < LGL_RUN; k++) any |= ids[k] > 0;
    return any;
}
// the depths of a run that holds leaf pixels: four 16-byte loads (same conditions as load_run's vector path)
__device__ inline void load_run_depth(const float* __restrict__ depth, int W, int y, int x0, int x1, bool vec, float (&dv)[LGL_RUN]) {
    const float* p = depth + (size_t)y * W + x0;
    if (vec && x1 - x0 == LGL_RUN) {
#pragma unroll
        for (int k = 0; k < LGL_RUN / 4; k++) {
            const float4 v = reinterpret_cast<const float4*>(p)[k];
            dv[4 * k] = v.x; dv[4 * k + 1] = v.y; dv[4 * k + 2] = v.z; dv[4 * k + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < LGL_RUN; k++) dv[k] = (x0 + k < x1) ? p[k] : 0.0f;
    }
}
__device__ inline bool run_vec_ok(const int16_t* lab, const float* depth, int W) {
    return (W % LGL_RUN) == 0 && (reinterpret_cast<uintptr_t>(lab) & 15u) == 0 && (reinterpret_cast<uintptr_t>(depth) & 15u) == 0;
}


// one thread per 16-pixel run: presence bits of its labels, its 16 mask bits; four lanes assemble a 64-bit word of the mask
// (rows are padded to whole words: WW = ceil(W / 64), 4 WW runs per row, bits past W are 0)
__global__ __launch_bounds__(256) void k_presence_bits(const int16_t* __restrict__ lab, const float* __restrict__ depth_unused,
                                                       int H, int W, int WW, unsigned long long* __restrict__ pres,
                                                       unsigned long long* __restrict__ bits,
                                                       unsigned long long* __restrict__ first_leaf) {
    __shared__ unsigned long long s_p[512];
    __shared__ unsigned long long s_first;
    {
        const size_t fr = blockIdx.y;
        lab += fr * H * W; pres += fr * 512; bits += fr * H * WW; first_leaf += fr;
    }
    for (int i = threadIdx.x; i < 512; i += 256) s_p[i] = 0;
    if (threadIdx.x == 0) s_first = ~0ull;
    __syncthreads();
    const int rpr = 4 * WW;                               // runs per padded row
    const long long nruns = (long long)H * rpr;           // a multiple of 4: the four lanes of a word stay together
    const bool vec = (W % LGL_RUN) == 0 && (reinterpret_cast<uintptr_t>(lab) & 15u) == 0;
    for (long long r0 = (long long)blockIdx.x * 256; r0 < nruns; r0 += (long long)gridDim.x * 256) {
        const long long r = r0 + threadIdx.x;
        unsigned piece = 0;
        int y = 0, x0 = 0;
        if (r < nruns) {
            y = (int)(r / rpr); x0 = (int)(r % rpr) * LGL_RUN;
            if (x0 < W) {
                const int x1 = min(x0 + LGL_RUN, W);
                int16_t ids[LGL_RUN];
                if (load_run(lab, W, y, x0, x1, vec, ids)) {
                    int last = 0;
#pragma unroll
                    for (int k = 0; k < LGL_RUN; k++) {
                        const int id = ids[k];
                        if (id > 0) {
                            piece |= 1u << k;
                            if (id != last) {
                                const unsigned long long bit = 1ull << (id & 63);
                                if (!(s_p[id >> 6] & bit)) atomicOr(&s_p[id >> 6], bit);
                                last = id;
                            }
                        }
                    }
                }
            }
        }
        // word = pieces of lanes 4q .. 4q+3 (quad exchange)
        const unsigned p1 = __shfl_down(piece, 1, 64), p2 = __shfl_down(piece, 2, 64), p3 = __shfl_down(piece, 3, 64);
        if ((threadIdx.x & 3) == 0 && r < nruns) {
            const unsigned long long word = (unsigned long long)piece | ((unsigned long long)p1 << 16) | ((unsigned long long)p2 << 32) |
                                            ((unsigned long long)p3 << 48);
            bits[(size_t)y * WW + (x0 >> 6)] = word;
            if (word) atomicMin(&s_first, (unsigned long long)y * W + x0 + __builtin_ctzll(word));
        }
    }
    __syncthreads();
    for (int w = threadIdx.x; w < 512; w += 256)
        if (s_p[w]) atomicOr(&pres[w], s_p[w]);
    if (threadIdx.x == 0 && s_first != ~0ull) atomicMin(first_leaf, s_first);
}

// prefix popcounts of the presence words: slot(id) = pre[id>>6] + popc(pres[id>>6] & ((1<<(id&63))-1)); one wave per frame
__global__ __launch_bounds__(64) void k_prefix(const unsigned long long* __restrict__ pres, int* __restrict__ pre, int* __restrict__ nlab) {
    pres += (size_t)blockIdx.x * 512;
    pre += (size_t)blockIdx.x * 512;
    const int lane = threadIdx.x;
    int c[8], mine = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) { c[k] = __popcll(pres[8 * lane + k]); mine += c[k]; }
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    int acc = incl - mine;
#pragma unroll
    for (int k = 0; k < 8; k++) { pre[8 * lane + k] = acc; acc += c[k]; }
    if (lane == 63) nlab[blockIdx.x] = incl;
}

__device__ __forceinline__ int slot_of(int id, const unsigned long long* pres, const int* pre) {
    unsigned long long w = pres[id >> 6];
    return pre[id >> 6] + __popcll(w & ((1ull << (id & 63)) - 1ull));
}

struct LeafAcc {  // device accumulators, one per slot
    unsigned long long area;
    long long sum_x, sum_y;
    double sum_depth, sum_ray;
    int border;
    unsigned kmax, nkmin;   // largest depth key, ~(smallest depth key): both grow by atomicMax from the all-zero state
    int pad;
};


// ---------------------------------------------------------------- per-slot sums + the (slot, key) list of the leaf pixels
__global__ __launch_bounds__(256) void k_accumulate(const int16_t* __restrict__ lab, const float* __restrict__ depth,
                                                    int H, int W, const unsigned long long* __restrict__ pres,
                                                    const int* __restrict__ pre, float cx, float cy, float f,
                                                    LeafAcc* __restrict__ acc, unsigned long long* __restrict__ comp,
                                                    unsigned int* __restrict__ comp_n, size_t comp_stride, unsigned int segcap, int ablate) {
    __shared__ LeafAcc s_acc[LGL_ACC_LDS];
    __shared__ unsigned int s_cnt;                          // entries of this workgroup's list segment so far
    {
        const size_t fr = blockIdx.y;
        lab += fr * H * W; depth += fr * H * W; pres += fr * 512; pre += fr * 512; acc += fr * LGL_MAXL;
        // the list of a frame is one segment per workgroup (capacity = the pixels the workgroup can meet): space is reserved
        // with an LDS atomic per wave and step, the segment's length is written once at the end -- no global atomics
        comp += fr * comp_stride + (size_t)blockIdx.x * segcap; comp_n += fr * gridDim.x + blockIdx.x;
    }
    if (threadIdx.x == 0) s_cnt = 0;
    for (int i = threadIdx.x; i < LGL_ACC_LDS; i += 256) {
        s_acc[i].area = 0; s_acc[i].sum_x = 0; s_acc[i].sum_y = 0;
        s_acc[i].sum_depth = 0.0; s_acc[i].sum_ray = 0.0; s_acc[i].border = 0; s_acc[i].kmax = 0; s_acc[i].nkmin = 0;
    }
    __syncthreads();
    // A wave's 64 runs form a TILE of 64 x 16 pixels (4 runs across, 16 rows), not 1024 pixels of one row: every wave that meets
    // a leaf pixel walks the whole per-pixel loop for all its lanes (61 % of the SIMDs' vector issue cycles went there,
    // profiles/r04_leaf_pmc.txt), and a 1024-pixel strip meets a leaf in most rows of the image while a tile only does where one is.
    const int runs_per_row = (W + LGL_RUN - 1) / LGL_RUN;
    const int tiles_x = (runs_per_row + 3) / 4, tiles_y = (H + 15) / 16;
    const long long nruns = (long long)tiles_x * tiles_y * 64;
    const double f2 = (double)f * (double)f;
    const bool vec = run_vec_ok(lab, depth, W);
    const int lane = threadIdx.x & 63;
    for (long long r0 = (long long)blockIdx.x * 256; r0 < nruns; r0 += (long long)gridDim.x * 256) {
        const long long r = r0 + threadIdx.x;
        int16_t ids[LGL_RUN];
        float dv[LGL_RUN];
        int y = 0, x0 = 0, x1 = 0;
        bool any = false;
        if (r < nruns) {
            const long long tile = r >> 6;
            const int ty = (int)(tile / tiles_x), tx = (int)(tile - (long long)ty * tiles_x);
            const int xi = 4 * tx + (lane & 3);
            y = 16 * ty + (lane >> 2); x0 = xi * LGL_RUN;
            x1 = min(x0 + LGL_RUN, W);
            if (y < H && xi < runs_per_row) any = load_run(lab, W, y, x0, x1, vec, ids);
        }
        if (__ballot(any) == 0ull) continue;                 // wave-uniform: no leaf pixel in this tile
        if (any) {
            load_run_depth(depth, W, y, x0, x1, vec, dv);
        } else {
#pragma unroll
            for (int k = 0; k < LGL_RUN; k++) ids[k] = 0;
        }
        // List positions (the list is unordered: a median only needs the multiset).  Pixel k of every lane is written by ONE store
        // instruction whose active lanes take consecutive entries (ballot + prefix popcount): 512 contiguous bytes per
        // instruction.  (Each lane writing its own 16 entries back to back made every one of the 16 stores touch 64 different
        // cache lines: 0.30 of the kernel's 0.77 ms per 128 frames.)  One LDS atomic per wave reserves the space.
        unsigned long long bal[LGL_RUN];
        unsigned total = 0;
#pragma unroll
        for (int k = 0; k < LGL_RUN; k++) {
            bal[k] = __ballot(ids[k] > 0);
            total += (unsigned)__popcll(bal[k]);
        }
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(&s_cnt, total);
        base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
        const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;   // lanes below this one
        int cur = 0, cur_slot = 0;
        unsigned long long a = 0; long long sx = 0; double sd = 0.0, sr = 0.0; int bd = 0;
        unsigned kmx = 0, nkmn = 0;
        const double dy = (double)y - (double)cy;
        auto flush = [&]() {
            if (cur > 0 && a && !(ablate & 2)) {
                const int s = cur_slot;
                if (s < LGL_ACC_LDS) {
                    if (kmx > s_acc[s].kmax) atomicMax(&s_acc[s].kmax, kmx);      // (plain pre-test: a leaf's range is set by few runs)
                    if (nkmn > s_acc[s].nkmin) atomicMax(&s_acc[s].nkmin, nkmn);
                    atomicAdd(&s_acc[s].area, a);
                    atomicAdd((unsigned long long*)&s_acc[s].sum_x, (unsigned long long)sx);
                    atomicAdd((unsigned long long*)&s_acc[s].sum_y, (unsigned long long)((long long)a * y));
                    atomicAdd(&s_acc[s].sum_depth, sd);
                    atomicAdd(&s_acc[s].sum_ray, sr);
                    if (bd) atomicOr(&s_acc[s].border, 1);
                } else if (s < LGL_MAXL) {
                    atomicMax(&acc[s].kmax, kmx);
                    atomicMax(&acc[s].nkmin, nkmn);
                    atomicAdd(&acc[s].area, a);
                    atomicAdd((unsigned long long*)&acc[s].sum_x, (unsigned long long)sx);
                    atomicAdd((unsigned long long*)&acc[s].sum_y, (unsigned long long)((long long)a * y));
                    atomicAdd(&acc[s].sum_depth, sd);
                    atomicAdd(&acc[s].sum_ray, sr);
                    if (bd) atomicOr(&acc[s].border, 1);
                }
            }
        };
#pragma unroll
        for (int k = 0; k < LGL_RUN; k++) {
            const int x = x0 + k;
            int id = (int)ids[k];
            if (id < 0) id = 0;
            if (id != cur) {
                flush();
                cur = id; a = 0; sx = 0; sd = 0.0; sr = 0.0; bd = 0; kmx = 0; nkmn = 0;
                if (id > 0) cur_slot = slot_of(id, pres, pre);
            }
            if (id > 0) {
                a++;
                sx += x;
                sd += (double)dv[k];
                const double dx = (double)x - (double)cx;
                // f64 square root = f32 square root (1 ulp) + one Newton step in f64 (relative error ~1e-14; the correctly rounded
                // v_sqrt_f64 sequence was the single most expensive thing in this kernel's inner loop)
                const double v = dx * dx + dy * dy + f2;
                const float r0 = __builtin_amdgcn_sqrtf((float)v);
                const double rd = (double)r0;
                if (!(ablate & 4)) sr += r0 > 0.0f ? rd + (v - rd * rd) * (double)(0.5f * __frcp_rn(r0)) : 0.0;   // (v == 0: f == 0 and the pixel is the optical centre)
                bd |= (x == 0) | (x == W - 1) | (y == 0) | (y == H - 1);
                const unsigned key = f2key(dv[k]);
                kmx = max(kmx, key); nkmn = max(nkmn, ~key);
                if (!(ablate & 1)) comp[base + (unsigned)__popcll(bal[k] & lt)] = ((unsigned long long)(unsigned)cur_slot << 32) | key;
            }
            base += (unsigned)__popcll(bal[k]);
        }
        flush();
    }
    __syncthreads();
    for (int i = threadIdx.x; i < LGL_ACC_LDS; i += 256) {
        if (s_acc[i].area) {
            atomicAdd(&acc[i].area, s_acc[i].area);
            atomicAdd((unsigned long long*)&acc[i].sum_x, (unsigned long long)s_acc[i].sum_x);
            atomicAdd((unsigned long long*)&acc[i].sum_y, (unsigned long long)s_acc[i].sum_y);
            atomicAdd(&acc[i].sum_depth, s_acc[i].sum_depth);
            atomicAdd(&acc[i].sum_ray, s_acc[i].sum_ray);
            if (s_acc[i].border) atomicOr(&acc[i].border, 1);
            atomicMax(&acc[i].kmax, s_acc[i].kmax);
            atomicMax(&acc[i].nkmin, s_acc[i].nkmin);
        }
    }
    if (threadIdx.x == 0) *comp_n = s_cnt;
}

// ---------------------------------------------------------------- radix select (median) over the (slot, key) list
// Keys are taken RELATIVE to the leaf's smallest key (k_accumulate delivers min and max): a leaf's depths span a fraction of a
// metre, i.e. ~2^22 float32 steps, so the most significant byte of (key - min) is 0 for every pixel and that pass has nothing
// to decide -- three passes over the list instead of four (four for a leaf whose depths span more than 2^24 steps, fewer for a
// flat one).  A slot takes part in pass p only while p < npass; a frame none of whose slots needs a pass leaves it at once.
struct SelState {       // per slot
    uint32_t prefix;    // bits of (key - base) fixed so far (high bits)
    uint32_t rank;      // remaining 0-based rank inside the current prefix bucket
    uint32_t n_le;      // (after the last pass) number of elements equal to the key ranked ABOVE the selected one
    uint32_t key;       // selected key (after the last pass)
    uint32_t base;      // smallest key of the slot
    uint32_t npass;     // 8-bit digits of (largest key - base) that can differ: 0 .. 4
};

// median ranks from the per-slot areas (lower median index), key range -> passes; grid (LGL_MAXL / 256, B)
__global__ void k_seed(const LeafAcc* __restrict__ acc, SelState* __restrict__ st, int* __restrict__ maxpass) {
    const size_t fr = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= LGL_MAXL) return;
    const LeafAcc& A = acc[fr * LGL_MAXL + i];
    const unsigned long long a = A.area;
    SelState z;
    z.prefix = 0; z.n_le = 0;
    z.rank = a ? (uint32_t)((a - 1) / 2) : 0;
    z.base = a ? ~A.nkmin : 0u;
    const uint32_t range = a ? A.kmax - z.base : 0u;
    z.npass = range ? (uint32_t)(32 - __builtin_clz(range) + 7) / 8u : 0u;
    z.key = z.base;                                     // npass == 0: every element equals the smallest key
    if (z.npass == 0 && a) z.n_le = (uint32_t)(a - 1) - z.rank;
    st[fr * LGL_MAXL + i] = z;
    if (z.npass) atomicMax(&maxpass[fr], (int)z.npass);
}

// histogram of digit `pass` (3 = most significant byte) of (key - base) among the list entries of one slot group whose
// higher bytes match their slot's prefix.  GROUP slots share the LDS histogram (16: 16 KB, eight workgroups per CU -- the pass
// is bound by the list's load latency, not by the LDS atomics: with 64 slots = 64 KB two workgroups fit and a pass took 0.17 ms
// per 128 frames whether the atomics ran or not); frames with more labels run further slot groups.  The last pass (0) also
// finds, per slot, the smallest relative key ABOVE the selected key's 256-key bucket (successor for even counts, when the
// bucket itself holds nothing above the selected key).
template <int GROUP>
__global__ __launch_bounds__(256) void k_hist(const unsigned long long* __restrict__ comp, const unsigned int* __restrict__ comp_n,
                                              size_t comp_stride, unsigned int segcap, const int* __restrict__ nlab,
                                              const SelState* __restrict__ st, const int* __restrict__ maxpass, int pass,
                                              uint32_t* __restrict__ hist, uint32_t* __restrict__ above, int ablate) {
    __shared__ uint32_t s_h[GROUP * 256];
    __shared__ uint32_t s_prefix[GROUP], s_base[GROUP], s_np[GROUP], s_above[GROUP];
    const size_t fr = blockIdx.y;
    if (pass >= maxpass[fr]) return;                          // no slot of this frame needs the pass (whole workgroup)
    comp += fr * comp_stride + (size_t)blockIdx.x * segcap;   // workgroup x reads the segment k_accumulate's workgroup x wrote
    const unsigned n = comp_n[fr * gridDim.x + blockIdx.x];
    const int nl = min(nlab[fr], LGL_MAXL);
    const int shift = 8 * pass;
    const uint32_t himask = (pass == 3) ? 0u : (0xFFFFFFFFu << (shift + 8));
    const int lane = threadIdx.x & 63;
    for (int g = 0; g * GROUP < nl; g++) {
        const int nsl = min(GROUP, nl - g * GROUP);
        const SelState* stg = st + fr * LGL_MAXL + g * GROUP;
        uint32_t* histg = hist + (fr * LGL_MAXL + (size_t)g * GROUP) * 256;
        __syncthreads();
        for (int i = threadIdx.x; i < nsl * 256; i += 256) s_h[i] = 0;
        if (threadIdx.x < nsl) {
            const SelState z = stg[threadIdx.x];
            s_prefix[threadIdx.x] = z.prefix; s_base[threadIdx.x] = z.base; s_np[threadIdx.x] = z.npass;
            s_above[threadIdx.x] = 0xFFFFFFFFu;
        }
        __syncthreads();
        // One entry per lane and load (512 contiguous bytes per wave instruction), four loads per thread in flight.  Neighbouring
        // entries mostly belong to the same slot and, in the upper digits, to the same bin: each maximal group of equal
        // neighbours inside the wave becomes one LDS atomic of its first lane (ballot of the heads).
        constexpr int UNR = 4;
        for (unsigned i0 = 0; i0 < n; i0 += 256 * UNR) {
            unsigned long long e_[UNR];
#pragma unroll
            for (int k = 0; k < UNR; k++) {
                const unsigned i = i0 + k * 256 + threadIdx.x;
                e_[k] = i < n ? __builtin_nontemporal_load(comp + i) : ~0ull;
            }
#pragma unroll
            for (int k = 0; k < UNR; k++) {
                int idx = -1;
                if (i0 + k * 256 + threadIdx.x < n) {
                    const unsigned long long e = e_[k];
                    const int slot = (int)(e >> 32) - g * GROUP;
                    if ((unsigned)slot < (unsigned)nsl && (uint32_t)pass < s_np[slot]) {
                        const uint32_t rel = (uint32_t)e - s_base[slot], pf = s_prefix[slot];
                        if ((rel & himask) == (pf & himask)) idx = slot * 256 + (int)((rel >> shift) & 0xFFu);
                        else if (pass == 0 && rel > pf && rel < s_above[slot]) atomicMin(&s_above[slot], rel);   // (rel > pf: a higher bucket)
                    }
                }
                const int prev = __shfl_up(idx, 1, 64);
                const bool head = lane == 0 || idx != prev;
                const unsigned long long hm = __ballot(head);
                if (head && idx >= 0 && !(ablate & 8)) {
                    const unsigned long long later = lane == 63 ? 0ull : (hm >> (lane + 1));
                    const int len = later ? __builtin_ctzll(later) + 1 : 64 - lane;
                    atomicAdd(&s_h[idx], (uint32_t)len);
                }
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < nsl * 256; i += 256)
            if (s_h[i]) atomicAdd(&histg[i], s_h[i]);
        if (pass == 0 && threadIdx.x < nsl && s_above[threadIdx.x] != 0xFFFFFFFFu)
            atomicMin(&above[fr * LGL_MAXL + g * GROUP + threadIdx.x], s_above[threadIdx.x]);
    }
}

// pick the bin holding the wanted rank, descend; one WAVE per slot (lane = four bins, wave prefix sum); grid (16, B).
// Pass 0 finishes the slot: selected key, how many equal elements rank above it, and the successor key (smallest key above
// the selected one: the next non-empty bin of the same bucket, else what k_hist found above the bucket) into succ.
__global__ __launch_bounds__(256) void k_select(SelState* __restrict__ st, uint32_t* __restrict__ hist, const int* __restrict__ nlab,
                                                const int* __restrict__ maxpass, int pass, const uint32_t* __restrict__ above,
                                                uint32_t* __restrict__ succ) {
    const size_t fr = blockIdx.y;
    if (pass >= maxpass[fr]) return;
    const int nl = min(nlab[fr], LGL_MAXL);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    st += fr * LGL_MAXL; hist += fr * LGL_MAXL * 256; above += fr * LGL_MAXL; succ += fr * LGL_MAXL;
    for (int s = blockIdx.x * 4 + wave; s < nl; s += gridDim.x * 4) {
        SelState z = st[s];
        if ((uint32_t)pass >= z.npass) continue;              // (wave-uniform)
        uint4* h4 = reinterpret_cast<uint4*>(hist + (size_t)s * 256) + lane;
        const uint4 c = *h4;
        *h4 = make_uint4(0u, 0u, 0u, 0u);                     // the histogram workspace is left clean: no memset per call
        const uint32_t mine = c.x + c.y + c.z + c.w;
        uint32_t incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = __shfl_up(incl, o, 64);
            if (lane >= o) incl += v;
        }
        const uint32_t excl = incl - mine, rank = z.rank;
        const bool here = rank >= excl && rank < incl;        // exactly one lane (the counts add up to more than the rank)
        int bin = -1; uint32_t accum = 0, inbin = 0;
        if (here) {
            const uint32_t cc[4] = {c.x, c.y, c.z, c.w};
            uint32_t acc2 = excl;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (bin < 0 && rank < acc2 + cc[q]) { bin = 4 * lane + q; accum = acc2; inbin = cc[q]; }
                acc2 += cc[q];
            }
        }
        const unsigned long long who = __ballot(here);
        const int src = who ? __builtin_ctzll(who) : 0;
        bin = __shfl(bin, src, 64); accum = __shfl(accum, src, 64); inbin = __shfl(inbin, src, 64);
        if (bin < 0) bin = 255;                               // (unreachable: kept from the serial form)
        z.prefix |= ((uint32_t)bin) << (8 * pass);
        z.rank = rank - accum;
        if (pass == 0) {
            z.key = z.base + z.prefix;
            z.n_le = inbin - z.rank - 1;                      // elements equal to the key ranked ABOVE the selected one
            // successor: the first non-empty bin above `bin` in this bucket, else the smallest key above the bucket
            const uint32_t cc[4] = {c.x, c.y, c.z, c.w};
            int nb = 256;
#pragma unroll
            for (int q = 3; q >= 0; q--)
                if (cc[q] && 4 * lane + q > bin) nb = 4 * lane + q;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) nb = min(nb, __shfl_xor(nb, o, 64));
            if (lane == 0) {
                const uint32_t ab = above[s];
                succ[s] = nb < 256 ? z.base + ((z.prefix & ~0xFFu) | (uint32_t)nb) : (ab != 0xFFFFFFFFu ? z.base + ab : 0xFFFFFFFFu);
            }
        }
        if (lane == 0) st[s] = z;
    }
}

// ---------------------------------------------------------------- exact EDT arg-max by branch and bound on the bit mask
// ---- word occupancy of the bit rows: occ[y] bit i = word i of row y holds a leaf pixel (W <= 4096: a row is at most 64 words).
// With it the leaf pixel of a row nearest to column x is found with ONE more round trip -- the own word and the nearest
// non-empty word on either side, all three requested together -- instead of a walk over the row's words outwards from x (up
// to WW dependent loads per row for a point far from every leaf, which is exactly where the arg-max search spends its
// evaluations: 1.16 -> 0.45 ms per 128 frames).  One wave per row.
__global__ __launch_bounds__(256) void k_rowocc(const unsigned long long* __restrict__ bits, int H, int WW,
                                                unsigned long long* __restrict__ occ) {
    const size_t fr = blockIdx.y;
    bits += fr * H * WW; occ += fr * H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int y = blockIdx.x * 4 + wave; y < H; y += gridDim.x * 4) {
        const unsigned long long z = lane < WW ? bits[(size_t)y * WW + lane] : 0ull;
        const unsigned long long nz = __ballot(z != 0ull);
        if (lane == 0) occ[y] = nz;
    }
}

// Exact squared distance from (y, x) to the nearest leaf pixel, given that it is <= ub2 (0xFFFFFFFF when nothing is known).
// One WAVE per point: the 64 lanes take 64 rows at a time (first the 64 rows around y, then 64 more above and 64 more below,
// ...), each lane finds the leaf pixel of its row nearest to x, a wave minimum closes the round.  All lanes return the same
// value.  need2: the caller only cares about values >= need2 (a cell whose bound cannot reach the best value is dropped
// whatever its exact value): the search stops once it is below.
__device__ inline uint32_t edt_point_wave_occ(const unsigned long long* __restrict__ bits, const unsigned long long* __restrict__ occ,
                                              int H, int W, int WW, int y, int x, uint32_t ub2, uint32_t need2, int lane) {
    unsigned long long best = (unsigned long long)ub2 + 1ull;   // search for d2 < best
    const int w0 = x >> 6, b = x & 63;
    auto row_probe = [&](int yy) -> unsigned long long {        // d2 of row yy's nearest leaf pixel, or ~0
        if (yy < 0 || yy >= H) return ~0ull;
        const long long dyl = (long long)yy - y;
        const unsigned long long dy2 = (unsigned long long)(dyl * dyl);
        if (dy2 >= best) return ~0ull;
        const unsigned long long oc = occ[yy];
        if (!oc) return ~0ull;
        const unsigned long long* row = bits + (size_t)yy * WW;
        const unsigned long long lm = w0 ? oc & ((1ull << w0) - 1ull) : 0ull, rm = w0 < 63 ? oc >> (w0 + 1) : 0ull;
        const int kl = lm ? 63 - __builtin_clzll(lm) : w0, kr = rm ? w0 + 1 + __builtin_ctzll(rm) : w0;
        const unsigned long long wc = row[w0], wl = row[kl], wr = row[kr];   // one round trip for the three
        long long dl = -1, dr = -1;
        const unsigned long long ml = wc & (b == 63 ? ~0ull : ((2ull << b) - 1ull));   // bits <= b
        const unsigned long long mr = wc & (~0ull << b);                               // bits >= b
        if (ml) dl = b - (63 - __builtin_clzll(ml));
        else if (lm) dl = x - (kl * 64 + (63 - __builtin_clzll(wl)));
        if (mr) dr = __builtin_ctzll(mr) - b;
        else if (rm) dr = kr * 64 + __builtin_ctzll(wr) - x;
        long long dx = -1;
        if (dl >= 0) dx = dl;
        if (dr >= 0 && (dx < 0 || dr < dx)) dx = dr;
        return dx >= 0 ? (unsigned long long)(dx * dx) + dy2 : ~0ull;
    };
    auto wave_min = [&](unsigned long long v) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const unsigned long long u = __shfl_xor(v, o, 64);
            v = u < v ? u : v;
        }
        return v;
    };
    {
        const unsigned long long v = wave_min(row_probe(y - 32 + lane));
        if (v < best) best = v;
    }
    for (int m = 1;; m++) {
        const long long near = 32 + 64ll * (m - 1);             // smallest |dy| of this round's rows
        if ((unsigned long long)(near * near) >= best) break;
        if (best < (unsigned long long)need2) break;            // already below what the caller cares about
        if (y - near < 0 && y + near >= H) break;               // both chunks outside the image from here on
        unsigned long long v = row_probe(y - 32 - 64 * m + lane);
        const unsigned long long v2 = row_probe(y + 32 + 64 * (m - 1) + lane);
        v = wave_min(v2 < v ? v2 : v);
        if (v < best) best = v;
    }
    return best > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)best;
}

// One 1024-thread workgroup (16 waves, one point per wave at a time) per frame.  out: ((u64)D2 << 32) | (0xFFFFFFFF - flat
// index) of the first farthest background pixel, 0 when the frame has no leaf pixel or no background pixel; flag = 1 when a
// survivor list overflowed (the caller runs the fallback)
#define LGL_BB_T 1024
__global__ __launch_bounds__(LGL_BB_T) void k_edt_bb(const unsigned long long* __restrict__ bits, const unsigned long long* __restrict__ occ,
                                                     int H, int W, int WW, unsigned long long* __restrict__ qa, unsigned long long* __restrict__ qb,
                                                     unsigned long long* __restrict__ out, int* __restrict__ flag) {
    __shared__ unsigned int s_lb2, s_na, s_nb, s_any, s_ovf;
    __shared__ unsigned long long s_res;
    const int fr = blockIdx.x, t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    constexpr int NW = LGL_BB_T / 64;
    bits += (size_t)fr * H * WW; occ += (size_t)fr * H;
    qa += (size_t)fr * LGL_QCAP; qb += (size_t)fr * LGL_QCAP;
    if (t == 0) { s_lb2 = 0; s_na = 0; s_nb = 0; s_any = 0; s_ovf = 0; s_res = 0; }
    __syncthreads();
    {   // any leaf pixel at all?
        unsigned any = 0;
        for (long long i = t; i < (long long)H * WW; i += LGL_BB_T) any |= bits[i] != 0ull;
        if (any) s_any = 1;
    }
    __syncthreads();
    if (!s_any) { if (t == 0) { out[fr] = 0; flag[fr] = 0; } return; }
    // entry: d2 (32) | y0 (16) | x0 (16) of a cell; level 0: cells of S x S pixels
    int S = 64;
    while (((H + S - 1) / S) * ((W + S - 1) / S) > 8192) S *= 2;
    auto centre = [&](int y0, int x0, int s, int& cy, int& cx, double& rad) {
        const int y1 = min(y0 + s, H) - 1, x1 = min(x0 + s, W) - 1;      // last pixel of the cell inside the image
        cy = min(y0 + s / 2, y1); cx = min(x0 + s / 2, x1);
        const int ry = max(cy - y0, y1 - cy), rx = max(cx - x0, x1 - cx);
        rad = sqrt((double)(ry * ry + rx * rx));
    };
    const int ny = (H + S - 1) / S, nx = (W + S - 1) / S;
    // Level 0.  (1) occupancy of the cells (any leaf pixel inside).  (2) every 4th cell in both directions exactly -> a first
    // best value.  (3) the others: a cell next to occupied cells cannot hold the maximum of a frame whose best value is
    // hundreds of pixels -- its centre is at most (distance to the farthest corner of the nearest occupied cell) from a leaf
    // pixel; only cells whose bound from that survives are evaluated, and their search may stop as soon as the cell cannot
    // reach the best value (the list keeps such a cell's inexact, but low enough, value).
    __shared__ unsigned char s_occ[8192];
    for (int c = t; c < ny * nx; c += LGL_BB_T) {
        const int y0 = (c / nx) * S, x0 = (c % nx) * S;
        const int y1 = min(y0 + S, H), x1 = min(x0 + S, W);
        // (S is a multiple of 64: a cell covers whole words, and the rows' word-occupancy masks answer for them)
        const int wq0 = x0 >> 6, nwq = ((x1 - 1) >> 6) - wq0 + 1;
        const unsigned long long wm = (nwq >= 64 ? ~0ull : ((1ull << nwq) - 1ull)) << wq0;
        unsigned long long any = 0;
        for (int yy = y0; yy < y1; yy++) any |= occ[yy] & wm;
        s_occ[c] = any != 0;
    }
    __syncthreads();
    auto need_of = [&](double rad) -> uint32_t {
        const double nd = sqrt((double)s_lb2) - rad - 1e-6;
        return nd <= 0.0 ? 0u : (uint32_t)(nd * nd);
    };
    for (int round = 0; round < 2; round++) {
        for (int c = wave; c < ny * nx; c += NW) {
            const bool coarse = ((c / nx) & 3) == 0 && ((c % nx) & 3) == 0;
            if (coarse != (round == 0)) continue;
            const int y0 = (c / nx) * S, x0 = (c % nx) * S;
            int cy, cx; double rad;
            centre(y0, x0, S, cy, cx, rad);
            uint32_t ub2 = 0xFFFFFFFFu;
            if (round) {   // upper bound of the centre value from the occupied cells (all lanes: 64 cells per step, wave minimum)
                unsigned long long m = ~0ull;
                for (int o = lane; o < ny * nx; o += 64) {
                    if (!s_occ[o]) continue;
                    const int oy0 = (o / nx) * S, ox0 = (o % nx) * S;
                    const int oy1 = min(oy0 + S, H) - 1, ox1 = min(ox0 + S, W) - 1;
                    const long long dyf = max(abs(cy - oy0), abs(cy - oy1)), dxf = max(abs(cx - ox0), abs(cx - ox1));
                    const unsigned long long d = (unsigned long long)(dyf * dyf + dxf * dxf);
                    m = d < m ? d : m;
                }
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) {
                    const unsigned long long u = __shfl_xor(m, o, 64);
                    m = u < m ? u : m;
                }
                ub2 = m >= 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)m;
            }
            uint32_t d2;
            if (round && sqrt((double)ub2) + rad + 1e-6 < sqrt((double)s_lb2)) d2 = ub2;   // cannot reach the best value: not evaluated
            else d2 = edt_point_wave_occ(bits, occ, H, W, WW, cy, cx, ub2, round ? need_of(rad) : 0u, lane);
            if (lane == 0) {
                if (!(round && d2 == ub2 && sqrt((double)ub2) + rad + 1e-6 < sqrt((double)s_lb2))) atomicMax(&s_lb2, d2);
                qb[c] = ((unsigned long long)d2 << 32) | ((unsigned long long)y0 << 16) | (unsigned long long)x0;
            }
        }
        __syncthreads();
    }
    if (t == 0) s_nb = ny * nx;
    __syncthreads();
    for (;;) {
        // filter list B (cells of size S with their centre values) into list A: keep a cell iff its bound reaches the best value
        const unsigned nb = s_nb;
        const double lb = sqrt((double)s_lb2);
        for (unsigned i = t; i < nb; i += LGL_BB_T) {
            const unsigned long long e = qb[i];
            const int y0 = (int)((e >> 16) & 0xFFFF), x0 = (int)(e & 0xFFFF);
            int cy, cx; double rad;
            centre(y0, x0, S, cy, cx, rad);
            if (sqrt((double)(uint32_t)(e >> 32)) + rad + 1e-6 >= lb) {
                const unsigned p = atomicAdd(&s_na, 1u);
                if (p < LGL_QCAP / 4) qa[p] = e; else s_ovf = 1;   // (a quarter: every survivor has four children)
            }
        }
        __syncthreads();
        if (s_ovf) { if (t == 0) { out[fr] = 0; flag[fr] = 1; } return; }
        const unsigned na = s_na;
        if (S == 1) {   // cells are pixels, bounds are exact: the survivors are the farthest pixels; first occurrence wins
            const uint32_t lb2 = s_lb2;
            for (unsigned i = t; i < na; i += LGL_BB_T) {
                const unsigned long long e = qa[i];
                if ((uint32_t)(e >> 32) == lb2) {
                    const uint32_t idx = (uint32_t)((e >> 16) & 0xFFFF) * (uint32_t)W + (uint32_t)(e & 0xFFFF);
                    atomicMax(&s_res, ((unsigned long long)lb2 << 32) | (uint32_t)(0xFFFFFFFFu - idx));
                }
            }
            __syncthreads();
            if (t == 0) { out[fr] = s_lb2 ? s_res : 0ull; flag[fr] = 0; }
            return;
        }
        __syncthreads();
        if (t == 0) { s_nb = 0; }
        __syncthreads();
        // children of the survivors: exact centre values (bounded by the parent's value + the centres' distance)
        const int Sc = S / 2;
        for (unsigned i = wave; i < na * 4; i += NW) {
            const unsigned long long e = qa[i >> 2];
            const int py0 = (int)((e >> 16) & 0xFFFF), px0 = (int)(e & 0xFFFF);
            const int y0 = py0 + ((i >> 1) & 1) * Sc, x0 = px0 + (i & 1) * Sc;
            if (y0 >= H || x0 >= W) continue;
            int pcy, pcx, cy, cx; double prad, rad;
            centre(py0, px0, S, pcy, pcx, prad);
            centre(y0, x0, Sc, cy, cx, rad);
            // the parent's value is exact (it survived the filter: it was not cut short): Lipschitz bounds for the child
            const double pd = sqrt((double)(uint32_t)(e >> 32)), cd = sqrt((double)((cy - pcy) * (cy - pcy) + (cx - pcx) * (cx - pcx)));
            const double ubd = pd + cd + 1e-6, lod = pd - cd - 1e-6;
            const double ub2d = ceil(ubd * ubd);
            const uint32_t ub2 = ub2d >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)ub2d;
            const uint32_t lo2 = lod <= 1.0 ? 0u : (uint32_t)floor((lod - 1e-6) * (lod - 1e-6));
            (void)lo2;
            const uint32_t d2 = edt_point_wave_occ(bits, occ, H, W, WW, cy, cx, ub2, need_of(rad), lane);
            if (lane == 0) {
                atomicMax(&s_lb2, d2);
                const unsigned p = atomicAdd(&s_nb, 1u);
                qb[p] = ((unsigned long long)d2 << 32) | ((unsigned long long)y0 << 16) | (unsigned long long)x0;   // p < 4 * QCAP / 4
            }
        }
        __syncthreads();
        if (t == 0) s_na = 0;
        S = Sc;
        __syncthreads();
    }
}

// ---------------------------------------------------------------- results in their final form, one copy back
struct LeafHdr { int n_leaves, status, ext[4], bbflag, pad; };
// one workgroup per frame: lg_leaf_stat of every slot (id from the presence bits, np.median from the selected key / successor)
// and the frame header (label count, extrema) -> pinned host memory in ONE copy
__global__ __launch_bounds__(256) void k_pack(const unsigned long long* __restrict__ pres, const int* __restrict__ pre,
                                              const int* __restrict__ nlab, const LeafAcc* __restrict__ acc,
                                              const SelState* __restrict__ st, const uint32_t* __restrict__ succ,
                                              const unsigned long long* __restrict__ first_leaf,
                                              const unsigned long long* __restrict__ best, const int* __restrict__ bbflag, int W,
                                              int max_leaves, lg_leaf_stat* __restrict__ out, LeafHdr* __restrict__ hdr) {
    const size_t fr = blockIdx.x;
    pres += fr * 512; pre += fr * 512; acc += fr * LGL_MAXL; st += fr * LGL_MAXL; succ += fr * LGL_MAXL;
    out += fr * (size_t)max_leaves;
    const int nl = nlab[fr];
    const int status = nl > LGL_MAXL ? LG_ERR_UNSUPPORTED : (nl > max_leaves ? LG_ERR_INVALID : LG_OK);
    if (threadIdx.x == 0) {
        LeafHdr h;
        h.n_leaves = status == LG_ERR_UNSUPPORTED ? 0 : nl;   // (too small a result array: the count tells the caller how much room it needs)
        h.status = status;
        h.ext[0] = h.ext[1] = h.ext[2] = h.ext[3] = 0;
        h.bbflag = bbflag[fr]; h.pad = 0;
        const unsigned long long fl = first_leaf[fr], bs = best[fr];
        if (fl != ~0ull) { h.ext[0] = (int)(fl / (unsigned)W); h.ext[1] = (int)(fl % (unsigned)W); }
        if (bs != 0) {   // (no leaf / no background pixel: the field is constant -> arg-max index 0)
            const uint32_t idx = 0xFFFFFFFFu - (uint32_t)(bs & 0xFFFFFFFFull);
            h.ext[2] = (int)(idx / (unsigned)W); h.ext[3] = (int)(idx % (unsigned)W);
        }
        hdr[fr] = h;
    }
    if (status != LG_OK) return;
    for (int slot = threadIdx.x; slot < nl; slot += 256) {
        // id of the slot: the (slot - pre[w])-th set bit of the presence word w with pre[w] <= slot < pre[w + 1]
        int lo = 0, hi = 511;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (pre[mid] <= slot) lo = mid; else hi = mid - 1;
        }
        unsigned long long wv = pres[lo];
        for (int k = slot - pre[lo]; k > 0; k--) wv &= wv - 1;
        lg_leaf_stat o;
        o.id = lo * 64 + __builtin_ctzll(wv);
        o.area = (int32_t)acc[slot].area;
        o.touches_border = acc[slot].border;
        o.pad_ = 0;
        o.sum_x = (double)acc[slot].sum_x;
        o.sum_y = (double)acc[slot].sum_y;
        o.sum_depth = acc[slot].sum_depth;
        o.sum_ray = acc[slot].sum_ray;
        // np.median: odd n -> middle element; even n -> float32 mean of the two middle elements
        const float lo_v = key2f(st[slot].key);
        if (acc[slot].area % 2 == 1) {
            o.median_depth = lo_v;
        } else {
            const float hi_v = (st[slot].n_le > 0) ? lo_v : key2f(succ[slot]);   // duplicates of the key cover the upper index
            o.median_depth = __fdiv_rn(__fadd_rn(lo_v, hi_v), 2.0f);           // float32 add, then / 2 (np.mean of 2 float32)
        }
        o.pad2_ = 0.f;
        out[slot] = o;
    }
}

// ---------------------------------------------------------------- exact EDT extrema, full-transform form (fallback)
// phase 1: g[y][x] = vertical distance to the nearest leaf pixel of column x (sentinel when none)
#define LGL_GINF 16384
__global__ __launch_bounds__(256) void k_coldist(const int16_t* __restrict__ lab, int H, int W, uint16_t* __restrict__ g) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    lab += (size_t)blockIdx.y * H * W;
    g += (size_t)blockIdx.y * H * W;
    int d = LGL_GINF;
#pragma unroll 8
    for (int y = 0; y < H; y++) {
        d = (lab[(size_t)y * W + x] >= 1) ? 0 : min(d + 1, LGL_GINF);
        g[(size_t)y * W + x] = (uint16_t)d;
    }
    d = LGL_GINF;
#pragma unroll 8
    for (int y = H - 1; y >= 0; y--) {
        int v = g[(size_t)y * W + x];
        d = (v == 0) ? 0 : min(d + 1, LGL_GINF);
        if (d < v) g[(size_t)y * W + x] = (uint16_t)d;
    }
}

// phase 2: one wave per row.  D2(x) = min_x' (x-x')^2 + g(x')^2 ; the leftmost arg-min is monotone in x
// (Monge), so positions are solved in bisection order, each searching only between its solved neighbours.
template <int WP2>  // padded power-of-two width
__global__ __launch_bounds__(64) void k_rowedt(const uint16_t* __restrict__ g, int H, int W,
                                               unsigned long long* __restrict__ best) {
    // 16-bit LDS images (g <= 16384, columns < 8192): 8.5 KB per row-workgroup -> 18 resident rows per CU instead of 9;
    // the pass is a chain of dependent LDS round trips, more rows in flight is what hides them
    __shared__ uint16_t s_g[WP2];
    __shared__ uint16_t s_opt[WP2 + 2];
    __shared__ int s_big[WP2 / 64 + 2];
    __shared__ int s_nbig;
    const int y = blockIdx.x, lane = threadIdx.x;
    g += (size_t)blockIdx.y * H * W;
    best += (size_t)blockIdx.y * H;     // per-row results: H x B same-address atomics would serialise at one L2 channel
    for (int x = lane; x < WP2; x += 64) {
        s_g[x] = (x < W) ? g[(size_t)y * W + x] : (uint16_t)LGL_GINF;
    }
    __syncthreads();
    unsigned long long mykey = 0;
    auto cost = [&](int x, int xp) { const int d = x - xp, gv = (int)s_g[xp]; return d * d + gv * gv; };
    // level l solves positions p = (2i+1) * WP2 / 2^(l+1); neighbours p -/+ half are solved (or the borders)
    for (int half = WP2 / 2; half >= 1; half >>= 1) {
        const int nsub = WP2 / (2 * half);
        {
            // One position per lane and round.  The brackets of a level add up to <= W + nsub columns, but where the
            // arg-min jumps between two leaves a single bracket spans the whole gap (hundreds of columns) at EVERY level:
            // left to its lane it serialises the wave (measured 300 us per row).  Brackets wider than 64 columns are
            // queued and searched afterwards by all 64 lanes together.
            if (lane == 0) s_nbig = 0;
            __syncthreads();
            for (int sub = lane; sub < nsub; sub += 64) {
                const int p = (2 * sub + 1) * half;
                int res = W - 1;
                if (p < W) {
                    const int lo = (p - half > 0) ? s_opt[p - half] : 0;
                    const int hi = (p + half < WP2) ? s_opt[p + half] : W - 1;
                    if (hi - lo > 64) {
                        const int q = atomicAdd(&s_nbig, 1);
                        s_big[q] = p;          // at most W / 64 such brackets per level
                        continue;              // s_opt[p] is written by the cooperative pass (nobody reads it before)
                    }
                    int bc = INT_MAX, bx = lo;
                    for (int xp = lo; xp <= hi; xp++) {
                        int c = cost(p, xp);
                        if (c < bc) { bc = c; bx = xp; }
                    }
                    res = bx;
                    if (bc > 0) {
                        unsigned long long key = ((unsigned long long)(uint32_t)bc << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)(y * W + p));
                        mykey = key > mykey ? key : mykey;
                    }
                }
                s_opt[p] = (uint16_t)res;
            }
            __syncthreads();
            const int nbig = s_nbig;
            for (int q = 0; q < nbig; q++) {
                const int p = s_big[q];
                const int lo = (p - half > 0) ? s_opt[p - half] : 0;
                const int hi = (p + half < WP2) ? s_opt[p + half] : W - 1;
                int bc = INT_MAX, bx = lo;
                for (int xp = lo + lane; xp <= hi; xp += 64) {
                    int c = cost(p, xp);
                    if (c < bc) { bc = c; bx = xp; }
                }
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {   // leftmost arg-min across the wave
                    int oc = __shfl_xor(bc, o, 64), ox = __shfl_xor(bx, o, 64);
                    if (oc < bc || (oc == bc && ox < bx)) { bc = oc; bx = ox; }
                }
                if (lane == 0) {
                    s_opt[p] = (uint16_t)bx;
                    if (bc > 0) {
                        unsigned long long key = ((unsigned long long)(uint32_t)bc << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)(y * W + p));
                        mykey = key > mykey ? key : mykey;
                    }
                }
            }
            __syncthreads();
        }
    }
    // position 0 is never a bisection midpoint: solve it against [0, opt(1)]
    if (lane == 0) {
        const int hi = (WP2 > 1) ? s_opt[1] : W - 1;
        int bc = INT_MAX;
        for (int xp = 0; xp <= min(hi, W - 1); xp++) bc = min(bc, cost(0, xp));
        if (bc > 0) {
            unsigned long long key = ((unsigned long long)(uint32_t)bc << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)(y * W));
            mykey = key > mykey ? key : mykey;
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        unsigned long long w = __shfl_xor(mykey, o, 64);
        mykey = w > mykey ? w : mykey;
    }
    if (lane == 0) best[y] = mykey;
}

// max over the rows of a frame (one workgroup per frame)
__global__ __launch_bounds__(256) void k_rowbest(const unsigned long long* __restrict__ rowbest, int H,
                                                 unsigned long long* __restrict__ best) {
    __shared__ unsigned long long s_b[4];
    const unsigned long long* rb = rowbest + (size_t)blockIdx.x * H;
    unsigned long long m = 0;
    for (int y = threadIdx.x; y < H; y += 256) m = rb[y] > m ? rb[y] : m;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        unsigned long long w = __shfl_xor(m, o, 64);
        m = w > m ? w : m;
    }
    if ((threadIdx.x & 63) == 0) s_b[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; i++) m = s_b[i] > m ? s_b[i] : m;
        best[blockIdx.x] = m;
    }
}

}  // namespace

// ---- per-kernel timing (event pairs on the launching stream, resolved after the call's synchronise)
struct LgLeafProf {
    struct Slot { std::string name; std::vector<hipEvent_t> ev; size_t used = 0; int launches = 0; double ms = 0.0; };
    bool on = false;
    std::vector<Slot> slots;
};
LgLeafProf* lg_leaf_prof_new() { return new (std::nothrow) LgLeafProf(); }
void lg_leaf_prof_free(LgLeafProf* p) {
    if (!p) return;
    for (auto& sl : p->slots)
        for (auto e : sl.ev) hipEventDestroy(e);
    delete p;
}
void lg_leaf_prof_enable(LgLeafProf* p, int on) {
    if (!p) return;
    p->on = on != 0;
    for (auto& sl : p->slots) { sl.used = 0; sl.launches = 0; sl.ms = 0.0; }
}
static void leaf_prof_flush(LgLeafProf* p) {
    if (!p) return;
    for (auto& sl : p->slots) {
        for (size_t i = 0; i + 1 < sl.used; i += 2) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, sl.ev[i], sl.ev[i + 1]) == hipSuccess) { sl.ms += ms; sl.launches++; }
        }
        sl.used = 0;
    }
}
int lg_leaf_prof_read(LgLeafProf* p, const char* name, int* launches, double* total_ms) {
    if (!p || strncmp(name, "leaf_", 5) != 0) return 0;
    for (auto& sl : p->slots)
        if (sl.name == name) { if (launches) *launches = sl.launches; if (total_ms) *total_ms = sl.ms; return 1; }
    return 1;
}
namespace {
struct LeafProfScope {   // an event pair around the launches of its scope
    hipEvent_t e1 = nullptr;
    hipStream_t s;
    LeafProfScope(LgLeafProf* p, const char* name, hipStream_t s_) : s(s_) {
        if (!p || !p->on) return;
        LgLeafProf::Slot* sl = nullptr;
        for (auto& x : p->slots)
            if (x.name == name) sl = &x;
        if (!sl) { p->slots.emplace_back(); sl = &p->slots.back(); sl->name = name; }
        if (sl->used + 2 > sl->ev.size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
            sl->ev.push_back(a); sl->ev.push_back(b);
        }
        hipEventRecord(sl->ev[sl->used], s);
        e1 = sl->ev[sl->used + 1];
        sl->used += 2;
    }
    ~LeafProfScope() { if (e1) hipEventRecord(e1, s); }
};
}  // namespace

struct LgLeafWs {   // every array holds capB frames back to back
    unsigned long long* pres;  // 512
    int* pre;                  // 512
    int* nlab;                 // 1
    LeafAcc* acc;              // MAXL
    unsigned long long* first_leaf;  // 1
    SelState* st;              // MAXL
    uint32_t* hist;            // MAXL*256, kept all-zero between calls (k_select clears what k_hist fills)
    uint32_t* succ;            // MAXL  successor keys (k_select, last pass)
    uint32_t* above;           // MAXL  smallest relative key above the selected key's bucket (k_hist, last pass)
    int* maxpass;              // 1     radix passes the frame needs (k_seed)
    unsigned long long* best;  // 1
    int* bbflag;               // 1
    unsigned int* comp_n;      // 256 (one per k_accumulate workgroup)
    unsigned long long *qa, *qb;  // LGL_QCAP each
    unsigned long long* bits;  // H * ceil(W/64)
    unsigned long long* comp;  // H*W  (slot, key) list of the leaf pixels
    unsigned long long* occ;   // H     word occupancy of the bit rows (k_rowocc)
    size_t occ_cap;
    uint16_t* g;               // H*W   (fallback only, allocated on first use)
    unsigned long long* rowbest;  // H  (fallback only)
    lg_leaf_stat* d_out;       // [capB][out_cap] results in their final form (device) ...
    LeafHdr* d_hdr;            // [capB]
    lg_leaf_stat* h_out;       // ... and their pinned host copies
    LeafHdr* h_hdr;
    int out_cap;
    size_t px_cap, bits_cap, g_cap, rb_cap;   // pixels (comp list), 64-bit words of the bit rows, pixels, rows
    int capB;
    hipEvent_t ev_in, ev_side; // fences of the side chain (the side stream belongs to the handle)
};

void lg_leaf_free(LgLeafWs*& w) {
    if (!w) return;
    void* ps[] = {w->pres, w->pre, w->nlab, w->acc, w->first_leaf, w->st, w->hist, w->succ, w->above, w->maxpass, w->best, w->bbflag, w->comp_n,
                  w->qa, w->qb, w->bits, w->comp, w->g, w->rowbest, w->d_out, w->d_hdr, w->occ};
    for (void* p : ps)
        if (p) hipFree(p);
    if (w->h_out) hipHostFree(w->h_out);
    if (w->h_hdr) hipHostFree(w->h_hdr);
    if (w->ev_in) hipEventDestroy(w->ev_in);
    if (w->ev_side) hipEventDestroy(w->ev_side);
    delete w;
    w = nullptr;
}

static int leaf_ws(LgLeafWs*& w, int B, int H, int W) {
    if (w && B > w->capB) lg_leaf_free(w);
    if (!w) {
        w = new LgLeafWs();
        memset(w, 0, sizeof(*w));
        const size_t nb = (size_t)B;
        if (hipMalloc((void**)&w->pres, nb * 512 * 8) || hipMalloc((void**)&w->pre, nb * 512 * 4) ||
            hipMalloc((void**)&w->nlab, nb * 4) || hipMalloc((void**)&w->acc, nb * sizeof(LeafAcc) * LGL_MAXL) ||
            hipMalloc((void**)&w->first_leaf, nb * 8) || hipMalloc((void**)&w->st, nb * sizeof(SelState) * LGL_MAXL) ||
            hipMalloc((void**)&w->hist, nb * 4 * LGL_MAXL * 256) || hipMalloc((void**)&w->succ, nb * 4 * LGL_MAXL) ||
            hipMalloc((void**)&w->above, nb * 4 * LGL_MAXL) || hipMalloc((void**)&w->maxpass, nb * 4) ||
            hipMalloc((void**)&w->best, nb * 8) || hipMalloc((void**)&w->bbflag, nb * 4) || hipMalloc((void**)&w->comp_n, nb * 4 * 256) ||
            hipMalloc((void**)&w->qa, nb * 8 * LGL_QCAP) || hipMalloc((void**)&w->qb, nb * 8 * LGL_QCAP) ||
            hipMemset(w->hist, 0, nb * 4 * LGL_MAXL * 256) || hipMalloc((void**)&w->d_hdr, nb * sizeof(LeafHdr)) ||
            hipHostMalloc((void**)&w->h_hdr, nb * sizeof(LeafHdr)))
            return LG_ERR_NOMEM;
        if (hipEventCreateWithFlags(&w->ev_in, hipEventDisableTiming) || hipEventCreateWithFlags(&w->ev_side, hipEventDisableTiming))
            return LG_ERR_HIP;
        w->capB = B;
    }
    // the bit rows are H rows of ceil(W / 64) words: their word count does not follow the pixel count (480 x 640 needs 4800
    // words per frame, 640 x 480 needs 5120), so it has a capacity of its own
    const size_t need = (size_t)w->capB * H * W, need_words = (size_t)w->capB * H * ((W + 63) / 64);
    if (need_words > w->bits_cap) {
        if (w->bits) hipFree(w->bits);
        w->bits = nullptr;
        w->bits_cap = 0;
        if (hipMalloc((void**)&w->bits, need_words * 8)) return LG_ERR_NOMEM;
        w->bits_cap = need_words;
    }
    if ((size_t)w->capB * H > w->occ_cap) {
        if (w->occ) hipFree(w->occ);
        w->occ = nullptr;
        w->occ_cap = 0;
        if (hipMalloc((void**)&w->occ, (size_t)w->capB * H * 8)) return LG_ERR_NOMEM;
        w->occ_cap = (size_t)w->capB * H;
    }
    if (need > w->px_cap) {
        if (w->comp) hipFree(w->comp);
        w->comp = nullptr;
        w->px_cap = 0;
        // (comp: per frame the segments of up to 256 workgroups, each rounded up to whole 4096-pixel steps)
        if (hipMalloc((void**)&w->comp, (need + (size_t)w->capB * 256 * 4096 * 2) * 8)) return LG_ERR_NOMEM;
        w->px_cap = need;
    }
    return LG_OK;
}

// the full-transform pass (column scan + per-row lower envelope) for a batch: only when a frame's branch-and-bound list overflowed
static int leaf_edt_fallback(LgLeafWs* w, const int16_t* labels, int B, int H, int W, hipStream_t s) {
    const size_t need = (size_t)w->capB * H * W, need_rb = (size_t)w->capB * H;
    if (need > w->g_cap) {
        if (w->g) hipFree(w->g);
        w->g = nullptr; w->g_cap = 0;
        if (hipMalloc((void**)&w->g, need * 2)) return LG_ERR_NOMEM;
        w->g_cap = need;
    }
    if (need_rb > w->rb_cap) {
        if (w->rowbest) hipFree(w->rowbest);
        w->rowbest = nullptr; w->rb_cap = 0;
        if (hipMalloc((void**)&w->rowbest, need_rb * 8)) return LG_ERR_NOMEM;
        w->rb_cap = need_rb;
    }
    hipLaunchKernelGGL(k_coldist, dim3((W + 255) / 256, B), dim3(256), 0, s, labels, H, W, w->g);
    if (W <= 512) hipLaunchKernelGGL(k_rowedt<512>, dim3(H, B), dim3(64), 0, s, w->g, H, W, w->rowbest);
    else if (W <= 1024) hipLaunchKernelGGL(k_rowedt<1024>, dim3(H, B), dim3(64), 0, s, w->g, H, W, w->rowbest);
    else if (W <= 2048) hipLaunchKernelGGL(k_rowedt<2048>, dim3(H, B), dim3(64), 0, s, w->g, H, W, w->rowbest);
    else hipLaunchKernelGGL(k_rowedt<4096>, dim3(H, B), dim3(64), 0, s, w->g, H, W, w->rowbest);
    hipLaunchKernelGGL(k_rowbest, dim3(B), dim3(256), 0, s, w->rowbest, H, w->best);
    return LG_OK;
}

// B frames per call: every kernel carries the frame in blockIdx.y (blockIdx.x for the one-workgroup-per-frame steps), no
// host round trip between the passes (the median ranks are seeded on the device), one copy-back at the end.
int lg_leaf_run_batch(LgLeafWs*& w, const int16_t* labels, const float* depth, int B, int H, int W, float cx, float cy,
                      float f, lg_leaf_stat* stats, int max_leaves, int* n_leaves, int32_t* extrema, int* status,
                      hipStream_t s, hipStream_t side, std::string* err, LgLeafProf* prof) {
    int rc = leaf_ws(w, B, H, W);
    if (rc) { *err = "lg_leaf_stats: workspace allocation failed"; return rc; }
    if (W > 4096 || H > 65535) { *err = "lg_leaf_stats: width > 4096 unsupported"; return LG_ERR_UNSUPPORTED; }
    const size_t nb = (size_t)B;
    const int WW = (W + 63) / 64;
    // streaming passes: grid-stride workgroups per frame; fewer per frame for large batches (every workgroup ends with a
    // flush of its LDS accumulators / 64 KB histogram into global atomics)
    static const int gx_budget = getenv("LG_LEAF_GX") ? atoi(getenv("LG_LEAF_GX")) : 2048;   // workgroups per batch of the streaming passes
    const int gx = std::max(16, std::min(256, gx_budget / B));
    const dim3 grid(gx, B);
    // list segments: workgroup x of a frame meets at most ceil(runs / (gx * 256)) steps of 256 runs of 16 pixels (k_accumulate
    // walks tiles of 4 x 16 runs: the run count is the padded one)
    const long long nruns_acc = (long long)(((W + LGL_RUN - 1) / LGL_RUN + 3) / 4) * ((H + 15) / 16) * 64;
    const unsigned segcap = (unsigned)((nruns_acc + (long long)gx * 256 - 1) / ((long long)gx * 256)) * 256u * LGL_RUN;
    const size_t comp_stride = (size_t)segcap * gx;          // <= H * W + gx * 4096 entries per frame
    static const int ablate = getenv("LG_LEAF_ABLATE") ? atoi(getenv("LG_LEAF_ABLATE")) : 0;   // timing experiments only (wrong results)
    hipMemsetAsync(w->pres, 0, nb * 512 * 8, s);
    hipMemsetAsync(w->acc, 0, nb * sizeof(LeafAcc) * LGL_MAXL, s);
    hipMemsetAsync(w->first_leaf, 0xFF, nb * 8, s);
    hipMemsetAsync(w->above, 0xFF, nb * 4 * LGL_MAXL, s);
    hipMemsetAsync(w->maxpass, 0, nb * 4, s);
    {
        LeafProfScope ps(prof, "leaf_presence", s);
        hipLaunchKernelGGL(k_presence_bits, grid, dim3(256), 0, s, labels, depth, H, W, WW, w->pres, w->bits, w->first_leaf);
    }
    // clutter extrema: needs only the bit mask; runs beside the statistics chain on a second stream of the handle
    // (side = one of the handle's own streams: a stream more per process changes how ROCm maps streams to its few hardware
    //  queues -- an extra stream here made an unrelated trainer in the same process 2.5x slower)
    hipStream_t s2 = side ? side : s;
    hipEventRecord(w->ev_in, s);
    hipStreamWaitEvent(s2, w->ev_in, 0);
    {
        LeafProfScope ps(prof, "leaf_edt", s2);
        hipLaunchKernelGGL(k_rowocc, dim3(std::max(4, std::min(64, 2048 / B)), B), dim3(256), 0, s2, w->bits, H, WW, w->occ);
        hipLaunchKernelGGL(k_edt_bb, dim3(B), dim3(LGL_BB_T), 0, s2, w->bits, w->occ, H, W, WW, w->qa, w->qb, w->best, w->bbflag);
    }
    hipEventRecord(w->ev_side, s2);
    hipLaunchKernelGGL(k_prefix, dim3(B), dim3(64), 0, s, w->pres, w->pre, w->nlab);
    {
        LeafProfScope ps(prof, "leaf_accumulate", s);
        hipLaunchKernelGGL(k_accumulate, grid, dim3(256), 0, s, labels, depth, H, W, w->pres, w->pre, cx, cy, f, w->acc, w->comp, w->comp_n, comp_stride, segcap, ablate);
    }
    // exact medians: ranks from the areas, 4 radix passes over the leaf-pixel list, successor for even counts
    hipLaunchKernelGGL(k_seed, dim3(LGL_MAXL / 256, B), dim3(256), 0, s, w->acc, w->st, w->maxpass);
    for (int pass = 3; pass >= 0; pass--) {
        {
            LeafProfScope ps(prof, "leaf_hist", s);
            hipLaunchKernelGGL(k_hist<16>, grid, dim3(256), 0, s, w->comp, w->comp_n, comp_stride, segcap, w->nlab, w->st, w->maxpass, pass, w->hist,
                               w->above, ablate);
        }
        LeafProfScope ps(prof, "leaf_select", s);
        hipLaunchKernelGGL(k_select, dim3(16, B), dim3(256), 0, s, w->st, w->hist, w->nlab, w->maxpass, pass, w->above, w->succ);
    }
    hipStreamWaitEvent(s, w->ev_side, 0);   // join: `best` is read back below

    // results in their final form: one pack kernel, one copy into pinned memory
    if (max_leaves > w->out_cap) {
        if (w->d_out) hipFree(w->d_out);
        if (w->h_out) hipHostFree(w->h_out);
        w->d_out = nullptr; w->h_out = nullptr; w->out_cap = 0;
        const int cap = std::max(max_leaves, 64);
        if (hipMalloc((void**)&w->d_out, (size_t)w->capB * cap * sizeof(lg_leaf_stat)) ||
            hipHostMalloc((void**)&w->h_out, (size_t)w->capB * cap * sizeof(lg_leaf_stat))) {
            *err = "lg_leaf_stats: workspace allocation failed";
            return LG_ERR_NOMEM;
        }
        w->out_cap = cap;
    }
    hipLaunchKernelGGL(k_pack, dim3(B), dim3(256), 0, s, w->pres, w->pre, w->nlab, w->acc, w->st, w->succ, w->first_leaf, w->best,
                       w->bbflag, W, max_leaves, w->d_out, w->d_hdr);
    if (hipMemcpyAsync(w->h_hdr, w->d_hdr, nb * sizeof(LeafHdr), hipMemcpyDeviceToHost, s) ||
        hipMemcpyAsync(w->h_out, w->d_out, nb * max_leaves * sizeof(lg_leaf_stat), hipMemcpyDeviceToHost, s) ||
        hipStreamSynchronize(s) || hipGetLastError()) {
        *err = "lg_leaf_stats: device copy / kernel failed";
        return LG_ERR_HIP;
    }
    if (prof && prof->on) leaf_prof_flush(prof);
    bool any_flag = false;
    for (int b = 0; b < B; b++) any_flag |= w->h_hdr[b].bbflag != 0;
    if (any_flag) {   // a survivor list overflowed somewhere: the full-transform pass for the batch, then the headers once more
        rc = leaf_edt_fallback(w, labels, B, H, W, s);
        if (rc) { *err = "lg_leaf_stats: workspace allocation failed"; return rc; }
        hipMemsetAsync(w->bbflag, 0, nb * 4, s);
        hipLaunchKernelGGL(k_pack, dim3(B), dim3(256), 0, s, w->pres, w->pre, w->nlab, w->acc, w->st, w->succ, w->first_leaf, w->best,
                           w->bbflag, W, max_leaves, w->d_out, w->d_hdr);
        if (hipMemcpyAsync(w->h_hdr, w->d_hdr, nb * sizeof(LeafHdr), hipMemcpyDeviceToHost, s) || hipStreamSynchronize(s)) {
            *err = "lg_leaf_stats: device copy / kernel failed";
            return LG_ERR_HIP;
        }
    }
    int worst = LG_OK;
    for (int b = 0; b < B; b++) {
        const LeafHdr& h = w->h_hdr[b];
        n_leaves[b] = h.n_leaves;
        for (int k = 0; k < 4; k++) extrema[4 * (size_t)b + k] = h.ext[k];
        if (status) status[b] = h.status;
        if (h.status == LG_ERR_UNSUPPORTED) *err = "lg_leaf_stats: more than 1024 distinct leaf labels in one frame";
        else if (h.status == LG_ERR_INVALID) *err = "lg_leaf_stats: stats capacity too small";
        if (h.status) { worst = h.status; continue; }
        memcpy(stats + (size_t)b * max_leaves, w->h_out + (size_t)b * max_leaves, sizeof(lg_leaf_stat) * (size_t)h.n_leaves);
    }
    return status ? LG_OK : worst;
}

// ---------------------------------------------------------------- the selection itself (host, O(#leaves) per frame)
// OptimalLeafSelector.select_optimal_leaf after the per-leaf passes (scripts/utils/leaf_scorer.py:53-62, 74-181), the arithmetic
// of leaf_scorer.py::_select_from_statistics restated in C++ with the same types and operation order: float32 medians and their
// float32 mean with numpy's pairwise summation, float32 mean depth, float64 scores, non-dominated "max" filter keeping the first
// of identical rows (paretoset 1.2.3), weighted 0.35 / 0.35 / 0.30 pick, first best wins.
namespace {
float np_sum_f32(const float* a, int n) {   // numpy's pairwise sum for n < 128 (one block): 8 partial sums, then the tail
    if (n < 8) {
        float r = 0.0f;     // numpy starts from -0.0; adding +0.0 vs -0.0 differs only for an all-(-0.0) input
        for (int i = 0; i < n; i++) r += a[i];
        return r;
    }
    float r[8];
    for (int k = 0; k < 8; k++) r[k] = a[k];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
        for (int k = 0; k < 8; k++) r[k] += a[i + k];
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
}
}  // namespace

// returns the chosen leaf id (-1: no result = the reference's None); tall: ids with median < mean of medians (capacity tall_cap)
int lg_leaf_select_host(const lg_leaf_stat* st, int n, const int32_t ext[4], int H, int W, double cx, double cy, double f,
                        int32_t* tall, int tall_cap, int* n_tall) {
    *n_tall = 0;
    // torch.unique(mask)[1:] skips the smallest value: the background 0 when present, else the first label (:32)
    long long total = 0;
    for (int i = 0; i < n; i++) total += st[i].area;
    if (total == (long long)H * W && n > 0) { st++; n--; }
    std::vector<float> med;
    med.reserve(n);
    for (int i = 0; i < n; i++)
        if (st[i].area > 0) med.push_back(st[i].median_depth);
    if (med.empty()) return -1;
    const float depth_mean = med.size() >= 128 ? (float)0 : np_sum_f32(med.data(), (int)med.size()) / (float)med.size();
    if (med.size() >= 128) return -2;   // (numpy switches to recursive halving there: the Python path handles such frames)
    struct Cand { int id; double s[3]; bool tall; };
    std::vector<Cand> cands;
    std::vector<int> tall_ids;
    for (int i = 0; i < n; i++)
        if (st[i].median_depth < depth_mean) tall_ids.push_back(st[i].id);
    for (int i = 0; i < n; i++) {
        const double area = (double)st[i].area;
        if (st[i].area < 10000) continue;                                            // :79-81
        const double ctx = st[i].sum_x / area, cty = st[i].sum_y / area;              // :84-88
        const double dmin = sqrt((ctx - ext[1]) * (ctx - ext[1]) + (cty - ext[0]) * (cty - ext[0]));
        const double dmax = sqrt((ctx - ext[3]) * (ctx - ext[3]) + (cty - ext[2]) * (cty - ext[2]));
        const double tot = dmin + dmax;
        const double clutter = tot > 0 ? dmin / tot : 0.0;                            // :91-101
        const float mean_depth = (float)(st[i].sum_depth / area);                     // np.mean of float32 depths (:105-106)
        const double mean_distance = (double)mean_depth / f * (st[i].sum_ray / area);  // :109-115
        const double dist_score = exp(-mean_distance / 0.3);                          // :117
        double vis = 0.0;                                                             // :277-306
        if (!st[i].touches_border) {
            const double d = sqrt((ctx - W / 2.0) * (ctx - W / 2.0) + (cty - H / 2.0) * (cty - H / 2.0));
            vis = 1.0 - d / sqrt((W / 2.0) * (W / 2.0) + (H / 2.0) * (H / 2.0));
        }
        Cand c;
        c.id = st[i].id; c.s[0] = clutter; c.s[1] = dist_score; c.s[2] = vis;
        c.tall = false;
        for (int t_ : tall_ids) c.tall |= t_ == c.id;
        cands.push_back(c);
    }
    if (cands.empty()) return -1;
    bool any_tall = false;
    for (auto& c : cands) any_tall |= c.tall;
    std::vector<const Cand*> pool;
    for (auto& c : cands)
        if (c.tall == any_tall) pool.push_back(&c);
    const double scale = any_tall ? 1.1 : 1.0;                                        // :150-160
    const int m = (int)pool.size();
    std::vector<char> keep(m, 1);
    for (int i = 0; i < m; i++)
        for (int j = 0; j < m; j++) {
            if (i == j) continue;
            bool ge = true, gt = false;
            for (int k = 0; k < 3; k++) {
                const double a = pool[j]->s[k] * scale, b = pool[i]->s[k] * scale;
                if (a < b) { ge = false; break; }
                if (a > b) gt = true;
            }
            if (ge && (gt || j < i)) { keep[i] = 0; break; }
        }
    bool any_keep = false;
    for (int i = 0; i < m; i++) any_keep |= keep[i] != 0;
    double best_s = -INFINITY;
    int best = -1;
    for (int i = 0; i < m; i++) {
        if (any_keep && !keep[i]) continue;
        const double ws = 0.35 * pool[i]->s[0] + 0.35 * pool[i]->s[1] + 0.3 * pool[i]->s[2];   // :170-181
        if (ws > best_s) { best_s = ws; best = pool[i]->id; }
    }
    *n_tall = (int)tall_ids.size();
    for (int i = 0; i < (int)tall_ids.size() && i < tall_cap; i++) tall[i] = tall_ids[i];
    return best;
}

// statistics + selection for B frames, results only: ids [B] (-1 none, -2 "use the Python path"), n_tall [B], tall [B][tall_cap]
int lg_leaf_select_batch_run(LgLeafWs*& w, const int16_t* labels, const float* depth, int B, int H, int W, double cx, double cy,
                             double f, int32_t* ids, int32_t* n_tall, int32_t* tall, int tall_cap, hipStream_t s, hipStream_t side,
                             std::string* err, LgLeafProf* prof) {
    const int cap = 256;   // result rows per frame of this entry point (frames with more labels: -2, the caller's general path)
    std::vector<lg_leaf_stat> stats((size_t)B * cap);
    std::vector<int> nl(B), status(B);
    std::vector<int32_t> ext((size_t)B * 4);
    int rc = lg_leaf_run_batch(w, labels, depth, B, H, W, (float)cx, (float)cy, (float)f, stats.data(), cap, nl.data(), ext.data(), status.data(), s, side, err, prof);
    if (rc) return rc;
    for (int b = 0; b < B; b++) {
        n_tall[b] = 0;
        if (status[b] == LG_ERR_INVALID) { ids[b] = -2; continue; }
        if (status[b] != LG_OK) { ids[b] = -1; continue; }
        ids[b] = lg_leaf_select_host(stats.data() + (size_t)b * cap, nl[b], ext.data() + 4 * (size_t)b, H, W, cx, cy, f,
                                     tall + (size_t)b * tall_cap, tall_cap, &n_tall[b]);
    }
    return LG_OK;
}

int lg_leaf_run(LgLeafWs*& w, const int16_t* labels, const float* depth, int H, int W, float cx, float cy, float f,
                lg_leaf_stat* stats, int max_leaves, int* n_leaves, int32_t* extrema, hipStream_t s, hipStream_t side,
                std::string* err, LgLeafProf* prof) {
    return lg_leaf_run_batch(w, labels, depth, 1, H, W, cx, cy, f, stats, max_leaves, n_leaves, extrema, nullptr, s, side, err, prof);
}
