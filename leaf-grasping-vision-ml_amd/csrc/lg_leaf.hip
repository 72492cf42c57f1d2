// Leaf-selection statistics on gfx950: the per-leaf full-frame passes of
// OptimalLeafSelector.select_optimal_leaf (scripts/utils/leaf_scorer.py:25-203) in a fixed number of
// streaming passes over the int16 label image + f32 depth:
//   * label presence bitmap -> compact slot per id (torch.unique, :32)
//   * per-slot area, sum x, sum y (int64, exact), sum depth, sum ray length (f64), border flag, first leaf pixel
//   * per-slot exact median depth by 4x8-bit radix select on order-preserving keys (+1 successor pass for
//     even counts)  (np.median, :41-47)
//   * global clutter extrema (:66-71): exact squared Euclidean distance to the nearest leaf pixel
//     (column scan + per-row monotone-minima divide and conquer), arg-max with first-occurrence ties.
#include "lg_leaf.h"

#include <limits.h>
#include <math.h>
#include <string.h>

#include <algorithm>
#include <vector>

#define LGL_MAXL 64          // labels per frame supported by the LDS histograms
#define LGL_RUN 16           // consecutive pixels per thread (run-length aggregation before atomics)

namespace {

__device__ __forceinline__ uint32_t f2key(float f) {  // order-preserving float -> uint32
    uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) {
    uint32_t b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __uint_as_float(b);
}

// ---------------------------------------------------------------- presence bitmap of ids 1..32767
__global__ __launch_bounds__(256) void k_presence(const int16_t* __restrict__ lab, long long n,
                                                  unsigned long long* __restrict__ pres) {
    __shared__ unsigned long long s_p[512];
    lab += (size_t)blockIdx.y * n;     // frame = blockIdx.y
    pres += (size_t)blockIdx.y * 512;
    for (int i = threadIdx.x; i < 512; i += 256) s_p[i] = 0;
    __syncthreads();
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long stride = (long long)gridDim.x * 256;
    int last = 0;
    for (; i < n; i += stride) {
        int id = lab[i];
        if (id > 0 && id != last) {
            unsigned long long bit = 1ull << (id & 63);
            if (!(s_p[id >> 6] & bit)) atomicOr(&s_p[id >> 6], bit);
            last = id;
        }
    }
    __syncthreads();
    for (int w = threadIdx.x; w < 512; w += 256)
        if (s_p[w]) atomicOr(&pres[w], s_p[w]);
}

// prefix popcounts of the presence words: slot(id) = pre[id>>6] + popc(pres[id>>6] & ((1<<(id&63))-1))
__global__ void k_prefix(const unsigned long long* __restrict__ pres, int* __restrict__ pre, int* __restrict__ nlab) {
    pres += (size_t)blockIdx.x * 512;  // one workgroup per frame
    pre += (size_t)blockIdx.x * 512;
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int w = 0; w < 512; w++) {
            pre[w] = acc;
            acc += __popcll(pres[w]);
        }
        nlab[blockIdx.x] = acc;
    }
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
    int pad;
};

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

// ---------------------------------------------------------------- per-slot sums
__global__ __launch_bounds__(256) void k_accumulate(const int16_t* __restrict__ lab, const float* __restrict__ depth,
                                                    int H, int W, const unsigned long long* __restrict__ pres,
                                                    const int* __restrict__ pre, float cx, float cy, float f,
                                                    LeafAcc* __restrict__ acc, unsigned long long* __restrict__ first_leaf) {
    __shared__ LeafAcc s_acc[LGL_MAXL];
    __shared__ unsigned long long s_first;
    {
        const size_t fr = blockIdx.y;
        lab += fr * H * W; depth += fr * H * W; pres += fr * 512; pre += fr * 512; acc += fr * LGL_MAXL; first_leaf += fr;
    }
    for (int i = threadIdx.x; i < LGL_MAXL; i += 256) {
        s_acc[i].area = 0; s_acc[i].sum_x = 0; s_acc[i].sum_y = 0;
        s_acc[i].sum_depth = 0.0; s_acc[i].sum_ray = 0.0; s_acc[i].border = 0;
    }
    if (threadIdx.x == 0) s_first = ~0ull;
    __syncthreads();
    const int runs_per_row = (W + LGL_RUN - 1) / LGL_RUN;
    const long long nruns = (long long)H * runs_per_row;
    const double f2 = (double)f * (double)f;
    const bool vec = run_vec_ok(lab, depth, W);
    for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < nruns; r += (long long)gridDim.x * 256) {
        const int y = (int)(r / runs_per_row), x0 = (int)(r % runs_per_row) * LGL_RUN;
        const int x1 = min(x0 + LGL_RUN, W);
        int16_t ids[LGL_RUN];
        if (!load_run(lab, W, y, x0, x1, vec, ids)) continue;
        float dv[LGL_RUN];
        load_run_depth(depth, W, y, x0, x1, vec, dv);
        int cur = 0;
        unsigned long long a = 0; long long sx = 0; double sd = 0.0, sr = 0.0; int bd = 0;
        const double dy = (double)y - (double)cy;
#pragma unroll
        for (int k = 0; k <= LGL_RUN; k++) {
            const int x = x0 + k;
            int id = (k < LGL_RUN) ? (int)ids[k < LGL_RUN ? k : 0] : 0;   // k == LGL_RUN: flush (labels past x1 are 0)
            if (id < 0) id = 0;
            if (id != cur) {
                if (cur > 0 && a) {
                    int s = slot_of(cur, pres, pre);
                    if (s < LGL_MAXL) {
                        atomicAdd(&s_acc[s].area, a);
                        atomicAdd((unsigned long long*)&s_acc[s].sum_x, (unsigned long long)sx);
                        atomicAdd((unsigned long long*)&s_acc[s].sum_y, (unsigned long long)((long long)a * y));
                        atomicAdd(&s_acc[s].sum_depth, sd);
                        atomicAdd(&s_acc[s].sum_ray, sr);
                        if (bd) atomicOr(&s_acc[s].border, 1);
                    }
                }
                cur = id; a = 0; sx = 0; sd = 0.0; sr = 0.0; bd = 0;
            }
            if (id > 0) {
                if (a == 0) atomicMin(&s_first, (unsigned long long)y * W + x);
                a++;
                sx += x;
                sd += (double)dv[k < LGL_RUN ? k : 0];
                const double dx = (double)x - (double)cx;
                sr += sqrt(dx * dx + dy * dy + f2);
                bd |= (x == 0) | (x == W - 1) | (y == 0) | (y == H - 1);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < LGL_MAXL; i += 256) {
        if (s_acc[i].area) {
            atomicAdd(&acc[i].area, s_acc[i].area);
            atomicAdd((unsigned long long*)&acc[i].sum_x, (unsigned long long)s_acc[i].sum_x);
            atomicAdd((unsigned long long*)&acc[i].sum_y, (unsigned long long)s_acc[i].sum_y);
            atomicAdd(&acc[i].sum_depth, s_acc[i].sum_depth);
            atomicAdd(&acc[i].sum_ray, s_acc[i].sum_ray);
            if (s_acc[i].border) atomicOr(&acc[i].border, 1);
        }
    }
    if (threadIdx.x == 0 && s_first != ~0ull) atomicMin(first_leaf, s_first);
}

// ---------------------------------------------------------------- radix select (median)
struct SelState {       // per slot
    uint32_t prefix;    // key bits fixed so far (high bits)
    uint32_t rank;      // remaining 0-based rank inside the current prefix bucket
    uint32_t n_le;      // (after the last pass) number of elements <= selected key
    uint32_t key;       // selected key (after the last pass)
};

// median ranks from the per-slot areas (lower median index), one workgroup per frame
__global__ void k_seed(const LeafAcc* __restrict__ acc, SelState* __restrict__ st) {
    const size_t fr = blockIdx.x;
    const int i = threadIdx.x;
    if (i >= LGL_MAXL) return;
    const unsigned long long a = acc[fr * LGL_MAXL + i].area;
    SelState z;
    z.prefix = 0; z.key = 0; z.n_le = 0;
    z.rank = a ? (uint32_t)((a - 1) / 2) : 0;
    st[fr * LGL_MAXL + i] = z;
}

// histogram of digit `pass` (3 = most significant byte) among elements whose higher bytes match the prefix
__global__ __launch_bounds__(256) void k_hist(const int16_t* __restrict__ lab, const float* __restrict__ depth, int H, int W,
                                              const unsigned long long* __restrict__ pres, const int* __restrict__ pre,
                                              const SelState* __restrict__ st, int pass, uint32_t* __restrict__ hist) {
    __shared__ uint32_t s_h[LGL_MAXL * 256];
    {
        const size_t fr = blockIdx.y;
        lab += fr * H * W; depth += fr * H * W; pres += fr * 512; pre += fr * 512; st += fr * LGL_MAXL;
        hist += fr * LGL_MAXL * 256;
    }
    for (int i = threadIdx.x; i < LGL_MAXL * 256; i += 256) s_h[i] = 0;
    __syncthreads();
    const int shift = 8 * pass;
    const uint32_t himask = (pass == 3) ? 0u : (0xFFFFFFFFu << (shift + 8));
    const bool vec = run_vec_ok(lab, depth, W);
    const int runs_per_row = (W + LGL_RUN - 1) / LGL_RUN;
    const long long nruns = (long long)H * runs_per_row;
    for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < nruns; r += (long long)gridDim.x * 256) {
        const int y = (int)(r / runs_per_row), x0 = (int)(r % runs_per_row) * LGL_RUN;
        const int x1 = min(x0 + LGL_RUN, W);
        int16_t ids[LGL_RUN];
        if (!load_run(lab, W, y, x0, x1, vec, ids)) continue;
        float dv[LGL_RUN];
        load_run_depth(depth, W, y, x0, x1, vec, dv);
        int cur_id = 0, cur_slot = -1, cur_bin = -1;
        uint32_t cnt = 0, cur_prefix = 0;
#pragma unroll
        for (int k = 0; k < LGL_RUN; k++) {
            int id = (int)ids[k];
            if (id <= 0) continue;
            if (id != cur_id) {
                if (cnt) { atomicAdd(&s_h[cur_slot * 256 + cur_bin], cnt); cnt = 0; }
                cur_id = id;
                cur_slot = slot_of(id, pres, pre);
                cur_bin = -1;
                if (cur_slot < LGL_MAXL) cur_prefix = st[cur_slot].prefix;
            }
            if (cur_slot >= LGL_MAXL) continue;
            const uint32_t key = f2key(dv[k]);
            if ((key & himask) != (cur_prefix & himask)) continue;
            const int bin = (int)((key >> shift) & 0xFFu);
            if (bin != cur_bin) {
                if (cnt) atomicAdd(&s_h[cur_slot * 256 + cur_bin], cnt);
                cnt = 0;
                cur_bin = bin;
            }
            cnt++;
        }
        if (cnt) atomicAdd(&s_h[cur_slot * 256 + cur_bin], cnt);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < LGL_MAXL * 256; i += 256)
        if (s_h[i]) atomicAdd(&hist[i], s_h[i]);
}

// pick the bin holding the wanted rank, descend; one thread per slot
__global__ void k_select(SelState* __restrict__ st, uint32_t* __restrict__ hist, int pass, int nslots_max) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nslots_max) return;
    st += (size_t)blockIdx.y * LGL_MAXL;
    hist += (size_t)blockIdx.y * LGL_MAXL * 256;
    uint32_t* h = hist + s * 256;
    uint32_t rank = st[s].rank, accum = 0;
    int bin = 255;
    for (int b = 0; b < 256; b++) {
        uint32_t c = h[b];
        if (rank < accum + c) { bin = b; break; }
        accum += c;
    }
    uint32_t inbin = h[bin];
    st[s].prefix |= ((uint32_t)bin) << (8 * pass);
    st[s].rank = rank - accum;
    if (pass == 0) {
        st[s].key = st[s].prefix;
        st[s].n_le = inbin - (rank - accum) - 1;  // elements equal to the key ranked ABOVE the selected one
    }
    for (int b = 0; b < 256; b++) h[b] = 0;  // ready for the next pass
}

// smallest key strictly greater than the selected key, per slot (upper median for even counts)
__global__ __launch_bounds__(256) void k_successor(const int16_t* __restrict__ lab, const float* __restrict__ depth, int H,
                                                   int W, const unsigned long long* __restrict__ pres,
                                                   const int* __restrict__ pre, const SelState* __restrict__ st,
                                                   uint32_t* __restrict__ succ) {
    __shared__ uint32_t s_m[LGL_MAXL];
    {
        const size_t fr = blockIdx.y;
        lab += fr * H * W; depth += fr * H * W; pres += fr * 512; pre += fr * 512; st += fr * LGL_MAXL; succ += fr * LGL_MAXL;
    }
    for (int i = threadIdx.x; i < LGL_MAXL; i += 256) s_m[i] = 0xFFFFFFFFu;
    __syncthreads();
    const long long n = (long long)H * W;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        int id = lab[i];
        if (id <= 0) continue;
        int s = slot_of(id, pres, pre);
        if (s >= LGL_MAXL) continue;
        uint32_t key = f2key(depth[i]);
        if (key > st[s].key && key < s_m[s]) atomicMin(&s_m[s], key);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < LGL_MAXL; i += 256)
        if (s_m[i] != 0xFFFFFFFFu) atomicMin(&succ[i], s_m[i]);
}

// ---------------------------------------------------------------- exact EDT extrema
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

struct LgLeafWs {   // every array holds capB frames back to back
    unsigned long long* pres;  // 512
    int* pre;                  // 512
    int* nlab;                 // 1
    LeafAcc* acc;              // MAXL
    unsigned long long* first_leaf;  // 1
    SelState* st;              // MAXL
    uint32_t* hist;            // MAXL*256
    uint32_t* succ;            // MAXL
    unsigned long long* best;  // 1
    uint16_t* g;               // H*W
    unsigned long long* rowbest;  // H
    size_t g_cap, rb_cap;
    int capB;
    hipEvent_t ev_in, ev_side; // fences of the side chain (the side stream belongs to the handle)
};

void lg_leaf_free(LgLeafWs*& w) {
    if (!w) return;
    void* ps[] = {w->pres, w->pre, w->nlab, w->acc, w->first_leaf, w->st, w->hist, w->succ, w->best, w->g, w->rowbest};
    for (void* p : ps)
        if (p) hipFree(p);
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
            hipMalloc((void**)&w->best, nb * 8))
            return LG_ERR_NOMEM;
        if (hipEventCreateWithFlags(&w->ev_in, hipEventDisableTiming) || hipEventCreateWithFlags(&w->ev_side, hipEventDisableTiming))
            return LG_ERR_HIP;
        w->capB = B;
    }
    size_t need = (size_t)w->capB * H * W;
    if (need > w->g_cap) {
        if (w->g) hipFree(w->g);
        w->g = nullptr;
        w->g_cap = 0;
        if (hipMalloc((void**)&w->g, need * 2)) return LG_ERR_NOMEM;
        w->g_cap = need;
    }
    const size_t need_rb = (size_t)w->capB * H;
    if (need_rb > w->rb_cap) {
        if (w->rowbest) hipFree(w->rowbest);
        w->rowbest = nullptr;
        w->rb_cap = 0;
        if (hipMalloc((void**)&w->rowbest, need_rb * 8)) return LG_ERR_NOMEM;
        w->rb_cap = need_rb;
    }
    return LG_OK;
}

// B frames per call: every kernel carries the frame in blockIdx.y (blockIdx.x for the one-workgroup-per-frame steps), no
// host round trip between the passes (the median ranks are seeded on the device), one copy-back at the end.
int lg_leaf_run_batch(LgLeafWs*& w, const int16_t* labels, const float* depth, int B, int H, int W, float cx, float cy,
                      float f, lg_leaf_stat* stats, int max_leaves, int* n_leaves, int32_t* extrema, int* status,
                      hipStream_t s, hipStream_t side, std::string* err) {
    int rc = leaf_ws(w, B, H, W);
    if (rc) { *err = "lg_leaf_stats: workspace allocation failed"; return rc; }
    if (W > 4096) { *err = "lg_leaf_stats: width > 4096 unsupported"; return LG_ERR_UNSUPPORTED; }
    const long long n = (long long)H * W;
    const size_t nb = (size_t)B;
    // streaming passes: grid-stride workgroups per frame; fewer per frame for large batches (every workgroup ends with a
    // flush of its LDS accumulators / 64 KB histogram into global atomics)
    const int gx = std::max(16, std::min(256, 2048 / B));   // (one per CU for a single frame: every workgroup of a frame ends
                                                             //  with atomics on that frame's few accumulator addresses)
    const dim3 grid(gx, B);
    hipMemsetAsync(w->pres, 0, nb * 512 * 8, s);
    hipMemsetAsync(w->acc, 0, nb * sizeof(LeafAcc) * LGL_MAXL, s);
    hipMemsetAsync(w->first_leaf, 0xFF, nb * 8, s);
    hipMemsetAsync(w->hist, 0, nb * 4 * LGL_MAXL * 256, s);
    hipMemsetAsync(w->succ, 0xFF, nb * 4 * LGL_MAXL, s);
    // clutter extrema: independent of the statistics, and like the median chain bound by latency (dependent LDS round trips
    // there, LDS atomics here) rather than by bandwidth -- the two chains run side by side on two streams
    // (side = one of the handle's own streams: a stream more per process changes how ROCm maps streams to its few hardware
    //  queues -- an extra stream here made an unrelated trainer in the same process 2.5x slower)
    hipStream_t s2 = side ? side : s;
    hipEventRecord(w->ev_in, s);            // the caller's inputs are ready on s
    hipStreamWaitEvent(s2, w->ev_in, 0);
    hipLaunchKernelGGL(k_coldist, dim3((W + 255) / 256, B), dim3(256), 0, s2, labels, H, W, w->g);
    if (W <= 512) hipLaunchKernelGGL(k_rowedt<512>, dim3(H, B), dim3(64), 0, s2, w->g, H, W, w->rowbest);
    else if (W <= 1024) hipLaunchKernelGGL(k_rowedt<1024>, dim3(H, B), dim3(64), 0, s2, w->g, H, W, w->rowbest);
    else if (W <= 2048) hipLaunchKernelGGL(k_rowedt<2048>, dim3(H, B), dim3(64), 0, s2, w->g, H, W, w->rowbest);
    else hipLaunchKernelGGL(k_rowedt<4096>, dim3(H, B), dim3(64), 0, s2, w->g, H, W, w->rowbest);
    hipLaunchKernelGGL(k_rowbest, dim3(B), dim3(256), 0, s2, w->rowbest, H, w->best);
    hipEventRecord(w->ev_side, s2);
    hipLaunchKernelGGL(k_presence, grid, dim3(256), 0, s, labels, n, w->pres);
    hipLaunchKernelGGL(k_prefix, dim3(B), dim3(64), 0, s, w->pres, w->pre, w->nlab);
    hipLaunchKernelGGL(k_accumulate, grid, dim3(256), 0, s, labels, depth, H, W, w->pres, w->pre, cx, cy, f, w->acc,
                       w->first_leaf);
    // exact medians: ranks from the areas, 4 radix passes, successor for even counts
    hipLaunchKernelGGL(k_seed, dim3(B), dim3(LGL_MAXL), 0, s, w->acc, w->st);
    for (int pass = 3; pass >= 0; pass--) {
        hipLaunchKernelGGL(k_hist, grid, dim3(256), 0, s, labels, depth, H, W, w->pres, w->pre, w->st, pass, w->hist);
        hipLaunchKernelGGL(k_select, dim3(1, B), dim3(LGL_MAXL), 0, s, w->st, w->hist, pass, LGL_MAXL);
    }
    hipLaunchKernelGGL(k_successor, grid, dim3(256), 0, s, labels, depth, H, W, w->pres, w->pre, w->st, w->succ);
    hipStreamWaitEvent(s, w->ev_side, 0);   // join: `best` is read back below

    std::vector<int> nlab(nb);
    std::vector<LeafAcc> acc(nb * LGL_MAXL);
    std::vector<unsigned long long> pres(nb * 512), first_leaf(nb), best(nb);
    std::vector<SelState> st(nb * LGL_MAXL);
    std::vector<uint32_t> succ(nb * LGL_MAXL);
    if (hipMemcpyAsync(nlab.data(), w->nlab, nb * 4, hipMemcpyDeviceToHost, s) ||
        hipMemcpyAsync(acc.data(), w->acc, nb * sizeof(LeafAcc) * LGL_MAXL, hipMemcpyDeviceToHost, s) ||
        hipMemcpyAsync(pres.data(), w->pres, nb * 512 * 8, hipMemcpyDeviceToHost, s) ||
        hipMemcpyAsync(st.data(), w->st, nb * sizeof(SelState) * LGL_MAXL, hipMemcpyDeviceToHost, s) ||
        hipMemcpyAsync(succ.data(), w->succ, nb * 4 * LGL_MAXL, hipMemcpyDeviceToHost, s) ||
        hipMemcpyAsync(first_leaf.data(), w->first_leaf, nb * 8, hipMemcpyDeviceToHost, s) ||
        hipMemcpyAsync(best.data(), w->best, nb * 8, hipMemcpyDeviceToHost, s) || hipStreamSynchronize(s) || hipGetLastError()) {
        *err = "lg_leaf_stats: device copy / kernel failed";
        return LG_ERR_HIP;
    }
    int worst = LG_OK;
    for (int b = 0; b < B; b++) {
        const int nl = nlab[b];
        lg_leaf_stat* fstats = stats + (size_t)b * max_leaves;
        int32_t* ext = extrema + 4 * (size_t)b;
        n_leaves[b] = 0;
        ext[0] = ext[1] = ext[2] = ext[3] = 0;
        int fs = LG_OK;
        if (nl > LGL_MAXL) { *err = "lg_leaf_stats: more than 64 distinct leaf labels in one frame"; fs = LG_ERR_UNSUPPORTED; }
        else if (nl > max_leaves) { *err = "lg_leaf_stats: stats capacity too small"; fs = LG_ERR_INVALID; }
        if (status) status[b] = fs;
        if (fs) { worst = fs; continue; }
        const LeafAcc* facc = &acc[(size_t)b * LGL_MAXL];
        const SelState* fst = &st[(size_t)b * LGL_MAXL];
        const uint32_t* fsucc = &succ[(size_t)b * LGL_MAXL];
        // ids in ascending order = slots in ascending order
        int slot = 0;
        for (int wi = 0; wi < 512 && slot < nl; wi++) {
            unsigned long long bits = pres[(size_t)b * 512 + wi];
            while (bits && slot < nl) {
                int bb = __builtin_ctzll(bits);
                bits &= bits - 1;
                lg_leaf_stat& o = fstats[slot];
                memset(&o, 0, sizeof(o));
                o.id = wi * 64 + bb;
                o.area = (int32_t)facc[slot].area;
                o.touches_border = facc[slot].border;
                o.sum_x = (double)facc[slot].sum_x;
                o.sum_y = (double)facc[slot].sum_y;
                o.sum_depth = facc[slot].sum_depth;
                o.sum_ray = facc[slot].sum_ray;
                // np.median: odd n -> middle element; even n -> float32 mean of the two middle elements
                auto k2f = [](uint32_t k) { uint32_t v = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k; float fv; memcpy(&fv, &v, 4); return fv; };
                float lo = k2f(fst[slot].key);
                if (facc[slot].area % 2 == 1) {
                    o.median_depth = lo;
                } else {
                    float hi = (fst[slot].n_le > 0) ? lo : k2f(fsucc[slot]);  // duplicates of the key cover the upper index
                    float sum = lo + hi;                                       // float32 add, then /2 (np.mean of 2 float32)
                    o.median_depth = sum / 2.0f;
                }
                slot++;
            }
        }
        n_leaves[b] = nl;
        // extrema: argmin = first leaf pixel (row-major); argmax = farthest background pixel, first occurrence
        if (first_leaf[b] != ~0ull) { ext[0] = (int32_t)(first_leaf[b] / W); ext[1] = (int32_t)(first_leaf[b] % W); }
        if (best[b] != 0) {   // (no background pixel: the field is all zeros -> argmax index 0)
            uint32_t idx = 0xFFFFFFFFu - (uint32_t)(best[b] & 0xFFFFFFFFull);
            ext[2] = (int32_t)(idx / W); ext[3] = (int32_t)(idx % W);
        }
    }
    return status ? LG_OK : worst;
}

int lg_leaf_run(LgLeafWs*& w, const int16_t* labels, const float* depth, int H, int W, float cx, float cy, float f,
                lg_leaf_stat* stats, int max_leaves, int* n_leaves, int32_t* extrema, hipStream_t s, hipStream_t side,
                std::string* err) {
    return lg_leaf_run_batch(w, labels, depth, 1, H, W, cx, cy, f, stats, max_leaves, n_leaves, extrema, nullptr, s, side, err);
}
