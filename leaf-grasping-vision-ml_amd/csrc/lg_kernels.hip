// gfx950 kernels of the per-pixel grasp-scoring path (score planes, fusion, top-k, patch gather).
// Reference semantics: scripts/utils/grasp_point_selector.py (cited per kernel); design: DESIGN.md.
#include "lg_internal.h"

#include <hip/hip_ext.h>

#include <limits.h>
#include <algorithm>
#include <stdlib.h>

// ============================================================================ helpers
__device__ __forceinline__ int lg_dpp_row_shr(int old, int v, int n) {
    switch (n) {  // dpp_ctrl must be an immediate
        case 1: return __builtin_amdgcn_update_dpp(old, v, 0x111, 0xf, 0xf, false);
        case 2: return __builtin_amdgcn_update_dpp(old, v, 0x112, 0xf, 0xf, false);
        case 4: return __builtin_amdgcn_update_dpp(old, v, 0x114, 0xf, 0xf, false);
        default: return __builtin_amdgcn_update_dpp(old, v, 0x118, 0xf, 0xf, false);
    }
}
// inclusive prefix-min over the 64 lanes of a wave (DPP row shifts + row broadcasts, gfx9 family)
__device__ __forceinline__ int lg_wave_prefix_min(int v) {
    const int id = INT_MAX;
    v = min(v, lg_dpp_row_shr(id, v, 1));
    v = min(v, lg_dpp_row_shr(id, v, 2));
    v = min(v, lg_dpp_row_shr(id, v, 4));
    v = min(v, lg_dpp_row_shr(id, v, 8));
    v = min(v, __builtin_amdgcn_update_dpp(id, v, 0x142, 0xa, 0xf, false));  // row_bcast:15 -> rows 1,3
    v = min(v, __builtin_amdgcn_update_dpp(id, v, 0x143, 0xc, 0xf, false));  // row_bcast:31 -> rows 2,3
    return v;
}
__device__ __forceinline__ int lg_wave_shr1(int old, int v) {  // lane i <- lane i-1 (lane 0 keeps old)
    return __builtin_amdgcn_update_dpp(old, v, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ int lg_wave_shl1(int old, int v) {  // lane i <- lane i+1 (lane 63 keeps old)
    return __builtin_amdgcn_update_dpp(old, v, 0x130, 0xf, 0xf, false);
}
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt(0), which would stall
// every image row of the distance-transform sweeps on its own global prefetch (rows are ~1000 cycles apart,
// HBM latency is longer): LDS operations are complete at lgkmcnt(0), global loads/stores stay in flight.
__device__ __forceinline__ void lg_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// max of a 64-bit key over the wave, returned in every lane.  DPP row shifts / row broadcasts on the two
// halves (a ds_bpermute shuffle costs ~100+ cycles of latency per step; the top-k walk does 13 of these per round).
#define LG_MAX64_STEP(CTRL, RMASK)                                                            \
    {                                                                                         \
        const uint32_t oh = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, CTRL, RMASK, 0xf, false); \
        const uint32_t ol = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, CTRL, RMASK, 0xf, false); \
        const bool take = (oh > hi) || (oh == hi && ol > lo);                                 \
        hi = take ? oh : hi;                                                                  \
        lo = take ? ol : lo;                                                                  \
    }
__device__ __forceinline__ unsigned long long lg_wave_max_u64(unsigned long long v) {
    uint32_t hi = (uint32_t)(v >> 32), lo = (uint32_t)v;
    LG_MAX64_STEP(0x111, 0xf)  // row_shr:1
    LG_MAX64_STEP(0x112, 0xf)  // row_shr:2
    LG_MAX64_STEP(0x114, 0xf)  // row_shr:4
    LG_MAX64_STEP(0x118, 0xf)  // row_shr:8
    LG_MAX64_STEP(0x142, 0xa)  // row_bcast:15 -> rows 1,3
    LG_MAX64_STEP(0x143, 0xc)  // row_bcast:31 -> rows 2,3
    hi = (uint32_t)__builtin_amdgcn_readlane((int)hi, 63);
    lo = (uint32_t)__builtin_amdgcn_readlane((int)lo, 63);
    return ((unsigned long long)hi << 32) | lo;
}
// (score, index) arg-max over the wave where the index grows with the lane: max of the 32-bit scores by DPP (12 instructions),
// then the highest lane holding it -- a third of the instructions of the 64-bit key reduction above, same result as
// max over (score << 32 | index).  Returns the key in every lane; all-zero scores give 0.
__device__ __forceinline__ unsigned long long lg_wave_argmax_lane_ordered(uint32_t score, uint32_t index) {
    uint32_t m = score;
#define LG_MAX32_STEP(CTRL, RMASK) { const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, CTRL, RMASK, 0xf, false); m = o > m ? o : m; }
    LG_MAX32_STEP(0x111, 0xf)  // row_shr:1
    LG_MAX32_STEP(0x112, 0xf)  // row_shr:2
    LG_MAX32_STEP(0x114, 0xf)  // row_shr:4
    LG_MAX32_STEP(0x118, 0xf)  // row_shr:8
    LG_MAX32_STEP(0x142, 0xa)  // row_bcast:15 -> rows 1,3
    LG_MAX32_STEP(0x143, 0xc)  // row_bcast:31 -> rows 2,3
#undef LG_MAX32_STEP
    m = (uint32_t)__builtin_amdgcn_readlane((int)m, 63);
    if (m == 0) return 0ull;
    const unsigned long long who = __ballot(score == m);
    const int lane = 63 - __builtin_clzll(who);
    const uint32_t idx = (uint32_t)__builtin_amdgcn_readlane((int)index, lane);
    return ((unsigned long long)m << 32) | idx;
}
__device__ __forceinline__ uint32_t lg_wave_max_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        uint32_t w = __shfl_xor(v, o, 64);
        v = w > v ? w : v;
    }
    return v;
}
__device__ __forceinline__ uint32_t lg_orderable(float s) {  // monotone float -> uint32 map
    if (s == 0.0f) s = 0.0f;  // -0.0 -> +0.0 (numpy sorts them as equal)
    if (s != s) return 0xFFFFFFFFu;   // NaN of either sign: np.argsort puts NaN last, the reference's [::-1] puts it FIRST (:454)
    uint32_t b = __float_as_uint(s);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
// valid_scores = traditional_score * valid_regions (grasp_point_selector.py:451) -- a product, not a selection: a NaN or
// infinite score (non-finite depth on the leaf) stays NaN where the pixel is not valid, and NaN leads the candidate order
__device__ __forceinline__ float lg_valid_score(float trad, bool valid) { return trad * (valid ? 1.0f : 0.0f); }
__device__ __forceinline__ int lg_reflect(int v, int n) {  // torch 'reflect' index, clamped for safety
    if (v < 0) v = -v;
    if (v >= n) v = 2 * (n - 1) - v;
    return v < 0 ? 0 : (v >= n ? n - 1 : v);
}

// ============================================================================ mask -> bit rows
// bits[b][y][w] bit j = mask[b][y][64w + j] != 0.  One wave-ballot per 64 pixels.
__global__ __launch_bounds__(256) void lg_pack_bits_kernel(const uint8_t* __restrict__ mask,
                                                           unsigned long long* __restrict__ bits, int H, int W, int WW,
                                                           long long nwords_total) {
    const int lane = threadIdx.x & 63;
    long long wid = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6;  // one 64-bit word per wave
    const long long stride = ((long long)gridDim.x * 256) >> 6;
    for (; wid < nwords_total; wid += stride) {
        long long row = wid / WW;  // b*H + y
        int w = (int)(wid - row * WW);
        int x = w * 64 + lane;
        uint8_t m = (x < W) ? mask[row * W + x] : (uint8_t)0;
        unsigned long long b = __ballot(m != 0);
        if (lane == 0) bits[wid] = b;
    }
}

// Vector variant (W % 16 == 0): each lane loads 16 mask bytes (1 KiB per wave instruction), four adjacent
// lanes assemble one 64-bit word.  Slot s of a row covers pixels [16s, 16s+16); slots beyond W are empty.
__global__ __launch_bounds__(256) void lg_pack_bits16_kernel(const uint8_t* __restrict__ mask,
                                                             unsigned long long* __restrict__ bits, int W, int WW,
                                                             long long nslots_total) {
    const int slots_per_row = 4 * WW;
    long long sid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long stride = (long long)gridDim.x * 256;  // multiple of 4: a word's four slots stay in adjacent lanes
    for (; sid < nslots_total; sid += stride) {
        const long long row = sid / slots_per_row;
        const int slot = (int)(sid - row * slots_per_row);
        const int x = slot * 16;
        unsigned b16 = 0;
        if (x < W) {
            const uint4 v = *reinterpret_cast<const uint4*>(mask + row * W + x);
            const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; q++) {
                b16 |= ((w4[q] & 0x000000ffu) ? 1u : 0u) << (4 * q);
                b16 |= ((w4[q] & 0x0000ff00u) ? 2u : 0u) << (4 * q);
                b16 |= ((w4[q] & 0x00ff0000u) ? 4u : 0u) << (4 * q);
                b16 |= ((w4[q] & 0xff000000u) ? 8u : 0u) << (4 * q);
            }
        }
        unsigned long long word = (unsigned long long)b16 << (16 * (slot & 3));
        word |= __shfl_xor(word, 1, 64);
        word |= __shfl_xor(word, 2, 64);
        if ((slot & 3) == 0) bits[row * WW + (slot >> 2)] = word;
    }
}

void lg_launch_pack_bits(const uint8_t* mask, unsigned long long* bits, int B, int H, int W, int WW, hipStream_t s) {
    if ((W & 15) == 0 && ((uintptr_t)mask & 15) == 0) {
        long long nslots = (long long)B * H * WW * 4;
        long long blocks = (nslots + 255) / 256;
        if (blocks > 16384) blocks = 16384;
        // all lanes of a wave must run the same number of iterations (shuffles): pad the slot space
        long long per_iter = blocks * 256;
        long long padded = ((nslots + per_iter - 1) / per_iter) * per_iter;
        (void)padded;
        hipLaunchKernelGGL(lg_pack_bits16_kernel, dim3((unsigned)blocks), dim3(256), 0, s, mask, bits, W, WW, nslots);
        return;
    }
    long long nwords = (long long)B * H * WW;
    long long waves = nwords;
    int blocks = (int)((waves + 3) / 4);
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(lg_pack_bits_kernel, dim3(blocks), dim3(256), 0, s, mask, bits, H, W, WW, nwords);
}

// The node's `optimal_mask = mask_tensor == optimal_leaf_id` (leaf_grasp_node_v3.py:118) folded into the bit-row pass: labels
// [B][H][W] int16 and one leaf id per frame in, the 0 / 1 byte mask (what the sweeps and the patch gather read) and the bit rows
// out.  One pass over the labels instead of torch's comparison (labels in, mask out) + lg_pack_bits (mask in).  A lane takes 16
// labels (two 16-byte loads), four adjacent lanes assemble a word; W % 16 == 0 (the launcher falls back otherwise).
__global__ __launch_bounds__(256) void lg_pack_labels16_kernel(const int16_t* __restrict__ labels, const int32_t* __restrict__ ids,
                                                               uint8_t* __restrict__ mask, unsigned long long* __restrict__ bits,
                                                               int H, int W, int WW, long long nslots_total) {
    const int slots_per_row = 4 * WW;
    long long sid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long stride = (long long)gridDim.x * 256;  // multiple of 4: a word's four slots stay in adjacent lanes
    for (; sid < nslots_total; sid += stride) {
        const long long row = sid / slots_per_row;
        const int slot = (int)(sid - row * slots_per_row);
        const int x = slot * 16;
        unsigned b16 = 0;
        if (x < W) {
            const int id = ids[row / H];
            const uint4 v0 = *reinterpret_cast<const uint4*>(labels + row * W + x);
            const uint4 v1 = *reinterpret_cast<const uint4*>(labels + row * W + x + 8);
            const uint32_t w8[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            uint32_t mb[4];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const unsigned lo = (int)(int16_t)(w8[q] & 0xffffu) == id ? 1u : 0u, hi = (int)(int16_t)(w8[q] >> 16) == id ? 1u : 0u;
                b16 |= (lo | (hi << 1)) << (2 * q);
                const uint32_t two = lo | (hi << 8);
                if (q & 1) mb[q >> 1] |= two << 16; else mb[q >> 1] = two;
            }
            *reinterpret_cast<uint4*>(mask + row * W + x) = make_uint4(mb[0], mb[1], mb[2], mb[3]);
        }
        unsigned long long word = (unsigned long long)b16 << (16 * (slot & 3));
        word |= __shfl_xor(word, 1, 64);
        word |= __shfl_xor(word, 2, 64);
        if ((slot & 3) == 0) bits[row * WW + (slot >> 2)] = word;
    }
}
__global__ __launch_bounds__(256) void lg_labels_mask_kernel(const int16_t* __restrict__ labels, const int32_t* __restrict__ ids,
                                                             uint8_t* __restrict__ mask, long long px_per_frame, long long total) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256)
        mask[i] = (int)labels[i] == ids[i / px_per_frame] ? 1 : 0;
}
void lg_launch_pack_labels(const int16_t* labels, const int32_t* ids_dev, uint8_t* mask, unsigned long long* bits, int B, int H,
                           int W, int WW, hipStream_t s) {
    if ((W & 15) == 0 && ((uintptr_t)labels & 15) == 0 && ((uintptr_t)mask & 15) == 0) {
        const long long nslots = (long long)B * H * WW * 4;
        long long blocks = (nslots + 255) / 256;
        if (blocks > 16384) blocks = 16384;
        hipLaunchKernelGGL(lg_pack_labels16_kernel, dim3((unsigned)blocks), dim3(256), 0, s, labels, ids_dev, mask, bits, H, W, WW, nslots);
        return;
    }
    const long long total = (long long)B * H * W;
    hipLaunchKernelGGL(lg_labels_mask_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 16384)), dim3(256), 0, s, labels,
                       ids_dev, mask, (long long)H * W, total);
    lg_launch_pack_bits(mask, bits, B, H, W, WW, s);
}

// Export of the bits the host needs -- the rows and 64-bit words of each frame's bounding box only (a leaf spans a third
// of the frame in each direction: ~10x less PCIe traffic than the whole batch) -- by posted 8-byte writes into the pinned host image,
// which keeps its [B][H][WW] layout; rows outside the bounding box are all zero and never read by the host.
__global__ __launch_bounds__(256) void lg_export_rows_kernel(const unsigned long long* __restrict__ bits,
                                                             const LgWin* __restrict__ wins,
                                                             unsigned long long* __restrict__ dst_host, int H, int WW) {
    const int frame = blockIdx.y;
    const LgWin w = wins[frame];
    if (w.bx1 < w.bx0) return;
    const int w0 = w.bx0 >> 6, nw = (w.bx1 >> 6) - w0 + 1;          // words of the bounding box
    const size_t base = ((size_t)frame * H + w.by0) * WW + w0;
    const long long n = (long long)(w.by1 - w.by0 + 1) * nw;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const size_t o = base + (size_t)(i / nw) * WW + (size_t)(i % nw);
        dst_host[o] = bits[o];
    }
}

void lg_launch_export_rows(const unsigned long long* bits, const LgWin* wins, unsigned long long* dst_host_devptr, int B,
                           int H, int WW, hipStream_t s) {
    hipLaunchKernelGGL(lg_export_rows_kernel, dim3(2, B), dim3(256), 0, s, bits, wins, dst_host_devptr, H, WW);
}

// ============================================================================ stem penalty (binary dilation on bit rows)
// stem = dilate(mask & bottom_region, ellipse k) & mask     (grasp_point_selector.py:688-701)
// dilate(x,y) = OR over SE rows i, dx in [lo_i, hi_i] of src(x+dx, y+i-anchor)  (cv2.dilate, anchor k/2)
struct LgU192 {
    unsigned long long w0, w1, w2;  // bit p of the 192-bit row segment: w0 = bits 0..63 (word w-1), w1 = word w, w2 = word w+1
};
__device__ __forceinline__ LgU192 lg_shr192(LgU192 r, int s) {  // 0 < s < 64
    LgU192 o;
    o.w0 = (r.w0 >> s) | (r.w1 << (64 - s));
    o.w1 = (r.w1 >> s) | (r.w2 << (64 - s));
    o.w2 = (r.w2 >> s);
    return o;
}
__device__ __forceinline__ LgU192 lg_or192(LgU192 a, LgU192 b) {
    LgU192 o = {a.w0 | b.w0, a.w1 | b.w1, a.w2 | b.w2};
    return o;
}

__global__ __launch_bounds__(256) void lg_stem_bits_kernel(const unsigned long long* __restrict__ bits,
                                                           unsigned long long* __restrict__ stem, int H, int W, int WW,
                                                           int bottom_start, LgSeSpans se) {
    // one thread per (frame, y, word)
    long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    long long per_frame = (long long)H * WW;
    int b = blockIdx.y;
    if (gid >= per_frame) return;
    int y = (int)(gid / WW), w = (int)(gid % WW);
    const unsigned long long* fb = bits + (long long)b * per_frame;
    unsigned long long self = fb[(long long)y * WW + w];
    unsigned long long acc = 0;
    if (self != 0 && y + (se.n - 1 - se.anchor) >= bottom_start) {
        for (int i = 0; i < se.n; i++) {
            int yy = y + i - se.anchor;
            int lo = se.lo[i], hi = se.hi[i];
            if (lo > hi || yy < bottom_start || yy >= H) continue;
            const unsigned long long* row = fb + (long long)yy * WW;
            LgU192 r;
            r.w0 = (w > 0) ? row[w - 1] : 0ull;
            r.w1 = row[w];
            r.w2 = (w + 1 < WW) ? row[w + 1] : 0ull;
            if (!(r.w0 | r.w1 | r.w2)) continue;
            // T bit p = OR_{j=0..n} R[p+j]
            int n = hi - lo;
            LgU192 t = r;
            int covered = 1;  // t covers shifts [0, covered-1]
            while (covered * 2 <= n + 1) {
                t = lg_or192(t, lg_shr192(t, covered));
                covered *= 2;
            }
            if (covered < n + 1) t = lg_or192(t, lg_shr192(t, n + 1 - covered));
            // out bit b = T bit (64 + b + lo), lo in [-anchor, 0]
            int sh = 64 + lo;  // 64-anchor .. 64
            unsigned long long o = (sh == 64) ? t.w1 : ((t.w0 >> sh) | (t.w1 << (64 - sh)));
            if (lo > 0) o = (t.w1 >> lo) | (t.w2 << (64 - lo));
            acc |= o;
        }
    }
    stem[(long long)b * per_frame + gid] = acc & self;
}

void lg_launch_stem_bits(const unsigned long long* bits, unsigned long long* stem, int B, int H, int W, int WW,
                         int bottom_start, const LgSeSpans& se, hipStream_t s) {
    long long per_frame = (long long)H * WW;
    dim3 grid((unsigned)((per_frame + 255) / 256), B);
    hipLaunchKernelGGL(lg_stem_bits_kernel, grid, dim3(256), 0, s, bits, stem, H, W, WW, bottom_start, se);
}

// ============================================================================ sweep window (mask bounding box)
__device__ __forceinline__ uint32_t lg_norm5(int dx, int dy) {   // closed-form norm of the (1, 1.4, 2.1969) chamfer mask
    const uint32_t a = (uint32_t)max(dx, dy), b = (uint32_t)min(dx, dy);
    return 2u * b <= a ? (a - 2u * b) * LG_A5 + b * LG_C5 : (a - b) * LG_C5 + (2u * b - a) * LG_B5;
}
// One workgroup per frame over the bit rows: bounding box of the set bits -> LgWin (see lg_internal.h).  1024 threads, four per
// row (words q, q + 4, ...: the four read 32 adjacent bytes): a thread's loads are a dependent chain, and with one thread per
// row and 256 threads the kernel took 44 us for ONE frame -- 6 % of a single-frame call.
//
// Search or sweeps is decided for the BATCH (search_mode 2): the search's time grows like the sum over the frames of area^1.5
// (pixels x their depth), the sweeps' like the rows of the tallest window whatever the batch (one workgroup per frame, all at
// once) -- and a batch of which one half is searched while the other half is swept takes as long as both together (measured:
// 1.1 ms per 256 benchmark frames against 0.7 either way; the two forms do not run beside each other as their streams suggest).
// Every workgroup adds its frame to the batch's sums; the last one to finish compares them and, if the sweeps win, clears
// the flags of all frames (and resets the sums for the next launch).
__global__ __launch_bounds__(1024) void lg_bbox_kernel(const unsigned long long* __restrict__ bits, LgWin* __restrict__ wins,
                                                       int H, int W, int WW, int wc, int nw_max, int search_mode, float search_budget,
                                                       LgDtBatch* __restrict__ bt) {
    __shared__ int s_b[5];
    __shared__ int s_last;
    const int frame = blockIdx.x, t = threadIdx.x;
    if (t == 0) { s_b[0] = INT_MAX; s_b[1] = -1; s_b[2] = INT_MAX; s_b[3] = -1; s_b[4] = 0; }
    __syncthreads();
    const unsigned long long* fb = bits + (size_t)frame * H * WW;
    int x0 = INT_MAX, x1 = -1, y0 = INT_MAX, y1 = -1, cnt = 0;
    for (int y = t >> 2; y < H; y += 256) {
        const unsigned long long* row = fb + (size_t)y * WW;
        int first = INT_MAX, last = -1;
        for (int w = t & 3; w < WW; w += 4) {
            const unsigned long long v = row[w];
            if (v) {
                first = min(first, 64 * w + __builtin_ctzll(v));
                last = max(last, 64 * w + 63 - __builtin_clzll(v));
                cnt += __popcll(v);
            }
        }
        if (last >= 0) { x0 = min(x0, first); x1 = max(x1, last); y0 = min(y0, y); y1 = max(y1, y); }
    }
    if (x1 >= 0) {
        atomicMin(&s_b[0], x0); atomicMax(&s_b[1], x1);
        atomicMin(&s_b[2], y0); atomicMax(&s_b[3], y1);
        atomicAdd(&s_b[4], cnt);
    }
    __syncthreads();
    if (t == 0) {
        LgWin w;
        w.bx0 = s_b[0]; w.bx1 = s_b[1]; w.by0 = s_b[2]; w.by1 = s_b[3];
        w.area = s_b[4];
        w.search_in = 0;
        if (w.bx1 < 0) {   // empty mask: no window (d_out has no source: the closed form of the whole frame applies)
            w.bx0 = 0; w.bx1 = -1; w.by0 = 0; w.by1 = -1;
            w.wx0 = 0; w.nw = nw_max; w.wy0 = 0; w.wy1 = H;
            w.skip_out = 0;
        } else {
            // d_in by the row search (lg_dtsearch_kernel) needs a zero pixel in the image (a frame without one has OpenCV's
            // border-initialised result, which only the sweeps produce); its work grows like area^1.5 (pixels x their depth)
            // while the sweeps' time is set by the window's rows: `search_limit` is where the two meet for this batch.
            w.search_in = (search_mode != 0 && (long long)w.area < (long long)H * W) ? 1 : 0;
            if (search_mode == 2 && w.search_in) {
                const float a = (float)w.area;
                atomicAdd(&bt->cost, (unsigned long long)(a * __builtin_sqrtf(a)));
                atomicMax(&bt->rows, (unsigned)(w.by1 - w.by0 + 1));
            }
            w.wx0 = (w.bx0 / LG_TW) * LG_TW;
            w.nw = (w.bx1 + 1 - w.wx0 + wc - 1) / wc;
            w.wy0 = (w.by0 / LG_TH) * LG_TH;
            w.wy1 = min(H, ((w.by1 + 1 + LG_TH - 1) / LG_TH) * LG_TH);
            // Only max d_out is consumed (grasp_point_selector.py:531-533).  Inside the window d_out(p) <= N(p - q) for any leaf
            // pixel q, and both lie in the window: <= N(window width - 1, window height - 1).  At a frame corner every leaf pixel is
            // at least the corner's gap to the bounding box away in x and in y: d_out(corner) >= N(gap_x, gap_y) (N is monotone in
            // both).  When the best corner bound exceeds the window bound, the maximum is the frame-border maximum that
            // lg_dout_border_kernel computes exactly, and the two d_out sweeps of this frame have nothing to add: they are skipped
            // (the usual case: a leaf is a few hundred pixels across, the frame's far corner a thousand away).
            const int ww = min(W, w.wx0 + w.nw * wc) - w.wx0, wh = w.wy1 - w.wy0;
            const uint32_t ub_in = lg_norm5(ww - 1, wh - 1);
            const int gx = max(w.bx0, W - 1 - w.bx1), gy = max(w.by0, H - 1 - w.by1);
            // (strictly larger: the corner that achieves it then lies outside the window, on a border line lg_dout_border_kernel walks)
            w.skip_out = lg_norm5(gx, gy) > ub_in ? 1 : 0;
        }
        w.pad_[0] = 0;
        wins[frame] = w;
        s_last = 0;
        if (search_mode == 2) {
            __threadfence();
            s_last = atomicAdd(&bt->done, 1u) == gridDim.x - 1 ? 1 : 0;
        }
    }
    __syncthreads();
    if (s_last) {   // (one workgroup of the launch; every other one has published its frame and its sums)
        __threadfence();
        const unsigned long long cost = atomicAdd(&bt->cost, 0ull);
        const unsigned rows = atomicMax(&bt->rows, 0u);
        if ((float)cost > search_budget * (float)rows)
            for (int f = t; f < (int)gridDim.x; f += 1024) wins[f].search_in = 0;
        if (t == 0) { bt->cost = 0; bt->rows = 0; bt->done = 0; }
    }
}

void lg_launch_bbox(const unsigned long long* bits, LgWin* win, int B, int H, int W, int WW, int search_mode, LgDtBatch* batch,
                    hipStream_t s) {
    int nw = 0;
    const int wc = lg_dt_geometry(W, &nw);
    // mode 2: search while sum over the frames of area^1.5 <= LG_SEARCH_BUDGET * rows of the tallest window: 0.69 ms for 256
    // benchmark leaves of 100 k pixels in either form (anchors + bands; sweeps: ~1.6 us per row).  LG_DT_SEARCH_BUDGET=<x>
    // replaces the constant (experiments).
    static const float env_budget = getenv("LG_DT_SEARCH_BUDGET") ? (float)atof(getenv("LG_DT_SEARCH_BUDGET")) : 0.0f;
    hipLaunchKernelGGL(lg_bbox_kernel, dim3(B), dim3(1024), 0, s, bits, win, H, W, WW, wc, nw, search_mode,
                       env_budget > 0.0f ? env_budget : LG_SEARCH_BUDGET, batch);
}

// ============================================================================ max d_out outside the sweep window
// d_out(p) = min over leaf pixels q of the 5x5 chamfer norm N(p - q) (see LgWin).  Outside the window, moving p away
// from the leaf's bounding box along x or y increases |dx| (|dy|) for EVERY leaf pixel, and N is monotone in |dx|, |dy|:
// the maximum over the outside region sits on the frame border, and beyond the bounding box's span at a frame corner.
// grid (4, B): side 0 = top row, 1 = bottom row, 2 = left column, 3 = right column.  For a candidate on the top row
// only the topmost leaf pixel of each column can be the nearest one of that column (same dx, smallest dy), etc.
__global__ __launch_bounds__(256) void lg_dout_border_kernel(const unsigned long long* __restrict__ bits,
                                                             const LgWin* __restrict__ wins, uint32_t* __restrict__ maxfix,
                                                             int H, int W, int WW, int wc) {
    extern __shared__ int s_prof[];   // profile of the leaf seen from this side: distance from the border, -1 = no leaf pixel
    const int side = blockIdx.x, frame = blockIdx.y, t = threadIdx.x;
    const LgWin w = wins[frame];
    if (w.bx1 < w.bx0) return;                       // empty mask: handled by the full-frame sweep
    const int wxr = min(W, w.wx0 + w.nw * wc);
    const bool need = side == 0 ? w.wy0 > 0 : side == 1 ? w.wy1 < H : side == 2 ? w.wx0 > 0 : wxr < W;
    if (!need) return;                               // this border line lies inside the window: the sweep covers it
    const unsigned long long* fb = bits + (size_t)frame * H * WW;
    const bool horiz = side < 2;                     // candidates along x, profile indexed by column
    const int lo = horiz ? w.bx0 : w.by0, hi = horiz ? w.bx1 : w.by1, n = hi - lo + 1;
    if (horiz) {
        // thread q owns one 64-column word of the bounding box and walks the rows from the border inwards, 8 rows per
        // round trip; a column's profile entry is the row at which its bit is first seen
        for (int i = t; i < n; i += 256) s_prof[i] = -1;
        __syncthreads();
        const int q = (w.bx0 >> 6) + t;
        if (q <= (w.bx1 >> 6)) {
            unsigned long long seen = 0;
            const int rows = w.by1 - w.by0 + 1;
            for (int r0 = 0; r0 < rows; r0 += 8) {
                unsigned long long v[8];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int r = min(r0 + j, rows - 1);
                    const int y = side == 0 ? w.by0 + r : w.by1 - r;
                    v[j] = fb[(size_t)y * WW + q];
                }
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    unsigned long long nb = v[j] & ~seen;
                    seen |= v[j];
                    const int r = min(r0 + j, rows - 1);
                    const int dist = side == 0 ? w.by0 + r : H - 1 - (w.by1 - r);
                    while (nb) {
                        const int c = 64 * q + __builtin_ctzll(nb);
                        nb &= nb - 1;
                        s_prof[c - lo] = dist;
                    }
                }
            }
        }
    } else {
        for (int i = t; i < n; i += 256) {
            int d = -1;
            const unsigned long long* row = fb + (size_t)(lo + i) * WW;
            if (side == 2) {
                for (int q = w.bx0 >> 6; q <= w.bx1 >> 6; q++) if (row[q]) { d = 64 * q + __builtin_ctzll(row[q]); break; }
            } else {
                for (int q = w.bx1 >> 6; q >= w.bx0 >> 6; q--) if (row[q]) { d = W - 1 - (64 * q + 63 - __builtin_clzll(row[q])); break; }
            }
            s_prof[i] = d;
        }
    }
    __syncthreads();
    // candidates: the span of the bounding box plus the two ends of the line
    const int len = horiz ? W : H;
    uint32_t best = 0;
    for (int ci = t; ci < n + 2; ci += 256) {
        const int p = ci < n ? lo + ci : (ci == n ? 0 : len - 1);
        uint32_t dmin = 0xFFFFFFFFu;
        for (int i = 0; i < n; i++) {
            const int d = s_prof[i];
            if (d >= 0) dmin = min(dmin, lg_norm5(abs(p - (lo + i)), d));
        }
        best = max(best, dmin);
    }
    best = lg_wave_max_u32(best);
    if ((t & 63) == 0 && best) atomicMax(&maxfix[frame * 2 + 1], best);
}

void lg_launch_dout_border(const unsigned long long* bits, const LgWin* win, uint32_t* maxfix, int B, int H, int W, int WW,
                           hipStream_t s) {
    const int wc = lg_dt_geometry(W, nullptr);
    const size_t shm = sizeof(int) * (size_t)max(H, W);
    hipLaunchKernelGGL(lg_dout_border_kernel, dim3(4, B), dim3(256), shm, s, bits, win, maxfix, H, W, WW, wc);
}

// ============================================================================ chamfer 5x5 distance transform
// cv2.distanceTransform(src, DIST_L2, 5)  (grasp_point_selector.py:266, :529-530).
// Row-sequential emulation of the two raster sweeps; each image row is one segmented-free min-plus
// prefix scan across the workgroup (tmp[j] = min(u[j], tmp[j-1]+A)  ==  j*A + prefix-min(u[k]-k*A)).
// All arithmetic is exact 16.16 integers, so any evaluation order gives OpenCV's integers.
// grid = (2, B): blockIdx.x selects d_in (src = mask) or d_out (src = !mask).
// The backward sweep is the same code on the 180-degree-rotated image.
template <int T, int E, bool BWD, bool VEC>
__global__ __launch_bounds__(T) void lg_dt5_kernel(const uint8_t* __restrict__ mask, uint32_t* __restrict__ tmp,
                                                   float* __restrict__ dist_out, uint32_t* __restrict__ maxfix,
                                                   const LgWin* __restrict__ wins, int H, int W, uint32_t init0) {
    constexpr int NW = T / 64;
    constexpr int WC = 64 * E;                 // columns per wave
    constexpr int D = (BWD || E > 4) ? 4 : 8;  // rows per prefetch group (two groups are resident)
    // per row and wave: [0] inclusive wave total, [1..2] lane 0's first two pre-carry values,
    // [3..4] lane 63's last two pre-carry values, [5] lane 63's wave-local exclusive prefix
    __shared__ int s_x[2][NW][8];
    const int which = blockIdx.x;
    const int frame = blockIdx.y;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // ---- sweep window (see LgWin): waves beyond it leave; s_barrier only counts the surviving waves
    const LgWin win = wins[frame];
    const int nwa = win.nw;
    if (wave >= nwa) return;
    if (which == 1 && win.skip_out) return;   // max d_out lies on the frame border (lg_bbox_kernel): whole workgroup, before any barrier
    if (which == 0 && win.search_in) return;  // d_in of this frame comes from lg_dtsearch_kernel
    const int wx0 = win.wx0, wxe = wx0 + nwa * WC;   // window columns [wx0, wxe); columns >= W are outside the image
    const int wy0 = win.wy0, HW = win.wy1 - win.wy0; // window rows
    const size_t fo = (size_t)frame * H * W;
    const uint8_t* m = mask + fo;
    uint32_t* tp = tmp + ((size_t)frame * 2 + which) * H * W;
    float* dout = (dist_out && which == 0) ? dist_out + fo : nullptr;  // only d_in is a contract plane
    const int pc0 = BWD ? wxe - (t + 1) * E : wx0 + t * E;  // first physical column of this thread (multiple of E)
    constexpr bool vec_ok = VEC;  // host guarantees W % E == 0 when VEC
    const bool full = vec_ok && (pc0 + E <= W);
    const int c0 = t * E;  // first logical column (0 = first window column in sweep order)
    // value of a cell OUTSIDE the window: for d_in an in-image cell there is off the leaf, i.e. a source (0);
    // for d_out, and outside the image, "no path".
    auto outv = [&](int pc, int prow) -> uint32_t {
        return (which == 0 && pc >= 0 && pc < W && prow >= 0 && prow < H) ? 0u : LG_INF;
    };
    auto phys_col = [&](int j) { return BWD ? wxe - 1 - j : wx0 + j; };          // logical -> physical column
    auto phys_row = [&](int r) { return BWD ? win.wy1 - 1 - r : wy0 + r; };       // logical -> physical row

    uint32_t p1[E + 4], p2[E + 2];   // previous row (columns c0-2 .. c0+E+1) and the one before (c0-1 .. c0+E)
#pragma unroll
    for (int i = 0; i < E + 4; i++) p1[i] = outv(phys_col(c0 - 2 + i), phys_row(-1));
#pragma unroll
    for (int i = 0; i < E + 2; i++) p2[i] = outv(phys_col(c0 - 1 + i), phys_row(-2));
    // the cell left of the window in the current row (a source for d_in when it lies in the image)
    const uint32_t edge_l1 = outv(phys_col(-1), wy0), edge_l2 = outv(phys_col(-2), wy0);
    const uint32_t edge_r0 = outv(phys_col(nwa * WC), wy0), edge_r1 = outv(phys_col(nwa * WC + 1), wy0);

    // raw prefetch registers: forward = E mask bytes (packed in up to 2 dwords), backward = E dwords
    constexpr int RAWN = BWD ? E : (E + 3) / 4;
    uint32_t raw[D][RAWN], nxt[D][RAWN];  // group being consumed / group in flight

    // Branch-free prefetch: addresses are clamped into the image and out-of-image lanes are fixed up by
    // selects, so the compiler keeps the loads in flight (a load inside a divergent branch is waited for
    // with vmcnt(0) at the join, which serialises every image row on HBM latency).
    const int pc0c = full ? pc0 : (vec_ok ? max(0, min(pc0, W - E)) : 0);
    auto load_row = [&](int r, uint32_t* dst) {  // r = logical row
        const int prow = phys_row(min(r, HW - 1));
        if (vec_ok) {   // raw values only; out-of-image lanes are fixed up at the point of use
            if (BWD) {
                const uint32_t* src = tp + (size_t)prow * W + pc0c;
#pragma unroll
                for (int q = 0; q < E / 4; q++) {
                    uint4 v = *reinterpret_cast<const uint4*>(src + 4 * q);
                    dst[4 * q + 0] = v.x; dst[4 * q + 1] = v.y; dst[4 * q + 2] = v.z; dst[4 * q + 3] = v.w;
                }
            } else {
                const uint8_t* src = m + (size_t)prow * W + pc0c;
                if (E == 4) {
                    dst[0] = *reinterpret_cast<const uint32_t*>(src);
                } else {
                    uint2 v = *reinterpret_cast<const uint2*>(src);
                    dst[0] = v.x; dst[1] = v.y;
                }
            }
        } else {  // generic width: scalar, still branch-free
            if (BWD) {
#pragma unroll
                for (int k = 0; k < E; k++) {
                    const int pc = min(pc0 + k, W - 1);
                    uint32_t v = tp[(size_t)prow * W + pc];
                    dst[k] = (pc0 + k < W) ? v : LG_INF;
                }
            } else {
#pragma unroll
                for (int q = 0; q < RAWN; q++) dst[q] = 0;
#pragma unroll
                for (int k = 0; k < E; k++) {
                    const int pc = min(pc0 + k, W - 1);
                    uint32_t mb = m[(size_t)prow * W + pc];
                    mb = (pc0 + k < W) ? mb : (which ? 0u : 1u);
                    dst[k >> 2] |= (mb & 0xffu) << (8 * (k & 3));
                }
            }
        }
    };

    uint32_t mx = 0;
#pragma unroll
    for (int d = 0; d < D; d++) load_row(d, raw[d]);

    for (int rb = 0; rb < HW; rb += D) {
        // The whole next group is requested up front, so at the next iteration (where the compiler waits for
        // everything outstanding at the loop head) the youngest load is D rows old, not one.
#pragma unroll
        for (int d = 0; d < D; d++) load_row(rb + D + d, nxt[d]);
#pragma unroll
        for (int d = 0; d < D; d++) {
            const int r = rb + d;
            if (r < HW) {
                const int par = r & 1;
                // ---- per-pixel upper bound ("init"): forward: source ? 0 : +inf ; backward: forward result
                uint32_t init[E];
#pragma unroll
                for (int k = 0; k < E; k++) {
                    if (BWD) {
                        init[k] = (vec_ok && !full) ? LG_INF : raw[d][E - 1 - k];
                    } else {
                        uint32_t mb = (raw[d][k >> 2] >> (8 * (k & 3))) & 0xffu;
                        bool nz = which ? (mb == 0) : (mb != 0);
                        if (vec_ok && !full) nz = true;  // out-of-image columns: ordinary non-source pixels
                        init[k] = nz ? 0xFFFFFFFFu : 0u;
                    }
                }
                // ---- contributions of the two previous rows
                uint32_t v[E];
#pragma unroll
                for (int k = 0; k < E; k++) {
                    // group equal weights before adding them: min(x+w, y+w) == min(x,y)+w (no overflow, values < 2^31)
                    const uint32_t mc = min(min(p2[k], p2[k + 2]), min(p1[k], p1[k + 4])) + LG_C5;
                    const uint32_t mb = min(p1[k + 1], p1[k + 3]) + LG_B5;
                    const uint32_t ma = p1[k + 2] + LG_A5;
                    v[k] = min(min(mc, mb), min(ma, init[k]));
                }
                // ---- same-row chain: thread-local, then across the workgroup.  The cell left of the window enters through
                //      thread 0's first pixel, so the prefix scan carries it along the whole row.
                if (t == 0) v[0] = min(v[0], edge_l1 + LG_A5);
#pragma unroll
                for (int k = 1; k < E; k++) v[k] = min(v[k], v[k - 1] + LG_A5);
                int mloc = (int)v[E - 1] - (int)((uint32_t)(c0 + E - 1) * LG_A5);
                int pin = lg_wave_prefix_min(mloc);
                int excl = lg_wave_shr1(INT_MAX, pin);   // wave-local exclusive prefix
                // One LDS exchange per row: everything a neighbouring wave needs to rebuild this wave's border
                // pixels is published BEFORE the barrier (pre-carry values + prefixes); see "rotate" below.
                if (lane == 0) { s_x[par][wave][1] = (int)v[0]; s_x[par][wave][2] = (int)v[1]; }
                if (lane == 63) {
                    s_x[par][wave][0] = pin;
                    s_x[par][wave][3] = (int)v[E - 2]; s_x[par][wave][4] = (int)v[E - 1];
                    s_x[par][wave][5] = excl;
                }
                lg_lds_barrier();
                int wpre = INT_MAX, wpre_prev = INT_MAX;  // min of the totals of waves < wave, and < wave-1
#pragma unroll
                for (int w = 0; w < NW - 1; w++) {
                    const int tw = s_x[par][w][0];
                    if (w < wave) wpre = min(wpre, tw);
                    if (w < wave - 1) wpre_prev = min(wpre_prev, tw);
                }
                excl = min(excl, wpre);
                uint32_t cin = (t == 0) ? LG_INF : (uint32_t)(excl + (int)((uint32_t)(c0 - 1) * LG_A5));
#pragma unroll
                for (int k = 0; k < E; k++) v[k] = min(v[k], cin + (uint32_t)(k + 1) * LG_A5);
                // ---- write back
                const int prow = phys_row(r);
                if (!BWD) {
                    uint32_t* dst = tp + (size_t)prow * W + pc0;
                    if (full) {
#pragma unroll
                        for (int q = 0; q < E / 4; q++)
                            *reinterpret_cast<uint4*>(dst + 4 * q) =
                                make_uint4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
                    } else {
#pragma unroll
                        for (int k = 0; k < E; k++)
                            if (pc0 + k < W) dst[k] = v[k];
                    }
                } else {
                    float o[E];
#pragma unroll
                    for (int k = 0; k < E; k++) {
                        const int pc = pc0 + (E - 1 - k);
                        uint32_t val = v[k];
                        if (val >= LG_NOSRC) {  // image without any source pixel: OpenCV's border-initialised result
                            int dd = min(min(pc + 1, W - pc), min(prow + 1, H - prow));
                            val = init0 + (uint32_t)dd * LG_A5;
                        }
                        if (pc < W) mx = max(mx, val);
                        o[E - 1 - k] = (float)val * (1.0f / 65536.0f);
                    }
                    if (dout) {
                        float* dst = dout + (size_t)prow * W + pc0;
                        if (full) {
#pragma unroll
                            for (int q = 0; q < E / 4; q++)
                                *reinterpret_cast<float4*>(dst + 4 * q) =
                                    make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
                        } else {
#pragma unroll
                            for (int k = 0; k < E; k++)
                                if (pc0 + k < W) dst[k] = o[k];
                        }
                    }
                }
                // ---- rotate row registers, exchange 2-column halos with the neighbouring threads
#pragma unroll
                for (int j = 0; j < E + 2; j++) p2[j] = p1[j + 1];
                uint32_t l0 = (uint32_t)lg_wave_shr1((int)LG_INF, (int)v[E - 2]);
                uint32_t l1 = (uint32_t)lg_wave_shr1((int)LG_INF, (int)v[E - 1]);
                uint32_t r0 = (uint32_t)lg_wave_shl1((int)LG_INF, (int)v[0]);
                uint32_t r1 = (uint32_t)lg_wave_shl1((int)LG_INF, (int)v[1]);
                if (lane == 0 && wave > 0) {
                    // final values of thread t-1 (lane 63 of the previous wave), rebuilt from what it published
                    const int exl = min(s_x[par][wave - 1][5], wpre_prev);
                    const uint32_t cprev = (uint32_t)(exl + (int)((uint32_t)(c0 - E - 1) * LG_A5));
                    l0 = min((uint32_t)s_x[par][wave - 1][3], cprev + (uint32_t)(E - 1) * LG_A5);
                    l1 = min((uint32_t)s_x[par][wave - 1][4], cprev + (uint32_t)E * LG_A5);
                }
                if (t == 0) { l0 = edge_l2; l1 = edge_l1; }
                if (lane == 63 && wave == nwa - 1) { r0 = edge_r0; r1 = edge_r1; }
                if (lane == 63 && wave < nwa - 1) {
                    // final values of thread t+1 (lane 0 of the next wave): its carry is the prefix over all threads <= t
                    const int inc = min(s_x[par][wave][0], wpre);
                    const uint32_t cnext = (uint32_t)(inc + (int)((uint32_t)(c0 + E - 1) * LG_A5));
                    r0 = min((uint32_t)s_x[par][wave + 1][1], cnext + LG_A5);
                    r1 = min((uint32_t)s_x[par][wave + 1][2], cnext + 2u * LG_A5);
                }
                p1[0] = l0; p1[1] = l1;
#pragma unroll
                for (int k = 0; k < E; k++) p1[2 + k] = v[k];
                p1[E + 2] = r0; p1[E + 3] = r1;
            }
        }
#pragma unroll
        for (int d = 0; d < D; d++)
#pragma unroll
            for (int q = 0; q < RAWN; q++) raw[d][q] = nxt[d][q];
    }
    if (BWD) {
        mx = lg_wave_max_u32(mx);
        if (lane == 0) atomicMax(&maxfix[frame * 2 + which], mx);
    }
}

template <int T, int E>
static void lg_dt_launch_t(bool bwd, const uint8_t* mask, uint32_t* tmp, float* dist_out, uint32_t* maxfix,
                           const LgWin* win, int B, int H, int W, uint32_t init0, hipStream_t s) {
    dim3 grid(2, B), block(T);
    const bool vec = (W % E) == 0 && W >= E;
    if (bwd) {
        if (vec) hipLaunchKernelGGL((lg_dt5_kernel<T, E, true, true>), grid, block, 0, s, mask, tmp, dist_out, maxfix, win, H, W, init0);
        else hipLaunchKernelGGL((lg_dt5_kernel<T, E, true, false>), grid, block, 0, s, mask, tmp, dist_out, maxfix, win, H, W, init0);
    } else {
        if (vec) hipLaunchKernelGGL((lg_dt5_kernel<T, E, false, true>), grid, block, 0, s, mask, tmp, dist_out, maxfix, win, H, W, init0);
        else hipLaunchKernelGGL((lg_dt5_kernel<T, E, false, false>), grid, block, 0, s, mask, tmp, dist_out, maxfix, win, H, W, init0);
    }
}

// threads x columns-per-thread of the sweep workgroup for width W: returns 64 * E, *waves = T / 64 (0: unsupported)
int lg_dt_geometry(int W, int* waves) {
    static const int force_e = getenv("LG_DT_E") ? atoi(getenv("LG_DT_E")) : 0;
    int T, E;
    if (force_e == 8) {
        E = 8;
        T = W <= 512 ? 64 : W <= 1024 ? 128 : W <= 2048 ? 256 : W <= 4096 ? 512 : W <= 8192 ? 1024 : 0;
    } else if (W <= 2048) {
        E = 4;
        T = W <= 256 ? 64 : W <= 512 ? 128 : W <= 1024 ? 256 : 512;
    } else {
        E = 8;
        T = W <= 4096 ? 512 : W <= 8192 ? 1024 : 0;
    }
    if (waves) *waves = T / 64;
    return T ? 64 * E : 0;
}

int lg_launch_dt(bool bwd, const uint8_t* mask, uint32_t* tmp, float* dist_out, uint32_t* maxfix, const LgWin* win, int B,
                 int H, int W, uint32_t init0, hipStream_t s) {
    // threads * E columns must cover the row.  E = 4 keeps the per-row dependency chain short (best up to
    // 2048 columns: 0.70/0.82 ms vs 0.82/0.91 ms at 1080p); at 4K 512x8 beats 1024x4 (2.04/2.54 vs 2.65/3.12 ms).
    // LG_DT_E=8 forces 8 columns per thread (experiments).
    static const int force_e = getenv("LG_DT_E") ? atoi(getenv("LG_DT_E")) : 0;
    if (force_e == 8) {
        if (W <= 512) lg_dt_launch_t<64, 8>(bwd, mask, tmp, dist_out, maxfix, win, B, H, W, init0, s);
        else if (W <= 1024) lg_dt_launch_t<128, 8>(bwd, mask, tmp, dist_out, maxfix, win, B, H, W, init0, s);
        else if (W <= 2048) lg_dt_launch_t<256, 8>(bwd, mask, tmp, dist_out, maxfix, win, B, H, W, init0, s);
        else if (W <= 4096) lg_dt_launch_t<512, 8>(bwd, mask, tmp, dist_out, maxfix, win, B, H, W, init0, s);
        else if (W <= 8192) lg_dt_launch_t<1024, 8>(bwd, mask, tmp, dist_out, maxfix, win, B, H, W, init0, s);
        else return -1;
        return 0;
    }
    if (W <= 256) lg_dt_launch_t<64, 4>(bwd, mask, tmp, dist_out, maxfix, win, B, H, W, init0, s);
    else if (W <= 512) lg_dt_launch_t<128, 4>(bwd, mask, tmp, dist_out, maxfix, win, B, H, W, init0, s);
    else if (W <= 1024) lg_dt_launch_t<256, 4>(bwd, mask, tmp, dist_out, maxfix, win, B, H, W, init0, s);
    else if (W <= 2048) lg_dt_launch_t<512, 4>(bwd, mask, tmp, dist_out, maxfix, win, B, H, W, init0, s);
    else if (W <= 4096) lg_dt_launch_t<512, 8>(bwd, mask, tmp, dist_out, maxfix, win, B, H, W, init0, s);   // 4K: 8 waves beat 16 (measured)
    else if (W <= 8192) lg_dt_launch_t<1024, 8>(bwd, mask, tmp, dist_out, maxfix, win, B, H, W, init0, s);
    else return -1;
    return 0;
}

// ============================================================================ d_in without the sweeps: run distances + row search
// The two raster passes of the 5x5 chamfer transform give  d(p) = min over zero pixels q of N(p - q)  with the closed-form
// chamfer norm N (LgWin; pinned in tests/test_oracle_c.py), and N is monotone in |dx| and in |dy|.  For a fixed row y' only the
// zero pixel of that row nearest to column x can matter:
//     d(x, y) = min over rows y' of N(h[y'][x], |y - y'|),   h[y'][x] = distance from x to the nearest zero pixel of row y'
// h comes straight from the bit rows (count-leading/trailing-zeros, no sequential dependence: lg_hrun_kernel); the search over
// rows is bounded by N(., dy) >= a * dy: rows further away than the best value found so far cannot improve it
// (lg_dtsearch_kernel).  Every leaf pixel is independent -- the work spreads over all CUs instead of one workgroup per frame
// walking the window's rows one after the other (~1 us per row).  Rows outside [by0 - 1, by1 + 1] never matter: the rows next
// to the bounding box are all zero (h = 0) and nearer than any row beyond them.  Out-of-image pixels are not sources.
// Both kernels: workgroup ids are dealt round-robin to the 8 XCDs, and a frame's workgroups all carry the same id % 8, so a
// frame's run distances are written and read through ONE XCD's L2.  grid = 8 * G * ceil(B / 8).
__device__ __forceinline__ bool lg_frame_of_block(int G, int B, int* frame, int* j) {
    const int id = (int)blockIdx.x, q = id >> 3;
    const int fq = q / G;
    *frame = (id & 7) + 8 * fq;
    *j = q - fq * G;
    return *frame < B;
}

__device__ __forceinline__ unsigned long long lg_readlane_u64(unsigned long long v, int l) {
    return ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) |
           (uint32_t)__builtin_amdgcn_readlane((int)v, l);
}
// One wave per row of [by0 - 1, by1 + 1]: lane k loads word k of the bounding box (one coalesced load, nothing else is read:
// in-image words outside the bounding box are all zero pixels), the nearest word with a zero pixel on either side of a word
// comes from a ballot, and the row's words are then written one after the other, lane = pixel.  Rows wider than 64 words
// (W > 4096 and a leaf spanning them) are processed in 64-word pieces whose neighbours are looked up in memory.
__global__ __launch_bounds__(256) void lg_hrun_kernel(const unsigned long long* __restrict__ bits, const LgWin* __restrict__ wins,
                                                      uint32_t* __restrict__ tmp, int H, int W, int WW, int G, int B) {
    int frame, j;
    if (!lg_frame_of_block(G, B, &frame, &j)) return;
    const LgWin w = wins[frame];
    if (!w.search_in) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r0 = max(w.by0 - 1, 0), r1 = min(w.by1 + 1, H - 1);
    const int w0 = w.bx0 >> 6, w1 = w.bx1 >> 6;
    const unsigned long long* fb = bits + (size_t)frame * H * WW;
    uint16_t* hd = reinterpret_cast<uint16_t*>(tmp + (size_t)frame * 2 * H * W);
    const int tail = W & 63;   // the last word's bits beyond the image are neither leaf nor source
    for (int y = r0 + j * 4 + wave; y <= r1; y += 4 * G) {
        const unsigned long long* row = fb + (size_t)y * WW;
        auto zeros = [&](int k) -> unsigned long long {   // zero pixels of word k (k in [0, WW))
            unsigned long long z = ~row[k];
            if (k == WW - 1 && tail) z &= (1ull << tail) - 1ull;
            return z;
        };
        for (int c0 = w0; c0 <= w1; c0 += 64) {
            const int nc = min(64, w1 - c0 + 1);
            const unsigned long long z = lane < nc ? zeros(c0 + lane) : 0ull;
            const unsigned long long nz = __ballot(z != 0ull);
            // nearest zero pixel left of the piece's first pixel / right of its last one, as distances from those pixels
            unsigned edge_l = LG_HCAP, edge_r = LG_HCAP;
            for (int k = c0 - 1; k >= 0; k--) {
                const unsigned long long zk = zeros(k);
                if (zk) { edge_l = 64u * (unsigned)(c0 - 1 - k) + (unsigned)__builtin_clzll(zk) + 1u; break; }
            }
            for (int k = c0 + nc; k < WW; k++) {
                const unsigned long long zk = zeros(k);
                if (zk) { edge_r = 64u * (unsigned)(k - c0 - nc) + (unsigned)__builtin_ctzll(zk) + 1u; break; }
            }
            for (int i = 0; i < nc; i++) {
                const unsigned long long zi = lg_readlane_u64(z, i);
                const unsigned long long ml = i ? nz & ((1ull << i) - 1ull) : 0ull, mr = i < 63 ? nz >> (i + 1) : 0ull;
                unsigned lc, rc;
                if (ml) {
                    const int k = 63 - __builtin_clzll(ml);
                    lc = 64u * (unsigned)(i - 1 - k) + (unsigned)__builtin_clzll(lg_readlane_u64(z, k)) + 1u;
                } else {
                    lc = min(LG_HCAP, 64u * (unsigned)i + edge_l);
                }
                if (mr) {
                    const int k = i + 1 + __builtin_ctzll(mr);
                    rc = 64u * (unsigned)(k - i - 1) + (unsigned)__builtin_ctzll(lg_readlane_u64(z, k)) + 1u;
                } else {
                    rc = min(LG_HCAP, 64u * (unsigned)(nc - 1 - i) + edge_r);
                }
                const unsigned long long zl = zi << (63 - lane);   // pixels <= lane, the pixel itself in bit 63
                const unsigned long long zr = zi >> lane;          // pixels >= lane, the pixel itself in bit 0
                const unsigned dl = zl ? (unsigned)__builtin_clzll(zl) : (unsigned)lane + lc;
                const unsigned dr = zr ? (unsigned)__builtin_ctzll(zr) : (unsigned)(63 - lane) + rc;
                const int x = 64 * (c0 + i) + lane;
                if (x < W) hd[(unsigned)(y * W) + (unsigned)x] = (uint16_t)min(min(dl, dr), LG_HCAP);
            }
        }
    }
}

// N(h, dy) as the maximum of its four linear pieces (a norm is the maximum of its supporting functionals; for this mask
// 2a <= c <= a + b makes them a*M + (c-2a)*m and (c-b)*M + (2b-c)*m with M, m the larger / smaller of the two arguments):
// dy is wave-uniform, so its four products live in scalar registers and a candidate costs four 24-bit multiply-adds.
#define LG_N5_AL (LG_C5 - 2u * LG_A5)   // 12904
#define LG_N5_BE (LG_C5 - LG_B5)        // 52226
#define LG_N5_GA (2u * LG_B5 - LG_C5)   // 39524
struct LgH4 { uint32_t a, al, be, ga; };   // h times the four coefficients: shared by the lane's four pixels
__device__ __forceinline__ LgH4 lg_h4(uint32_t h) {   // h <= LG_HCAP
    LgH4 r = {h << 16, (uint32_t)__umul24(h, LG_N5_AL), (uint32_t)__umul24(h, LG_N5_BE), (uint32_t)__umul24(h, LG_N5_GA)};
    // (keeps the three products: hipcc otherwise re-fuses them into one multiply-add per pixel whose scalar addend costs a move)
    asm volatile("" : "+v"(r.al), "+v"(r.be), "+v"(r.ga));
    return r;
}
__device__ __forceinline__ uint32_t lg_norm5_h(const LgH4& h, uint32_t dy) {   // dy <= 16384 (wave-uniform): below 2^32
    const uint32_t t0 = h.a + __umul24(dy, LG_N5_AL);
    const uint32_t t1 = h.al + (dy << 16);
    const uint32_t t2 = h.be + __umul24(dy, LG_N5_GA);
    const uint32_t t3 = h.ga + __umul24(dy, LG_N5_BE);
    return max(max(t0, t1), max(t2, t3));
}

// Workgroup = one 64 x 16 tile of the window at a time (the score-plane kernel's tiles), wave = 4 rows, lane = column: a
// candidate row's h is loaded once (128 contiguous bytes per wave) and serves the lane's four pixels.
__global__ __launch_bounds__(256) void lg_dtsearch_kernel(const unsigned long long* __restrict__ bits,
                                                          const LgWin* __restrict__ wins, const uint32_t* __restrict__ tmp,
                                                          float* __restrict__ dist_out, uint32_t* __restrict__ maxfix, int H,
                                                          int W, int WW, int wc, int G, int B) {
    int frame, j;
    if (!lg_frame_of_block(G, B, &frame, &j)) return;
    const LgWin w = wins[frame];
    if (!w.search_in) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wxe = min(W, w.wx0 + w.nw * wc);
    const int ntx = (wxe - w.wx0 + 63) >> 6, nty = (w.wy1 - w.wy0 + LG_TH - 1) / LG_TH;
    const int lo = max(w.by0 - 1, 0), hi = min(w.by1 + 1, H - 1);   // candidate rows
    const unsigned long long* fb = bits + (size_t)frame * H * WW;
    const uint16_t* hd = reinterpret_cast<const uint16_t*>(tmp + (size_t)frame * 2 * H * W);
    float* dout = dist_out + (size_t)frame * H * W;
    uint32_t mx = 0;
    static_assert(LG_TH == 16, "four waves x four rows");
    const int ntile = __builtin_amdgcn_readfirstlane(ntx * nty);
    for (int tile = j; tile < ntile; tile += G) {
        const int tyi = __builtin_amdgcn_readfirstlane(tile / ntx), txi = tile - tyi * ntx;
        const int wi = (w.wx0 >> 6) + txi;
        const int x = 64 * wi + lane, yb = w.wy0 + LG_TH * tyi + 4 * wave;
        const bool xin = x < W;
        const unsigned xc = (unsigned)min(x, W - 1);
        unsigned long long rb[4];   // the tile is word aligned: one word per row, the same for every lane
#pragma unroll
        for (int i = 0; i < 4; i++) rb[i] = fb[(unsigned)(min(yb + i, H - 1) * WW + wi)];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            rb[i] = yb + i < H ? rb[i] : 0ull;
            rb[i] = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(rb[i] >> 32)) << 32) |
                    (uint32_t)__builtin_amdgcn_readfirstlane((int)rb[i]);
        }
        if (!(rb[0] | rb[1] | rb[2] | rb[3])) {   // no leaf pixel in these four rows: d_in = 0
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (xin && yb + i < H) dout[(unsigned)((yb + i) * W) + xc] = 0.0f;
            continue;
        }
        uint32_t best[4];
#pragma unroll
        for (int i = 0; i < 4; i++) best[i] = (xin && ((rb[i] >> lane) & 1ull)) ? 0xFFFFFFFFu : 0u;
        // the four rows themselves (a leaf pixel's own row always lies in [lo, hi]; rows beyond have no leaf pixel and, as
        // candidates, lose to the all-zero rows lo / hi)
        {
            uint32_t h4[4];
#pragma unroll
            for (int i2 = 0; i2 < 4; i2++) h4[i2] = hd[(unsigned)(max(lo, min(yb + i2, hi)) * W) + xc];
#pragma unroll
            for (int i2 = 0; i2 < 4; i2++) {
                if (yb + i2 >= lo && yb + i2 <= hi) {
                    const LgH4 hh = lg_h4(h4[i2]);
#pragma unroll
                    for (int i = 0; i < 4; i++) best[i] = min(best[i], lg_norm5_h(hh, (uint32_t)(i > i2 ? i - i2 : i2 - i)));
                }
            }
        }
        // rows above (yb - k) and below (yb + 3 + k), four steps per round so that eight loads are in flight
        for (int k0 = 1;; k0 += 4) {
            const int yu0 = yb - k0, yd0 = yb + 3 + k0;
            if (yu0 < lo && yd0 > hi) break;
            const uint32_t bm = max(max(best[0], best[1]), max(best[2], best[3]));
            if (!__any((uint32_t)k0 * LG_A5 < bm)) break;   // N(., dy) >= a * dy >= a * k0 from here on
            uint32_t hu[4], hv[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                hu[q] = hd[(unsigned)(max(yu0 - q, lo) * W) + xc];
                hv[q] = hd[(unsigned)(min(yd0 + q, hi) * W) + xc];
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t k = (uint32_t)(k0 + q);
                const uint32_t bq = max(max(best[0], best[1]), max(best[2], best[3]));
                // a row whose h is at least the best value everywhere in the wave cannot improve anything: N(h, .) >= a * h
                if (yu0 - q >= lo && __any((hu[q] << 16) < bq)) {
                    const LgH4 hh = lg_h4(hu[q]);
#pragma unroll
                    for (int i = 0; i < 4; i++) best[i] = min(best[i], lg_norm5_h(hh, k + (uint32_t)i));
                }
                if (yd0 + q <= hi && __any((hv[q] << 16) < bq)) {
                    const LgH4 hh = lg_h4(hv[q]);
#pragma unroll
                    for (int i = 0; i < 4; i++) best[i] = min(best[i], lg_norm5_h(hh, k + (uint32_t)(3 - i)));
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            mx = max(mx, best[i]);
            if (xin && yb + i < H) dout[(unsigned)((yb + i) * W) + xc] = (float)best[i] * (1.0f / 65536.0f);
        }
    }
    mx = lg_wave_max_u32(mx);
    if (lane == 0 && mx) atomicMax(&maxfix[frame * 2 + 0], mx);
}

// ---- the same search in two levels.  Along a column the minimising row is monotone in y: for two candidate rows r1 < r2,
// sign(N(h[r1], |y - r1|) - N(h[r2], |y - r2|)) never decreases with y (checked exhaustively for the mask's norm; it is what
// makes Hirata's / Meijster's lower-envelope scans work for chamfer metrics).  So if a1 minimises at row y1 and a2 at row
// y2 > y1 (ANY minimisers), every row y1 < y < y2 has a minimiser in [min(a1, a2), max(a1, a2)].
//   lg_dtanchor_kernel: rows y % 8 == 0 of the window by the bounded search, recording a minimising row per pixel;
//   lg_dtlevel_kernel:  the seven rows between two anchor rows, candidates = the union over the wave's columns of
//                       [anchor above's row, anchor below's row] -- a dozen rows where the nearest edge stays on one side,
//                       the leaf's whole thickness where the band crosses its medial axis (once per column).
// ~5x fewer candidate evaluations than the one-level search at the benchmark's leaf size; same integers.
template <int NP, int ST>   // NP anchor rows per lane, ST (= 8) rows apart: NP = 4 shares every candidate row's load among four
                            // pixels (large batches); 1 gives four times the waves and a quarter of the work per wave (small
                            // batches: latency)
__global__ __launch_bounds__(256) void lg_dtanchor_kernel(const unsigned long long* __restrict__ bits,
                                                          const LgWin* __restrict__ wins, uint32_t* __restrict__ tmp,
                                                          float* __restrict__ dist_out, uint32_t* __restrict__ maxfix, int H,
                                                          int W, int WW, int wc, int G, int B) {
    int frame, j;
    if (!lg_frame_of_block(G, B, &frame, &j)) return;
    const LgWin w = wins[frame];
    if (!w.search_in) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wxe = min(W, w.wx0 + w.nw * wc);
    constexpr int ROWS = 4 * NP * ST;   // rows per workgroup tile: four waves x NP anchors x ST
    const int ntx = (wxe - w.wx0 + 63) >> 6, nty = (w.wy1 - w.wy0 + ROWS - 1) / ROWS;
    const int lo = max(w.by0 - 1, 0), hi = min(w.by1 + 1, H - 1);   // candidate rows
    const unsigned long long* fb = bits + (size_t)frame * H * WW;
    const uint16_t* hd = reinterpret_cast<const uint16_t*>(tmp + (size_t)frame * 2 * H * W);
    uint16_t* argb = reinterpret_cast<uint16_t*>(tmp + (size_t)frame * 2 * H * W) + (size_t)H * W;   // [H][W] minimising rows
    float* dout = dist_out + (size_t)frame * H * W;
    uint32_t mx = 0;
    constexpr int SPAN = ST * (NP - 1) / 2;   // the anchors lie within SPAN rows of the scan's centre
    const int ntile = __builtin_amdgcn_readfirstlane(ntx * nty);
    for (int tile = j; tile < ntile; tile += G) {
        const int tyi = __builtin_amdgcn_readfirstlane(tile / ntx), txi = tile - tyi * ntx;
        const int wi = (w.wx0 >> 6) + txi;
        const int x = 64 * wi + lane, ya0 = w.wy0 + ROWS * tyi + ST * NP * wave;
        if (ya0 >= w.wy1) continue;   // (wave-uniform)
        const bool xin = x < W;
        const unsigned xc = (unsigned)min(x, W - 1);
        unsigned long long rb[NP];
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const int y = ya0 + ST * i;
            rb[i] = fb[(unsigned)(min(y, H - 1) * WW + wi)];
            rb[i] = y < w.wy1 ? lg_readlane_u64(rb[i], 0) : 0ull;
        }
        unsigned long long anyb = 0;
#pragma unroll
        for (int i = 0; i < NP; i++) anyb |= rb[i];
        uint32_t best[NP];
        int arow[NP];
#pragma unroll
        for (int i = 0; i < NP; i++) {
            best[i] = (xin && ((rb[i] >> lane) & 1ull)) ? 0xFFFFFFFFu : 0u;
            arow[i] = ya0 + ST * i;   // an off-leaf pixel is its own nearest zero pixel
        }
        if (anyb) {
            const int c = ya0 + SPAN;   // scan outwards from here: rows c - k and c + 1 + k
            for (int k0 = 0;; k0 += 4) {
                const int yu0 = c - k0, yd0 = c + 1 + k0;
                if (yu0 < lo && yd0 > hi) break;
                uint32_t bm = best[0];
#pragma unroll
                for (int i = 1; i < NP; i++) bm = max(bm, best[i]);
                if (!__any((uint32_t)max(k0 - SPAN, 0) * LG_A5 < bm)) break;   // every anchor is at least k0 - SPAN rows away from here on
                uint32_t hu[4], hv[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    hu[q] = hd[(unsigned)(min(max(yu0 - q, lo), hi) * W) + xc];
                    hv[q] = hd[(unsigned)(max(min(yd0 + q, hi), lo) * W) + xc];
                }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    uint32_t bq = best[0];
#pragma unroll
                    for (int i = 1; i < NP; i++) bq = max(bq, best[i]);
                    const int yu = yu0 - q, yd = yd0 + q;
                    if (yu >= lo && yu <= hi && __any((hu[q] << 16) < bq)) {
                        const LgH4 hh = lg_h4(hu[q]);
#pragma unroll
                        for (int i = 0; i < NP; i++) {
                            const int dy = ya0 + ST * i - yu;
                            const uint32_t v = lg_norm5_h(hh, (uint32_t)(dy < 0 ? -dy : dy));
                            arow[i] = v < best[i] ? yu : arow[i];
                            best[i] = min(best[i], v);
                        }
                    }
                    if (yd >= lo && yd <= hi && __any((hv[q] << 16) < bq)) {
                        const LgH4 hh = lg_h4(hv[q]);
#pragma unroll
                        for (int i = 0; i < NP; i++) {
                            const int dy = ya0 + ST * i - yd;
                            const uint32_t v = lg_norm5_h(hh, (uint32_t)(dy < 0 ? -dy : dy));
                            arow[i] = v < best[i] ? yd : arow[i];
                            best[i] = min(best[i], v);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const int y = ya0 + ST * i;
            mx = max(mx, best[i]);
            if (xin && y < w.wy1) {
                dout[(unsigned)(y * W) + xc] = (float)best[i] * (1.0f / 65536.0f);
                argb[(unsigned)(y * W) + xc] = (uint16_t)arow[i];
            }
        }
    }
    mx = lg_wave_max_u32(mx);
    if (lane == 0 && mx) atomicMax(&maxfix[frame * 2 + 0], mx);
}

__device__ __forceinline__ int lg_wave_min_i32(int v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int lg_wave_max_i32(int v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

// minimum over the wave in six DPP steps (quads, half rows, rows, then the two row broadcasts): every lane gets it
__device__ __forceinline__ uint32_t lg_wave_min_u32(uint32_t v) {
#define LG_DPP_MIN(CTRL, RM) v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, RM, 0xf, false))
    LG_DPP_MIN(0xB1, 0xf);    // quad_perm 1 0 3 2
    LG_DPP_MIN(0x4E, 0xf);    // quad_perm 2 3 0 1
    LG_DPP_MIN(0x141, 0xf);   // row_half_mirror
    LG_DPP_MIN(0x140, 0xf);   // row_mirror
    LG_DPP_MIN(0x142, 0xa);   // row_bcast15 into rows 1 and 3
    LG_DPP_MIN(0x143, 0xc);   // row_bcast31 into rows 2 and 3
#undef LG_DPP_MIN
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ uint32_t lg_absdiff(int a, int b) {   // |a - b| in one instruction (hipcc emits add, sub, max)
    uint32_t r;
    asm("v_sad_u32 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// One level of the row search between solved rows.  A group = the NT target rows ya + ST (r + 1), r < NT, between the solved
// rows ya and ya + GH (GH = ST (NT + 1)), whose minimising rows bound every target's candidates; wave = one group x 64 columns,
// lane = column, workgroup = four consecutive groups.  <7, 1>: the seven rows between anchors eight rows apart (distances
// only; ST > 1 also records the targets' minimising rows, for a further level -- measured slower, see lg_launch_dtsearch).
// Windows longer than GH + 8 rows -- the columns where the group crosses the leaf's medial axis: the row above is minimised from
// the leaf's upper edge, the row below from its lower edge, the window is the leaf's whole thickness -- are not walked by their
// lane (the other 63 lanes of the wave would wait for ~thickness / 4 round trips) but by the whole wave: lane t takes candidate row first + t of that column, wave minima fold the result.
template <int NT, int ST>
__global__ __launch_bounds__(256) void lg_dtlevel_kernel(const unsigned long long* __restrict__ bits,
                                                         const LgWin* __restrict__ wins, uint32_t* __restrict__ tmp,
                                                         float* __restrict__ dist_out, uint32_t* __restrict__ maxfix, int H, int W,
                                                         int WW, int wc, int G, int B) {
    constexpr int GH = ST * (NT + 1), LONG = GH + 8;
    constexpr bool ARG = ST > 1;   // the targets are solved rows of a later level
    int frame, j;
    if (!lg_frame_of_block(G, B, &frame, &j)) return;
    const LgWin w = wins[frame];
    if (!w.search_in) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wxe = min(W, w.wx0 + w.nw * wc);
    const int ntx = (wxe - w.wx0 + 63) >> 6, nty = (w.wy1 - w.wy0 + 4 * GH - 1) / (4 * GH);
    const int lo = max(w.by0 - 1, 0), hi = min(w.by1 + 1, H - 1);   // candidate rows
    const unsigned long long* fb = bits + (size_t)frame * H * WW;
    const uint16_t* hd = reinterpret_cast<const uint16_t*>(tmp + (size_t)frame * 2 * H * W);
    uint16_t* argb = reinterpret_cast<uint16_t*>(tmp + (size_t)frame * 2 * H * W) + (size_t)H * W;
    float* dout = dist_out + (size_t)frame * H * W;
    uint32_t mx = 0;
    const int ntile = __builtin_amdgcn_readfirstlane(ntx * nty);
    for (int tile = j; tile < ntile; tile += G) {
        const int tyi = __builtin_amdgcn_readfirstlane(tile / ntx), txi = tile - tyi * ntx;
        const int wi = (w.wx0 >> 6) + txi;
        const int x = 64 * wi + lane, ya = w.wy0 + 4 * GH * tyi + GH * wave;
        if (ya + ST >= w.wy1) continue;   // (wave-uniform)
        const bool xin = x < W;
        const unsigned xc = (unsigned)min(x, W - 1);
        unsigned long long rb[NT], anyb = 0;
#pragma unroll
        for (int r = 0; r < NT; r++) {
            const int y = ya + ST * (r + 1);
            rb[r] = fb[(unsigned)(min(y, H - 1) * WW + wi)];
            rb[r] = y < w.wy1 ? lg_readlane_u64(rb[r], 0) : 0ull;
            anyb |= rb[r];
        }
        uint32_t best[NT];
        int arow[NT];
#pragma unroll
        for (int r = 0; r < NT; r++) {
            best[r] = (xin && ((rb[r] >> lane) & 1ull)) ? 0xFFFFFFFFu : 0u;
            arow[r] = ya + ST * (r + 1);   // an off-leaf pixel is its own nearest zero pixel
        }
        if (anyb) {
            // this lane's candidate rows: between the minimising rows of the two solved rows (the row below the last leaf row
            // would be an off-leaf pixel minimised by itself: every row down to `hi` then).  Every lane walks ITS OWN rows
            // (row = first + t): along a tilted edge the windows of neighbouring columns are shifted against each other, and a
            // row shared by the whole wave would have to cover their union -- 64 columns x the edge's slope.  The loads are then
            // per-lane rows (neighbouring lanes mostly hit the same or the next line; NT evaluations pay for each).
            const int a1 = argb[(unsigned)(ya * W) + xc];
            const int yb = ya + GH;
            const int a2 = yb <= w.by1 ? (int)argb[(unsigned)(min(yb, H - 1) * W) + xc] : hi;
            const bool mine = xin && ((anyb >> lane) & 1ull);
            const int first = max(min(a1, a2), lo);
            const int len_all = mine ? min(max(a1, a2), hi) - first + 1 : 0;
            unsigned long long lm = __ballot(len_all > LONG);
            if (__popcll(lm) > 24) lm = 0;   // (a group ALONG the medial axis: every lane is long, the per-lane walk keeps them all busy)
            const int len = ((lm >> lane) & 1ull) ? 0 : len_all;
            const int maxlen = lg_wave_max_i32(len);
            for (int t0 = 0; t0 < maxlen; t0 += 4) {
                // a row past the lane's window counts as a row without a zero pixel (LG_HCAP: beats nothing, see lg_hrun_kernel)
                uint32_t hh4[4] = {LG_HCAP, LG_HCAP, LG_HCAP, LG_HCAP};
                if (t0 < len) {   // lanes whose window has ended fetch nothing: the rows are per lane (64 cache lines per wave
                                  // instruction), and a few columns with long windows would make all 64 lanes fetch lines for them
#pragma unroll
                    for (int q = 0; q < 4; q++) hh4[q] = hd[(unsigned)(min(first + t0 + q, hi) * W) + xc];
#pragma unroll
                    for (int q = 1; q < 4; q++) hh4[q] = t0 + q < len ? hh4[q] : (uint32_t)LG_HCAP;
                }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    uint32_t bq = best[0];
#pragma unroll
                    for (int r = 1; r < NT; r++) bq = max(bq, best[r]);
                    if (__any((hh4[q] << 16) < bq)) {   // N(h, .) >= a * h: a row that cannot improve any lane is skipped
                        const LgH4 hh = lg_h4(hh4[q]);
                        const int yc = first + t0 + q;
#pragma unroll
                        for (int r = 0; r < NT; r++) {
                            const uint32_t v = lg_norm5_h(hh, lg_absdiff(yc, ya + ST * (r + 1)));
                            if (ARG) arow[r] = v < best[r] ? yc : arow[r];
                            best[r] = min(best[r], v);
                        }
                    }
                }
            }
            while (lm) {   // the long windows, one column at a time, 64 candidate rows per step
                const int L = __builtin_ctzll(lm);
                lm &= lm - 1;
                const int fL = __builtin_amdgcn_readlane(first, L), nL = __builtin_amdgcn_readlane(len_all, L);
                const unsigned xL = (unsigned)__builtin_amdgcn_readlane((int)xc, L);
                uint32_t b[NT];
                int br[NT];
#pragma unroll
                for (int r = 0; r < NT; r++) { b[r] = 0xFFFFFFFFu; br[r] = 0xFFFF; }
                for (int t0 = 0; t0 < nL; t0 += 64) {
                    const int yc = fL + t0 + lane;
                    const bool act = t0 + lane < nL;
                    const LgH4 hh = lg_h4(act ? (uint32_t)hd[(unsigned)(yc * W) + xL] : (uint32_t)LG_HCAP);
#pragma unroll
                    for (int r = 0; r < NT; r++) {
                        const uint32_t v = lg_norm5_h(hh, lg_absdiff(yc, ya + ST * (r + 1)));
                        if (ARG) br[r] = v < b[r] ? yc : br[r];
                        b[r] = min(b[r], v);
                    }
                }
#pragma unroll
                for (int r = 0; r < NT; r++) {
                    const uint32_t m = lg_wave_min_u32(b[r]);
                    if (ARG) {   // a row that attains the minimum (any of them)
                        const int mr = (int)lg_wave_min_u32(b[r] == m ? (uint32_t)br[r] : 0xFFFFu);
                        arow[r] = (lane == L && m < best[r]) ? mr : arow[r];
                    }
                    best[r] = lane == L ? min(best[r], m) : best[r];   // (an off-leaf pixel of the column keeps its 0)
                }
            }
        }
#pragma unroll
        for (int r = 0; r < NT; r++) {
            const int y = ya + ST * (r + 1);
            mx = max(mx, best[r]);
            if (xin && y < w.wy1) {
                dout[(unsigned)(y * W) + xc] = (float)best[r] * (1.0f / 65536.0f);
                if (ARG) argb[(unsigned)(y * W) + xc] = (uint16_t)arow[r];
            }
        }
    }
    mx = lg_wave_max_u32(mx);
    if (lane == 0 && mx) atomicMax(&maxfix[frame * 2 + 0], mx);
}

static int lg_search_groups(int B, int per_batch, int lo, int hi) {
    const int bpad = 8 * ((B + 7) / 8);
    return std::max(lo, std::min(hi, per_batch / bpad));
}
void lg_launch_hrun(const unsigned long long* bits, uint32_t* tmp, const LgWin* win, int B, int H, int W, int WW, hipStream_t s) {
    const int G = lg_search_groups(B, 2048, 2, 64);
    hipLaunchKernelGGL(lg_hrun_kernel, dim3(8u * G * ((B + 7) / 8)), dim3(256), 0, s, bits, win, tmp, H, W, WW, G, B);
}
// algo 1: the one-level search of every row (one launch: small batches, where the launches' latency counts and the device is
// not full); algo 2: anchor rows every 8 (phase 0), then the rows between them (phase 1); 3 / 4: the same with four / one anchor
// rows per lane whatever the batch (tests).  Returns 0 when the algorithm has no such phase.  (Ladders -- anchors every 32 or 16
// rows, then the rows half-way between solved rows, level by level -- were built three times: six launches of one target row per
// wave, one kernel per 64-column strip with everything in LDS, and lg_dtlevel_kernel<1, 8> / <1, 4> / <3, 1> under anchors every
// 16 rows.  All exact, up to ten times fewer evaluations, all slower: a level's wave pays its fixed cost -- bit rows, the two
// solved rows' minimisers, window, stores: three dependent round trips -- for one or three rows of work instead of seven, and
// the anchors' cost does not halve with their number.  profiles/NOTES_r04.md.)
int lg_launch_dtsearch(int phase, int algo, const unsigned long long* bits, uint32_t* tmp, float* dist_out, uint32_t* maxfix,
                       const LgWin* win, int B, int H, int W, int WW, hipStream_t s) {
    static const int g_env = getenv("LG_DT_SEARCH_G") ? atoi(getenv("LG_DT_SEARCH_G")) : 0;
    const int wc = lg_dt_geometry(W, nullptr);
    const unsigned nb8 = 8u * (unsigned)((B + 7) / 8);
    const int tx = (W + 63) / 64;
#define LG_LAUNCH_SEARCH(K, G_) hipLaunchKernelGGL(K, dim3(nb8 * (G_)), dim3(256), 0, s, bits, win, tmp, dist_out, maxfix, H, W, WW, wc, G_, B)
    if (algo == 1) {
        if (phase) return 0;
        const int G = g_env > 0 ? g_env : lg_search_groups(B, 8192, 8, 256);
        LG_LAUNCH_SEARCH(lg_dtsearch_kernel, G);
        return 1;
    }
    if (phase == 0) {
        const bool np4 = algo == 3 || (algo == 2 && B >= 128);   // (32 frames: 0.071 vs 0.100 ms with one / four anchors per lane)
        // workgroups per frame: the window's tiles (64 columns x 128 / 32 rows) when the grid allows -- leaf tiles cluster, and a
        // workgroup that walks several of them with a fixed stride gets several heavy ones or none
        const int trows = np4 ? 128 : 32;
        const int Ga = g_env > 0 ? g_env : std::min(tx * ((H + trows - 1) / trows), lg_search_groups(B, 16384, 8, 512));
        if (np4) LG_LAUNCH_SEARCH((lg_dtanchor_kernel<4, 8>), Ga);
        else LG_LAUNCH_SEARCH((lg_dtanchor_kernel<1, 8>), Ga);
        return 1;
    }
    if (phase > 1) return 0;
    const int Gb = g_env > 0 ? g_env : std::min(tx * ((H + 31) / 32), lg_search_groups(B, 32768, 8, 512));
    LG_LAUNCH_SEARCH((lg_dtlevel_kernel<7, 1>), Gb);
#undef LG_LAUNCH_SEARCH
    return 1;
}

// ============================================================================ fused score planes
// One pass producing sdf_score, approach, flatness, isolation, accessibility, stem, traditional and the
// validity mask (grasp_point_selector.py:256-288) from depth + mask bits + distance_map (+ frame scalars).
// Tile 64 x LG_TH (16), 256 threads, each thread 4 consecutive pixels x LG_TH/16 rows (16-byte stores per lane).
//
// Launch shape: one workgroup per tile, or (LG_FINAL_PERSIST, lg_launch_final) resident workgroups that walk the tiles; the
// same code serves both -- blockIdx % 8 is the XCD, every XCD owns a contiguous range of tiles and its workgroups take them
// with the XCD's workgroup count as stride, so neighbouring workgroups work on neighbouring tiles at the same time (shared
// halos hit in the XCD's L2).
// What the tile loop needs from the code: nothing loop-invariant may stay in registers across tiles at 64 VGPRs (8 waves per
// SIMD; 66 gave 7) -- the arguments are read through the kernarg pointer, laundered once per tile (their ~60 scalar loads then
// belong to the iteration instead of being hoisted: without this 156 SGPRs + 39 VGPRs spilled), the thread index likewise
// (indices and LDS addresses are recomputed per tile), the tile walk is 32-bit scalar arithmetic, frame scalars come
// through the scalar cache, and the barriers order LDS only (s_waitcnt lgkmcnt(0)): __syncthreads() would also wait for the
// previous tile's plane stores to be acknowledged.
// What bounds it (tools/ubench/stream_mix.hip, tools/final_ablate.sh, one box): a kernel with ONLY this path's loads and
// stores in the same 64 x 16 tile shape reaches 0.72 of the 8 TB/s peak (1.70 ms per 128 frames; a plain 2-read + 1-write copy
// 0.65, the nine stores alone 0.84, 256 x 4 tiles 0.82); this kernel with its arithmetic removed 1.98 ms, complete 2.21 ms.
// Occupancy: the stencil path scales with the waves in flight -- every tile on the stencil path, 128 frames of 1080p, one
// workgroup per tile: 3.23 / 2.80 / 2.63 / 2.37 ms at 4 / 5 / 6-7 / 8 waves per SIMD (tools/final_dense.py).
#ifndef LG_FINAL_WPE
#define LG_FINAL_WPE 8
#endif
#define LG_FINAL_WPE_ATTR __attribute__((amdgpu_waves_per_eu(LG_FINAL_WPE, LG_FINAL_WPE)))
__device__ __forceinline__ float lg_uniform_f(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
// VEC: W % 4 == 0 (16-byte loads / stores everywhere); ALL: every plane and the validity plane are wanted (the grasp-selection
// call with the CNN): the instantiation for the usual case carries no per-plane null checks and no scalar store paths.
// R: radius of the separable Gaussian of ImageProcessor.smooth_depth (gaussian_size = 2R+1: 1, 3, 5 = the node's, 7); the
// stencil reaches HALO = R + 1 pixels (Gaussian, then the 3x3 Sobel).
template <bool VEC, bool ALL, int R>
__global__ __launch_bounds__(256) LG_FINAL_WPE_ATTR void lg_final_kernel(LgFinalArgs a_) {
    constexpr int HALO = R + 1;
    static_assert(R >= 0 && HALO <= 4, "the depth tile carries 4 halo columns");
    constexpr int DW = 72;             // dm tile: cols tx0-4 .. tx0+67
    constexpr int DH = LG_TH + 2 * HALO;   // rows ty0-HALO .. ty0+LG_TH-1+HALO
    constexpr int GW = LG_TW + 2;      // g tile: cols tx0-1 .. tx0+64
    constexpr int GH = LG_TH + 2;
    __shared__ __attribute__((aligned(16))) float s_dm[DH * DW];
    __shared__ float s_h[DH * GW];
    float* const s_g = s_dm;  // the smoothed tile reuses the depth tile's LDS (dead after the horizontal pass)
    static_assert(GH * (GW + 2) <= DH * DW, "g tile must fit in the depth tile");
    __shared__ unsigned long long s_key[4];
    __shared__ int s_any[2];
    typedef float lg_f4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(4))) LgFinalArgs* lg_args_ptr;
    lg_args_ptr ap = (lg_args_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    (void)a_;

    // ---- the tile walk, in scalar registers (total < 2^31: lg_launch_final)
    const int ntile = ap->tiles_x * ap->tiles_y;
    const int total = ntile * ap->B;
    const int xcd = (int)(blockIdx.x & 7u), tq = total >> 3, tr = total & 7;
    const int xcd_first = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    const int xcd_count = tq + (xcd < tr ? 1 : 0);
    // tiles per workgroup (ap->tpw > 0): workgroup j of the XCD takes the tpw consecutive tiles j * tpw ...; otherwise one tile
    // each, or, as resident workgroups, tiles j, j + n, j + 2n ... of the XCD's range
    const int tpw = ap->tpw;
    const int stride = tpw > 0 ? 1 : (int)((gridDim.x + 7u) >> 3);
    int it = (int)(blockIdx.x >> 3) * (tpw > 0 ? tpw : 1);
    const int it_end = tpw > 0 ? min(it + tpw, xcd_count) : xcd_count;
    if (it >= it_end) return;
    int frame = __builtin_amdgcn_readfirstlane((xcd_first + it) / ntile);
    int tile = xcd_first + it - frame * ntile;
    const int step_f = __builtin_amdgcn_readfirstlane(stride / ntile), step_t = stride - step_f * ntile;
    for (int par = 0; it < it_end;
         par ^= 1, it += stride, frame += step_f, tile += step_t, frame += (tile >= ntile ? 1 : 0), tile -= (tile >= ntile ? ntile : 0)) {
    asm volatile("" : "+s"(ap));
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    const int tiles_x = ap->tiles_x;
    // tile < 8192, tiles_x <= 128: the float quotient of (tile + 0.5) is at least 0.5 / 128 away from an integer, its error < 0.001
    const int by = __builtin_amdgcn_readfirstlane((int)(((float)tile + 0.5f) * __frcp_rn((float)tiles_x)));
    const int bx = tile - by * tiles_x;
    const int tx0 = bx * LG_TW, ty0 = by * LG_TH;
    const int H = ap->H, W = ap->W, WW = ap->WW;
    const size_t fo = (size_t)frame * H * W;
    const char* depth = (const char*)(ap->depth + fo);
    const char* bits = (const char*)(ap->bits + (size_t)frame * H * WW);
    const char* stemb = (const char*)(ap->stem_bits + (size_t)frame * H * WW);
    auto bits_at = [&](const char* base, int y, int w) {   // uniform base + 32-bit lane offset (saddr addressing)
        return *reinterpret_cast<const unsigned long long*>(base + (unsigned)(y * WW + w) * 8u);
    };

    const int txi = t & 15, tyi = t >> 4;
    constexpr bool vec = VEC;         // x0 % 4 == 0 always; x0 + 3 < W when W % 4 == 0
    constexpr int RPT = LG_TH / 16;   // rows per thread (16 thread rows per tile)
    static_assert(LG_TH % 16 == 0 && RPT >= 1, "tile height must be a multiple of 16");
    // distance_map is only computed inside the frame's sweep window (LgWin, tile aligned); outside it is exactly 0 and
    // this kernel writes the plane instead of reading it
    // (win / fp / maxfix were written by earlier kernels and are only read here: constant-address-space loads, i.e. the
    //  scalar cache, instead of a vector load + readfirstlane per value)
    bool in_win;
    {
        const __attribute__((address_space(4))) LgWin* wp = (const __attribute__((address_space(4))) LgWin*)(ap->win + frame);
        const int wx0 = wp->wx0;
        in_win = tx0 >= wx0 && tx0 < min(W, wx0 + wp->nw * ap->win_wc) && ty0 >= wp->wy0 && ty0 < wp->wy1;
    }
    // 4 floats of one plane at pixel offset `off` of this frame (uniform plane base + 32-bit byte offset)
    auto st4 = [&](int mi, unsigned off, int x0, const float* v) {
        float* dst = ap->maps[mi];
        if (!ALL && !dst) return;
#ifdef LG_FINAL_ABLATE   // timing ablation builds only (tools/build_variants.sh): wrong results by design
        if ((LG_FINAL_ABLATE & 8) && mi != LG_MAP_TRADITIONAL) return;   // arithmetic without the plane stores
#endif
        char* p = (char*)(dst + fo) + off * 4u;
        if (vec) {
            lg_f4 pk = {v[0], v[1], v[2], v[3]};
            if (ap->nt_stores) __builtin_nontemporal_store(pk, reinterpret_cast<lg_f4*>(p));  // write-once stream (measured slower)
            else *reinterpret_cast<lg_f4*>(p) = pk;
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (x0 + j < W) reinterpret_cast<float*>(p)[j] = v[j];
        }
    };
    auto st_valid = [&](unsigned off, int x0, uint32_t vbytes) {
        if (!ALL && !ap->valid) return;
        uint8_t* p = ap->valid + fo + off;
        if (vec) {
            __builtin_nontemporal_store(vbytes, reinterpret_cast<uint32_t*>(p));
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (x0 + j < W) p[j] = (uint8_t)((vbytes >> (8 * j)) & 1u);
        }
    };
    float zero = 0.0f;
    asm volatile("" : "+v"(zero));   // (a zero the compiler cannot keep in registers across tiles)
    // ---- tile-level fast path.  A leaf covers a few per cent of the frame: when no mask bit lies in this tile's extended
    //      region (tile + the HALO-pixel reach of the Gaussian and the 3x3 Sobel), depth * mask is 0 all over it, the smoothed
    //      plane and both gradients are 0 and flatness = exp(-5 * 0) = 1 exactly; every other plane is "* mask" = 0,
    //      traditional = w_flat * 1, nothing is valid.  Such tiles never read depth and skip the stencil phases.
    //      Wave 0 looks at the DH x 3 (row, word) pairs and posts the verdict; s_any alternates between two slots so that a
    //      wave that runs ahead into the next tile cannot overwrite a verdict the others have not read yet.
    if (!ap->no_skip) {
        if (t < 64) {
            unsigned long long nz = 0;
            for (int e = t; e < DH * 3; e += 64) {
                const int er = e / 3, wq = e % 3;                 // extended row, word (left neighbour, own, right neighbour)
                const int y = lg_reflect(ty0 - HALO + er, H), wi = bx - 1 + wq;
                if (wi >= 0 && wi < WW) {
                    unsigned long long v = bits_at(bits, y, wi);
                    if (wq == 0) v >>= 56;                        // columns tx0-8 .. tx0-1 (halo 4 + reflection slack)
                    if (wq == 2) v &= 0xffull;                    // columns tx0+64 .. tx0+71
                    nz |= v;
                }
            }
            const bool any = __ballot(nz != 0) != 0ull;
            if (t == 0) s_any[par] = any ? 1 : 0;
        }
        lg_lds_barrier();
        if (!s_any[par]) {
            const float flat1 = __expf(-ap->flat_scale * __builtin_amdgcn_sqrtf(zero));
            const float tr1 = ap->w_flat * flat1;
            const float c_flat[4] = {flat1, flat1, flat1, flat1};
            const float c_trad[4] = {tr1, tr1, tr1, tr1};
            const float c_zero[4] = {zero, zero, zero, zero};
#pragma unroll
            for (int rr = 0; rr < RPT; rr++) {
                const int y = ty0 + tyi + 16 * rr, x0 = tx0 + 4 * txi;
                if (y < H && x0 < W) {
                    const unsigned off = (unsigned)(y * W + x0);
                    st4(LG_MAP_SDF, off, x0, c_zero); st4(LG_MAP_APPROACH, off, x0, c_zero); st4(LG_MAP_FLATNESS, off, x0, c_flat);
                    st4(LG_MAP_ISOLATION, off, x0, c_zero); st4(LG_MAP_ACCESS, off, x0, c_zero); st4(LG_MAP_STEM, off, x0, c_zero);
                    st4(LG_MAP_TRADITIONAL, off, x0, c_trad);
                    if (!in_win) st4(LG_MAP_DISTANCE, off, x0, c_zero);
                    st_valid(off, x0, 0u);
                }
            }
            if (t == 0) {   // arg-max key of a tile without valid pixels: score 0, largest flat index (top-k tie rule)
                const int ymax = min(ty0 + LG_TH, H) - 1, xmax = min(tx0 + LG_TW, W) - 1;
                ap->tilekeys[(size_t)frame * ntile + tile] =
                    ((unsigned long long)lg_orderable(0.0f) << 32) | (uint32_t)(ymax * W + xmax);
            }
            continue;
        }
    }

    // ---- per-pixel operands of the full path, requested before the stencil phases so their latency overlaps them
    float din_pre[RPT][4];
    unsigned mnib_pre[RPT], snib_pre[RPT];
#pragma unroll
    for (int rr = 0; rr < RPT; rr++) {
        const int y = ty0 + tyi + 16 * rr, x0 = tx0 + 4 * txi;
        mnib_pre[rr] = snib_pre[rr] = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) din_pre[rr][j] = 0.0f;
        if (y < H && x0 < W) {
            const unsigned sh = (unsigned)(x0 & 63);
            mnib_pre[rr] = (unsigned)(bits_at(bits, y, bx) >> sh) & 0xfu;
            snib_pre[rr] = (unsigned)(bits_at(stemb, y, bx) >> sh) & 0xfu;
            const char* dsrc = (const char*)(ap->maps[LG_MAP_DISTANCE] + fo) + (unsigned)(y * W + x0) * 4u;
            if (!in_win) {
            } else if (vec) {
                float4 v = *reinterpret_cast<const float4*>(dsrc);
                din_pre[rr][0] = v.x; din_pre[rr][1] = v.y; din_pre[rr][2] = v.z; din_pre[rr][3] = v.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) din_pre[rr][j] = (x0 + j < W) ? reinterpret_cast<const float*>(dsrc)[j] : 0.0f;
            }
        }
    }

    // ---- stage dm = depth * mask over the extended tile (reflect padding of smooth_depth, image_processor.py:60)
    const bool fast = VEC && (tx0 >= 4) && (tx0 + LG_TW + 4 <= W);
    if (fast) {
        for (int idx = t; idx < DH * (DW / 4); idx += 256) {
            int er = idx / (DW / 4), g4 = idx % (DW / 4);
            int y = lg_reflect(ty0 - HALO + er, H);
            int x = tx0 - 4 + 4 * g4;
            float4 d = *reinterpret_cast<const float4*>(depth + (unsigned)(y * W + x) * 4u);
            unsigned long long wbits = bits_at(bits, y, x >> 6);
            unsigned nib = (unsigned)(wbits >> (x & 63)) & 0xfu;
            float4 o;
            o.x = (nib & 1u) ? d.x : 0.0f;
            o.y = (nib & 2u) ? d.y : 0.0f;
            o.z = (nib & 4u) ? d.z : 0.0f;
            o.w = (nib & 8u) ? d.w : 0.0f;
            *reinterpret_cast<float4*>(&s_dm[er * DW + 4 * g4]) = o;
        }
    } else {
        for (int idx = t; idx < DH * DW; idx += 256) {
            int er = idx / DW, ec = idx % DW;
            int y = lg_reflect(ty0 - HALO + er, H);
            int x = lg_reflect(tx0 - 4 + ec, W);
            float d = *reinterpret_cast<const float*>(depth + (unsigned)(y * W + x) * 4u);
            unsigned long long wbits = bits_at(bits, y, x >> 6);
            s_dm[idx] = ((wbits >> (x & 63)) & 1ull) ? d : 0.0f;
        }
    }
    lg_lds_barrier();
#ifdef LG_FINAL_ABLATE
    if (LG_FINAL_ABLATE & 16) {   // ablation build: the memory traffic of the dense path without its arithmetic
#pragma unroll
        for (int rr = 0; rr < RPT; rr++) {
            const int y = ty0 + tyi + 16 * rr, x0 = tx0 + 4 * txi;
            if (y < H && x0 < W) {
                const unsigned off = (unsigned)(y * W + x0);
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; j++) v[j] = din_pre[rr][j] + s_dm[(tyi + 16 * rr + HALO) * DW + 4 + 4 * txi + j] + (float)mnib_pre[rr] + (float)snib_pre[rr];
                st4(LG_MAP_SDF, off, x0, v); st4(LG_MAP_APPROACH, off, x0, v); st4(LG_MAP_FLATNESS, off, x0, v);
                st4(LG_MAP_ISOLATION, off, x0, v); st4(LG_MAP_ACCESS, off, x0, v); st4(LG_MAP_STEM, off, x0, v);
                st4(LG_MAP_TRADITIONAL, off, x0, v);
                if (!in_win) st4(LG_MAP_DISTANCE, off, x0, v);
                st_valid(off, x0, 0u);
            }
        }
        lg_lds_barrier();
        continue;
    }
#endif
    // ---- separable (2R+1)-tap Gaussian; g is stored at the *reflect-padded* coordinates the Sobel stage reads
    //      (F.pad(g,(1,1,1,1),'reflect'), grasp_point_selector.py:648): g_ext(e) = G(reflect1(e)).
    // Work split without div/mod: thread t owns column (t & 63) for rows (t >> 6) + 4k; the two extra halo
    // columns (64, 65) are covered by the first threads afterwards.
    {
        float kw[2 * R + 1];
#pragma unroll
        for (int i = 0; i < 2 * R + 1; i++) kw[i] = ap->k1[i];
        auto tap_row = [&](const float* p) {   // taps in ascending order, like the 5-tap form this generalises
            float acc = kw[0] * p[0];
#pragma unroll
            for (int i = 1; i < 2 * R + 1; i++) acc += kw[i] * p[i];
            return acc;
        };
        auto tap_col = [&](const float* p) {
            float acc = kw[0] * p[0];
#pragma unroll
            for (int i = 1; i < 2 * R + 1; i++) acc += kw[i] * p[i * GW];
            return acc;
        };
        const int ec = t & 63;
        int lc = lg_reflect(tx0 - 1 + ec, W) - (tx0 - 4);
        lc = lc < R ? R : (lc > DW - 1 - R ? DW - 1 - R : lc);
#pragma unroll
        for (int k = 0; k < (DH + 3) / 4; k++) {
            const int er = (t >> 6) + 4 * k;
            if (er < DH) s_h[er * GW + ec] = tap_row(&s_dm[er * DW + lc - R]);
        }
        if (t < 2 * DH) {
            const int er = t >> 1, ec2 = 64 + (t & 1);
            int lc2 = lg_reflect(tx0 - 1 + ec2, W) - (tx0 - 4);
            lc2 = lc2 < R ? R : (lc2 > DW - 1 - R ? DW - 1 - R : lc2);
            s_h[er * GW + ec2] = tap_row(&s_dm[er * DW + lc2 - R]);
        }
        lg_lds_barrier();
#pragma unroll
        for (int k = 0; k < (GH + 3) / 4; k++) {
            const int gr = (t >> 6) + 4 * k;
            if (gr < GH) {
                int lr = lg_reflect(ty0 - 1 + gr, H) - (ty0 - HALO);
                lr = lr < R ? R : (lr > DH - 1 - R ? DH - 1 - R : lr);
                s_g[gr * (GW + 2) + ec] = tap_col(&s_h[(lr - R) * GW + ec]);
            }
        }
        if (t < 2 * GH) {
            const int gr = t >> 1, ec2 = 64 + (t & 1);
            int lr = lg_reflect(ty0 - 1 + gr, H) - (ty0 - HALO);
            lr = lr < R ? R : (lr > DH - 1 - R ? DH - 1 - R : lr);
            s_g[gr * (GW + 2) + ec2] = tap_col(&s_h[(lr - R) * GW + ec2]);
        }
    }
    lg_lds_barrier();

    // ---- per-pixel planes
    const __attribute__((address_space(4))) LgFrameParams* fpp = (const __attribute__((address_space(4))) LgFrameParams*)(ap->fp + frame);
    const int has_angle = fpp->has_angle;
    const float sin_t = fpp->sin_t, cos_t = fpp->cos_t;
    const __attribute__((address_space(4))) uint32_t* mfp = (const __attribute__((address_space(4))) uint32_t*)(ap->maxfix + frame * 2);
    const uint32_t mfi = mfp[0], mfo = mfp[1];
    const float maxabs = fmaxf((float)mfi * (1.0f / 65536.0f), (float)mfo * (1.0f / 65536.0f));
    const float inv_maxabs = lg_uniform_f(__frcp_rn(maxabs));
    const float opt_d = ap->optimal_distance;
    const float inv_2s2 = ap->inv_2s2;
    const float focal = ap->f;
    const float f2 = focal * focal;
    const float ramp_step = ap->iso_ramp_step;
    unsigned long long best = 0;
#pragma unroll
    for (int rr = 0; rr < RPT; rr++) {
        const int ly = tyi + 16 * rr;
        const int y = ty0 + ly;
        const int x0 = tx0 + 4 * txi;
        if (y < H && x0 < W) {
            const unsigned off = (unsigned)(y * W + x0);
            const unsigned mnib = mnib_pre[rr], snib = snib_pre[rr];
            const float* din = din_pre[rr];
            // flatness first: Sobel cross-correlation on the smoothed plane, exp(-5 |grad|) (:646-655); its 18 operands
            // are dead before the other planes' arithmetic starts
            float o_flat[4];
            {
                const float* g0 = &s_g[(ly + 0) * (GW + 2) + 4 * txi];
                const float* g1 = g0 + (GW + 2);
                const float* g2 = g1 + (GW + 2);
                float ga[6], gb[6], gc[6];
                // 16-byte + 8-byte LDS reads (row stride 272 B keeps them aligned): no bank conflicts
                const float4 a4 = *reinterpret_cast<const float4*>(g0), b4 = *reinterpret_cast<const float4*>(g1),
                             c4 = *reinterpret_cast<const float4*>(g2);
                const float2 a2 = *reinterpret_cast<const float2*>(g0 + 4), b2 = *reinterpret_cast<const float2*>(g1 + 4),
                             c2 = *reinterpret_cast<const float2*>(g2 + 4);
                ga[0] = a4.x; ga[1] = a4.y; ga[2] = a4.z; ga[3] = a4.w; ga[4] = a2.x; ga[5] = a2.y;
                gb[0] = b4.x; gb[1] = b4.y; gb[2] = b4.z; gb[3] = b4.w; gb[4] = b2.x; gb[5] = b2.y;
                gc[0] = c4.x; gc[1] = c4.y; gc[2] = c4.z; gc[3] = c4.w; gc[4] = c2.x; gc[5] = c2.y;
                const float flat_scale = ap->flat_scale;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    float sx = (ga[j + 2] - ga[j]) + 2.0f * (gb[j + 2] - gb[j]) + (gc[j + 2] - gc[j]);
                    float sy = (gc[j] + 2.0f * gc[j + 1] + gc[j + 2]) - (ga[j] + 2.0f * ga[j + 1] + ga[j + 2]);
                    o_flat[j] = __expf(-flat_scale * __builtin_amdgcn_sqrtf(sx * sx + sy * sy));  // v_sqrt_f32, 1 ulp
                }
            }
            st4(LG_MAP_FLATNESS, off, x0, o_flat);
            if (!in_win) { const float z4[4] = {zero, zero, zero, zero}; st4(LG_MAP_DISTANCE, off, x0, z4); }
            float o_trad[4];
            uint32_t vbytes = 0;
            const float w_flat = ap->w_flat;
            // A leaf covers a few per cent of a frame: when no lane of this wave sits on the mask every plane but
            // flatness is exactly zero (they are all "* mask"), traditional = w_flat * flatness and nothing is
            // valid.  The wave-uniform branch skips the geometry / SDF / isolation arithmetic for those rows.
            const bool wave_on_mask = (ap->no_skip & 2) || __ballot(mnib != 0) != 0ull;
            if (wave_on_mask) {
                float o_sdf[4], o_app[4], o_iso[4], o_acc[4], o_stem[4];
                const float dyp = (float)(y - ap->cyi) - ap->cyf;  // exact integer part first: no cancellation near the centre
                const int dyb = min(y + 1, H - y);
                const float ramp = ap->iso_ramp_top + ramp_step * (float)y;
                const float w_approach = ap->w_approach, w_sdf = ap->w_sdf, w_access = ap->w_access;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int x = x0 + j;
                    const float m = ((mnib >> j) & 1u) ? 1.0f : 0.0f;
                    const float st = ((snib >> j) & 1u) ? 1.0f : 0.0f;
                    // closed-form geometry planes                                           (:502-524, :569-593)
                    const float dxp = (float)(x - ap->cxi) - ap->cxf;
                    const float r2 = dxp * dxp + dyp * dyp;
                    const float inv_r = r2 > 0.0f ? rsqrtf(r2) : 0.0f;
                    const float r = r2 * inv_r;
                    const float app = focal * rsqrtf(r2 + f2) * m;
                    const float cosang = r2 > 0.0f ? dxp * inv_r : 1.0f;
                    const float acc = (ap->access_w_dist * (1.0f - r * ap->inv_maxd) + ap->access_w_dir * cosang) * m;
                    // SDF / edge term                                                        (:526-567)
                    const float align = has_angle ? fabsf(dxp * inv_r * sin_t - dyp * inv_r * cos_t) : 1.0f;
                    const float dd = din[j] - opt_d;
                    const float interior = __expf(-(dd * dd) * inv_2s2);
                    const float sdfn = din[j] * inv_maxabs;  // inside the mask d_out == 0
                    const float sdf = (ap->sdf_w_interior * interior + ap->sdf_w_align * align + ap->sdf_w_sdf * sdfn) * m;
                    // degenerate isolation map: chamfer-3 transform of an image with no zero pixel (:595-633)
                    const int dbrd = min(min(x + 1, W - x), dyb);
                    const float dt3 = (float)(ap->init0 + (uint32_t)dbrd * LG_A3) * (1.0f / 65536.0f);
                    const float s = dt3 * ap->iso_inv_max;
                    const float iso = (ap->iso_w_close * s + ap->iso_w_wide * s) * ramp * m;
                    // fusion + validity                                                      (:272-288)
                    const float trad = (w_approach * app + w_sdf * sdf + w_flat * o_flat[j] + w_access * acc) * (1.0f - st);
                    const bool valid = (din[j] > ap->min_edge_distance) && (m > 0.0f) && (st < ap->stem_valid_thresh);
                    o_sdf[j] = sdf; o_app[j] = app; o_iso[j] = iso; o_acc[j] = acc; o_stem[j] = st; o_trad[j] = trad;
                    if (valid) vbytes |= 1u << (8 * j);
                    if (x < W) {
                        unsigned long long key =
                            ((unsigned long long)lg_orderable(lg_valid_score(trad, valid)) << 32) | (uint32_t)(y * W + x);
                        best = key > best ? key : best;
                    }
                }
                st4(LG_MAP_SDF, off, x0, o_sdf);
                st4(LG_MAP_APPROACH, off, x0, o_app);
                st4(LG_MAP_ISOLATION, off, x0, o_iso);
                st4(LG_MAP_ACCESS, off, x0, o_acc);
                st4(LG_MAP_STEM, off, x0, o_stem);
            } else {
                const float z4[4] = {zero, zero, zero, zero};
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    o_trad[j] = w_flat * o_flat[j];  // (0.4*0 + 0.3*0 + 0.2*flat + 0.1*0) * (1 - 0), same float32 operations' result
                    if (x0 + j < W) {
                        unsigned long long key = ((unsigned long long)lg_orderable(0.0f) << 32) | (uint32_t)(y * W + x0 + j);
                        best = key > best ? key : best;
                    }
                }
                st4(LG_MAP_SDF, off, x0, z4);
                st4(LG_MAP_APPROACH, off, x0, z4);
                st4(LG_MAP_ISOLATION, off, x0, z4);
                st4(LG_MAP_ACCESS, off, x0, z4);
                st4(LG_MAP_STEM, off, x0, z4);
            }
            st4(LG_MAP_TRADITIONAL, off, x0, o_trad);
            st_valid(off, x0, vbytes);
        }
    }
    best = lg_wave_max_u64(best);
    if ((t & 63) == 0) s_key[t >> 6] = best;
    lg_lds_barrier();   // (also orders this tile's LDS reads before the next tile's writes)
    if (t == 0) {
        unsigned long long k = s_key[0];
        k = s_key[1] > k ? s_key[1] : k;
        k = s_key[2] > k ? s_key[2] : k;
        k = s_key[3] > k ? s_key[3] : k;
        ap->tilekeys[(size_t)frame * ntile + tile] = k;
    }
    }   // tile walk
}

void lg_launch_final(const LgFinalArgs& a_in, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
    long long total = (long long)a_in.tiles_x * a_in.tiles_y * a_in.B;   // < 2^31: tiles_x * tiles_y <= 8192 (make_plan), B is an int count of frames that fit in memory
    // Launch form.  One workgroup per tile by default: on real frames most tiles take the constant path and the dispatcher's
    // own load balancing beats a fixed stride (2.71 vs 3.12 ms per 256 frames).  LG_FINAL_PERSIST=n (or a.persist): n resident
    // workgroups per CU (a multiple of 8 workgroups, so that blockIdx % 8 stays the XCD) walk the tiles.  With every tile on
    // the stencil path the walk measured 5 % faster on three boxes (2.10 vs 2.21 ms per 128 frames; its arithmetic alone 1.14
    // vs 1.42 ms) and 25 % SLOWER on a fourth, faster one (2.28-2.33 vs 1.80-1.87 ms): a wave's loads for its next tile queue
    // behind its own plane stores (vmcnt retires in order), which costs more the faster the memory side is.  Off by default.
    static const int persist_env = getenv("LG_FINAL_PERSIST") ? atoi(getenv("LG_FINAL_PERSIST")) : -1;
    static const int cus = [] {
        int dev = 0, n = 256;
        hipGetDevice(&dev);
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n;
    }();
    const int per_cu = persist_env >= 0 ? persist_env : (a_in.persist ? LG_FINAL_WPE : 0);
    const int resident = per_cu > 0 ? std::max(8, cus * per_cu / 8 * 8) : 0;
    // LG_FINAL_TPW=n: n consecutive tiles per workgroup (experiment: 2 / 4 / 8 change the benchmark launch by -1 / +1 / +3 %,
    // the dense launch within the noise: the workgroup launch rate is not what limits either)
    static const int tpw_env = getenv("LG_FINAL_TPW") ? atoi(getenv("LG_FINAL_TPW")) : 0;
    LgFinalArgs a = a_in;
    const bool walk = resident && total > 2ll * resident;
    a.tpw = walk ? 0 : std::max(0, tpw_env);
    const long long per_xcd = (total + 7) / 8;
    const unsigned grid = (unsigned)(walk ? resident : a.tpw > 1 ? 8 * ((per_xcd + a.tpw - 1) / a.tpw) : total);
    if (a.tpw == 1) a.tpw = 0;
    bool all = a.valid != nullptr;
    for (int i = 0; i < LG_NUM_MAPS; i++) all = all && a.maps[i] != nullptr;
    const bool vec = (a.W & 3) == 0;
    void (*k)(LgFinalArgs) = nullptr;
#define LG_FINAL_PICK(RR)                                                                                  \
    k = vec ? (all ? lg_final_kernel<true, true, RR> : lg_final_kernel<true, false, RR>)                   \
            : (all ? lg_final_kernel<false, true, RR> : lg_final_kernel<false, false, RR>)
    switch (a.gauss_r) {   // make_plan admits gaussian_size 1, 3, 5, 7 only
        case 0: LG_FINAL_PICK(0); break;
        case 1: LG_FINAL_PICK(1); break;
        case 3: LG_FINAL_PICK(3); break;
        default: LG_FINAL_PICK(2); break;
    }
#undef LG_FINAL_PICK
    if (ev_start && ev_stop)  // events stamped by the command processor right around this dispatch
        hipExtLaunchKernelGGL(k, dim3(grid), dim3(256), 0, s, ev_start, ev_stop, 0, a);
    else
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, s, a);
}

// ============================================================================ the tail of select_grasp_point, per frame
// grasp_point_selector.py:205-245 after the candidates exist: CNN rescoring (ml = tanh(3 sigmoid(logit)) / 2 + 1/2, confidence
// weights, "take it if the combined score beats the best so far", :210-237), get_3d_grasp_point (:152-180) and
// calculate_pre_grasp_point (:754-819: five probes of the leaf mask dilated by the clearance ellipse along the viewing ray,
// fallback at 0.10 m).  One wave per frame; the same float64 operations in the same order as the host code this replaces (fp
// contraction off), the selection loop sequential in lane 0 -- the call no longer ships candidate lists, logits and bit rows to
// the host and walks the frames there (0.1 ms of host work + five copies per 256-frame call, with the GPU idle behind them).
__global__ __launch_bounds__(64) void lg_finish_kernel(LgFinishArgs a) {
#pragma clang fp contract(off)
    __shared__ double s_comb[64];
    __shared__ int s_elig[64], s_best, s_ml;
    __shared__ double s_bs;
    const int b = blockIdx.x, lane = threadIdx.x;
    const int n = a.cand_n[b], K = a.K, H = a.H, W = a.W, WW = a.WW;
    lg_grasp_result R;
    R.found = 0; R.x = 0; R.y = 0; R.X = 0.f; R.Y = 0.f; R.Z = 0.f; R.has_pre = 0; R.pX = 0.f; R.pY = 0.f; R.pZ = 0.f;
    R.n_candidates = n; R.ml_used = 0; R.best_score = 0.f; R.theta = a.fp[b].theta;
    if (n <= 0) {   // reference: "No valid candidate points found" -> (None, None, None)
        if (lane == 0) a.out[b] = R;
        return;
    }
    const int32_t* xy = a.cand_xy + (size_t)b * K * 2;
    const float* info = a.cand_info + (size_t)b * K * 2;
    const bool rescoring = a.use_cnn && n > 1;
    double comb = 0.0;
    int elig = 0;
    if (rescoring && lane < n) {
        const int x = xy[2 * lane], y = xy[2 * lane + 1];
        if (!(a.mask_is_bool && (x < 16 || y < 16 || x + 16 > W || y + 16 > H))) {
            const double logit = (double)a.logits[(size_t)b * K + lane];
            const double sg = 1.0 / (1.0 + exp(-logit));
            const double ml = tanh(sg * 3.0) * 0.5 + 0.5;               // :133-136
            const double conf = 1.0 - fabs(ml - 0.5) * 2.0;              // :222
            const double wml = fmin(0.3, conf * 0.6);                    // :223
            comb = (1.0 - wml) * (double)info[2 * lane] + wml * ml;      // :226
            elig = 1;
        }
    }
    s_comb[lane] = comb;
    s_elig[lane] = elig;
    __syncthreads();
    if (lane == 0) {
        int best = 0, ml_used = 0;
        double best_score = (double)info[0];  // candidate 0's traditional score (:205-206)
        if (rescoring)
            for (int i = 0; i < n; i++)
                if (s_elig[i] && s_comb[i] > best_score) { best_score = s_comb[i]; best = i; ml_used = 1; }
        s_best = best; s_ml = ml_used; s_bs = best_score;
    }
    __syncthreads();
    const int best = s_best;
    R.found = 1;
    R.ml_used = s_ml;
    R.x = xy[2 * best]; R.y = xy[2 * best + 1];
    R.best_score = (float)s_bs;
    // get_3d_grasp_point (:152-180)
    const double Z = (double)info[2 * best + 1];
    const double X = Z * ((double)R.x - a.cx) / a.f;
    const double Y = Z * ((double)R.y - a.cy) / a.f;
    R.X = (float)X; R.Y = (float)Y; R.Z = (float)Z;
    // calculate_pre_grasp_point (:754-819)
    const double nrm = sqrt(X * X + Y * Y + Z * Z);
    if (!(nrm > 0.0) || !isfinite(nrm)) {    // reference: exception -> None
        if (lane == 0) a.out[b] = R;
        return;
    }
    const double dxn = X / nrm, dyn = Y / nrm;
    const unsigned long long* fb = a.bits + (size_t)b * H * WW;
    bool done = false;
    for (int step = 0; step < 5 && !done; step++) {
        // np.arange(0.05, 0.10, 0.01)[step] = start + step * ((start + delta) - start)
        const double dist = 0.05 + (double)step * ((0.05 + 0.01) - 0.05);
        const double tx = X - dxn * dist, ty = Y - dyn * dist, tz = Z;
        const int u = (int)((tx * a.f / tz) + a.cx);
        const int v = (int)((ty * a.f / tz) + a.cy);
        if (!(u >= 0 && u < W && v >= 0 && v < H)) continue;
        // dilated[v, u] != 0  <=>  some set pixel under the ellipse centred there: lane i tests row i of the structuring element
        bool hit = false;
        if (lane < a.se.n && a.se.lo[lane] <= a.se.hi[lane]) {
            const int y = v + lane - a.se.anchor;
            const int x0 = max(u + a.se.lo[lane], 0), x1 = min(u + a.se.hi[lane], W - 1);
            if (y >= 0 && y < H && x0 <= x1) {
                const unsigned long long* row = fb + (size_t)y * WW;
                for (int w = x0 >> 6; w <= (x1 >> 6); w++) {
                    unsigned long long m = ~0ull;
                    if (w == (x0 >> 6)) m &= ~0ull << (x0 & 63);
                    if (w == (x1 >> 6)) m &= ~0ull >> (63 - (x1 & 63));
                    hit |= (row[w] & m) != 0ull;
                }
            }
        }
        if (__ballot(hit) == 0ull) {
            const double dg = sqrt((tx - X) * (tx - X) + (ty - Y) * (ty - Y));
            if (dg >= 0.05) { R.pX = (float)tx; R.pY = (float)ty; R.pZ = (float)tz; done = true; }
        }
    }
    if (!done) { R.pX = (float)(X - dxn * 0.10); R.pY = (float)(Y - dyn * 0.10); R.pZ = (float)Z; }
    R.has_pre = 1;
    if (lane == 0) a.out[b] = R;
}

void lg_launch_finish(const LgFinishArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(lg_finish_kernel, dim3(a.B), dim3(64), 0, s, a);
}

// ============================================================================ ImageProcessor.smooth_depth on its own
// image_processor.py:56-64: F.pad(depth, size // 2 on every side, 'reflect') then F.conv2d with the size x size Gaussian (a
// cross-correlation; the kernel is symmetric).  Output (H + 2P - S + 1) x (W + 2P - S + 1), P = S / 2: the input's shape for odd
// S, one row and one column more for even S (what torch returns there).  Separable: rows, then columns, taps in ascending
// order -- the same two passes as the stencil phase of lg_final_kernel.  Tile 64 x 16 outputs, 256 threads, any 1 <= S <= 15.
__global__ __launch_bounds__(256) void lg_smooth_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, int W,
                                                        int Ho, int Wo, int S, LgGaussTaps taps) {
    constexpr int TW = 64, TH = 16, MAXS = LG_MAX_GAUSS;
    __shared__ float s_in[(TH + MAXS - 1) * (TW + MAXS - 1)];
    __shared__ float s_row[(TH + MAXS - 1) * TW];
    const int t = threadIdx.x, P = S >> 1;
    const int tx0 = blockIdx.x * TW, ty0 = blockIdx.y * TH;
    const float* in = src + (size_t)blockIdx.z * H * W;
    float* out = dst + (size_t)blockIdx.z * Ho * Wo;
    const int IW = TW + S - 1, IH = TH + S - 1;
    for (int i = t; i < IH * IW; i += 256) {
        const int er = i / IW, ec = i - er * IW;
        // output (y, x) reads padded[y + i][x + j] = in[reflect(y + i - P)][reflect(x + j - P)]; rows / columns past the output
        // (partial tiles) are clamped by lg_reflect and never stored
        s_in[i] = in[(size_t)lg_reflect(ty0 + er - P, H) * W + lg_reflect(tx0 + ec - P, W)];
    }
    __syncthreads();
    for (int i = t; i < IH * TW; i += 256) {
        const int er = i >> 6, c = i & 63;
        const float* p = &s_in[er * IW + c];
        float acc = taps.k[0] * p[0];
        for (int j = 1; j < S; j++) acc += taps.k[j] * p[j];
        s_row[i] = acc;
    }
    __syncthreads();
    for (int i = t; i < TH * TW; i += 256) {
        const int r = i >> 6, c = i & 63;
        const int y = ty0 + r, x = tx0 + c;
        if (y < Ho && x < Wo) {
            const float* p = &s_row[r * TW + c];
            float acc = taps.k[0] * p[0];
            for (int j = 1; j < S; j++) acc += taps.k[j] * p[j * TW];
            out[(size_t)y * Wo + x] = acc;
        }
    }
}

void lg_launch_smooth(const float* src, float* dst, int B, int H, int W, int S, const LgGaussTaps& taps, hipStream_t s) {
    const int P = S / 2, Ho = H + 2 * P - S + 1, Wo = W + 2 * P - S + 1;
    hipLaunchKernelGGL(lg_smooth_kernel, dim3((Wo + 63) / 64, (Ho + 15) / 16, B), dim3(256), 0, s, src, dst, H, W, Ho, Wo, S, taps);
}

__global__ __launch_bounds__(256) void lg_tilekeys_kernel(const float* __restrict__ trad,
                                                          const uint8_t* __restrict__ valid,
                                                          unsigned long long* __restrict__ tilekeys, int H, int W,
                                                          int tiles_x, int tiles_y) {
    __shared__ unsigned long long s_key[4];
    const int ntile = tiles_x * tiles_y;
    const int frame = blockIdx.y, tile = blockIdx.x;
    const int bx = tile % tiles_x, by = tile / tiles_x;
    const size_t fo = (size_t)frame * H * W;
    unsigned long long best = 0;
    for (int i = threadIdx.x; i < LG_TW * LG_TH; i += 256) {
        int x = bx * LG_TW + (i % LG_TW), y = by * LG_TH + (i / LG_TW);
        if (x < W && y < H) {
            size_t o = fo + (size_t)y * W + x;
            float sc = lg_valid_score(trad[o], valid[o] != 0);
            unsigned long long key = ((unsigned long long)lg_orderable(sc) << 32) | (uint32_t)(y * W + x);
            best = key > best ? key : best;
        }
    }
    best = lg_wave_max_u64(best);
    if ((threadIdx.x & 63) == 0) s_key[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long k = s_key[0];
        for (int w = 1; w < 4; w++) k = s_key[w] > k ? s_key[w] : k;
        tilekeys[(size_t)frame * ntile + tile] = k;
    }
}

// ============================================================================ greedy spaced top-k
// GraspPointSelector._get_candidate_points (grasp_point_selector.py:447-482) without the full sort:
// repeat k times { argmax over not-yet-suppressed pixels (score desc, flat index desc);
//                  suppress every pixel within Chebyshev distance 2*min_dist of the pick }.
// Equivalent to the reference's greedy walk over the descending argsort (a pixel is accepted iff its
// (2d+1)^2 window meets no earlier window, i.e. iff it is > 2d away from every accepted point).
// One 1024-thread workgroup per frame; per-tile maxima live in LDS and only the tiles touched by a new suppression
// window (<= 8 of 64x16 pixels for the reference's 41x41 window; any number for larger min_distance) are recomputed.
#define LG_TOPK_T 1024
#define LG_MAX_TILES 8192
#define LG_MAX_K 64
__global__ __launch_bounds__(LG_TOPK_T) void lg_topk_kernel(const float* __restrict__ trad,
                                                            const uint8_t* __restrict__ valid,
                                                            const float* __restrict__ depth,
                                                            const unsigned long long* __restrict__ tilekeys, int H,
                                                            int W, int tiles_x, int tiles_y, int k, int md,
                                                            int32_t* __restrict__ out_xy, int32_t* __restrict__ out_n,
                                                            float* __restrict__ out_info) {
    __shared__ unsigned long long s_keys[LG_MAX_TILES];
    __shared__ unsigned long long s_best;
    __shared__ int s_cx[LG_MAX_K], s_cy[LG_MAX_K];
    const int frame = blockIdx.x;
    const int ntile = tiles_x * tiles_y;
    const int t = threadIdx.x;
    const size_t fo = (size_t)frame * H * W;
    const int sup = 2 * md;
    const float inv_w = __frcp_rn((float)W);
    for (int i = t; i < ntile; i += LG_TOPK_T) s_keys[i] = tilekeys[(size_t)frame * ntile + i];
    if (t == 0) s_best = 0;
    __syncthreads();
    int n = 0;
    for (int r = 0; r < k; r++) {
        unsigned long long b = 0;
        for (int i = t; i < ntile; i += LG_TOPK_T) b = s_keys[i] > b ? s_keys[i] : b;
        b = lg_wave_max_u64(b);
        if ((t & 63) == 0 && b) atomicMax(&s_best, b);
        __syncthreads();
        const unsigned long long bk = s_best;
        if (bk == 0) break;  // every pixel is suppressed
        const int idx = (int)(uint32_t)(bk & 0xffffffffull);
        // idx < 2^24 (at most 8192 tiles of 1024 pixels): the float quotient is within 1 of idx / W
        int py = (int)(((float)idx + 0.5f) * inv_w);
        py -= (py * W > idx) ? 1 : 0;
        py += ((py + 1) * W <= idx) ? 1 : 0;
        const int px = idx - py * W;
        if (t == 0) {
            s_cx[r] = px; s_cy[r] = py;
            out_xy[((size_t)frame * k + r) * 2 + 0] = px;
            out_xy[((size_t)frame * k + r) * 2 + 1] = py;
        }
        n = r + 1;
        // tiles touched by the new suppression window
        const int tx_lo = max(px - sup, 0) / LG_TW, tx_hi = min(px + sup, W - 1) / LG_TW;
        const int ty_lo = max(py - sup, 0) / LG_TH, ty_hi = min(py + sup, H - 1) / LG_TH;
        const int ntx = tx_hi - tx_lo + 1, nty = ty_hi - ty_lo + 1;
        const int naff = ntx * nty;
        __syncthreads();  // s_cx/s_cy visible; everyone has read s_best
        if (t == 0) s_best = 0;
        for (int i = t; i < naff; i += LG_TOPK_T)   // any window size: a large min_distance touches > 1024 tiles
            s_keys[(ty_lo + i / ntx) * tiles_x + tx_lo + i % ntx] = 0;
        __syncthreads();
        // earlier picks whose suppression window reaches the affected tiles at all (usually the new pick and a neighbour or
        // two): the per-pixel test below walks these, not all r + 1 picks -- at the late rounds that test was most of the round
        unsigned long long rel = 0;
        {
            const int xlo = tx_lo * LG_TW, xhi = tx_hi * LG_TW + LG_TW - 1, ylo = ty_lo * LG_TH, yhi = ty_hi * LG_TH + LG_TH - 1;
            for (int q = 0; q <= r; q++)
                if (s_cx[q] + sup >= xlo && s_cx[q] - sup <= xhi && s_cy[q] + sup >= ylo && s_cy[q] - sup <= yhi) rel |= 1ull << q;
        }
        constexpr int TPT = LG_TOPK_T / 8;                 // threads per tile on the fast path: 8 tiles x 128 threads x 8 pixels
        static_assert(LG_TW * LG_TH == 8 * TPT, "one tile = 128 threads x 8 pixels");
        if (naff <= 8 && (W & 3) == 0) {
            // The usual case (min_distance 10: a 41 x 41 window touches at most 2 x 4 tiles), kept short: the whole kernel runs on
            // ONE CU per frame, every round is a dependent chain, and the general form below spent ~2500 instructions per wave and
            // round on per-chunk index arithmetic (runtime divisions) -- 15 us of instruction issue per round, 0.3 ms per call at
            // any batch size.  Thread t: tile t >> 7 of the window, row (t & 127) >> 3, eight consecutive pixels: two 16-byte score
            // loads + one 8-byte validity load, all issued before anything is looked at.
            const int ta = t >> 7, u = t & 127;
            if (ta < naff) {
                int ry = 0, rx = ta;
                while (rx >= ntx) { rx -= ntx; ry++; }       // (ntx <= 8: a few wave-uniform iterations instead of a division)
                const int tile = (ty_lo + ry) * tiles_x + tx_lo + rx;
                const int y = (ty_lo + ry) * LG_TH + (u >> 3), x0 = (tx_lo + rx) * LG_TW + (u & 7) * 8;
                unsigned long long best = 0;
                if (y < H && x0 < W) {
                    const size_t o = fo + (size_t)y * W + x0;
                    const bool two = x0 + 4 < W;             // W % 4 == 0: groups of four pixels are inside or outside as a whole
                    const float4 s0 = *reinterpret_cast<const float4*>(trad + o);
                    const float4 s1 = two ? *reinterpret_cast<const float4*>(trad + o + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                    const uint32_t v0 = *reinterpret_cast<const uint32_t*>(valid + o);
                    const uint32_t v1 = two ? *reinterpret_cast<const uint32_t*>(valid + o + 4) : 0u;
                    const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
                    // rows: which of the relevant picks reach this row at all (the column test is then per pixel)
                    unsigned long long rowrel = 0;
                    for (unsigned long long m = rel; m; m &= m - 1) {
                        const int q = __builtin_ctzll(m);
                        if (abs(y - s_cy[q]) <= sup) rowrel |= 1ull << q;
                    }
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const int x = x0 + j;
                        const bool inside = j < 4 || two;
                        const uint32_t vb = ((j < 4 ? v0 : v1) >> (8 * (j & 3))) & 0xffu;
                        bool dead = false;
                        for (unsigned long long m = rowrel; m; m &= m - 1) dead |= abs(x - s_cx[__builtin_ctzll(m)]) <= sup;
                        if (inside && !dead) {
                            const unsigned long long key = ((unsigned long long)lg_orderable(lg_valid_score(sc[j], vb != 0)) << 32) | (uint32_t)(y * W + x);
                            best = key > best ? key : best;
                        }
                    }
                }
                best = lg_wave_max_u64(best);
                if ((t & 63) == 0 && best) atomicMax(&s_keys[tile], best);
            }
            __syncthreads();
            continue;
        }
        // General form (any window size, any width): 1024-pixel chunks, one per tile; the loads of up to 12 chunks are issued
        // together (one round trip to L2 / HBM instead of one per chunk).
        constexpr int CPT = LG_TW * LG_TH / LG_TOPK_T;
        const int chunks = naff * CPT;
        constexpr int GRP = 12;
        for (int cb = 0; cb < chunks; cb += GRP) {
            float sc_[GRP];
            int idx_[GRP], tile_[GRP], x_[GRP], y_[GRP];
            unsigned off_[GRP];
            // addresses first, then every load of the group back to back with nothing conditional on a loaded value in between
            // (written as `valid[o] ? trad[o] : 0` per chunk, hipcc issued byte load -> wait -> branch -> float load -> wait for
            // each chunk)
#pragma unroll
            for (int g = 0; g < GRP; g++) {
                const int c = min(cb + g, chunks - 1);
                const int ta = c / CPT;
                const int tile = (ty_lo + ta / ntx) * tiles_x + tx_lo + ta % ntx;
                const int li = (c % CPT) * LG_TOPK_T + t;
                const int x = (tile % tiles_x) * LG_TW + (li % LG_TW), y = (tile / tiles_x) * LG_TH + (li / LG_TW);
                const bool inb = x < W && y < H;
                tile_[g] = tile; x_[g] = x; y_[g] = y;
                idx_[g] = inb ? y * W + x : -1;
                off_[g] = inb ? (unsigned)(y * W + x) : 0u;   // (outside the image: pixel 0 of the frame, loaded and ignored)
            }
            uint8_t vv_[GRP];
            float tt_[GRP];
#pragma unroll
            for (int g = 0; g < GRP; g++) {
                vv_[g] = __builtin_nontemporal_load(valid + fo + off_[g]);
                tt_[g] = __builtin_nontemporal_load(trad + fo + off_[g]);
            }
#pragma unroll
            for (int g = 0; g < GRP; g++) sc_[g] = idx_[g] >= 0 ? lg_valid_score(tt_[g], vv_[g] != 0) : 0.0f;
#pragma unroll
            for (int g = 0; g < GRP; g++) {
                if (cb + g < chunks) {   // uniform across the workgroup
                    uint32_t score = 0;   // orderable(anything alive) >= 0x00800000 > 0: zero = suppressed / outside the image
                    if (idx_[g] >= 0) {
                        const int x = x_[g], y = y_[g];
                        bool dead = false;
                        for (unsigned long long m = rel; m; m &= m - 1) {
                            const int q = __builtin_ctzll(m);
                            dead |= (abs(x - s_cx[q]) <= sup) && (abs(y - s_cy[q]) <= sup);
                        }
                        if (!dead) score = lg_orderable(sc_[g]);
                    }
                    // within a chunk the flat index grows with the lane (li = chunk * 1024 + t): the cheaper lane-ordered arg-max
                    const unsigned long long key = lg_wave_argmax_lane_ordered(score, (uint32_t)idx_[g]);
                    if ((t & 63) == 0 && key) atomicMax(&s_keys[tile_[g]], key);
                }
            }
        }
        __syncthreads();
    }
    if (t == 0) out_n[frame] = n;
    // traditional score + depth at the picks, gathered once after the walk (a dependent global load inside
    // every round would sit on the critical path of the next arg-max)
    if (out_info && t < n) {
        const size_t o = fo + (size_t)s_cy[t] * W + s_cx[t];
        out_info[((size_t)frame * k + t) * 2 + 0] = trad[o];
        out_info[((size_t)frame * k + t) * 2 + 1] = depth ? depth[o] : 0.0f;
    }
}

void lg_launch_topk(const float* trad, const uint8_t* valid, const float* depth, unsigned long long* tilekeys,
                    bool keys_ready, int B, int H, int W, int k, int min_dist, int32_t* out_xy, int32_t* out_n,
                    float* out_info, hipStream_t s) {
    int tiles_x = (W + LG_TW - 1) / LG_TW, tiles_y = (H + LG_TH - 1) / LG_TH;
    if (!keys_ready)
        hipLaunchKernelGGL(lg_tilekeys_kernel, dim3(tiles_x * tiles_y, B), dim3(256), 0, s, trad, valid, tilekeys, H, W,
                           tiles_x, tiles_y);
    hipLaunchKernelGGL(lg_topk_kernel, dim3(B), dim3(LG_TOPK_T), 0, s, trad, valid, depth, tilekeys, H, W, tiles_x,
                       tiles_y, k, min_dist, out_xy, out_n, out_info);
}

// ============================================================================ 9-channel patch gather
// get_ml_score feature assembly (grasp_point_selector.py:59-127): 32x32 window [y-16,y+16) x [x-16,x+16),
// replicate-padded (:392-445); channel 0 depth and channels 2..8 (sdf, approach, flatness, isolation,
// distance, accessibility, stem) are min-max normalised per patch when max > min; channel 1 = raw mask.
struct LgGatherMaps { const float* p[7]; };
// HALO: write the interior of the first 9 of the 12 haloed planes [12][34][36] (pixel (y,x) at [y+1][x+1]; the CNN's staging layout, lg_cnn.hip;
// the halo itself is zeroed once when the workspace is allocated) instead of dense [9][32][32].
template <bool HALO>
__global__ __launch_bounds__(256) void lg_gather_kernel(const float* __restrict__ depth,
                                                        const uint8_t* __restrict__ mask, LgGatherMaps maps, int H,
                                                        int W, int k, const int32_t* __restrict__ xy,
                                                        const int32_t* __restrict__ n, float* __restrict__ patches) {
    __shared__ float s_mn[4], s_mx[4];
    constexpr int PL = HALO ? 34 * 36 : 1024, RP = HALO ? 36 : 32, O0 = HALO ? 37 : 0;
    const int ci = blockIdx.x, frame = blockIdx.y;
    const int t = threadIdx.x;
    float* outp = patches + ((size_t)frame * k + ci) * (HALO ? 12 : 9) * PL + O0;   // haloed patches carry 3 more (zero) planes
    if (ci >= n[frame]) {
        for (int c = 0; c < 9; c++)
            for (int i = t; i < 1024; i += 256) outp[c * PL + (i >> 5) * RP + (i & 31)] = 0.0f;
        return;
    }
    const int px = xy[((size_t)frame * k + ci) * 2], py = xy[((size_t)frame * k + ci) * 2 + 1];
    const size_t fo = (size_t)frame * H * W;
    for (int c = 0; c < 9; c++) {
        float v[4];
        float mn = INFINITY, mx = -INFINITY;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            int i = t + 256 * q;
            int yy = py - 16 + (i >> 5), xx = px - 16 + (i & 31);
            yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
            xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
            size_t o = fo + (size_t)yy * W + xx;
            float val = (c == 0) ? depth[o] : (c == 1) ? (mask[o] ? 1.0f : 0.0f) : maps.p[c - 2][o];
            v[q] = val;
            mn = fminf(mn, val);
            mx = fmaxf(mx, val);
        }
        if (c != 1) {
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                mn = fminf(mn, __shfl_xor(mn, o, 64));
                mx = fmaxf(mx, __shfl_xor(mx, o, 64));
            }
            __syncthreads();  // previous channel's readers are done
            if ((t & 63) == 0) { s_mn[t >> 6] = mn; s_mx[t >> 6] = mx; }
            __syncthreads();
            mn = fminf(fminf(s_mn[0], s_mn[1]), fminf(s_mn[2], s_mn[3]));
            mx = fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3]));
            if (mx > mn) {
                const float range = mx - mn;
#pragma unroll
                for (int q = 0; q < 4; q++) v[q] = (v[q] - mn) / range;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int i = t + 256 * q;
            outp[c * PL + (i >> 5) * RP + (i & 31)] = v[q];
        }
        if (HALO && t < 200) {
            // the plane's 200 halo floats, zero as they already are: a cache line the kernel writes only in part leaves
            // the L2 as a masked write (tools/ubench/partial_lines.hip)
            float* pl = outp + c * PL - O0;
            const int hi = t < 72 ? (t < 36 ? t : 33 * 36 + (t - 36))              // top and bottom rows
                                  : (t < 104 ? (t - 72 + 1) * 36                     // left column of rows 1..32
                                             : ((t - 104) / 3 + 1) * 36 + 33 + (t - 104) % 3);   // columns 33..35
            pl[hi] = 0.0f;
        }
    }
}

void lg_launch_gather(const float* depth, const uint8_t* mask, const float* const* maps_host, int B, int H, int W, int k,
                      const int32_t* xy, const int32_t* n, float* patches, bool haloed, hipStream_t s) {
    LgGatherMaps gm;
    for (int i = 0; i < 7; i++) gm.p[i] = maps_host[i];
    if (haloed) hipLaunchKernelGGL(lg_gather_kernel<true>, dim3(k, B), dim3(256), 0, s, depth, mask, gm, H, W, k, xy, n, patches);
    else hipLaunchKernelGGL(lg_gather_kernel<false>, dim3(k, B), dim3(256), 0, s, depth, mask, gm, H, W, k, xy, n, patches);
}

// ============================================================================ training-sample harvesting
// EnhancedGraspDataCollector (scripts/utils/ml_grasp_optimizer/data_collector.py): raw, un-normalised 32x32 windows
// [y-16, y+16) x [x-16, x+16) of depth, mask and the seven score planes around n points of ONE frame
// (_extract_patches :91-173), optionally rotated by rot[i] quarter turns like torch.rot90(t, k, dims=(-2,-1))
// (_generate_augmented_samples :250-293).  flags[i]: bit 0 non-finite depth, bit 1 empty mask patch, bit 2 non-finite
// score value, bit 3 window not inside the frame (nothing is written for that point).
__global__ __launch_bounds__(256) void lg_harvest_kernel(const float* __restrict__ depth, const uint8_t* __restrict__ mask,
                                                         LgGatherMaps maps, int H, int W, const int32_t* __restrict__ xy,
                                                         const int32_t* __restrict__ rot, float* __restrict__ out_depth,
                                                         float* __restrict__ out_mask, float* __restrict__ out_scores,
                                                         int32_t* __restrict__ flags) {
    __shared__ int s_flag, s_any;
    const int i = blockIdx.x, t = threadIdx.x;
    const int px = xy[2 * i], py = xy[2 * i + 1];
    const int k = rot ? (rot[i] & 3) : 0;
    if (t == 0) { s_flag = 0; s_any = 0; }
    __syncthreads();
    if (px < 16 || py < 16 || px + 16 > W || py + 16 > H) {   // _check_boundaries :83-89
        if (t == 0) flags[i] = 8;
        return;
    }
    int fl = 0, any = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int o = t + 256 * q, r = o >> 5, c = o & 31;
        // torch.rot90: k=1: out[r][c] = in[c][31-r]; k=2: in[31-r][31-c]; k=3: in[31-c][r]
        const int sr = k == 0 ? r : k == 1 ? c : k == 2 ? 31 - r : 31 - c;
        const int sc = k == 0 ? c : k == 1 ? 31 - r : k == 2 ? 31 - c : r;
        const size_t src = (size_t)(py - 16 + sr) * W + (px - 16 + sc);
        const float d = depth[src];
        const float m = mask[src] ? 1.0f : 0.0f;
        if (!isfinite(d)) fl |= 1;
        any |= (m != 0.0f);
        out_depth[(size_t)i * 1024 + o] = d;
        out_mask[(size_t)i * 1024 + o] = m;
#pragma unroll
        for (int ch = 0; ch < 7; ch++) {
            const float v = maps.p[ch][src];
            if (!isfinite(v)) fl |= 4;
            out_scores[((size_t)i * 7 + ch) * 1024 + o] = v;
        }
    }
    if (fl) atomicOr(&s_flag, fl);
    if (any) atomicOr(&s_any, 1);
    __syncthreads();
    if (t == 0) flags[i] = s_flag | (s_any ? 0 : 2);
}

void lg_launch_harvest(const float* depth, const uint8_t* mask, const float* const* maps_host, int H, int W, int n,
                       const int32_t* xy, const int32_t* rot, float* out_depth, float* out_mask, float* out_scores,
                       int32_t* flags, hipStream_t s) {
    LgGatherMaps gm;
    for (int i = 0; i < 7; i++) gm.p[i] = maps_host[i];
    hipLaunchKernelGGL(lg_harvest_kernel, dim3(n), dim3(256), 0, s, depth, mask, gm, H, W, xy, rot, out_depth, out_mask,
                       out_scores, flags);
}

// Negative-sample regions (data_collector.py:426-466):
//   tip  = (cv2.dilate(dist, ones(5,5)) == dist) & mask      local maxima of the distance transform (:430-437)
//   stem = erode(erode(mask with rows < int(0.75 H) cleared, ellipse 5x5), ellipse 5x5)  (:449-456), iteration `pass`
// cv2's morphology border is "ignore": a dilation never sees values outside the frame, an erosion is not eroded by it.
__global__ __launch_bounds__(256) void lg_tip_kernel(const float* __restrict__ dist, const uint8_t* __restrict__ mask,
                                                     uint8_t* __restrict__ tip, int H, int W) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const float d = dist[(size_t)y * W + x];
    float mx = d;
    for (int dy = -2; dy <= 2; dy++) {
        const int yy = y + dy;
        if (yy < 0 || yy >= H) continue;
        for (int dx = -2; dx <= 2; dx++) {
            const int xx = x + dx;
            if (xx >= 0 && xx < W) mx = fmaxf(mx, dist[(size_t)yy * W + xx]);
        }
    }
    tip[(size_t)y * W + x] = (mx == d && mask[(size_t)y * W + x]) ? 1 : 0;
}
__global__ __launch_bounds__(256) void lg_erode5_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int H,
                                                        int W, int first_row) {
    // 5x5 MORPH_ELLIPSE: rows -2 / +2 hold the centre column only, rows -1..1 all five columns; rows < first_row of src
    // count as background (stem_region[:int(0.75*height)] = 0)
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    bool keep = true;
    for (int dy = -2; dy <= 2 && keep; dy++) {
        const int yy = y + dy;
        if (yy < 0 || yy >= H) continue;
        const int r = (dy == -2 || dy == 2) ? 0 : 2;
        for (int dx = -r; dx <= r; dx++) {
            const int xx = x + dx;
            if (xx < 0 || xx >= W) continue;
            if (yy < first_row || !src[(size_t)yy * W + xx]) { keep = false; break; }
        }
    }
    dst[(size_t)y * W + x] = keep ? 1 : 0;
}

void lg_launch_negative_masks(const float* dist, const uint8_t* mask, uint8_t* tip, uint8_t* stem, uint8_t* scratch, int H,
                              int W, hipStream_t s) {
    dim3 grid((W + 63) / 64, (H + 3) / 4), block(256);
    hipLaunchKernelGGL(lg_tip_kernel, grid, block, 0, s, dist, mask, tip, H, W);
    const int first_row = (int)(0.75 * (double)H);   // int(0.75 * height)
    hipLaunchKernelGGL(lg_erode5_kernel, grid, block, 0, s, mask, scratch, H, W, first_row);
    hipLaunchKernelGGL(lg_erode5_kernel, grid, block, 0, s, scratch, stem, H, W, 0);
}
