// Leaf orientation on the device: GraspPointSelector.estimate_leaf_orientation
// (scripts/utils/grasp_point_selector.py:718-752) = cv2.findContours(RETR_EXTERNAL) -> max by cv2.contourArea ->
// cv2.minAreaRect -> direction of the longer side, per frame, from the bit rows of the mask's bounding box.
//
// One workgroup per frame, the same steps as the host analysis (lg_contour.cpp) and the same arithmetic:
//   A  runs of every bounding-box row (popcount of the run starts, workgroup prefix sum, extraction)
//   B  8-connected components of the runs: union-find in LDS (hook the larger root under the smaller with atomicMin; a
//      component's root is its first run in raster order = the start of its outer border)
//   C  more than one component: outer-border following per component (one lane each, Suzuki-Abe like the host), shoelace
//      sum in int64 (exact: the host's double sum never rounds either); largest area wins, first in raster order on ties
//   D  leftmost / rightmost pixel of the chosen component per row: only these can be hull vertices
//   E  strict vertices of the left (convex) and right (concave) chains by exact integer slope comparisons, compacted in the
//      order Andrew's monotone chain produces (counter-clockwise from the lexicographically smallest point)
//   F  min-area rectangle over the hull edges in float64, no fused multiply-add (the host code has none): first minimum in
//      edge order, then atan2 / fmod, sin and cos for the score-plane kernel
// A frame with more runs than the scratch holds (or more hull vertices than LDS holds) gets status 1: the caller analyses
// that frame with the host code.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <string>

#include "lg_internal.h"
#include "lg_orient.h"

namespace {

typedef unsigned long long u64;
constexpr int OT = 256;     // threads per frame
constexpr int OHULL = 2048; // hull vertices held in LDS (a convex lattice polygon in a 16384-box has < 2300 / 2 of them)

struct Band { const u64* rows; int hy, W, WW, w0, w1; };   // rows -> image row by0; only words [w0, w1] hold leaf pixels

__device__ __forceinline__ u64 band_word(const Band& bd, int y, int w) {
    u64 v = (y < 0 || y >= bd.hy || w < bd.w0 || w > bd.w1) ? 0ull : bd.rows[(size_t)y * bd.WW + w];
    if (w == bd.WW - 1 && (bd.W & 63)) v &= (~0ull) >> (64 - (bd.W & 63));
    return v;
}

// pixels x-1, x, x+1 of band row y in bits 0..2
__device__ __forceinline__ unsigned row3(const Band& bd, int y, int x) {
    if (y < 0 || y >= bd.hy) return 0;
    const int xa = x - 1, xb = x + 1;
    const int wa = xa >> 6, wb = xb >> 6, wm = x >> 6;   // xa = -1 -> word -1: outside
    const u64 lo = band_word(bd, y, wa);
    const u64 hi = wb == wa ? lo : band_word(bd, y, wb);
    unsigned r = 0;
    if (xa >= 0) r |= (unsigned)((lo >> (xa & 63)) & 1ull);
    r |= (unsigned)((((wm == wa) ? lo : hi) >> (x & 63)) & 1ull) << 1;
    if (xb < bd.W) r |= (unsigned)((hi >> (xb & 63)) & 1ull) << 2;
    return r;
}

// neighbours of (x, y), bit d = direction d: clockwise on the screen starting at West (W NW N NE E SE S SW), lg_contour.cpp
__device__ __forceinline__ unsigned ring_at(const Band& bd, int x, int y) {
    const unsigned up = row3(bd, y - 1, x), mid = row3(bd, y, x), dn = row3(bd, y + 1, x);
    return (mid & 1u) | ((up & 1u) << 1) | (((up >> 1) & 1u) << 2) | (((up >> 2) & 1u) << 3) | (((mid >> 2) & 1u) << 4) |
           (((dn >> 2) & 1u) << 5) | (((dn >> 1) & 1u) << 6) | ((dn & 1u) << 7);
}
__device__ __forceinline__ int dir_dx(int d) { return (int)((0x1A90u >> (2 * d)) & 3u) - 1; }   // -1 -1 0 1 1 1 0 -1
__device__ __forceinline__ int dir_dy(int d) { return (int)((0xA901u >> (2 * d)) & 3u) - 1; }   //  0 -1 -1 -1 0 1 1 1

// twice the shoelace area of the outer border that starts at (sx, sy), the top-most / left-most pixel of its component
__device__ long long trace_area2(const Band& bd, int sx, int sy) {
    unsigned ring = ring_at(bd, sx, sy);
    if (!ring) return 0;                                 // single pixel
    const int first = __builtin_ctz(ring);               // first neighbour clockwise from West
    const int fx = sx + dir_dx(first), fy = sy + dir_dy(first);
    int cx = sx, cy = sy, dprev = first, lastx = 0, lasty = 0;
    bool have_last = false;
    long long acc = 0, guard = 8ll * ((long long)bd.hy * bd.W + 16);
    for (;;) {
        // counter-clockwise from the previous pixel: directions dprev-1, dprev-2, ... = bits 7, 6, ... of the rotated ring
        const unsigned r = ((ring >> dprev) | (ring << (8 - dprev))) & 0xffu;
        if (have_last) acc += (long long)lastx * cy - (long long)cx * lasty;
        lastx = cx; lasty = cy; have_last = true;
        if (!r) break;
        const int m = 31 - __builtin_clz(r);
        const int d = (dprev - (8 - m)) & 7;
        const int nx = cx + dir_dx(d), ny = cy + dir_dy(d);
        if (nx == sx && ny == sy && cx == fx && cy == fy) break;
        cx = nx; cy = ny; dprev = (d + 4) & 7;
        ring = ring_at(bd, cx, cy);
        if (--guard < 0) break;
    }
    acc += (long long)lastx * sy - (long long)sx * lasty;
    return acc < 0 ? -acc : acc;
}

// exclusive prefix sum of one int per thread; total to every thread
__device__ int block_scan(int v, int* s, int* total) {
    const int t = threadIdx.x;
    s[t] = v;
    __syncthreads();
    for (int d = 1; d < OT; d <<= 1) {
        const int a = t >= d ? s[t - d] : 0;
        __syncthreads();
        s[t] += a;
        __syncthreads();
    }
    const int incl = s[t];
    *total = s[OT - 1];
    __syncthreads();
    return incl - v;
}

__device__ __forceinline__ int uf_find(int* p, int i) {
    int r;
    while ((r = __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) != i) i = r;
    return i;
}
__device__ void uf_union(int* p, int a, int b) {
    for (;;) {
        a = uf_find(p, a);
        b = uf_find(p, b);
        if (a == b) return;
        if (a > b) { const int t = a; a = b; b = t; }
        const int old = atomicMin(p + b, a);   // b was a root: hook it under the smaller root
        if (old == b) return;
        b = old;                               // somebody hooked b meanwhile: (a, old) still have to meet
    }
}

struct FrameScratch {
    uint32_t* run_x;   // x0 | x1 << 16
    uint16_t* run_y;   // row inside the band
    int* roots;
    int* row_start;    // [hy + 1]
    short* rowL;       // leftmost / rightmost pixel of the chosen component per band row (-1: none)
    short* rowR;
    unsigned char* vflag;   // [2][hy] strict hull vertex on the left / right chain
    int* comp;              // [4][cap] per component (indexed by its root run): interior pixels, min x, max x, last row
};

__device__ void write_none(LgFrameParams* fp, double* out, int* status, int b, int st) {
    LgFrameParams f;
    f.sin_t = 0.f; f.cos_t = 0.f; f.has_angle = 0; f.theta = __builtin_nanf("");
    fp[b] = f;
    for (int i = 0; i < 5; i++) out[5 * (size_t)b + i] = __builtin_nan("");
    status[b] = st;
}

__global__ __launch_bounds__(OT) void lg_orient_kernel(const u64* __restrict__ bits, const LgWin* __restrict__ win, int H, int W,
                                                       int WW, int cap, uint32_t* run_x_, uint16_t* run_y_, int* roots_,
                                                       int* row_start_, short* rowL_, short* rowR_, unsigned char* vflag_,
                                                       int* comp_, LgFrameParams* fp, double* out, int* status) {
#pragma clang fp contract(off)
    extern __shared__ __align__(16) unsigned char s_dyn[];   // parent[cap] | L[n], R[n] shorts | hull x[OHULL], y[OHULL] doubles
    __shared__ int s_scan[OT];
    __shared__ int s_nroots, s_ybot, s_rank, s_lb, s_ncand, s_cand;
    __shared__ u64 s_key, s_p0;
    const int b = blockIdx.x, tid = threadIdx.x;
    const LgWin w = win[b];
    const int hyb = w.by1 - w.by0 + 1;
    if (w.bx1 < w.bx0 || hyb <= 0) {                      // empty mask: the reference returns None
        if (tid == 0) write_none(fp, out, status, b, 0);
        return;
    }
    FrameScratch fs;
    fs.run_x = run_x_ + (size_t)b * cap; fs.run_y = run_y_ + (size_t)b * cap; fs.roots = roots_ + (size_t)b * cap;
    fs.row_start = row_start_ + (size_t)b * (H + 1); fs.rowL = rowL_ + (size_t)b * H; fs.rowR = rowR_ + (size_t)b * H;
    fs.vflag = vflag_ + (size_t)b * 2 * H;
    fs.comp = comp_ + (size_t)b * 4 * cap;
    Band bd;
    bd.rows = bits + ((size_t)b * H + w.by0) * WW; bd.hy = hyb; bd.W = W; bd.WW = WW; bd.w0 = w.bx0 >> 6; bd.w1 = w.bx1 >> 6;
    if (tid == 0) { s_nroots = 0; s_ybot = 0; s_key = 0; s_p0 = ~0ull; s_rank = 0x7fffffff; s_lb = 0; s_ncand = 0; s_cand = 0; }
    // ---- A: runs.  Thread t owns the rows [r0, r1)
    const int chunk = (hyb + OT - 1) / OT, r0 = min(tid * chunk, hyb), r1 = min(r0 + chunk, hyb);
    int sum = 0;
    for (int r = r0; r < r1; r++) {
        int cnt = 0;
        u64 in = 0;
        for (int q = bd.w0; q <= bd.w1; q++) {
            const u64 v = band_word(bd, r, q);
            cnt += __popcll(v & ~((v << 1) | in));
            in = v >> 63;
        }
        fs.row_start[r] = cnt;
        sum += cnt;
    }
    int R = 0;
    int base = block_scan(sum, s_scan, &R);
    if (R > cap) {
        if (tid == 0) write_none(fp, out, status, b, 1);
        return;
    }
    int* parent = (int*)s_dyn;
    for (int r = r0; r < r1; r++) {
        const int cnt = fs.row_start[r];
        fs.row_start[r] = base;
        int idx = base, start = 0;
        base += cnt;
        u64 in = 0;
        for (int q = bd.w0; q <= bd.w1; q++) {
            const u64 v = band_word(bd, r, q);
            u64 tr = v ^ ((v << 1) | in);               // bit k: pixel k differs from pixel k-1
            bool inside = in != 0;
            while (tr) {
                const int x = q * 64 + __builtin_ctzll(tr);
                tr &= tr - 1;
                if (!inside) { inside = true; start = x; }
                else {
                    inside = false;
                    fs.run_x[idx] = (uint32_t)start | ((uint32_t)(x - 1) << 16);
                    fs.run_y[idx] = (uint16_t)r;
                    parent[idx] = idx;
                    idx++;
                }
            }
            in = v >> 63;
        }
        if (in) {
            fs.run_x[idx] = (uint32_t)start | ((uint32_t)min(W - 1, 64 * (bd.w1 + 1) - 1) << 16);
            fs.run_y[idx] = (uint16_t)r;
            parent[idx] = idx;
        }
    }
    if (tid == 0) fs.row_start[hyb] = R;
    __syncthreads();
    // ---- B: union runs that touch the previous row (8-connectivity: overlap after growing by one pixel)
    for (int i = tid; i < R; i += OT) {
        const int r = fs.run_y[i];
        if (r == 0) continue;
        const uint32_t xi = fs.run_x[i];
        const int x0 = (int)(xi & 0xffffu), x1 = (int)(xi >> 16);
        int lo = fs.row_start[r - 1];
        const int hi = fs.row_start[r];
        int l = lo, h = hi;                              // first run a of the previous row with x1a + 1 >= x0
        while (l < h) {
            const int m = (l + h) >> 1;
            if ((int)(fs.run_x[m] >> 16) + 1 >= x0) h = m; else l = m + 1;
        }
        for (int a = l; a < hi; a++) {
            if ((int)(fs.run_x[a] & 0xffffu) > x1 + 1) break;
            uf_union(parent, a, i);
        }
    }
    __syncthreads();
    for (int i = tid; i < R; i += OT) {
        const int root = uf_find(parent, i);
        if (root == i) {
            fs.roots[atomicAdd(&s_nroots, 1)] = i;
            fs.comp[i] = 0; fs.comp[cap + i] = 0x7fffffff; fs.comp[2 * cap + i] = -1; fs.comp[3 * cap + i] = -1;
        }
    }
    __syncthreads();
    for (int i = tid; i < R; i += OT) parent[i] = uf_find(parent, i);   // (a root stays a root: concurrent reads still end there)
    const int nroots = s_nroots;
    __syncthreads();
    // ---- C: the component with the largest outer contour
    int best;
    if (nroots == 1) {
        best = fs.roots[0];
    } else {
        // Which components can hold the largest outer contour at all?  Following a border is serial (one lane per component, a
        // dependent chain of bit-row loads per step: ~1 us per border pixel), so a leaf with a few specks beside it would cost a
        // few thousand steps for a decision that is obvious.  Two bounds on a component's contour area A (the area of the polygon
        // through the centres of its outer border pixels): A >= the number of its pixels whose whole 3 x 3 neighbourhood is set
        // (their unit squares are disjoint and lie inside that polygon) and A <= (width - 1) (height - 1) of its bounding box.
        // A component whose upper bound is below the best lower bound is out -- it cannot even tie.  One survivor: no border is
        // followed at all.
        for (int i = tid; i < R; i += OT) {
            const int root = parent[i], r = fs.run_y[i];
            const uint32_t xi = fs.run_x[i];
            const int x0 = (int)(xi & 0xffffu), x1 = (int)(xi >> 16);
            int inner = 0;
            if (x1 - x0 >= 2 && r > 0 && r < hyb - 1) {
                for (int q = x0 >> 6; q <= (x1 >> 6); q++) {
                    const u64 f0 = band_word(bd, r - 1, q - 1) & band_word(bd, r, q - 1) & band_word(bd, r + 1, q - 1);
                    const u64 f1 = band_word(bd, r - 1, q) & band_word(bd, r, q) & band_word(bd, r + 1, q);
                    const u64 f2 = band_word(bd, r - 1, q + 1) & band_word(bd, r, q + 1) & band_word(bd, r + 1, q + 1);
                    u64 in3 = f1 & ((f1 << 1) | (f0 >> 63)) & ((f1 >> 1) | (f2 << 63));
                    const int lo = max(x0 - q * 64, 0), hi = min(x1 - q * 64, 63);
                    in3 &= (~0ull << lo) & (~0ull >> (63 - hi));
                    inner += __popcll(in3);
                }
            }
            if (inner) atomicAdd(&fs.comp[root], inner);
            atomicMin(&fs.comp[cap + root], x0);
            atomicMax(&fs.comp[2 * cap + root], x1);
            atomicMax(&fs.comp[3 * cap + root], r);
        }
        __threadfence_block();
        __syncthreads();
        for (int k = tid; k < nroots; k += OT) atomicMax(&s_lb, fs.comp[fs.roots[k]]);
        __syncthreads();
        const long long lb2 = 2ll * s_lb;
        for (int k = tid; k < nroots; k += OT) {
            const int root = fs.roots[k];
            const long long ub2 = 2ll * (fs.comp[2 * cap + root] - fs.comp[cap + root]) * (fs.comp[3 * cap + root] - (int)fs.run_y[root]);
            if (ub2 >= lb2) { atomicAdd(&s_ncand, 1); atomicMax(&s_cand, root); } else fs.roots[k] = -1;
        }
        __syncthreads();
        if (s_ncand == 1) {
            if (tid == 0) s_key = (u64)(0xfffff - s_cand);
        } else {
            for (int k = tid; k < nroots; k += OT) {
                const int root = fs.roots[k];
                if (root < 0) continue;
                const long long a2 = trace_area2(bd, (int)(fs.run_x[root] & 0xffffu), (int)fs.run_y[root]);
                atomicMax(&s_key, ((u64)a2 << 20) | (u64)(0xfffff - root));   // a2 <= 2 H W < 2^29; equal areas: first in raster order
            }
        }
        __syncthreads();
        best = 0xfffff - (int)(s_key & 0xfffffull);
    }
    // ---- D: per row, leftmost / rightmost pixel of that component
    for (int r = r0; r < r1; r++) {
        int L = -1, Rr = -1;
        for (int i = fs.row_start[r]; i < fs.row_start[r + 1]; i++)
            if (parent[i] == best) {
                const uint32_t xi = fs.run_x[i];
                if (L < 0) L = (int)(xi & 0xffffu);
                Rr = (int)(xi >> 16);
            }
        fs.rowL[r] = (short)L; fs.rowR[r] = (short)Rr;
        if (L >= 0) atomicMax(&s_ybot, r);
    }
    __syncthreads();                                      // parent[] is dead from here on
    const int ytop = fs.run_y[best], n = s_ybot - ytop + 1;   // a component covers a contiguous range of rows
    short* Ls = (short*)s_dyn;
    short* Rs = Ls + n;
    for (int i = tid; i < n; i += OT) { Ls[i] = fs.rowL[ytop + i]; Rs[i] = fs.rowR[ytop + i]; }
    __syncthreads();
    // ---- E: strict hull vertices.  Left chain = lower convex envelope of L over the rows: row i is a vertex iff the
    // steepest slope arriving from above is smaller than the flattest slope leaving downwards; right chain mirrored.
    for (int i = tid; i < n; i += OT) {
        bool vl = true, vr = true;
        if (i > 0 && i < n - 1) {
            const int li = Ls[i], ri = Rs[i];
            int lp = li - Ls[0], lq = i, rp = ri - Rs[0], rq = i;              // arriving slopes (x_i - x_a) / (i - a)
            for (int a = 1; a < i; a++) {
                const int q = i - a, pl = li - Ls[a], pr = ri - Rs[a];
                if (pl * lq > lp * q) { lp = pl; lq = q; }                         // max for the left chain
                if (pr * rq < rp * q) { rp = pr; rq = q; }                         // min for the right chain
            }
            int lp2 = Ls[i + 1] - li, lq2 = 1, rp2 = Rs[i + 1] - ri, rq2 = 1;  // leaving slopes (x_b - x_i) / (b - i)
            for (int c = i + 2; c < n; c++) {
                const int q = c - i, pl = Ls[c] - li, pr = Rs[c] - ri;
                if (pl * lq2 < lp2 * q) { lp2 = pl; lq2 = q; }                     // min
                if (pr * rq2 > rp2 * q) { rp2 = pr; rq2 = q; }                     // max
            }
            vl = lp * lq2 < lp2 * lq;
            vr = rp * rq2 > rp2 * rq;
        }
        // the chains meet at the top and bottom rows: a shared end point is listed once (with the left chain)
        if ((i == 0 || i == n - 1) && Ls[i] == Rs[i]) vr = false;
        fs.vflag[i] = vl; fs.vflag[H + i] = vr;
    }
    __syncthreads();
    // order: left chain bottom -> top, then right chain top -> bottom (Andrew's monotone chain on (x, y) walks the hull
    // this way round); the start at the lexicographically smallest point is applied as a rotation below
    const int chn = (n + OT - 1) / OT, i0 = min(tid * chn, n), i1 = min(i0 + chn, n);
    int cl = 0, cr = 0;
    for (int i = i0; i < i1; i++) { cl += fs.vflag[i]; cr += fs.vflag[H + i]; }
    int totL = 0, totR = 0;
    int bl = block_scan(cl, s_scan, &totL);
    int br = block_scan(cr, s_scan, &totR);
    const int nh = totL + totR;
    if (nh > OHULL) {
        if (tid == 0) write_none(fp, out, status, b, 1);
        return;
    }
    double* hx = (double*)s_dyn;                          // (Ls / Rs are dead: every thread passed the scans' barriers)
    double* hyv = hx + OHULL;
    for (int i = i0; i < i1; i++) {
        const int yabs = w.by0 + ytop + i;
        if (fs.vflag[i]) {
            const int pos = totL - 1 - bl;
            bl++;
            const int x = fs.rowL[ytop + i];
            hx[pos] = (double)x; hyv[pos] = (double)yabs;
            atomicMin(&s_p0, ((u64)x << 32) | ((u64)yabs << 16) | (u64)pos);
        }
        if (fs.vflag[H + i]) {
            const int pos = totL + br;
            br++;
            hx[pos] = (double)fs.rowR[ytop + i]; hyv[pos] = (double)yabs;   // (never the smallest point: the left chain holds it)
        }
    }
    __syncthreads();
    if (nh == 1) {
        if (tid == 0) {
            LgFrameParams f;
            const double ang = M_PI / 2;
            f.sin_t = (float)sin(ang); f.cos_t = (float)cos(ang); f.has_angle = 1; f.theta = (float)ang;
            fp[b] = f;
            double* o = out + 5 * (size_t)b;
            o[0] = ang; o[1] = 0; o[2] = 0; o[3] = hx[0]; o[4] = hyv[0];
            status[b] = 0;
        }
        return;
    }
    // ---- F: min-area rectangle, one hull edge per thread (edge e starts at hull vertex e counted from the smallest point)
    const int pos0 = (int)(s_p0 & 0xffffull);
    double m_area = 1e300, m_ux = 0, m_uy = 0, m_w = 0, m_h = 0, m_sc = 0, m_tc = 0;
    int m_rank = 0x7fffffff;
    const int nedges = nh == 2 ? 1 : nh;
    for (int e = tid; e < nedges; e += OT) {
        int i = pos0 + e;
        if (i >= nh) i -= nh;
        const int i2 = i + 1 >= nh ? i + 1 - nh : i + 1;
        const double ex = hx[i2] - hx[i], ey = hyv[i2] - hyv[i], len = sqrt(ex * ex + ey * ey);
        if (len == 0) continue;
        const double ux = ex / len, uy = ey / len;
        double smin = 1e300, smax = -1e300, tmin = 1e300, tmax = -1e300;
        int j = pos0;
        for (int c = 0; c < nh; c++) {
            const double px = hx[j], py = hyv[j];
            const double s = px * ux + py * uy;
            const double t = -px * uy + py * ux;
            smin = s < smin ? s : smin; smax = s > smax ? s : smax;
            tmin = t < tmin ? t : tmin; tmax = t > tmax ? t : tmax;
            if (++j >= nh) j = 0;
        }
        const double area = (smax - smin) * (tmax - tmin);
        if (area < m_area) {
            m_area = area; m_rank = e; m_ux = ux; m_uy = uy;
            m_w = smax - smin; m_h = tmax - tmin;
            m_sc = 0.5 * (smin + smax); m_tc = 0.5 * (tmin + tmax);
        }
    }
    if (tid == 0) s_key = ~0ull;
    __syncthreads();
    if (m_rank != 0x7fffffff) atomicMin(&s_key, (u64)__double_as_longlong(m_area));   // areas are >= 0: bit order = value order
    __syncthreads();
    if (m_rank != 0x7fffffff && (u64)__double_as_longlong(m_area) == s_key) atomicMin(&s_rank, m_rank);
    __syncthreads();
    if (s_rank == 0x7fffffff) {                            // (cannot happen: distinct vertices give edges of non-zero length)
        if (tid == 0) write_none(fp, out, status, b, 1);
        return;
    }
    if (m_rank == s_rank) {
        const double bang = atan2(m_uy, m_ux);
        double ang = (m_w < m_h) ? bang + M_PI / 2 : bang;
        ang = fmod(ang, M_PI);
        if (ang <= 0) ang += M_PI;
        LgFrameParams f;
        f.sin_t = (float)sin(ang); f.cos_t = (float)cos(ang); f.has_angle = 1; f.theta = (float)ang;
        fp[b] = f;
        double* o = out + 5 * (size_t)b;
        o[0] = ang; o[1] = m_w > m_h ? m_w : m_h; o[2] = m_w < m_h ? m_w : m_h;
        o[3] = m_sc * m_ux - m_tc * m_uy; o[4] = m_sc * m_uy + m_tc * m_ux;
        status[b] = 0;
    }
}

template <typename T>
hipError_t dalloc(T** p, size_t n) { return hipMalloc((void**)p, n * sizeof(T)); }

}  // namespace

void lg_orient_free(LgOrientWs*& w) {
    if (!w) return;
    hipFree(w->run_x); hipFree(w->run_y); hipFree(w->roots); hipFree(w->row_start); hipFree(w->rowL); hipFree(w->rowR);
    hipFree(w->vflag); hipFree(w->comp); hipFree(w->out); hipFree(w->status);
    if (w->h_out) hipHostFree(w->h_out);
    if (w->h_status) hipHostFree(w->h_status);
    delete w;
    w = nullptr;
}

int lg_orient_ensure(LgOrientWs*& w, int B, int H, std::string* err) {
    if (w && B <= w->capB && H == w->H) return LG_OK;
    const int nB = std::max(B, w ? w->capB : 0);
    hipDeviceSynchronize();
    lg_orient_free(w);
    w = new LgOrientWs();
    const char* e = getenv("LG_ORIENT_CAP");   // runs per frame held by the scratch (tests lower it to reach the host hand-off)
    w->cap = e ? std::max(16, std::min(atoi(e), 16384)) : 16384;
    const size_t nb = (size_t)nB;
    hipError_t rc = hipSuccess;
    auto A = [&](hipError_t r) { if (rc == hipSuccess) rc = r; };
    A(dalloc(&w->run_x, nb * w->cap)); A(dalloc(&w->run_y, nb * w->cap)); A(dalloc(&w->roots, nb * w->cap));
    A(dalloc(&w->row_start, nb * (H + 1))); A(dalloc(&w->rowL, nb * H)); A(dalloc(&w->rowR, nb * H));
    A(dalloc(&w->vflag, nb * 2 * H)); A(dalloc(&w->comp, nb * 4 * w->cap)); A(dalloc(&w->out, nb * 5)); A(dalloc(&w->status, nb));
    A(hipHostMalloc((void**)&w->h_out, sizeof(double) * 5 * nb));
    A(hipHostMalloc((void**)&w->h_status, sizeof(int) * nb));
    if (rc != hipSuccess) {
        if (err) *err = std::string("orientation scratch: ") + hipGetErrorString(rc);
        lg_orient_free(w);
        return LG_ERR_NOMEM;
    }
    w->capB = nB; w->H = H;
    // LDS: parent[cap] ints, later two short rows [H] and the hull (2 * OHULL doubles)
    w->lds = std::max(std::max((size_t)w->cap * 4, (size_t)H * 4), (size_t)OHULL * 16);
    rc = hipFuncSetAttribute((const void*)lg_orient_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)w->lds);
    if (rc != hipSuccess) {
        if (err) *err = std::string("orientation kernel LDS: ") + hipGetErrorString(rc);
        lg_orient_free(w);
        return LG_ERR_HIP;
    }
    return LG_OK;
}

// frames [off, off + n) of the workspace: bits / win / fp point at frame `off`
void lg_launch_orient(LgOrientWs* w, const unsigned long long* bits, const LgWin* win, LgFrameParams* fp, int off, int n, int H,
                      int W, int WW, hipStream_t s) {
    const size_t o = (size_t)off;
    hipLaunchKernelGGL(lg_orient_kernel, dim3(n), dim3(OT), w->lds, s, bits, win, H, W, WW, w->cap, w->run_x + o * w->cap,
                       w->run_y + o * w->cap, w->roots + o * w->cap, w->row_start + o * (H + 1), w->rowL + o * H, w->rowR + o * H,
                       w->vflag + o * 2 * H, w->comp + o * 4 * w->cap, fp, w->out + 5 * o, w->status + o);
}
