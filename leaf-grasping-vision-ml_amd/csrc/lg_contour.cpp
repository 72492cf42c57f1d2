// Host-side leaf orientation on the bit-packed mask (product code; NOT the oracle).
//
// Semantics of GraspPointSelector.estimate_leaf_orientation (scripts/utils/grasp_point_selector.py:718-752):
//   cv2.findContours(mask, RETR_EXTERNAL, CHAIN_APPROX_NONE) -> max by cv2.contourArea -> cv2.minAreaRect,
//   "if size[0] < size[1]: angle += 90"  ==> direction of the LONGER side of the min-area rectangle of the
//   largest outer contour, reported in (0, pi].
// Implementation: run-length connected components (8-connectivity) on 64-bit rows, outer-border following
// only when more than one component competes for "largest contour area", convex hull of the run end points
// (identical to the hull of the contour), exhaustive min-area rectangle over hull edges.
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "lg_internal.h"

namespace {

struct Run { int y, x0, x1, parent; };  // inclusive span
struct P2 { long long x, y; };

// only the words [w0, w1] of a row are valid (the host image holds the bounding box of the leaf, everything else is 0)
inline bool bit_at(const unsigned long long* bits, int H, int W, int WW, int w0, int w1, int x, int y) {
    if (x < 0 || y < 0 || x >= W || y >= H) return false;
    const int w = x >> 6;
    if (w < w0 || w > w1) return false;
    return (bits[(size_t)y * WW + w] >> (x & 63)) & 1ull;
}

int find_root(std::vector<Run>& r, int i) {
    while (r[i].parent != i) {
        r[i].parent = r[r[i].parent].parent;
        i = r[i].parent;
    }
    return i;
}

// Shoelace area of the 8-connected outer border that starts at (sx, sy) (top-most/left-most pixel
// of its component), traced through pixel centres like Suzuki-Abe border following.
double outer_border_area(const unsigned long long* bits, int H, int W, int WW, int w0, int w1, int sx, int sy,
                         std::vector<int>* rec = nullptr) {
    // neighbours clockwise (screen coordinates, y down) starting at West
    static const int dx[8] = {-1, -1, 0, 1, 1, 1, 0, -1};
    static const int dy[8] = {0, -1, -1, -1, 0, 1, 1, 1};
    int first = -1;
    for (int k = 0; k < 8; k++)
        if (bit_at(bits, H, W, WW, w0, w1, sx + dx[k], sy + dy[k])) { first = k; break; }
    if (first < 0) {  // single pixel
        if (rec) { rec->push_back(sx); rec->push_back(sy); }
        return 0.0;
    }
    const int fx = sx + dx[first], fy = sy + dy[first];
    long long px = fx, py = fy, cx = sx, cy = sy;
    double acc = 0.0;
    long long lastx = 0, lasty = 0;
    bool have_last = false;
    long long firstx = sx, firsty = sy;
    long long guard = 8LL * ((long long)H * W + 16);
    for (;;) {
        int dprev = 0;
        for (int k = 0; k < 8; k++)
            if (cx + dx[k] == px && cy + dy[k] == py) { dprev = k; break; }
        long long nx = cx, ny = cy;
        bool found = false;
        for (int k = 1; k <= 8; k++) {
            int d = (dprev - k) & 7;  // counter-clockwise from the previous pixel
            if (bit_at(bits, H, W, WW, w0, w1, (int)cx + dx[d], (int)cy + dy[d])) {
                nx = cx + dx[d];
                ny = cy + dy[d];
                found = true;
                break;
            }
        }
        if (have_last) acc += (double)lastx * (double)cy - (double)cx * (double)lasty;
        lastx = cx; lasty = cy; have_last = true;
        if (rec) { rec->push_back((int)cx); rec->push_back((int)cy); }
        if (!found) break;
        if (nx == sx && ny == sy && cx == fx && cy == fy) break;
        px = cx; py = cy; cx = nx; cy = ny;
        if (--guard < 0) break;
    }
    acc += (double)lastx * (double)firsty - (double)firstx * (double)lasty;  // close the polygon
    return fabs(acc) * 0.5;
}

inline double crossp(const P2& o, const P2& a, const P2& b) {
    return (double)(a.x - o.x) * (double)(b.y - o.y) - (double)(a.y - o.y) * (double)(b.x - o.x);
}

}  // namespace

void lg_make_se_spans(int k, LgSeSpans* out) {
    // cv2.getStructuringElement(MORPH_ELLIPSE, (k,k)): r = c = k/2, row i: dy = i - r,
    // dx = round(c * sqrt((r*r - dy*dy) / (r*r))), ones in [max(c-dx,0), min(c+dx+1,k))
    memset(out, 0, sizeof(*out));
    if (k > 64) k = 64;
    out->n = k;
    out->anchor = k / 2;
    const int r = k / 2, c = k / 2;
    const double inv_r2 = r ? 1.0 / ((double)r * r) : 0.0;
    for (int i = 0; i < k; i++) {
        int j1 = 0, j2 = 0;
        const int dy = i - r;
        if (k == 1) {
            j2 = 1;
        } else if (abs(dy) <= r) {
            int dx = (int)lrint(c * sqrt((r * r - dy * dy) * inv_r2));
            j1 = std::max(c - dx, 0);
            j2 = std::min(c + dx + 1, k);
        }
        if (j2 > j1) {
            out->lo[i] = (signed char)(j1 - out->anchor);
            out->hi[i] = (signed char)(j2 - 1 - out->anchor);
        } else {
            out->lo[i] = 1;
            out->hi[i] = 0;
        }
    }
}

int lg_host_ellipse_hit_se(const unsigned long long* bits, int H, int W, int WW, int u, int v, const LgSeSpans& se) {
    return lg_host_ellipse_hit_band(bits, H, W, WW, 0, WW - 1, u, v, se);
}

int lg_host_ellipse_hit_band(const unsigned long long* bits, int H, int W, int WW, int w0, int w1, int u, int v,
                             const LgSeSpans& se) {
    // dilated[v,u] != 0  <=>  some set pixel (u+dx, v+dy) with (dy,dx) in the (2c+1) ellipse
    // (calculate_pre_grasp_point, grasp_point_selector.py:777-779,804); word-wise span tests on the bit rows
    for (int i = 0; i < se.n; i++) {
        int y = v + i - se.anchor;
        if (y < 0 || y >= H || se.lo[i] > se.hi[i]) continue;
        int x0 = std::max(u + se.lo[i], 0), x1 = std::min(u + se.hi[i], W - 1);
        if (x0 > x1) continue;
        const unsigned long long* row = bits + (size_t)y * WW;
        for (int w = std::max(x0 >> 6, w0); w <= std::min(x1 >> 6, w1); w++) {
            unsigned long long m = ~0ull;
            if (w == (x0 >> 6)) m &= ~0ull << (x0 & 63);
            if (w == (x1 >> 6)) m &= ~0ull >> (63 - (x1 & 63));
            if (row[w] & m) return 1;
        }
    }
    return 0;
}

int lg_host_ellipse_hit(const unsigned long long* bits, int H, int W, int WW, int u, int v, int clearance) {
    LgSeSpans se;
    lg_make_se_spans(2 * clearance + 1, &se);
    return lg_host_ellipse_hit_se(bits, H, W, WW, u, v, se);
}

int lg_host_orientation(const unsigned long long* bits, int H, int W, int WW, double* out) {
    return lg_host_orientation_band(bits, H, W, WW, 0, 0, WW - 1, out);
}

int lg_host_orientation_rows(const unsigned long long* bits, int H, int W, int WW, int y_off, double* out) {
    return lg_host_orientation_band(bits, H, W, WW, y_off, 0, WW - 1, out);
}

// Same analysis on a band of rows and words: `bits` points at image row y_off, H is the band height and only the words
// [w0, w1] of a row are read (every other pixel of the image is empty).  Hull and rectangle are computed in ABSOLUTE image coordinates, so ties between candidate rectangles
// (decided by the last bits of double arithmetic) fall exactly as they do for the whole image.
// runs of the band + the root run of the component with the largest outer contour (cv2.findContours EXTERNAL +
// max(contourArea)); returns -1 for an empty band
static int lg_best_component(const unsigned long long* bits, int H, int W, int WW, int w0, int w1, std::vector<Run>& runs) {
    std::vector<int> row_start(H + 1, 0);
    // ---- runs per row
    for (int y = 0; y < H; y++) {
        row_start[y] = (int)runs.size();
        const unsigned long long* row = bits + (size_t)y * WW;
        bool in = false;
        int start = 0;
        for (int w = w0; w <= w1; w++) {
            unsigned long long v = row[w];
            if (w == WW - 1 && (W & 63)) v &= (~0ull) >> (64 - (W & 63));  // ignore padding bits past W
            // transitions inside the word: bit b of tr is set when pixel b differs from pixel b-1
            unsigned long long tr = v ^ ((v << 1) | (in ? 1ull : 0ull));
            while (tr) {
                const int b = __builtin_ctzll(tr);
                tr &= tr - 1;
                const int x = w * 64 + b;
                if (!in) { in = true; start = x; }
                else { in = false; runs.push_back({y, start, x - 1, (int)runs.size()}); }
            }
        }
        if (in) runs.push_back({y, start, std::min(W - 1, 64 * (w1 + 1) - 1), (int)runs.size()});
    }
    row_start[H] = (int)runs.size();
    if (runs.empty()) return -1;
    // ---- union runs that touch (8-connectivity: overlap after growing by one pixel)
    for (int y = 1; y < H; y++) {
        int a = row_start[y - 1], ae = row_start[y], b = row_start[y], be = row_start[y + 1];
        while (a < ae && b < be) {
            if (runs[a].x1 + 1 >= runs[b].x0 && runs[b].x1 + 1 >= runs[a].x0) {
                int ra = find_root(runs, a), rb = find_root(runs, b);
                if (ra != rb) runs[std::max(ra, rb)].parent = std::min(ra, rb);
            }
            if (runs[a].x1 < runs[b].x1) a++; else b++;
        }
    }
    // ---- components: root = first run in raster order (min index) => its (x0, y) is the border start
    std::vector<int> roots;
    for (int i = 0; i < (int)runs.size(); i++)
        if (find_root(runs, i) == i) roots.push_back(i);
    int best_root = roots[0];
    if (roots.size() > 1) {
        double best_area = -1.0;
        for (int r : roots) {
            double a = outer_border_area(bits, H, W, WW, w0, w1, runs[r].x0, runs[r].y);
            if (a > best_area) { best_area = a; best_root = r; }
        }
    }
    return best_root;
}

int lg_host_orientation_band(const unsigned long long* bits, int H, int W, int WW, int y_off, int w0, int w1, double* out) {
    std::vector<Run> runs;
    const int best_root = lg_best_component(bits, H, W, WW, w0, w1, runs);
    if (best_root < 0) return 0;
    // ---- hull of the component's run end points (Andrew monotone chain, same orientation/ordering
    //      convention as the documented spec: sort by x then y, counter-clockwise, collinear dropped)
    std::vector<P2> pts;
    for (int i = 0; i < (int)runs.size(); i++)
        if (find_root(runs, i) == best_root) {
            pts.push_back({runs[i].x0, runs[i].y + y_off});
            if (runs[i].x1 != runs[i].x0) pts.push_back({runs[i].x1, runs[i].y + y_off});
        }
    std::sort(pts.begin(), pts.end(), [](const P2& a, const P2& b) { return a.x != b.x ? a.x < b.x : a.y < b.y; });
    std::vector<P2> hull(2 * pts.size() + 2);
    int n = (int)pts.size(), k = 0;
    if (n < 3) {
        for (int i = 0; i < n; i++) hull[k++] = pts[i];
    } else {
        for (int i = 0; i < n; i++) {
            while (k >= 2 && crossp(hull[k - 2], hull[k - 1], pts[i]) <= 0) k--;
            hull[k++] = pts[i];
        }
        for (int i = n - 2, t = k + 1; i >= 0; i--) {
            while (k >= t && crossp(hull[k - 2], hull[k - 1], pts[i]) <= 0) k--;
            hull[k++] = pts[i];
        }
        k--;
    }
    const int nh = k;
    if (nh == 1) {
        out[0] = M_PI / 2; out[1] = 0; out[2] = 0; out[3] = (double)hull[0].x; out[4] = (double)hull[0].y;
        return 1;
    }
    double min_area = 1e300, bw = 0, bh = 0, bang = 0, bcx = 0, bcy = 0;
    // structure-of-arrays copy of the hull: the O(nh^2) projection loop below is the hot spot of the host hand-off
    // (nh ~ 150 at 1080p) and vectorises in this form; same expressions, same values
    std::vector<double> hx(nh), hy(nh);
    for (int j = 0; j < nh; j++) { hx[j] = (double)hull[j].x; hy[j] = (double)hull[j].y; }
    for (int i = 0; i < nh; i++) {
        const P2 p = hull[i], q = hull[(i + 1) % nh];
        double ex = (double)(q.x - p.x), ey = (double)(q.y - p.y), len = sqrt(ex * ex + ey * ey);
        if (len == 0) continue;
        double ux = ex / len, uy = ey / len;
        double smin = 1e300, smax = -1e300, tmin = 1e300, tmax = -1e300;
        const double* __restrict__ px = hx.data();
        const double* __restrict__ py = hy.data();
        for (int j = 0; j < nh; j++) {
            const double s = px[j] * ux + py[j] * uy;
            const double t = -px[j] * uy + py[j] * ux;
            smin = s < smin ? s : smin; smax = s > smax ? s : smax;
            tmin = t < tmin ? t : tmin; tmax = t > tmax ? t : tmax;
        }
        double area = (smax - smin) * (tmax - tmin);
        if (area < min_area) {
            double sc = 0.5 * (smin + smax), tc = 0.5 * (tmin + tmax);
            min_area = area;
            bw = smax - smin; bh = tmax - tmin;
            bang = atan2(uy, ux);
            bcx = sc * ux - tc * uy;
            bcy = sc * uy + tc * ux;
        }
        if (nh == 2) break;
    }
    double ang = (bw < bh) ? bang + M_PI / 2 : bang;
    ang = fmod(ang, M_PI);
    if (ang <= 0) ang += M_PI;
    out[0] = ang;
    out[1] = std::max(bw, bh);
    out[2] = std::min(bw, bh);
    out[3] = bcx;
    out[4] = bcy;
    return 1;
}

// Outer contour of the component with the largest contour area, every border pixel in tracing order (cv2.findContours
// RETR_EXTERNAL / CHAIN_APPROX_NONE restated: start at the component's top-most, then left-most pixel, first look from
// the West neighbour clockwise, then walk counter-clockwise from the previous pixel).  xy receives x0,y0,x1,y1,...
int lg_host_contour_points(const unsigned long long* bits, int H, int W, int WW, std::vector<int>& xy) {
    std::vector<Run> runs;
    xy.clear();
    const int best_root = lg_best_component(bits, H, W, WW, 0, WW - 1, runs);
    if (best_root < 0) return 0;
    outer_border_area(bits, H, W, WW, 0, WW - 1, runs[best_root].x0, runs[best_root].y, &xy);
    return (int)(xy.size() / 2);
}

// ---- the node's result message (leaf_grasp_node_v3.py:170-176): "x,y,X,Y,Z[,pX,pY,pZ]" with Python's str() of every number.
// For the floats that is repr(float(np.float32)): the SHORTEST decimal string that reads back as the same double, in fixed
// notation while -4 <= exponent < 16 (always with a fractional part: "5.0") and as d[.ddd]e+XX (at least two exponent digits)
// outside -- CPython's float_repr_style 'short'.  std::to_chars yields the same shortest digits (both pick the closest of the
// shortest candidates); the layout rules are applied here.  One '\n'-terminated line per frame, empty when nothing was found.
#include <charconv>
#include <cstring>

#include "../../include/leafgrasp.h"

namespace {
char* put_py_float(char* p, float f32) {
    const double v = (double)f32;
    if (v != v) { memcpy(p, "nan", 3); return p + 3; }
    if (v == HUGE_VAL) { memcpy(p, "inf", 3); return p + 3; }
    if (v == -HUGE_VAL) { memcpy(p, "-inf", 4); return p + 4; }
    char tmp[40];
    const std::to_chars_result r = std::to_chars(tmp, tmp + sizeof(tmp), v, std::chars_format::scientific);
    const char* q = tmp;
    if (*q == '-') *p++ = *q++;
    char dig[24];
    int nd = 0;
    for (; q < r.ptr && *q != 'e'; q++)
        if (*q != '.') dig[nd++] = *q;
    int e10 = 0;
    {
        q++;   // 'e'
        const bool neg = *q == '-';
        q++;   // sign
        for (; q < r.ptr; q++) e10 = 10 * e10 + (*q - '0');
        if (neg) e10 = -e10;
    }
    if (nd == 1 && dig[0] == '0') e10 = 0;   // 0.0 / -0.0
    if (e10 < -4 || e10 >= 16) {             // d[.ddd]e+XX
        *p++ = dig[0];
        if (nd > 1) { *p++ = '.'; memcpy(p, dig + 1, nd - 1); p += nd - 1; }
        *p++ = 'e';
        *p++ = e10 < 0 ? '-' : '+';
        const int a = e10 < 0 ? -e10 : e10;
        if (a >= 100) *p++ = (char)('0' + a / 100);
        *p++ = (char)('0' + (a / 10) % 10);
        *p++ = (char)('0' + a % 10);
    } else if (e10 < 0) {                    // 0.000ddd
        *p++ = '0'; *p++ = '.';
        for (int i = 0; i < -e10 - 1; i++) *p++ = '0';
        memcpy(p, dig, nd); p += nd;
    } else {                                 // ddd.ddd / ddd000.0
        const int ni = e10 + 1;
        for (int i = 0; i < ni; i++) *p++ = i < nd ? dig[i] : '0';
        *p++ = '.';
        if (nd > ni) { memcpy(p, dig + ni, nd - ni); p += nd - ni; }
        else *p++ = '0';
    }
    return p;
}
char* put_int(char* p, int v) {
    const std::to_chars_result r = std::to_chars(p, p + 12, v);
    return r.ptr;
}
}  // namespace

extern "C" int lg_format_grasp_results(const lg_grasp_result* res, int n, char* buf, int64_t cap, int64_t* used) {
    if (!res || n < 0 || !buf || !used) return LG_ERR_INVALID;
    char* p = buf;
    char* const end = buf + cap;
    for (int i = 0; i < n; i++) {
        if (end - p < 8 * 32) return LG_ERR_INVALID;   // (a line is at most 2 x 11 + 6 x 25 + 8 characters)
        const lg_grasp_result& r = res[i];
        if (r.found) {
            p = put_int(p, r.x); *p++ = ',';
            p = put_int(p, r.y); *p++ = ',';
            p = put_py_float(p, r.X); *p++ = ',';
            p = put_py_float(p, r.Y); *p++ = ',';
            p = put_py_float(p, r.Z);
            if (r.has_pre) {
                *p++ = ','; p = put_py_float(p, r.pX);
                *p++ = ','; p = put_py_float(p, r.pY);
                *p++ = ','; p = put_py_float(p, r.pZ);
            }
        }
        *p++ = '\n';
    }
    *used = p - buf;
    return LG_OK;
}
