// Internal declarations shared by the HIP translation units of liblgrasp.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/leafgrasp.h"

// ---- chamfer constants: 16.16 fixed point of OpenCV's DIST_L2 masks ("Distance transform")
#define LG_A5 65536u        // 1.0
#define LG_B5 91750u        // 1.4
#define LG_C5 143976u       // 2.1969
#define LG_A3 62587u        // 0.955
#define LG_B3 89738u        // 1.3693
#define LG_INIT0 0x1FFFFFFFu  // INT_MAX >> 2 : OpenCV's border initialiser, the default of lg_params.chamfer_init_dist0
#define LG_INF 0x3FFFFFFFu    // "no path yet" inside the DT sweeps (never wins against a real distance)
#define LG_NOSRC 0x20000000u  // values >= this after the backward sweep mean: image has no source pixel
#define LG_HCAP 16383u        // run distance of a row without any zero pixel in the image (row search; W <= 8192)
// lg_bbox_kernel, search_mode 2: a batch's d_in comes from the row search while sum over its frames of area^1.5 <= this *
// (rows of its tallest bounding box)
#ifndef LG_SEARCH_BUDGET
#define LG_SEARCH_BUDGET 2.0e7f
#endif
struct LgDtBatch { unsigned long long cost; unsigned rows, done; };   // the sums behind that decision (device; lg_bbox_kernel resets them)


#define LG_MAX_GAUSS 15   // largest smoothing kernel lg_smooth_depth takes (the fused plane kernel: 1, 3, 5, 7)
struct LgGaussTaps { float k[LG_MAX_GAUSS]; };   // 1-D factor of ImageProcessor's Gaussian (by value in the kernel arguments)

// ---- final-kernel tile
#define LG_TW 64
#ifndef LG_TH
#define LG_TH 16   // measured at 1080p B=128: 16 -> 1.78 ms (0.69 of HBM peak), 32 -> 1.86 ms, 64 -> 2.11 ms (round 1); round 2, whole library
                   // built with -DLG_TH=32: planes 2.60-2.71 vs 2.66-2.74 ms per 256 frames, top-k 0.33 vs 0.27 ms, dense launch the same
#endif

// Window of the distance-transform sweeps (per frame, written by lg_bbox_kernel).  The binary leaf mask covers a few
// per cent of a frame, so the sweeps run on the tile-aligned window around its bounding box only:
//   d_in  (distance to the nearest zero pixel): every pixel outside the window is itself a zero pixel -> exactly 0, and
//         enters the window as a 0-valued halo: the windowed two-pass recurrence is the full-frame one, bit for bit.
//   d_out (distance to the leaf; only max d_out is consumed): the two-pass 5x5 chamfer transform equals
//         min over sources of the chamfer NORM (weights satisfy 2a <= c <= a+b, 3b <= 2c), shortest paths stay inside the
//         bounding rectangle of their end points, so the window computed as a stand-alone image is exact inside; outside
//         the window the distance grows monotonically towards the frame border, where lg_dout_border_kernel evaluates
//         the norm in closed form against the leaf's row / column profiles.
// Columns [wx0, wx0 + nw * 64 * E) (clipped to W), rows [wy0, wy1): wx0 % 64 == 0, wy0 % LG_TH == 0, so every tile of
// the fused score-plane kernel is either inside (reads distance_map) or outside (writes zeros to it).
struct LgWin {
    int wx0, nw, wy0, wy1;   // window: first column, active waves (64*E columns each), row range
    int bx0, bx1, by0, by1;  // bounding box of the mask (bx1 < bx0: empty mask -> window = whole frame)
    int skip_out;            // 1: max d_out cannot lie inside the window (see lg_bbox_kernel): the d_out sweeps of this frame are skipped
    int search_in;           // 1: d_in of this frame comes from the row search (lg_hrun_kernel + lg_dtsearch_kernel), its d_in sweeps are skipped
    int area;                // set bits of the mask
    int pad_[1];
};

struct LgFrameParams {  // per frame: leaf orientation, written by lg_orient_kernel (or by the host analysis for frames it hands back)
    float sin_t, cos_t;
    int has_angle;
    float theta;
};

struct LgSeSpans {  // run-length form of an elliptical structuring element (one span per SE row)
    int n;          // rows
    int anchor;     // k/2
    signed char lo[64];  // first set column - anchor (inclusive); lo > hi => empty row
    signed char hi[64];  // last set column - anchor (inclusive)
};

struct LgFinalArgs {
    const float* depth;
    const unsigned long long* bits;
    const unsigned long long* stem_bits;
    const uint32_t* maxfix;       // [B][2] max fixed-point d_in / d_out
    const LgWin* win;             // [B] sweep windows
    int win_wc;                   // columns per sweep wave (64 * E)
    const LgFrameParams* fp;      // [B]
    float* maps[LG_NUM_MAPS];     // [B][H][W] each (may be null except DISTANCE/TRADITIONAL)
    uint8_t* valid;               // [B][H][W] or null
    unsigned long long* tilekeys; // [B][tiles]
    int B, H, W, WW, tiles_x, tiles_y;
    int cxi, cyi;      // floor of the optical centre; (x - cxi) is exact, the fraction is subtracted afterwards
    float cxf, cyf, f;  // fractions in [0,1) and the focal length
    float w_approach, w_sdf, w_flat, w_access;
    float sdf_w_interior, sdf_w_align, sdf_w_sdf, optimal_distance;
    float access_w_dist, access_w_dir, flat_scale;
    float iso_w_close, iso_w_wide, iso_ramp_top, iso_ramp_bottom, iso_inv_max;
    float min_edge_distance, stem_valid_thresh;
    float inv_maxd;
    uint32_t init0;    // lg_params.chamfer_init_dist0
    float inv_2s2, iso_ramp_step;   // 1 / (2 optimal_distance^2) (float32 reciprocal), (ramp_bottom - ramp_top) / (H - 1)
    float k1[7];  // separable 1-D Gaussian: 2 * gauss_r + 1 taps, sigma = size / 6 (image_processor.py:25-32)
    int gauss_r;    // radius of that Gaussian: 0..3 (lg_params.gaussian_size 1, 3, 5, 7)
    int no_skip;    // bit 0: tile-level constant path off, bit 1: wave-level off-leaf shortcut off (LG_NO_SKIP; same results, A/B timing)
    int nt_stores;  // 0: plain stores (default); 1: non-temporal plane stores (LG_NT_STORES=1; measured slower)
    int persist;    // 1: resident workgroups walk the tiles (lg_launch_final); 0: one workgroup per tile
    int tpw;        // > 0: consecutive tiles per workgroup (set by lg_launch_final)
};

// kernel launchers (lg_kernels.hip)
void lg_launch_pack_bits(const uint8_t* mask, unsigned long long* bits, int B, int H, int W, int WW, hipStream_t s);
// mask[b] = labels[b] == ids[b] (0 / 1 bytes) and its bit rows in one pass (ids: DEVICE, one per frame)
void lg_launch_pack_labels(const int16_t* labels, const int32_t* ids_dev, uint8_t* mask, unsigned long long* bits, int B, int H,
                           int W, int WW, hipStream_t s);
void lg_launch_export_rows(const unsigned long long* bits, const LgWin* wins, unsigned long long* dst_host_devptr, int B,
                           int H, int WW, hipStream_t s);
void lg_launch_stem_bits(const unsigned long long* bits, unsigned long long* stem, int B, int H, int W, int WW,
                         int bottom_start, const LgSeSpans& se, hipStream_t s);
// columns per sweep wave / waves per sweep workgroup for width W (0 if unsupported)
int lg_dt_geometry(int W, int* waves);
// search_mode: 0 = d_in by the two sweeps for every frame, 1 = by the row search wherever it applies (a non-empty mask with
// at least one zero pixel), 2 = the row search when the batch's estimated search work stays below the sweeps' latency
void lg_launch_bbox(const unsigned long long* bits, LgWin* win, int B, int H, int W, int WW, int search_mode, LgDtBatch* batch,
                    hipStream_t s);
// d_in without the row-sequential sweeps (frames with LgWin::search_in): horizontal run distances of the bounding-box rows into
// `tmp` (the d_in half of the sweep workspace, as uint16), then the bounded search over rows, which writes distance_map inside
// the window and the maximum into maxfix[b][0]
void lg_launch_hrun(const unsigned long long* bits, uint32_t* tmp, const LgWin* win, int B, int H, int W, int WW, hipStream_t s);
// algo 1: one-level search (phase 0 only); algo 2: phase 0 = anchor rows (every 8th), phase 1 = the rows between them
int lg_launch_dtsearch(int phase, int algo, const unsigned long long* bits, uint32_t* tmp, float* dist_out, uint32_t* maxfix,
                       const LgWin* win, int B, int H, int W, int WW, hipStream_t s);
int lg_launch_dt(bool bwd, const uint8_t* mask, uint32_t* tmp, float* dist_out, uint32_t* maxfix, const LgWin* win, int B,
                 int H, int W, uint32_t init0, hipStream_t s);   // init0: lg_params.chamfer_init_dist0 (frames without a zero pixel)
// max d_out outside the sweep windows (closed-form chamfer norm on the frame border) -> atomicMax into maxfix[b][1]
void lg_launch_dout_border(const unsigned long long* bits, const LgWin* win, uint32_t* maxfix, int B, int H, int W, int WW,
                           hipStream_t s);
void lg_launch_final(const LgFinalArgs& a, hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
void lg_launch_smooth(const float* src, float* dst, int B, int H, int W, int S, const LgGaussTaps& taps, hipStream_t s);
void lg_launch_topk(const float* trad, const uint8_t* valid, const float* depth, unsigned long long* tilekeys,
                    bool keys_ready, int B, int H, int W, int k, int min_dist, int32_t* out_xy, int32_t* out_n,
                    float* out_info, hipStream_t s);
void lg_launch_gather(const float* depth, const uint8_t* mask, const float* const* maps_dev, int B, int H, int W, int k,
                      const int32_t* xy, const int32_t* n, float* patches, bool haloed, hipStream_t s);

// The host half of select_grasp_point on the device (lg_finish_kernel): CNN rescoring of the candidates, 3-D point, pre-grasp point
struct LgFinishArgs {
    const int32_t* cand_n;        // [B]
    const int32_t* cand_xy;       // [B][K][2]
    const float* cand_info;       // [B][K][2] traditional score, depth at the candidate
    const float* logits;          // [B][K] (read only when use_cnn)
    const LgFrameParams* fp;      // [B] (theta)
    const unsigned long long* bits;   // [B][H][WW] mask bit rows
    lg_grasp_result* out;         // [B] DEVICE
    int B, H, W, WW, K, use_cnn, mask_is_bool;
    double cx, cy, f;
    LgSeSpans se;                 // (2 * pregrasp_clearance + 1) ellipse
};
void lg_launch_finish(const LgFinishArgs& a, hipStream_t s);

// host-side contour analysis on the bit-packed mask (lg_contour.cpp)
// returns 1 and fills out[0..4] = angle(rad,(0,pi]), major, minor, cx, cy ; 0 if the mask is empty
int lg_host_orientation(const unsigned long long* bits, int H, int W, int WW, double* out);
// the same on the band of rows [y_off, y_off + H) (bits -> row y_off); results in absolute image coordinates
int lg_host_orientation_rows(const unsigned long long* bits, int H, int W, int WW, int y_off, double* out);
// ... and reading only the words [w0, w1] of every row (all other pixels are empty)
int lg_host_orientation_band(const unsigned long long* bits, int H, int W, int WW, int y_off, int w0, int w1, double* out);
// 1 if any set bit of `bits` lies under the (2c+1)^2 ellipse centred at (u,v)  (pre-grasp clearance probe)
int lg_host_ellipse_hit(const unsigned long long* bits, int H, int W, int WW, int u, int v, int clearance);
int lg_host_ellipse_hit_se(const unsigned long long* bits, int H, int W, int WW, int u, int v, const LgSeSpans& se);
int lg_host_ellipse_hit_band(const unsigned long long* bits, int H, int W, int WW, int w0, int w1, int u, int v,
                             const LgSeSpans& se);
void lg_make_se_spans(int k, LgSeSpans* out);
#ifdef __cplusplus
#include <vector>
// outer contour (all border pixels, tracing order) of the component with the largest contour area; returns #points
int lg_host_contour_points(const unsigned long long* bits, int H, int W, int WW, std::vector<int>& xy);
#endif
void lg_launch_harvest(const float* depth, const uint8_t* mask, const float* const* maps_host, int H, int W, int n,
                       const int32_t* xy, const int32_t* rot, float* out_depth, float* out_mask, float* out_scores,
                       int32_t* flags, hipStream_t s);
void lg_launch_negative_masks(const float* dist, const uint8_t* mask, uint8_t* tip, uint8_t* stem, uint8_t* scratch, int H,
                              int W, hipStream_t s);
