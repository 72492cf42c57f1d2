/*
 * leafgrasp.h -- C-ABI of the MI355X (gfx950) grasp-scoring library, liblgrasp.so.
 *
 * Drop-in boundary for the per-pixel grasp-scoring hot path of
 * Srecharan/Leaf-Grasping-Vision-ML (reference paths are relative to the reference repo root).
 * Plain pointers and sizes only; no torch types.  All image pointers are DEVICE pointers owned by
 * the caller (e.g. torch tensor.data_ptr()); frames are dense row-major [B][H][W].
 * Every entry point returns 0 (LG_OK) or a negative lg_status; nothing throws, nothing exits.
 * One handle <-> one device; calls on one handle are stream-ordered, one at a time (a second thread entering the same
 * handle gets LG_ERR_BUSY); handles are independent -- threads may drive different handles concurrently -- and
 * the library keeps no global mutable state (SURVEY.md 8b "Threading").
 *
 * Reference interfaces replaced (what a ctypes/cffi binding in the reference would call):
 *   lg_score_maps      GraspPointSelector._calculate_all_scores + _get_valid_regions
 *                      (scripts/utils/grasp_point_selector.py:256-288) and everything they call:
 *                      calculate_sdf_score :526-567, calculate_approach_vector_score :569-593,
 *                      _calculate_flatness_map :635-657 + ImageProcessor.smooth_depth
 *                      (scripts/utils/image_processor.py:56-64), _calculate_isolation_score :595-633,
 *                      cv2.distanceTransform :266, _calculate_accessibility_score :502-524,
 *                      _calculate_stem_penalty :688-701, estimate_leaf_orientation :718-752
 *   lg_smooth_depth    ImageProcessor.smooth_depth (scripts/utils/image_processor.py:56-64)
 *   lg_topk_nms        GraspPointSelector._get_candidate_points  :447-482
 *   lg_gather_patches  get_ml_score feature assembly :59-127 + _extract_local_patch :392-445
 *   lg_cnn_load / lg_cnn_forward   GraspPointCNN.forward (eval)
 *                      (scripts/utils/ml_grasp_optimizer/model.py:101-128), load_ml_model :43-57
 *   lg_select_grasp    GraspPointSelector.select_grasp_point :184-253 (whole path, batched)
 *   lg_harvest_patches / lg_negative_masks / lg_leaf_contour   EnhancedGraspDataCollector's patch extraction and
 *                      negative-region helpers (scripts/utils/ml_grasp_optimizer/data_collector.py:91-173,426-490)
 *   lg_leaf_stats      the per-leaf passes of OptimalLeafSelector.select_optimal_leaf
 *                      (scripts/utils/leaf_scorer.py:32-47,66-71,74-138)
 */
#ifndef LEAFGRASP_H
#define LEAFGRASP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lg_ctx* lg_handle;

typedef enum lg_status {
    LG_OK = 0,
    LG_ERR_INVALID = -1,  /* bad argument (null pointer, non-positive size, unsupported shape) */
    LG_ERR_HIP = -2,      /* a HIP runtime call failed; see lg_last_error */
    LG_ERR_NOMEM = -3,
    LG_ERR_NO_MODEL = -4, /* lg_cnn_forward without lg_cnn_load (reference: ml_predictor is None) */
    LG_ERR_UNSUPPORTED = -5,
    LG_ERR_BUSY = -6      /* another thread is inside a call on this handle (one call in flight per handle; the error string
                             of the running call is left alone) */
} lg_status;

/* Indices into out_maps[] -- the keys of the reference's `scores` dict
   (grasp_point_selector.py:258-280), in get_ml_score's channel order (:95-99) + traditional. */
enum {
    LG_MAP_SDF = 0,        /* 'sdf_score' */
    LG_MAP_APPROACH = 1,   /* 'approach_score' */
    LG_MAP_FLATNESS = 2,   /* 'flatness_map' */
    LG_MAP_ISOLATION = 3,  /* 'isolation_map' */
    LG_MAP_DISTANCE = 4,   /* 'distance_map' */
    LG_MAP_ACCESS = 5,     /* 'accessibility_map' */
    LG_MAP_STEM = 6,       /* 'stem_penalty' */
    LG_MAP_TRADITIONAL = 7,/* 'traditional_score' */
    LG_NUM_MAPS = 8
};

/* Every constant of the path; lg_default_params fills the reference values. */
typedef struct lg_params {
    double cx, cy, f;                /* camera: P[0,2], P[1,2], P[0,0]  (:145-150); defaults 707, 494, 0 (unset).
                                        double: x - cx cancels catastrophically in float32 next to the optical centre */
    float w_approach, w_sdf, w_flat, w_access;      /* 0.4 0.3 0.2 0.1            (:272-277) */
    float sdf_w_interior, sdf_w_align, sdf_w_sdf;   /* 0.4 0.4 0.2                (:563-565) */
    float optimal_distance;                         /* 20 px                      (:535-536) */
    float access_w_dist, access_w_dir;              /* 0.7 0.3                    (:522)     */
    float flat_scale;                               /* 5                          (:655)     */
    float iso_w_close, iso_w_wide;                  /* 0.7 0.3                    (:620)     */
    float iso_ramp_top, iso_ramp_bottom;            /* 1.0 0.2                    (:623)     */
    float min_edge_distance;                        /* 20                         (:25,285)  */
    float stem_valid_thresh;                        /* 0.8                        (:287)     */
    int32_t stem_se;                                /* ellipse 30                 (:696)     */
    int32_t stem_bottom_div;                        /* bottom H//3 rows           (:693)     */
    int32_t top_k;                                  /* 20                         (:197)     */
    int32_t nms_min_distance;                       /* 10                         (:198)     */
    int32_t pregrasp_clearance;                     /* 15 px (SE 31)              (:777-778) */
    int32_t mask_is_bool;                           /* 1: torch.bool mask => border patches give no ML score */
    int32_t gaussian_size;                          /* 5: the ImageProcessor's smoothing kernel that _calculate_flatness_map applies
                                                       (:635-657, image_processor.py:25-32,56-64; sigma = size / 6).  1, 3, 5, 7;
                                                       anything else => LG_ERR_UNSUPPORTED (an even size raises in the reference) */
    int32_t chamfer_init_dist0;                     /* OpenCV's INIT_DIST0, the value cv2.distanceTransform's border cells start from.
                                                       It only shows in the transform of an image WITHOUT any zero pixel, which is
                                                       what _calculate_isolation_score asks for (:605-616: other_leaves == 0) and
                                                       what dist_inside of an all-ones mask is: INIT_DIST0 / 65536 + weight * (distance
                                                       to the frame).  The constant depends on the OpenCV revision: INT_MAX >> 2
                                                       (536870911, default; distransform.cpp of the 2.4 / 3.x lines) or INT_MAX
                                                       (2147483647; revisions that lifted the 8192-pixel ceiling).  The pinned
                                                       opencv-python 4.10.0.84 is absent here: parity unpinned.
                                                       Range [INT_MAX >> 2, INT_MAX] */
    int32_t reserved_;                              /* 0 */
} lg_params;

/* Raw GraspPointCNN(in_channels=9, attention_type, encoder_filters) state_dict tensors, HOST pointers,
   float32, PyTorch layouts (conv: [Cout][Cin][3][3], linear: [out][in]); BN is folded at load.
   attention_type (scripts/utils/ml_grasp_optimizer/model.py:30-60): LG_ATT_SPATIAL is what the node
   instantiates (grasp_point_selector.py:40); CHANNEL / HYBRID / NONE are the sweep variants. */
enum { LG_ATT_SPATIAL = 0, LG_ATT_CHANNEL = 1, LG_ATT_HYBRID = 2, LG_ATT_NONE = 3 };
typedef struct lg_cnn_weights {
    const float* conv_w[8];  const float* conv_b[8];       /* encoder.{b}.{0,3}, b < n_blocks (2 per block) */
    const float* bn_g[8]; const float* bn_b[8]; const float* bn_m[8]; const float* bn_v[8]; /* encoder.{b}.{1,4} */
    const float* att_w; const float* att_b;               /* attention.0 : [1][256][1][1], [1] */
    const float* fc_w[4]; const float* fc_b[4];            /* classifier.{0,4,8,12} */
    const float* fbn_g[3]; const float* fbn_b[3]; const float* fbn_m[3]; const float* fbn_v[3]; /* classifier.{1,5,9} */
    float bn_eps;                                          /* 1e-5 */
    int32_t attention_type;                                /* LG_ATT_*; att_w/att_b: SPATIAL, HYBRID (spatial_attention.0) */
    const float* ca_w1; const float* ca_b1;                /* CHANNEL: attention.1, HYBRID: channel_attention.1 : [16][256][1][1], [16] */
    const float* ca_w2; const float* ca_b2;                /*          attention.3,         channel_attention.3 : [256][16][1][1], [256] */
    int32_t n_blocks;                                      /* 0 = the default encoder [64,128,256]; else 3 or 4 */
    int32_t filters[4];                                    /* encoder_filters: [32,64,128] [64,128,256] [64,128,256,512] [128,256,512]
                                                              (train_model_mlflow.py:177-182); attention / classifier sizes follow filters[n_blocks-1] */
} lg_cnn_weights;

/* Per-frame result of lg_select_grasp (HOST memory). */
typedef struct lg_grasp_result {
    int32_t found;            /* 0 => reference returns (None, None, None) */
    int32_t x, y;             /* grasp_point_2d */
    float   X, Y, Z;          /* grasp_point_3d   (:152-180) */
    int32_t has_pre;          /* pre-grasp point present */
    float   pX, pY, pZ;       /* pre_grasp_point  (:754-819) */
    int32_t n_candidates;
    int32_t ml_used;          /* a CNN-rescored candidate replaced the best traditional one */
    float   best_score;
    float   theta;            /* leaf orientation (rad), NaN if none */
} lg_grasp_result;

/* Per-leaf statistics of lg_leaf_stats (HOST memory), one per label id present, ascending id. */
typedef struct lg_leaf_stat {
    int32_t id;
    int32_t area;             /* pixel count                          (leaf_scorer.py:79) */
    int32_t touches_border;   /* any pixel on the image border        (:286-291) */
    int32_t pad_;
    double  sum_x, sum_y;     /* centroid numerators                  (:84-88)   */
    double  sum_depth;        /* mean depth numerator                 (:105-106) */
    double  sum_ray;          /* sum over pixels of sqrt((x-cx)^2+(y-cy)^2+f^2)  (:109-115) */
    float   median_depth;     /* np.median of the leaf's depths       (:41-47)   */
    float   pad2_;
} lg_leaf_stat;

int lg_create(int device, lg_handle* out);
int lg_destroy(lg_handle h);
const char* lg_last_error(lg_handle h);
/* "" normally; why the device-side contour analysis (estimate_leaf_orientation, :718-752) could not be set up for this handle's
   workspace -- scoring then goes on with the host analysis of every frame (same results). */
const char* lg_orientation_note(lg_handle h);
const char* lg_version(void);
void lg_default_params(lg_params* p);

/* Score planes for B frames.  depth [B][H][W] f32, mask [B][H][W] u8 (0/1), out_maps[i] [B][H][W] f32
   (any entry may be NULL = not wanted, except DISTANCE/TRADITIONAL which later stages need),
   out_valid [B][H][W] u8 (may be NULL).  theta_host (optional, HOST, B floats) receives the leaf
   axis angle per frame (NaN when the mask has no contour).  Work is enqueued on `stream`
   (a hipStream_t, NULL = default stream); the call synchronises internally only with its own
   copy stream (orientation hand-off), not with `stream`. */
int lg_score_maps(lg_handle h, const float* depth, const uint8_t* mask, int B, int H, int W,
                  const lg_params* p, float* const out_maps[LG_NUM_MAPS], uint8_t* out_valid,
                  float* theta_host, void* stream);

/* ImageProcessor.smooth_depth (scripts/utils/image_processor.py:56-64) on its own: reflect padding by gaussian_size / 2, then
   the gaussian_size x gaussian_size kernel of _create_gaussian_kernel (:25-32, sigma = size / 6) applied as its two 1-D
   factors.  depth [B][H][W] f32 DEVICE -> out [B][Ho][Wo] f32 DEVICE, Ho = H + 2 (size / 2) - size + 1 (= H for odd sizes,
   H + 1 for even ones, as F.conv2d returns); 1 <= gaussian_size <= 15, gaussian_size / 2 < min(H, W).
   Stateless: only the handle's device is used and nothing of the handle is written, so threads may share a handle here (it takes
   no part in the one-call-in-flight rule); the reason of a failure is per thread: lg_last_error(NULL).
   lg_gaussian_taps: that 1-D factor (HOST, gaussian_size floats) -- what the plane kernel and lg_smooth_depth multiply with. */
int lg_smooth_depth(lg_handle h, const float* depth, int B, int H, int W, int gaussian_size, float* out, void* stream);
int lg_gaussian_taps(int gaussian_size, float* taps);

/* Greedy spaced top-k on valid_scores = trad*valid (score desc, flat index desc on ties).
   out_xy [B][k][2] int32 (x,y) DEVICE, out_n [B] int32 DEVICE. */
int lg_topk_nms(lg_handle h, const float* trad, const uint8_t* valid, int B, int H, int W, int k,
                int min_dist, int32_t* out_xy, int32_t* out_n, void* stream);

/* 9-channel 32x32 patches around n_xy points per frame.  maps[] as produced by lg_score_maps
   (indices 0..6 are read).  xy [B][k][2] DEVICE, n [B] DEVICE; patches [B][k][9][32][32] f32 DEVICE. */
int lg_gather_patches(lg_handle h, const float* depth, const uint8_t* mask,
                      const float* const maps[LG_NUM_MAPS], int B, int H, int W, int k,
                      const int32_t* xy, const int32_t* n, float* patches, void* stream);

int lg_cnn_load(lg_handle h, const lg_cnn_weights* w);
int lg_cnn_unload(lg_handle h);
/* patches [N][9][32][32] f32 DEVICE -> logits [N] f32 DEVICE. */
int lg_cnn_forward(lg_handle h, const float* patches, int N, float* logits, void* stream);

/* Whole GraspPointSelector.select_grasp_point for B frames: maps -> valid -> top-k -> (CNN rescoring
   if a model is loaded) -> 3-D point -> pre-grasp.  out_maps / out_valid may be NULL (library
   workspace is used).  results: HOST array of B.  Synchronises `stream` before returning. */
int lg_select_grasp(lg_handle h, const float* depth, const uint8_t* mask, int B, int H, int W,
                    const lg_params* p, float* const out_maps[LG_NUM_MAPS], uint8_t* out_valid,
                    lg_grasp_result* results, void* stream);

/* lg_select_grasp on mask[b] = (labels[b] == leaf_ids[b]): the node's `optimal_mask = mask_tensor == optimal_leaf_id` followed
   by select_grasp_point (scripts/leaf_grasp_node_v3.py:118-125) with the comparison folded into the library's first pass over
   the frame.  labels [B][H][W] int16 DEVICE (the label image lg_leaf_select_batch reads), leaf_ids [B] HOST (an id no label
   carries, e.g. INT32_MIN, gives that frame an empty mask and found = 0).  Everything else as lg_select_grasp; the node's mask
   is a torch.bool tensor: set lg_params.mask_is_bool = 1 for its behaviour at the image border. */
int lg_select_grasp_labels(lg_handle h, const float* depth, const int16_t* labels, const int32_t* leaf_ids, int B, int H, int W,
                           const lg_params* p, float* const out_maps[LG_NUM_MAPS], uint8_t* out_valid,
                           lg_grasp_result* results, void* stream);

/* The node's result message of every frame (leaf_grasp_node_v3.py:170-176: "x,y,X,Y,Z[,pX,pY,pZ]", each number as Python's
   str() prints it -- the shortest decimal string of the float32 value as a double), '\n'-terminated, one line per frame in
   order, an empty line for a frame without a result.  buf: HOST, cap bytes (>= 256 per frame); *used = bytes written.
   No device work; needs no handle. */
int lg_format_grasp_results(const lg_grasp_result* results, int n, char* buf, int64_t cap, int64_t* used);

/* Per-label statistics + clutter extrema for one frame.  labels [H][W] int16 DEVICE, depth DEVICE.
   stats: HOST array of capacity max_leaves; n_leaves: HOST.  extrema (HOST, 4 ints): first leaf
   pixel (y,x) and the background pixel farthest (exact Euclidean) from any leaf (y,x). */
int lg_leaf_stats(lg_handle h, const int16_t* labels, const float* depth, int H, int W,
                  float cx, float cy, float f, lg_leaf_stat* stats, int max_leaves, int* n_leaves,
                  int32_t* extrema, void* stream);
/* The same for B frames per call: labels / depth [B][H][W] DEVICE; stats [B][max_leaves], n_leaves [B], extrema [B][4],
   status [B] HOST (per-frame lg_status: a frame with more than 1024 labels fails alone with LG_ERR_UNSUPPORTED; one with more
   than max_leaves labels with LG_ERR_INVALID and its label count in n_leaves, so the caller can come back with room).  Every pass carries the frame
   in its grid; there is no host round trip between the passes. */
int lg_leaf_stats_batch(lg_handle h, const int16_t* labels, const float* depth, int B, int H, int W, float cx, float cy,
                        float f, lg_leaf_stat* stats, int max_leaves, int* n_leaves, int32_t* extrema, int* status,
                        void* stream);

/* OptimalLeafSelector.select_optimal_leaf (scripts/utils/leaf_scorer.py:25-203) for B frames, the whole of it: the per-leaf
   passes of lg_leaf_stats_batch and then, on the host inside the library, tall-leaf split, scores, Pareto filter and weighted
   pick.  ids [B] HOST: the chosen label (-1: the reference's None; -2: a frame with more than 256 labels or 128 leaves -- take
   it through lg_leaf_stats + the caller's own selection); n_tall [B], tall [B][tall_cap] HOST: get_tall_leaves() per frame. */
int lg_leaf_select_batch(lg_handle h, const int16_t* labels, const float* depth, int B, int H, int W, double cx, double cy,
                         double f, int32_t* ids, int32_t* n_tall, int32_t* tall, int tall_cap, void* stream);

/* The host half of lg_leaf_select_batch on its own (no device, no handle): the selection of leaf_scorer.py:53-203 from the rows
   of lg_leaf_stats for ONE frame -- mean of medians -> tall leaves, the three scores of every candidate with area >= 10000,
   Pareto filter on the tall (x 1.1) or the regular set, weighted pick.  *id: the chosen label, -1 = the reference's None, -2 =
   128 or more leaves (numpy's summation order of the float32 mean changes there: the caller's own path decides).  tall
   [tall_cap], *n_tall: get_tall_leaves() (*n_tall may exceed tall_cap; only tall_cap ids are written). */
int lg_leaf_select_from_stats(const lg_leaf_stat* stats, int n, const int32_t* extrema, int H, int W, double cx, double cy,
                              double f, int32_t* id, int32_t* tall, int tall_cap, int32_t* n_tall);

/* GraspPointSelector.estimate_leaf_orientation (:718-752) for one frame: mask [H][W] u8 DEVICE.
   out (HOST, 5 floats): angle (rad, direction of the longer side of the min-area rectangle of the largest
   outer contour, in (0, pi]), major axis, minor axis, centre x, centre y.  Returns LG_OK and *found = 0
   when the mask is empty.  Synchronises `stream`. */
int lg_leaf_orientation(lg_handle h, const uint8_t* mask, int H, int W, float* out, int* found, void* stream);

/* ---- training-sample harvesting: EnhancedGraspDataCollector (scripts/utils/ml_grasp_optimizer/data_collector.py)
   lg_harvest_patches   _extract_patches :91-173 (+ the rot90 of _generate_augmented_samples :250-266): raw 32x32 windows
                        [y-16,y+16) x [x-16,x+16) of depth / mask / the seven score planes maps[0..6] of ONE frame around
                        n points xy [n][2] DEVICE, rotated by rot[i] quarter turns (torch.rot90; rot may be NULL).
                        out_depth, out_mask [n][32][32], out_scores [n][7][32][32] f32 DEVICE; flags [n] int32 DEVICE:
                        bit 0 non-finite depth, 1 empty mask patch, 2 non-finite score, 3 window outside the frame
                        (_check_boundaries :83-89; nothing written).
   lg_negative_masks    the regions negatives are drawn from: tip = (dilate5x5(distance_map) == distance_map) & mask
                        (_get_tip_points :426-443), stem = the mask's bottom quarter eroded twice by the 5x5 ellipse
                        (_get_stem_points :445-460); u8 [H][W] DEVICE each.
   lg_leaf_contour      cv2.findContours(EXTERNAL, CHAIN_APPROX_NONE) + max(contourArea) of _get_edge_points :462-490:
                        every border pixel of the largest outer contour in tracing order, out_xy [cap][2] int32 HOST;
                        *n_out = number of points (may exceed cap: only cap are written). */
int lg_harvest_patches(lg_handle h, const float* depth, const uint8_t* mask, const float* const maps[LG_NUM_MAPS],
                       int H, int W, int n, const int32_t* xy, const int32_t* rot, float* out_depth, float* out_mask,
                       float* out_scores, int32_t* flags, void* stream);
int lg_negative_masks(lg_handle h, const float* distance_map, const uint8_t* mask, int H, int W, uint8_t* out_tip,
                      uint8_t* out_stem, void* stream);
int lg_leaf_contour(lg_handle h, const uint8_t* mask, int H, int W, int32_t* out_xy, int cap, int* n_out, void* stream);

/* Per-kernel device timing with HIP events recorded on the launch stream (bench.py roofline).
   lg_profile_enable(h,1) starts collecting for every kernel (an event pair around each launch);
   lg_profile_enable(h,2) collects only "final", whose dispatch stamps its own start/stop events
   (hipExtLaunchKernelGGL: no extra packets in the stream); 0 stops.  lg_profile_read returns, for kernel `name`
   ("final", "dt_fwd", "dt_bwd", "prep", "stem", "topk", "gather", "cnn", ...), the number of
   launches and their summed duration in milliseconds since the last enable. */
int lg_profile_enable(lg_handle h, int on);
int lg_profile_read(lg_handle h, const char* name, int* launches, double* total_ms);

/* Inspection of the last lg_score_maps / lg_select_grasp call on this handle (synchronises the device):
   out[0], out[1] = max d_in, max d_out of frame `frame` in OpenCV's 16.16 fixed point (the normaliser
   max|d_in - d_out| of calculate_sdf_score, grasp_point_selector.py:531-533, is their maximum / 65536);
   win[0..3] (optional) = the distance-transform sweep window {x0, x1, y0, y1} (half open) used for that frame. */
int lg_debug_dt_max(lg_handle h, int frame, uint32_t out[2], int32_t win[4]);
/* form[0] = 1: d_in of that frame came from the row search (0: from the two sweeps); form[1] = 1: its d_out sweeps were
   skipped (the maximum provably lies on the frame border).  Which form a batch takes is decided on the device. */
int lg_debug_dt_form(lg_handle h, int frame, int32_t form[2]);

/* ---- GraspPointCNN training step: one call = one iteration of the inner loop of
   scripts/train_model.py:247-265 (zero_grad, forward in train mode, BCEWithLogitsLoss(pos_weight), backward,
   clip_grad_norm_(max_grad_norm), Adam step with L2 weight_decay) on the model of
   scripts/utils/ml_grasp_optimizer/model.py:5-128 with any LG_ATT_* attention (LG_ATT_SPATIAL = the script's model).
   Flat parameter vector = model.parameters() order; flat buffer vector = running_mean, running_var of every BatchNorm
   in module order.  Dropout keep masks: one [N][width] block per dropout layer, concatenated in module order
   (Dropout2d of every encoder block: width = filters[b]; classifier Dropout 0.5 / 0.5 / 0.4: widths F, F/2, F/4),
   values 0 or 1/(1-p); NULL = drawn on the device from `seed` and the step counter. */
typedef struct lg_trainer lg_trainer;
typedef struct lg_train_hparams {
    float lr, beta1, beta2, eps, weight_decay;   /* train_model.py:222: Adam(lr=0.0005, weight_decay=0.01), betas (0.9, 0.999), eps 1e-8 */
    float max_grad_norm;                         /* :256 clip_grad_norm_(max_norm=1.0); <= 0: no clipping */
    float pos_weight;                            /* :221 BCEWithLogitsLoss(pos_weight=2.0) */
} lg_train_hparams;
int lg_train_create(int device, int n_blocks, const int32_t* filters, int attention_type, int max_batch, lg_trainer** out);
int lg_train_destroy(lg_trainer* t);
const char* lg_train_last_error(lg_trainer* t);
int lg_train_sizes(lg_trainer* t, int64_t* n_params, int64_t* n_buffers, int64_t* mask_row);
/* HOST arrays; exp_avg / exp_avg_sq may be NULL (= zeros: a fresh optimizer) */
int lg_train_set_state(lg_trainer* t, const float* params, const float* buffers, const float* exp_avg,
                       const float* exp_avg_sq, int64_t step);
/* HOST arrays, any may be NULL; grads = the unclipped gradients of the last step */
int lg_train_get_state(lg_trainer* t, float* params, float* buffers, float* exp_avg, float* exp_avg_sq, float* grads,
                       int64_t* step);
/* x [N][9][32][32], labels [N], masks (or NULL), logits_dev (or NULL): DEVICE pointers.  apply_update = 0: loss and
   gradients only (BatchNorm running statistics still move, as in a train-mode forward).  loss_host / grad_norm_host
   (or NULL) make the call synchronous. */
int lg_train_step(lg_trainer* t, const float* x, const float* labels, int N, const float* masks, uint64_t seed,
                  const lg_train_hparams* hp, int apply_update, float* loss_host, float* grad_norm_host, float* logits_dev);
int lg_train_sync(lg_trainer* t);
/* Data-parallel training (one process per GPU): lg_train_step(apply_update = 0) on the rank's shard of the batch, average
   the flat gradient vector over the ranks (RCCL all-reduce on a tensor wrapping *dev_ptr; lg_train_sync first), then
   lg_train_apply = clip_grad_norm_ + Adam.step() on the averaged gradients (train_model.py:256-258).  BatchNorm statistics
   stay per rank, as with torch's DistributedDataParallel without SyncBatchNorm. */
int lg_train_grad_buffer(lg_trainer* t, float** dev_ptr, int64_t* n);
int lg_train_apply(lg_trainer* t, const lg_train_hparams* hp, float* grad_norm_host);

#ifdef __cplusplus
}
#endif
#endif /* LEAFGRASP_H */
