// GraspPointCNN (scripts/utils/ml_grasp_optimizer/model.py:5-128, eval mode; the four attention types and the four
// encoder_filters configurations of the reference's sweep) on gfx950: BN folded at load, 3x3 convs as implicit GEMM on
// v_mfma_f32_32x32x2_f32 (exact fp32), fused bias+ReLU(+2x2 max-pool) epilogues.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/leafgrasp.h"

struct RtLayer { int cin, cout, cinp, coutp, wi; bool pool; };   // real / padded channels, image width, 2x2 pool after it

struct LgCnn {
    bool loaded = false;
    int max_cus = 0;              // > 0: the persistent conv kernels take this many workgroups (CUs) instead of all (LG_CNN_CUS experiment)
    int n_layers = 6;             // 2 conv layers per encoder block
    RtLayer layers[8] = {};
    bool standard = true;         // encoder [64,128,256]: the direct kernels / A-B switches exist for this one only
    int F = 256, Fp = 256, npix = 16;   // final filters (real / padded), pixels of the last feature map
    size_t act_per_patch = 64 * 32 * 32;  // floats of the largest activation per patch
    float* wconv[8] = {nullptr};  // packed [ky][kx][cin_pad][cout], BN folded (layer 0; all layers of the standard model)
    float* bconv[8] = {nullptr};
    float* uwino[8] = {nullptr};  // Winograd F(2x2,3x3) weights [cin][cout][16] (layers 1..)
    float* uwino4[8] = {nullptr}; // Winograd F(4x4,3x3) weights in lg_wino4_kernel's fragment order (layers 1..)
    bool use_f23 = false;         // LG_CNN_F23 at load time: F(2x2,3x3) kernels instead of F(4x4,3x3)
    float* att_w = nullptr;       // [256] spatial attention (1x1 conv 256 -> 1)
    float att_b = 0.f;
    int att_type = 0;             // LG_ATT_*
    float* ca_w1 = nullptr;       // channel attention: [16][256], [16], [256][16], [256]
    float* ca_b1 = nullptr;
    float* ca_w2 = nullptr;
    float* ca_b2 = nullptr;
    float* fcw[4] = {nullptr};    // transposed [in][out], BN folded (first three)
    float* fcb[4] = {nullptr};
    float* act[8] = {nullptr};    // output of layer L for capN patches: haloed planes [coutp][(wo+2)][(wo+4)] (lg_cnn.hip), the
    size_t act_per[8] = {0};      //   last layer dense [coutp][wo*wo] for the head; one buffer per layer: halos stay zero
    float* in_halo = nullptr;     // haloed copy [capN][12][34][36] (9 feature planes + 3 zero planes) of dense input patches (lg_cnn_forward through the C-ABI)
    float* zeros = nullptr;       // 4096 zero floats
    float* kpart = nullptr;       // lg_wino4_kernel: partial accumulators of items split along the input channels, one slot per workgroup
    unsigned* kflag = nullptr;    //   and the counters of the parts that have arrived
    unsigned* kerr_host = nullptr;   // set by a split item whose parts did not arrive (pinned host word; kerr_dev = its device address)
    unsigned* kerr_dev = nullptr;
    int wino_mask = 0x3f;         // bit L = layer L on Winograd (LG_CNN_DIRECT / LG_CNN_WINO_MASK at load time; bits 1..5: standard encoder only)
    int capN = 0;
};

int lg_cnn_upload(LgCnn* c, const lg_cnn_weights* w, std::string* err);
void lg_cnn_free(LgCnn* c);
// patches: dense [N][9][32][32] (haloed_in = false) or haloed planes [N][12][34][36] (planes 9..11 and all halos zero) (what lg_select_grasp's
// gather writes directly; lg_cnn_halo_patch_floats() floats per patch)
int lg_cnn_run(LgCnn* c, const float* patches, bool haloed_in, int N, float* logits, hipStream_t s, std::string* err);
size_t lg_cnn_halo_patch_floats(void);
// after a synchronisation of the stream a forward ran on: true (once) if a split item of it gave up waiting for its parts
bool lg_cnn_take_error(LgCnn* c);
