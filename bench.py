#!/usr/bin/env python3
"""bench.py -- frames/sec of the full grasp-scoring hot path on synthetic 1080p depth+mask frames.

Workload (BASELINE.json configs[1]): 1080x1920 depth f32 + binary leaf mask, full per-pixel score planes
(8 f32 maps + validity), top-20 spaced candidates, 9-channel patch gather, GraspPointCNN (fp32) rescoring
of the 20 candidates, 3-D / pre-grasp points -- i.e. GraspPointSelector.select_grasp_point for every frame.
One "step" = one pass of that path over one batch of --batch frames already resident in HBM.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

Frames are independent: with N ranks every rank scores its own batch (weak scaling, no data-path
collective; torch.distributed is used only for the timing barrier and the max-over-ranks reduction).
Rank 0 prints ONE JSON line (see README / DESIGN.md "Measurement").
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
# compulsory HBM bytes per pixel of the dominant ("final" = fused score-plane) kernel:
#   read depth f32 4 + distance_map f32 4 + mask bits 1/8 + stem bits 1/8; write 7 f32 planes 28 + valid u8 1
FINAL_BYTES_PER_PX = 4 + 4 + 0.125 + 0.125 + 28 + 1
PATH_BYTES_PER_PX = 38.0  # whole-path compulsory traffic per pixel
BASELINE_METRIC = "frames/sec (1080p depth+mask) grasp scoring, 1/8 MI355X; % HBM roofline"  # BASELINE.json:metric


def _cpu_share():
    """Host cores this process may use (cgroup / affinity aware), capped at one GPU's share of an 8-GPU node's 256."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:  # noqa: BLE001
        n = os.cpu_count() or 1
    return max(1, min(32, n))


N_DISTINCT = 32  # distinct synthetic scenes in the benchmark batch (the batch cycles through them)
# the other BASELINE.json configs, measured outside the headline's timed region: name -> (H, W, frames per step, distinct scenes)
SECONDARY_CONFIGS = {
    "720p_b256": (720, 1280, 256, 16),       # configs[0]'s frame size (the reference's CPU-runnable case) on the HIP path
    "4k_b64": (2160, 3840, 64, 8),           # configs[3]: LDS-tiled SDF + distance-transform stress
    "1080p_b32": (1080, 1920, 32, 32),       # configs[2]'s per-GPU share: 256 frames sharded over 8 GPUs
}


def _scene(args):
    import synthetic_inputs as SI  # seeded scenes (inputs; the oracle is only touched by the cpu_baseline leg)

    H, W, seed = args
    labels, depth, P = SI.synthetic_scene(H, W, seed=seed)
    ids, counts = np.unique(labels[labels > 0], return_counts=True)
    chosen = int(ids[np.argmax(counts)])   # "binary optimal_mask = labels == chosen id" (SURVEY 8d): the largest leaf
    return labels, depth, P, chosen


def make_frames(B, H, W, n_distinct=N_DISTINCT, workers=1):
    """-> masks [B,H,W] bool, depths [B,H,W] f32, P, labels of the distinct scenes [n,H,W] int16.  Must run before the
    process touches the GPU when workers > 1 (fork)."""
    jobs = [(H, W, 100 + s) for s in range(min(B, n_distinct))]
    if workers > 1 and len(jobs) > 1:
        from concurrent.futures import ProcessPoolExecutor

        with ProcessPoolExecutor(max_workers=min(workers, len(jobs))) as ex:
            res = list(ex.map(_scene, jobs))
    else:
        res = [_scene(j) for j in jobs]
    masks = [r[0] == r[3] for r in res]
    depths = [r[1] for r in res]
    reps = (B + len(masks) - 1) // len(masks)
    m = np.stack((masks * reps)[:B])
    d = np.stack((depths * reps)[:B])
    return m, d, res[0][2], np.stack([r[0] for r in res]).astype(np.int16)


def final_kernel_bytes(masks_np, wins, H, W, tile_w=64, tile_h=16):
    """Compulsory HBM bytes of ONE lg_final_kernel launch for these masks.  Per 64x16 tile:
      full path  : read depth 4 + mask/stem bits 0.25 + distance_map 4 (inside the sweep window) ; write 7 planes 28 +
                   valid 1 (+ distance_map 4 outside the window)                                   = 37.25 B/px
      fast path  : (no mask bit within the stencil reach) write 7 planes 28 + valid 1 (+ distance_map 4 outside the
                   window), read mask bits 0.125                                                 = 33.125 / 29.125 B/px
    `wins[i]` = (x0, x1, y0, y1) sweep window of frame i as reported by the library (lg_debug_dt_max)."""
    total = 0.0
    for m, (wx0, wx1, wy0, wy1) in zip(masks_np, wins):
        for ty0 in range(0, H, tile_h):
            r0, r1 = max(0, ty0 - 3), min(H, ty0 + tile_h + 3)
            rows_any = m[r0:r1].any(axis=0)
            for tx0 in range(0, W, tile_w):
                px = (min(H, ty0 + tile_h) - ty0) * (min(W, tx0 + tile_w) - tx0)
                in_win = wx0 <= tx0 < wx1 and wy0 <= ty0 < wy1
                if rows_any[max(0, tx0 - 8):min(W, tx0 + tile_w + 8)].any():
                    total += 37.25 * px
                else:
                    total += (29.125 if in_win else 33.125) * px
    return total


def kernel_fracs(kern_ms, masks_np, wins, B, H, W, K=20, forms=None):
    """Compulsory HBM bytes (read once + written once) of ONE launch of every score-path kernel beside the
    plane kernel, for these masks, its average duration and the fraction of the 8 TB/s peak that makes.  Latency-bound kernels
    (one workgroup per frame, bit rows only) show as the small fractions they are.  `wins[i]` = (x0, x1, y0, y1): frame i's
    distance-transform window as the library reports it (lg_debug_dt_max); `forms[i]` = (searched, d_out sweeps skipped) as it
    reports them (lg_debug_dt_form): a searched frame's sweeps and a swept frame's search kernels move nothing.  A kernel that
    moved nothing in this batch (launched, every frame exits) is left out.  kern_ms: name -> average ms."""
    nd = len(masks_np)
    WW = (W + 63) // 64
    bits = H * WW * 8                               # one frame's bit rows
    per = {k: 0.0 for k in ("prep", "bbox", "stem", "orient", "dt_hrun", "dt_search", "dt_band", "dt_fwd", "dt_bwd", "dt_border",
                            "topk", "gather")}
    if forms is None:
        forms = [(False, False)] * nd
    for m, (wx0, wx1, wy0, wy1), (searched, skip_out) in zip(masks_np, wins, forms):
        ys, xs = np.nonzero(m)
        win_px = (wx1 - wx0) * (wy1 - wy0)
        if ys.size:
            bb_rows, bb_words = ys.max() - ys.min() + 3, xs.max() // 64 - xs.min() // 64 + 1
        else:
            bb_rows = bb_words = 0
        per["prep"] += H * W * 1 + bits             # mask bytes in, bit rows out
        per["bbox"] += bits
        per["stem"] += 2 * bits // 3 + bits // 3    # the bottom third's bit rows in (with the 15-row reach), stem rows out
        per["orient"] += bb_rows * bb_words * 8
        per["dt_border"] += bb_rows * bb_words * 8
        # row search: run distances (2 B per pixel of the bounding box's words) written once, read once; distance_map written
        if searched:
            per["dt_hrun"] += bb_rows * bb_words * (8 + 128)
            if "dt_band" in kern_ms:                # anchors (every 8th row: distance + minimising row) / the rows between
                per["dt_search"] += bb_rows * bb_words * 128 // 8 + win_px * (4 + 2) // 8
                per["dt_band"] += bb_rows * bb_words * 128 + win_px * 2 // 8 + win_px * 4 * 7 // 8
            else:
                per["dt_search"] += bb_rows * bb_words * 128 + win_px * 4
        n_sweeps = (0 if searched else 1) + (0 if skip_out else 1)    # d_in and d_out images this frame's sweeps work on
        per["dt_fwd"] += n_sweeps * win_px * (1 + 4)   # mask in, forward values out; those in, distance out
        per["dt_bwd"] += n_sweeps * win_px * (4 + 4)
        tiles = ((W + 63) // 64) * ((H + 15) // 16)
        per["topk"] += tiles * 8 + K * 8 * 1024 * 5  # tile keys + the <= 8 tiles a pick's suppression window touches (score + valid)
        per["gather"] += K * (9 * 32 * 32 * 4 + 12 * 34 * 36 * 4)
    out = {}
    for k, tot in per.items():
        if k in kern_ms and kern_ms[k] > 0 and tot > 0:
            b = tot * (B / nd)
            out[k] = {"bytes": round(b), "avg_ms": round(kern_ms[k], 4), "frac": round(b / (kern_ms[k] * 1e-3) / (HBM_PEAK_GBS * 1e9), 4)}
    return out


def cpu_baseline(H, W, P, params, n_frames, threads):
    """Restated reference-equivalent CPU path (oracle, NumPy/torch-CPU + C chamfer): same work as
    scripts/utils/grasp_point_selector.py::select_grasp_point.  NOT the reference's own timing
    (OpenCV / scikit-fmm are absent from this image).  Returns (per-frame seconds after one
    warm-up frame, single-thread milliseconds of the serial C pieces)."""
    from oracle import lg_oracle as O

    torch.set_num_threads(threads)
    ref = O.RefGraspPointSelector(cnn=lambda x: O.cnn_forward(params, x))
    ref.set_camera_params(P)
    times, serial = [], {}
    for s in range(n_frames + 1):
        labels, depth, _, chosen = _scene((H, W, 100 + (s % N_DISTINCT)))
        mask = (labels == chosen).astype(np.uint8)
        t0 = time.perf_counter()
        _, dbg = ref.select_grasp_point(mask, depth, tie_rule="numpy", return_debug=True)
        dt = time.perf_counter() - t0
        if s > 0:  # first frame is warm-up
            times.append(dt)
        if s == 1:   # single-thread C / NumPy timing of the serial pieces, one frame
            t0 = time.perf_counter()
            O.distance_transform(mask, 5)
            serial["chamfer_dt5_ms"] = round(1e3 * (time.perf_counter() - t0), 2)
            t0 = time.perf_counter()
            ref._get_candidate_points(dbg["scores"]["traditional_score"], dbg["valid"], 20, 10, "numpy")
            serial["argsort_nms_ms"] = round(1e3 * (time.perf_counter() - t0), 2)
            t0 = time.perf_counter()
            O.leaf_orientation_raw(mask)
            serial["contour_min_area_rect_ms"] = round(1e3 * (time.perf_counter() - t0), 2)
    return times, serial


def _time_config(dev, L, lib, params, P, m_np, d_np, B, H, W, steps):
    """fps, ms per step, per-kernel ms and the plane kernel's fraction of the HBM peak for B frames of H x W on a fresh handle."""
    sel = L.GraspPointSelector(dev, load_model=False)
    sel.set_camera_params(P)
    sel.set_cnn_state_dict(params)
    m, d = torch.from_numpy(m_np).to(dev), torch.from_numpy(d_np).to(dev)
    for _ in range(2):
        res = sel.select_grasp_points_batch(m, d)
    lib.lg_profile_enable(sel._h, 1)
    sel.select_grasp_points_batch(m, d)
    torch.cuda.synchronize(dev)
    kern = {}
    for name in ("prep", "bbox", "orient", "stem", "dt_hrun", "dt_search", "dt_band", "dt_fwd", "dt_bwd", "dt_border", "final", "topk", "gather", "cnn"):
        n, ms = C.c_int(0), C.c_double(0.0)
        lib.lg_profile_read(sel._h, name.encode(), C.byref(n), C.byref(ms))
        if n.value:
            kern[name] = round(ms.value / n.value, 4)
    lib.lg_profile_enable(sel._h, 2)
    sel.select_grasp_points_batch(m, d)
    lib.lg_profile_enable(sel._h, 2)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        res = sel.select_grasp_points_batch(m, d)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    n, ms = C.c_int(0), C.c_double(0.0)
    lib.lg_profile_read(sel._h, b"final", C.byref(n), C.byref(ms))
    lib.lg_profile_enable(sel._h, 0)
    nd = min(B, len(m_np))
    wins = [sel.dt_maxima(i)[2] for i in range(nd)]
    launch_bytes = final_kernel_bytes(m_np[:nd], wins, H, W) * (B / nd)
    final_ms = ms.value / max(1, n.value)
    kern["final"] = round(final_ms, 4)
    return {"value": round(B / dt, 1), "unit": "frames/s", "frames_per_step": B, "height": H, "width": W, "steps": steps,
            "ms_per_step": round(1e3 * dt, 4), "found": sum(r[0] is not None for r in res),
            "final_frac_of_hbm_peak": round(launch_bytes / (final_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "final_bytes_per_px": round(launch_bytes / (B * H * W), 3), "kernels_ms": kern,
            "path_hbm_frac": round(B / dt * PATH_BYTES_PER_PX * H * W / (HBM_PEAK_GBS * 1e9), 5)}


def secondary_configs(args, dev, L, lib, SI, params, P, extra_frames, masks_np, depths_np, labels_np, H, W):
    """The other BASELINE.json configs and the reference's own call pattern (one frame per call), all outside the headline's
    timed region.  Same workload definition as the headline (8 planes + valid + top-20 + fp32 CNN on 20 candidates)."""
    out = {}
    for name, (h2, w2, b2, nd2) in SECONDARY_CONFIGS.items():
        try:
            if (h2, w2) == (H, W):
                m_np, d_np, P2 = masks_np[:b2], depths_np[:b2], P
                if len(m_np) < b2:
                    continue
            else:
                m_np, d_np, P2, _ = extra_frames[name]
            out[name] = _time_config(dev, L, lib, params, P2, m_np, d_np, b2, h2, w2, args.config_steps)
        except Exception as e:  # noqa: BLE001
            out[name] = {"error": str(e)}
    if "1080p_b32" in out and "value" in out["1080p_b32"]:
        out["1080p_b32"]["what"] = ("BASELINE config 3 read literally (256 frames sharded over 8 GPUs = 32 per GPU, strong scaling): "
                                    "8 x this value is the 8-GPU figure when nothing but the frames is shared; the headline `value` is "
                                    "weak scaling at 256 frames per GPU")
    # the reference's call pattern: ONE frame per select_grasp_point call (leaf_grasp_node_v3.py:102-158), tensors resident
    try:
        sel = L.GraspPointSelector(dev, load_model=False)
        sel.set_camera_params(P)
        sel.set_cnn_state_dict(params)
        ip = L.ImageProcessor(H, W, 21, 5)
        m1, d1 = torch.from_numpy(masks_np[0]).to(dev), torch.from_numpy(depths_np[0]).to(dev)
        for _ in range(5):
            sel.select_grasp_point(m1, d1, ip)
        ts = []
        for _ in range(40):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            r1 = sel.select_grasp_point(m1, d1, ip)
            ts.append(time.perf_counter() - t0)
        lib.lg_profile_enable(sel._h, 1)
        sel.select_grasp_point(m1, d1, ip)
        k1 = {}
        for name in ("prep", "bbox", "orient", "stem", "dt_hrun", "dt_search", "dt_band", "dt_fwd", "dt_bwd", "dt_border", "final", "topk",
                     "gather", "cnn", "finish"):
            n_, ms_ = C.c_int(0), C.c_double(0.0)
            lib.lg_profile_read(sel._h, name.encode(), C.byref(n_), C.byref(ms_))
            if n_.value:
                k1[name] = round(ms_.value / n_.value, 4)
        lib.lg_profile_enable(sel._h, 0)
        hz = L.LeafGraspHarness(H, W, dev, load_model=False)
        hz.camera_info_callback(np.asarray(P).reshape(-1))
        hz.grasp_selector.set_cnn_state_dict(params)
        hz.latest_mask, hz.latest_depth = torch.from_numpy(labels_np[0]).to(dev), d1
        for _ in range(3):
            hz.select_optimal_leaf()
        tn = []
        for _ in range(20):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            csv = hz.select_optimal_leaf()
            tn.append(time.perf_counter() - t0)
        out["1080p_b1_latency_ms"] = {"select_grasp_point": round(1e3 * float(np.median(ts)), 4),
                                      "select_grasp_point_min": round(1e3 * float(np.min(ts)), 4),
                                      "node_sequence": round(1e3 * float(np.median(tn)), 4), "found": r1[0] is not None and csv is not None,
                                      "kernels_ms": k1,
                                      "what": "median wall time of one call on one resident frame (label / mask + depth tensors on the "
                                              "device): GraspPointSelector.select_grasp_point incl. CNN, and the node's whole "
                                              "select_optimal_leaf sequence (leaf selection + grasp selection)"}
        hz = sel = None
    except Exception as e:  # noqa: BLE001
        out["1080p_b1_latency_ms"] = {"error": str(e)}
    import gc
    gc.collect()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256,
                    help="frames per step per GPU (256 puts two sweep workgroups on every CU; 24 GB of the 288 GB)")
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--cpu-frames", type=int, default=5, help="frames timed for the cpu_baseline leg (0 = skip)")
    ap.add_argument("--no-cnn", action="store_true", help="CV-only (diagnostic; NOT the headline config)")
    ap.add_argument("--inflight", type=int, default=1,
                    help="batches in flight per GPU (each on its own handle, stream and host thread); >1 lets the "
                         "latency-bound sweeps of one batch run beside the CNN of another")
    ap.add_argument("--node-steps", type=int, default=3,
                    help="extra, untimed-for-the-headline leg: steps of the WHOLE node sequence (batched leaf selection + "
                         "grasp selection) reported as `node_sequence` (0 = skip)")
    ap.add_argument("--train-steps", type=int, default=30,
                    help="N=1 only: optimisation steps of the GraspPointCNN training step (lg_train_step: forward + backward + "
                         "clip + Adam) timed at the reference's batch size 16 and at 1024, reported as `train_step` (0 = skip)")
    ap.add_argument("--dense-steps", type=int, default=4,
                    help="N=1 only: untimed passes of lg_final_kernel with EVERY tile on the stencil path (all-leaves masks, "
                         "constant-tile fast path off), reported as `roofline_dense` (0 = skip)")
    ap.add_argument("--h2d-steps", type=int, default=3,
                    help="N=1 only: steps of pinned-host -> device copy of depth + mask followed by the scoring pass, reported "
                         "as `h2d_inclusive` (never `value`; 0 = skip)")
    ap.add_argument("--pipelined", type=int, default=2,
                    help="N=1 only: batches in flight for the secondary `pipelined` figure (the headline keeps --inflight; <= 1 = skip)")
    ap.add_argument("--config-steps", type=int, default=5,
                    help="N=1 only: timed steps per entry of the secondary `configs` leg (720p x 256, 4K x 64, 1080p x 32 = config 3's "
                         "per-GPU share, and single-frame latency of select_grasp_point / the node sequence; 0 = skip)")
    ap.add_argument("--per-step", action="store_true", help="diagnostic: print every step's wall time to stderr "
                                                            "(adds a device sync per step; not the headline mode)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # synthetic frames first: the scene generator forks worker processes, which must happen before this process touches the GPU
    H, W, B = args.height, args.width, args.batch
    cpu_share = _cpu_share()
    gen_workers = max(1, min(8, cpu_share // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))))
    masks_np, depths_np, P, labels_np = make_frames(B, H, W, workers=gen_workers)
    # frames of the other BASELINE configs (secondary `configs` leg, N=1 only): generated here, before the process touches the GPU
    extra_frames = {}
    if world == 1 and args.config_steps > 0 and not args.no_cnn:
        for name, (h2, w2, b2, nd2) in SECONDARY_CONFIGS.items():
            if (h2, w2) != (H, W):
                extra_frames[name] = make_frames(b2, h2, w2, n_distinct=nd2, workers=gen_workers)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # rehearsal switch for single-GPU boxes: all ranks on device 0 over gloo (exercises the barrier / MAX-reduction /
        # rank-0 reporting path; the real multi-GPU run uses one device per rank over RCCL)
        if os.environ.get("LG_BENCH_ONE_DEVICE"):
            local_rank = 0
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    import leafgrasp_amd as L
    from leafgrasp_amd._lib import lib
    import synthetic_inputs as SI  # seeded scenes + closed-form CNN weights (inputs)

    masks = torch.from_numpy(masks_np).to(dev)
    depths = torch.from_numpy(depths_np).to(dev)
    import threading

    params = SI.cnn_closed_form_params(seed=0)
    _idle = [torch.cuda.Stream(device=dev) for _ in range(int(os.environ.get("LG_BENCH_IDLE_STREAMS", "0")))]   # diagnostic:
    for _st in _idle:                          # streams alive before the selector decide the stream-to-queue mapping
        with torch.cuda.stream(_st):
            torch.zeros(1, device=dev)
    sels = []
    for _ in range(max(1, args.inflight)):
        sel = L.GraspPointSelector(dev, load_model=False)
        sel.set_camera_params(P)
        if not args.no_cnn:
            sel.set_cnn_state_dict(params)
        sels.append(sel)
    streams = [torch.cuda.Stream(dev) for _ in sels]

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    last = [None] * len(sels)

    def run_steps(n_steps):
        """n_steps passes over the batch; with --inflight N they are dealt round-robin to N host threads."""
        def worker(i):
            torch.cuda.set_device(dev)
            with torch.cuda.stream(streams[i]):
                for _ in range(i, n_steps, len(sels)):
                    t_s = time.perf_counter()
                    last[i] = sels[i].select_grasp_points_batch(masks, depths)
                    if args.per_step:
                        print(f"[bench] step {1e3 * (time.perf_counter() - t_s):.3f} ms", file=sys.stderr)
        if len(sels) == 1:
            worker(0)
        else:
            th = [threading.Thread(target=worker, args=(i,)) for i in range(len(sels))]
            for t in th:
                t.start()
            for t in th:
                t.join()

    # untimed passes: warm-up, then one pass with an event pair around EVERY kernel (kernels_ms, mfma);
    # the timed region only keeps the dominant kernel's self-stamped events (mode 2: no extra stream packets)
    run_steps(max(args.warmup, len(sels)) if args.warmup else 0)
    for sel in sels:
        lib.lg_profile_enable(sel._h, 1)
    run_steps(2 * len(sels))
    torch.cuda.synchronize(dev)
    kern_all = {}
    for name in ("prep", "bbox", "orient", "stem", "dt_hrun", "dt_search", "dt_band", "dt_fwd", "dt_bwd", "dt_border", "final", "topk", "gather", "cnn"):
        tot_n, tot_ms = 0, 0.0
        for sel in sels:
            n, ms = C.c_int(0), C.c_double(0.0)
            lib.lg_profile_read(sel._h, name.encode(), C.byref(n), C.byref(ms))
            tot_n += n.value
            tot_ms += ms.value
        if tot_n:
            kern_all[name] = {"launches": tot_n, "avg_ms": tot_ms / tot_n}
    for sel in sels:
        lib.lg_profile_enable(sel._h, 2)
    run_steps(len(sels))  # untimed rehearsal in exactly the timed configuration (the first such call costs a one-off ~40 ms)
    for sel in sels:
        lib.lg_profile_enable(sel._h, 2)  # reset the counters
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    res = [r for r in last if r is not None][0]
    assert all(r[0] is not None for r in res), "synthetic frames must yield a grasp point"

    # per-kernel device time (HIP events recorded on the launch stream inside the library)
    kern = dict(kern_all)
    tot_n, tot_ms = 0, 0.0
    for sel in sels:  # dominant kernel: HIP events over the TIMED region
        n, ms = C.c_int(0), C.c_double(0.0)
        lib.lg_profile_read(sel._h, b"final", C.byref(n), C.byref(ms))
        tot_n += n.value
        tot_ms += ms.value
    if tot_n:
        kern["final"] = {"launches": tot_n, "avg_ms": tot_ms / tot_n}
    for sel in sels:
        lib.lg_profile_enable(sel._h, 0)

    if rank == 0:
        frames = world * B * args.steps
        fps = frames / elapsed
        px = B * H * W
        out = {
            "metric": BASELINE_METRIC,
            "value": round(fps, 2),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{H}x{W} depth+mask, 8 score planes + valid + top-20 NMS + "
                                   f"{'CV only' if args.no_cnn else 'GraspPointCNN fp32 rescoring of 20 candidates'}",
                       "frames_per_step_per_gpu": B, "height": H, "width": W, "batches_in_flight": len(sels),
                       "parallelism": f"frames sharded over {world} GPU(s), no collective"},
            "path_hbm_frac": round(fps / world * PATH_BYTES_PER_PX * H * W / (HBM_PEAK_GBS * 1e9), 5),
            "kernels_ms": {k: round(v["avg_ms"], 4) for k, v in kern.items()},
        }
        if "final" in kern:
            # algorithmic bytes of THIS launch: tiles without a leaf pixel in stencil reach take the constant-store path
            # (no depth read); dense figure (every tile on the full path) = 37.25 B/px
            n_distinct = min(B, N_DISTINCT)
            wins = [sels[0].dt_maxima(i)[2] for i in range(n_distinct)]
            per_distinct = [final_kernel_bytes(masks_np[i:i + 1], wins[i:i + 1], H, W) for i in range(n_distinct)]
            launch_bytes = sum(per_distinct[i % n_distinct] for i in range(B))
            achieved = launch_bytes / (kern["final"]["avg_ms"] * 1e-3) / 1e9
            # HBM traffic of this kernel from rocprofv3 PMC passes (FETCH_SIZE doubled per the gfx950 note,
            # + WRITE_SIZE), recorded per pixel in profiles/ by tools/pmc_final.sh; scaled to this launch.
            traffic, traffic_file = None, None
            for cand in ("r04_pmc_counters.json", "r03_pmc_counters.json", "r02_pmc_counters.json"):
                try:
                    with open(os.path.join(REPO, "profiles", cand)) as fpmc:
                        traffic = round(json.load(fpmc)["lg_final_kernel_summary"]["traffic_bytes_per_px"] * px)
                    traffic_file = cand
                    break
                except Exception:  # noqa: BLE001
                    traffic = None
            out["roofline"] = {"kernel": "lg_final_kernel", "bound": "hbm", "achieved": round(achieved, 1),
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                               "traffic": traffic,
                               "traffic_source": f"profiles/{traffic_file}: rocprofv3 --pmc passes of a 32-frame 1080p launch of this "
                                                 "kernel (2 x FETCH_SIZE + WRITE_SIZE, tools/pmc_final.sh), bytes per pixel x this "
                                                 "launch's pixels -- NOT counted during this run",
                               "bytes_per_launch": round(launch_bytes),
                               "bytes_per_px": round(launch_bytes / px, 3), "dense_bytes_per_px": FINAL_BYTES_PER_PX,
                               "distinct_scenes": n_distinct}
        try:   # every other kernel of the path: compulsory bytes, average duration, fraction of the HBM peak
            n_distinct = min(B, N_DISTINCT)
            wins_k = [sels[0].dt_maxima(i)[2] for i in range(n_distinct)]
            forms_k = [sels[0].dt_form(i) for i in range(n_distinct)]
            out["kernel_fracs"] = kernel_fracs({k: v["avg_ms"] for k, v in kern.items()}, masks_np[:n_distinct], wins_k, B, H, W,
                                               forms=forms_k)
            out["kernel_fracs"]["dt_form"] = {"searched_frames": sum(f[0] for f in forms_k), "d_out_sweeps_skipped": sum(f[1] for f in forms_k),
                                              "of": n_distinct, "what": "which form of the distance transform the batch took (decided on "
                                                                        "the device, lg_bbox_kernel): row search or the two sweeps"}
        except Exception as e:  # noqa: BLE001
            out["kernel_fracs"] = {"error": str(e)}
        if "cnn" in kern and not args.no_cnn:
            # MFMA flops actually executed per 9x32x32 patch (2 x MACs): Winograd F(4x4,3x3) = 36 positions x Cout x Cin x
            # tiles (4x fewer than direct), layer 0 on 12 padded input planes; F(2x2,3x3) = 16 positions (2.25x fewer) with
            # layer 0 direct on 10 padded planes; direct = the reference network's 312.83 MFLOP / patch.
            # `direct_equivalent` prices the same launches with that 312.83 MFLOP.
            direct = os.environ.get("LG_CNN_DIRECT") is not None
            f23 = os.environ.get("LG_CNN_F23") is not None
            wino_macs = lambda pos, tiles: pos * (64 * 64 * tiles[0] + 128 * 64 * tiles[1] + 128 * 128 * tiles[1]  # noqa: E731
                                                  + 256 * 128 * tiles[2] + 256 * 256 * tiles[2])
            if direct:
                exec_fl, kname = 312.83e6, "lg_conv0_kernel + lg_conv3x3_kernel x5 (direct implicit GEMM) + head"
            elif f23:
                exec_fl = 2 * (64 * 90 * 1024 + wino_macs(16, (256, 64, 16)))
                kname = "lg_conv0_kernel (layer 0 direct) + lg_wino_kernel x5 (Winograd F(2x2,3x3)) + head"
            else:
                exec_fl = 2 * (36 * 64 * 12 * 64 + wino_macs(36, (64, 16, 4)))
                kname = "lg_wino4_kernel x6 (Winograd F(4x4,3x3), layer 0 on 12 padded planes) + head"
            sec = kern["cnn"]["avg_ms"] * 1e-3
            tf = exec_fl * 20 * B / sec / 1e12
            out["mfma"] = {"kernel": kname, "achieved": round(tf, 2), "peak": 157.3, "unit": "TFLOP/s", "frac": round(tf / 157.3, 4),
                           "executed_mflop_per_patch": round(exec_fl / 1e6, 2),
                           "direct_equivalent": round(312.83e6 * 20 * B / sec / 1e12, 2)}
        if world == 1 and args.cpu_frames > 0:
            times, serial = cpu_baseline(H, W, P, params, args.cpu_frames, cpu_share)
            out["cpu_baseline"] = {
                "value": round(len(times) / sum(times), 4), "unit": "frames/s",
                "cores": cpu_share, "kind": "port",
                "median_s_per_frame": round(float(np.median(times)), 4), "min_s_per_frame": round(float(np.min(times)), 4),
                "torch_threads": torch.get_num_threads(), "os_cpu_count": os.cpu_count(),
                "c_single_thread_ms": serial,
                "sample": f"{len(times)} synthetic {H}x{W} frames (after 1 warm-up) through oracle/lg_oracle.py: the restated "
                          f"reference-equivalent CPU path (NumPy float64 planes + C chamfer / contour code single-threaded, "
                          f"numpy argsort, 20 batch-1 torch-CPU CNN forwards on {cpu_share} threads); OpenCV / scikit-fmm "
                          f"replaced by in-repo equivalents -- not the reference's own timing"}
        if world == 1 and args.dense_steps > 0:
            # secondary roofline (never `value`): the STENCIL path of lg_final_kernel on every tile.  The headline launch is
            # ~96 % constant-store tiles (a leaf covers a few per cent of a frame); here the masks are "every leaf" (labels >= 1)
            # and the handle is created with the constant-tile fast path off (LG_NO_SKIP, read at lg_create), so each tile
            # reads depth (+ distance inside the sweep window, or writes it outside), runs the 5x5 Gaussian + 3x3 Sobel through
            # LDS and writes 7 planes + valid: 37.25 B/px for every pixel of the launch.
            try:
                os.environ["LG_NO_SKIP"] = "1"
                dsel = L.GraspPointSelector(dev, load_model=False)
                del os.environ["LG_NO_SKIP"]
                dsel.set_camera_params(P)
                nd = min(B, 128)
                dm = torch.from_numpy(np.stack([labels_np[i % len(labels_np)] >= 1 for i in range(nd)]).astype(np.uint8)).to(dev)
                dd = depths[:nd]
                dsel.score_maps(dm, dd)
                # which state of the part this leg sees: the headline's plane launch, timed right before it (2.6-2.7 ms per 256
                # frames on a fresh part, 3.2-3.3 after ~30 s of load: tools/final_variance.sh)
                try:
                    lib.lg_profile_enable(sels[0]._h, 2)
                    for _ in range(2):
                        sels[0].select_grasp_points_batch(masks, depths)
                    torch.cuda.synchronize(dev)
                    n_s, ms_s = C.c_int(0), C.c_double(0.0)
                    lib.lg_profile_read(sels[0]._h, b"final", C.byref(n_s), C.byref(ms_s))
                    lib.lg_profile_enable(sels[0]._h, 0)
                    state_ms = round(ms_s.value / max(1, n_s.value), 4)
                except Exception as e_state:  # noqa: BLE001
                    state_ms = f"not measured: {e_state}"
                lib.lg_profile_enable(dsel._h, 2)
                for _ in range(args.dense_steps):
                    dsel.score_maps(dm, dd)
                torch.cuda.synchronize(dev)
                n_l, ms_l = C.c_int(0), C.c_double(0.0)
                lib.lg_profile_read(dsel._h, b"final", C.byref(n_l), C.byref(ms_l))
                lib.lg_profile_enable(dsel._h, 0)
                dbytes = FINAL_BYTES_PER_PX * nd * H * W
                dach = dbytes / (ms_l.value / max(1, n_l.value) * 1e-3) / 1e9
                ceiling = None   # the same loads / stores without arithmetic (tools/ubench/stream_mix.hip), a committed measurement
                try:
                    with open(os.path.join(REPO, "profiles", "r02_ubench_stream_mix.txt")) as fmix:
                        for line in fmix:
                            if line.startswith("tiled    64x16  same mix"):
                                ceiling = float(line.split("(")[1].split()[0])
                except (OSError, ValueError, IndexError):
                    pass
                traffic_d = None
                for cand in ("r03_pmc_final_dense.json", "r02_pmc_final_dense.json"):
                    try:
                        with open(os.path.join(REPO, "profiles", cand)) as fpmc:
                            traffic_d = round(json.load(fpmc)["traffic_bytes_per_px"] * nd * H * W)
                        break
                    except Exception:  # noqa: BLE001
                        traffic_d = None
                out["roofline_dense"] = {"kernel": "lg_final_kernel (every tile on the stencil path)", "bound": "hbm",
                                         "achieved": round(dach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                         "frac": round(dach / HBM_PEAK_GBS, 4), "traffic": traffic_d,
                                         "bytes_per_px": FINAL_BYTES_PER_PX, "frames_per_launch": nd, "launches": n_l.value,
                                         "avg_ms": round(ms_l.value / max(1, n_l.value), 4),
                                         "masks": "labels >= 1 (all leaves, 20-30 % of the frame), LG_NO_SKIP=1",
                                         "headline_final_ms_just_before": state_ms,
                                         "part_state": "the headline's plane launch timed right before this leg: 2.6-2.7 ms per 256 "
                                                       "frames = fresh part, 3.2-3.3 = warmed-up part (tools/final_variance.sh); this "
                                                       "leg's fraction moves with it",
                                         "traffic_mix_ceiling_frac": ceiling,
                                         "traffic_mix_ceiling": "profiles/r02_ubench_stream_mix.txt: a kernel with only this "
                                                                "path's loads and stores (2 reads + 7 float planes + 1 byte plane "
                                                                "per pixel, the same 64x16 tiles), measured on another run"}
                dsel = dm = dd = None
            except Exception as e:  # noqa: BLE001
                out["roofline_dense"] = {"error": str(e)}
        if world == 1 and args.pipelined > 1 and len(sels) == 1 and not args.no_cnn:
            # secondary figure (never `value`): the same steps with several batches in flight -- one handle, stream and host thread
            # each, all on this GPU -- so that one batch's latency-bound stages (sweeps, top-k: one workgroup per frame) and host work
            # run beside another batch's CNN / plane kernels.  The stretched per-kernel times show what the overlap costs.
            try:
                psels, pstreams = list(sels), list(streams)
                while len(psels) < args.pipelined:
                    ps_ = L.GraspPointSelector(dev, load_model=False)
                    ps_.set_camera_params(P)
                    ps_.set_cnn_state_dict(params)
                    psels.append(ps_)
                    pstreams.append(torch.cuda.Stream(dev))
                plast = [None] * len(psels)

                def prun(n_steps):
                    def worker(i):
                        torch.cuda.set_device(dev)
                        with torch.cuda.stream(pstreams[i]):
                            for _ in range(i, n_steps, len(psels)):
                                plast[i] = psels[i].select_grasp_points_batch(masks, depths)
                    th = [threading.Thread(target=worker, args=(i,)) for i in range(len(psels))]
                    for t_ in th:
                        t_.start()
                    for t_ in th:
                        t_.join()
                prun(2 * len(psels))
                for ps_ in psels:
                    lib.lg_profile_enable(ps_._h, 1)
                prun(2 * len(psels))
                torch.cuda.synchronize(dev)
                pk = {}
                for name in ("dt_hrun", "dt_search", "dt_band", "dt_fwd", "dt_bwd", "final", "topk", "gather", "cnn"):
                    tn_, tm_ = 0, 0.0
                    for ps_ in psels:
                        n_, ms_ = C.c_int(0), C.c_double(0.0)
                        lib.lg_profile_read(ps_._h, name.encode(), C.byref(n_), C.byref(ms_))
                        tn_ += n_.value
                        tm_ += ms_.value
                    if tn_:
                        pk[name] = round(tm_ / tn_, 4)
                for ps_ in psels:
                    lib.lg_profile_enable(ps_._h, 0)
                prun(len(psels))
                torch.cuda.synchronize(dev)
                t_p = time.perf_counter()
                prun(args.steps)
                torch.cuda.synchronize(dev)
                dt_p = time.perf_counter() - t_p
                same = all(r == res for r in plast if r is not None)
                out["pipelined"] = {"value": round(B * args.steps / dt_p, 2), "unit": "frames/s", "batches_in_flight": len(psels),
                                    "steps": args.steps, "ms_per_step": round(1e3 * dt_p / args.steps, 4),
                                    "vs_single": round(B * args.steps / dt_p / fps, 4), "results_equal_single": bool(same),
                                    "kernels_ms_stretched": pk}
                psels = pstreams = plast = None
            except Exception as e:  # noqa: BLE001
                out["pipelined"] = {"error": str(e)}
        if world == 1 and args.config_steps > 0 and not args.no_cnn:
            out["configs"] = secondary_configs(args, dev, L, lib, SI, params, P, extra_frames, masks_np, depths_np, labels_np, H, W)
        if world == 1 and args.h2d_steps > 0 and not args.no_cnn:
            # secondary figure (never `value`, SURVEY 8d "Timing method"): the same scoring pass with depth f32 + mask u8 arriving
            # from PINNED host memory every step (10.4 MB per 1080p frame over PCIe Gen5)
            try:
                nh = min(B, 64)
                hm = masks_np[:nh].view(np.uint8)
                h_mask = torch.from_numpy(hm).pin_memory()
                h_depth = torch.from_numpy(depths_np[:nh]).pin_memory()
                d_mask = torch.empty_like(h_mask, device=dev)
                d_depth = torch.empty_like(h_depth, device=dev)

                def h2d_step():
                    d_mask.copy_(h_mask, non_blocking=True)
                    d_depth.copy_(h_depth, non_blocking=True)
                    return sels[0].select_grasp_points_batch(d_mask.view(torch.bool), d_depth)
                h2d_step()
                torch.cuda.synchronize(dev)
                t_h = time.perf_counter()
                for _ in range(args.h2d_steps):
                    r_h = h2d_step()
                torch.cuda.synchronize(dev)
                dt_h = time.perf_counter() - t_h
                torch.cuda.synchronize(dev)
                t_c = time.perf_counter()
                for _ in range(args.h2d_steps):
                    d_mask.copy_(h_mask, non_blocking=True)
                    d_depth.copy_(h_depth, non_blocking=True)
                torch.cuda.synchronize(dev)
                dt_c = time.perf_counter() - t_c
                # the same with the NEXT batch's copy in flight on a second stream (two device buffers) while this one is scored.
                # Which stream: ROCm deals a process's streams onto a few hardware queues in creation order, and a copy stream that
                # lands on the queue of the stream the scoring runs on (or of one of the selector's own streams) runs behind the
                # scoring instead of beside it -- whether it does depends on how many streams the process created before
                # (tools/h2d_probe.py: 5331 vs 4458 frames/s for the same code after three more selectors had come and gone; that,
                # not the library, is what BENCH_r03's h2d figure lost against BENCH_r02's).  So: a few candidate streams (one of
                # them high priority: a queue of its own), two overlapped steps with each, the fastest one carries the copies --
                # what a deployment would do once at start-up.
                bufs = [(d_mask, d_depth), (torch.empty_like(d_mask), torch.empty_like(d_depth))]
                evs = [torch.cuda.Event(), torch.cuda.Event()]

                def overlapped(cs_, n_):
                    def issue_copy(k):
                        with torch.cuda.stream(cs_):
                            bufs[k][0].copy_(h_mask, non_blocking=True)
                            bufs[k][1].copy_(h_depth, non_blocking=True)
                            evs[k].record(cs_)
                    torch.cuda.synchronize(dev)
                    t0_ = time.perf_counter()
                    issue_copy(0)
                    r_ = None
                    for i in range(n_):
                        evs[i % 2].synchronize()
                        if i + 1 < n_:
                            issue_copy((i + 1) % 2)
                        r_ = sels[0].select_grasp_points_batch(bufs[i % 2][0].view(torch.bool), bufs[i % 2][1])
                    torch.cuda.synchronize(dev)
                    return time.perf_counter() - t0_, r_
                cands = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev, priority=-1),
                         torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
                probe = [round(1e3 * overlapped(c_, 2)[0] / 2, 3) for c_ in cands]
                cs = cands[int(np.argmin(probe))]
                n_o = 2 * args.h2d_steps
                dt_o, r_o = overlapped(cs, n_o)
                out["h2d_inclusive"] = {"value": round(nh * n_o / dt_o, 1), "unit": "frames/s",
                                        "frames_per_step": nh, "steps": n_o,
                                        "serial_value": round(nh * args.h2d_steps / dt_h, 1),
                                        "copy_only_GBps": round(nh * H * W * 5 * args.h2d_steps / dt_c / 1e9, 2),
                                        "copy_stream_probe_ms_per_step": probe,
                                        "what": "pinned host depth f32 + mask u8 -> device, then the scoring pass incl. CNN; value: "
                                                "the next batch's copy runs on a second stream beside the scoring of this one (two "
                                                "device buffers); serial_value: copy and scoring on one stream, one after the other"}
                assert all(r[0] is not None for r in r_h) and all(r[0] is not None for r in r_o)
                h_mask = h_depth = d_mask = d_depth = bufs = cands = cs = None
            except Exception as e:  # noqa: BLE001
                out["h2d_inclusive"] = {"error": str(e)}
        if world == 1 and args.node_steps > 0 and not args.no_cnn:
            # secondary figure (never `value`): the node's whole per-frame sequence, leaf_grasp_node_v3.py:102-158 --
            # OptimalLeafSelector over the int16 label image, then GraspPointSelector on the chosen leaf -- batched
            try:
                nb = min(B, 128)
                nsc = len(labels_np)
                lab = torch.from_numpy(np.stack([labels_np[i % nsc] for i in range(nb)])).to(dev)
                dep = depths[:nb] if nb <= B and B % nsc == 0 else torch.from_numpy(np.stack([depths_np[i % nsc] for i in range(nb)])).to(dev)
                hz = L.LeafGraspHarness(H, W, dev, load_model=False)
                hz.camera_info_callback(np.asarray(P).reshape(-1))
                hz.grasp_selector.set_cnn_state_dict(params)
                hz.process_batch_device(lab, dep)
                torch.cuda.synchronize(dev)
                t_n = time.perf_counter()
                for _ in range(args.node_steps):
                    csvs = hz.process_batch_device(lab, dep)
                torch.cuda.synchronize(dev)
                dt_n = time.perf_counter() - t_n
                # the leaf stage on its own (lg_leaf_select_batch): wall time, per-kernel event times, fraction of its 6 B/px bound
                try:
                    ols = hz.leaf_scorer
                    ols.select_optimal_leaves_batch(lab, dep)
                    torch.cuda.synchronize(dev)
                    t_l = time.perf_counter()
                    for _ in range(args.node_steps):
                        ols.select_optimal_leaves_batch(lab, dep)
                    torch.cuda.synchronize(dev)
                    dt_l = (time.perf_counter() - t_l) / args.node_steps
                    lib.lg_profile_enable(ols._h, 1)
                    ols.select_optimal_leaves_batch(lab, dep)
                    lk = {}
                    for name in ("leaf_presence", "leaf_accumulate", "leaf_hist", "leaf_select", "leaf_edt", "leaf_pack"):
                        n_, ms_ = C.c_int(0), C.c_double(0.0)
                        lib.lg_profile_read(ols._h, name.encode(), C.byref(n_), C.byref(ms_))
                        if n_.value:
                            lk[name] = round(ms_.value, 4)      # total per call (leaf_hist / leaf_select: all passes)
                    lib.lg_profile_enable(ols._h, 0)
                    leaf_px = float(np.mean([(labels_np[i % nsc] > 0).sum() for i in range(nb)]))
                    lbytes = {"leaf_presence": nb * (H * W * 2 + H * ((W + 63) // 64) * 8),
                              "leaf_accumulate": nb * (H * W * 2 + leaf_px * (4 + 8)),
                              "leaf_hist": nb * leaf_px * 8 * 3,
                              "leaf_edt": nb * H * ((W + 63) // 64) * 8 * 2}
                    out["leaf_stage"] = {"ms_per_call": round(1e3 * dt_l, 4), "frames_per_call": nb,
                                         "frames_per_s": round(nb / dt_l, 1),
                                         "frac_of_6B_per_px_bound": round(nb * H * W * 6 / dt_l / (HBM_PEAK_GBS * 1e9), 4),
                                         "kernels_ms": lk,
                                         "kernel_fracs": {k: round(v / (lk[k] * 1e-3) / (HBM_PEAK_GBS * 1e9), 4) for k, v in lbytes.items() if lk.get(k)},
                                         "leaf_pixels_per_frame": round(leaf_px),
                                         "what": "lg_leaf_select_batch on the int16 label + depth frames (leaf_scorer.py:25-203): compulsory "
                                                 "traffic 6 B/px; kernels_ms = event time per call on the kernel's own stream (leaf_edt runs "
                                                 "beside the statistics chain); kernel_fracs = that kernel's compulsory bytes / its time / 8 TB/s"}
                except Exception as e:  # noqa: BLE001
                    out["leaf_stage"] = {"error": str(e)}
                out["node_sequence"] = {"value": round(nb * args.node_steps / dt_n, 1), "unit": "frames/s",
                                        "frames_per_step": nb, "steps": args.node_steps,
                                        "results": sum(c is not None for c in csvs),
                                        "what": "batched leaf selection (lg_leaf_select_batch: device statistics + native host Pareto pick) + grasp "
                                                "selection incl. CNN, int16 labels + depth resident in HBM"}
            except Exception as e:  # noqa: BLE001
                out["node_sequence"] = {"error": str(e)}
            # release the harness (two more library handles with their streams) before the next leg: how many streams a
            # process holds decides how ROCm maps them to its few hardware queues
            hz = lab = dep = csvs = None
            import gc
            gc.collect()
        if world == 1 and args.train_steps > 0 and not args.no_cnn:
            # secondary figure (never `value`): SURVEY 8f row 4, the inner loop body of scripts/train_model.py:247-265
            try:
                import ctypes as _C
                from leafgrasp_amd._lib import lib as _lib
                from leafgrasp_amd.trainer import GraspTrainer
                ts = {}
                for nb in (16, 1024):
                    tr = GraspTrainer(dev, max_batch=nb)
                    base = torch.from_numpy(SI.synthetic_patches(256, seed=1)).to(dev)
                    xb = base.repeat((nb + 255) // 256, 1, 1, 1)[:nb].contiguous()
                    yb = (torch.arange(nb, device=dev) % 3 == 0).float()
                    loss = _C.c_float()

                    def one(sync):
                        rc = _lib.lg_train_step(tr._h, xb.data_ptr(), yb.data_ptr(), nb, None, 1, _C.byref(tr.hp), 1,
                                                _C.byref(loss) if sync else None, None, None)
                        if rc != 0:
                            raise RuntimeError(_lib.lg_train_last_error(tr._h).decode())
                    torch.cuda.synchronize(dev)
                    for _ in range(3):
                        one(False)
                    one(True)
                    t_t = time.perf_counter()
                    for _ in range(args.train_steps - 1):
                        one(False)
                    one(True)
                    dt_t = (time.perf_counter() - t_t) / args.train_steps
                    ts[f"batch_{nb}"] = {"ms_per_step": round(dt_t * 1e3, 4), "samples_per_s": round(nb / dt_t, 1)}
                    del tr
                ts["what"] = ("GraspPointCNN([64,128,256], spatial attention) forward (train mode) + BCEWithLogits + backward + "
                              "clip_grad_norm + Adam per step, fp32 MFMA convolutions, patches resident in HBM, dropout masks "
                              "drawn on the device; batch 16 = scripts/train_model.py:207")
                ts["steps"] = args.train_steps
                out["train_step"] = ts
            except Exception as e:  # noqa: BLE001
                out["train_step"] = {"error": str(e)}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
