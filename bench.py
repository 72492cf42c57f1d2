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
CPU_BASELINE_THREADS = 16  # one GPU's share of the box's host cores (torch intra-op threads of the CPU leg)


def make_frames(B, H, W, n_distinct=4):
    import synthetic_inputs as SI  # seeded scenes (inputs; the oracle is only touched by the cpu_baseline leg)

    masks, depths, P = [], [], None
    for s in range(min(B, n_distinct)):
        labels, depth, P = SI.synthetic_scene(H, W, seed=100 + s)
        masks.append(labels == 1)
        depths.append(depth)
    reps = (B + len(masks) - 1) // len(masks)
    m = np.stack((masks * reps)[:B])
    d = np.stack((depths * reps)[:B])
    return m, d, P


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


def cpu_baseline(H, W, P, params, n_frames):
    """Restated reference-equivalent CPU path (oracle, NumPy/torch-CPU + C chamfer): same work as
    scripts/utils/grasp_point_selector.py::select_grasp_point.  NOT the reference's own timing
    (OpenCV / scikit-fmm are absent from this image)."""
    from oracle import lg_oracle as O

    torch.set_num_threads(CPU_BASELINE_THREADS)
    ref = O.RefGraspPointSelector(cnn=lambda x: O.cnn_forward(params, x))
    ref.set_camera_params(P)
    times = []
    for s in range(n_frames + 1):
        labels, depth, _ = O.synthetic_scene(H, W, seed=100 + (s % 4))
        mask = (labels == 1).astype(np.uint8)
        t0 = time.perf_counter()
        ref.select_grasp_point(mask, depth, tie_rule="numpy")
        dt = time.perf_counter() - t0
        if s > 0:  # first frame is warm-up
            times.append(dt)
    return times


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256,
                    help="frames per step per GPU (256 puts two sweep workgroups on every CU; 24 GB of the 288 GB)")
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--cpu-frames", type=int, default=3, help="frames timed for the cpu_baseline leg (0 = skip)")
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
    ap.add_argument("--per-step", action="store_true", help="diagnostic: print every step's wall time to stderr "
                                                            "(adds a device sync per step; not the headline mode)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
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

    H, W, B = args.height, args.width, args.batch
    masks_np, depths_np, P = make_frames(B, H, W)
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
    for name in ("prep", "bbox", "stem", "dt_fwd", "dt_bwd", "dt_border", "final", "topk", "gather", "cnn"):
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
            n_distinct = min(B, 4)
            wins = [sels[0].dt_maxima(i)[2] for i in range(n_distinct)]
            per_distinct = [final_kernel_bytes(masks_np[i:i + 1], wins[i:i + 1], H, W) for i in range(n_distinct)]
            launch_bytes = sum(per_distinct[i % n_distinct] for i in range(B))
            achieved = launch_bytes / (kern["final"]["avg_ms"] * 1e-3) / 1e9
            # HBM traffic of this kernel from rocprofv3 PMC passes (FETCH_SIZE doubled per the gfx950 note,
            # + WRITE_SIZE), recorded per pixel in profiles/ by tools/pmc_final.sh; scaled to this launch.
            traffic = None
            try:
                with open(os.path.join(REPO, "profiles", "r01_pmc_counters.json")) as fpmc:
                    traffic = round(json.load(fpmc)["lg_final_kernel_summary"]["traffic_bytes_per_px"] * px)
            except Exception:  # noqa: BLE001
                traffic = None
            out["roofline"] = {"kernel": "lg_final_kernel", "bound": "hbm", "achieved": round(achieved, 1),
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                               "traffic": traffic, "bytes_per_launch": round(launch_bytes),
                               "bytes_per_px": round(launch_bytes / px, 3), "dense_bytes_per_px": FINAL_BYTES_PER_PX}
        if "cnn" in kern and not args.no_cnn:
            # MFMA flops actually executed per 9x32x32 patch: layer 0 direct (K = 9 taps x 10 padded channels),
            # layers 1..5 Winograd F(2x2,3x3) = 16 positions x 2 x Cout x Cin x tiles (2.25x fewer than direct).
            # `direct_equivalent` prices the same launches with the reference network's 312.83 MFLOP / patch.
            direct = os.environ.get("LG_CNN_DIRECT") is not None
            exec_fl = 312.83e6 if direct else (2 * 64 * 90 * 1024 + 16 * 2 * (64 * 64 * 256 + 128 * 64 * 64 + 128 * 128 * 64
                                                                             + 256 * 128 * 16 + 256 * 256 * 16))
            sec = kern["cnn"]["avg_ms"] * 1e-3
            tf = exec_fl * 20 * B / sec / 1e12
            out["mfma"] = {"kernel": "lg_conv0_kernel (layer 0) + lg_wino_kernel x5 (Winograd F(2x2,3x3)) + head"
                                     if not direct else "lg_conv3x3_kernel x6 + head",
                           "achieved": round(tf, 2), "peak": 157.3, "unit": "TFLOP/s", "frac": round(tf / 157.3, 4),
                           "executed_mflop_per_patch": round(exec_fl / 1e6, 2),
                           "direct_equivalent": round(312.83e6 * 20 * B / sec / 1e12, 2)}
        if world == 1 and args.cpu_frames > 0:
            times = cpu_baseline(H, W, P, params, args.cpu_frames)
            out["cpu_baseline"] = {
                "value": round(len(times) / sum(times), 4), "unit": "frames/s",
                "cores": CPU_BASELINE_THREADS, "kind": "port",
                "sample": f"{len(times)} synthetic {H}x{W} frames through oracle/lg_oracle.py "
                          f"(restated NumPy/torch-CPU + C chamfer path, numpy argsort, 20 batch-1 CNN forwards; "
                          f"NumPy planes single-threaded, torch CPU ops on {CPU_BASELINE_THREADS} threads); "
                          f"os.cpu_count()={os.cpu_count()}"}
        if world == 1 and args.node_steps > 0 and not args.no_cnn:
            # secondary figure (never `value`): the node's whole per-frame sequence, leaf_grasp_node_v3.py:102-158 --
            # OptimalLeafSelector over the int16 label image, then GraspPointSelector on the chosen leaf -- batched
            try:
                nb = min(B, 128)
                scenes = [SI.synthetic_scene(H, W, seed=100 + s) for s in range(min(nb, 4))]
                lab = torch.from_numpy(np.stack([scenes[i % len(scenes)][0] for i in range(nb)]).astype(np.int16)).to(dev)
                dep = torch.from_numpy(np.stack([scenes[i % len(scenes)][1] for i in range(nb)])).to(dev)
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
                out["node_sequence"] = {"value": round(nb * args.node_steps / dt_n, 1), "unit": "frames/s",
                                        "frames_per_step": nb, "steps": args.node_steps,
                                        "results": sum(c is not None for c in csvs),
                                        "what": "batched leaf selection (lg_leaf_stats_batch + host Pareto) + grasp "
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
