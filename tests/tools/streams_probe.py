"""How the training step's time depends on the number of other streams alive in the process when the trainer is created
(ROCm maps streams to a few hardware queues in creation order).  With LG_TRAIN_STREAMS=2 (backward-weights on a second
stream) 2 of 9 counts gave a 2.2-2.5x slower step; the default one-stream step is flat (0.611-0.616 ms at batch 16).
    python tests/tools/streams_probe.py            # default build: one stream
    LG_TRAIN_STREAMS=2 python tests/tools/streams_probe.py"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import synthetic_inputs as S
from leafgrasp_amd._lib import lib
from leafgrasp_amd.trainer import GraspTrainer
dev = torch.device("cuda:0")
x = torch.from_numpy(S.synthetic_patches(16, seed=1)).to(dev)
y = (torch.arange(16, device=dev) % 3 == 0).float()
extra = []
for k in range(0, 9):
    tr = GraspTrainer(dev, max_batch=16)
    loss = C.c_float()
    def step(sync):
        lib.lg_train_step(tr._h, x.data_ptr(), y.data_ptr(), 16, None, 1, C.byref(tr.hp), 1, C.byref(loss) if sync else None, None, None)
    for _ in range(5): step(False)
    step(True)
    t0 = time.perf_counter()
    for _ in range(49): step(False)
    step(True)
    print(f"extra idle streams {len(extra)}: {(time.perf_counter()-t0)/50*1e3:.3f} ms/step", flush=True)
    del tr
    extra.append(torch.cuda.Stream(device=dev))     # one more idle non-default stream alive in the process
    with torch.cuda.stream(extra[-1]):
        torch.zeros(1, device=dev)                  # make sure the runtime has created it
    torch.cuda.synchronize()
