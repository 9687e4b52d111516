import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolo-fpga-accelerator_amd"))
import numpy as np, torch
from yolo2_amd import hipdrv, synth
model = synth.SynthModel(seed=1)
def run(nctx, B, steps=6):
    ctxs = [hipdrv.Yolo2Hip(0) for _ in range(nctx)]
    for c in ctxs: c.load_model(model); c.set_batch(B)
    frames = [torch.from_numpy(synth.frames(7, B, first=i*B)).cuda() for i in range(nctx)]
    regs = [torch.empty((B,425,13,13), dtype=torch.int16, device="cuda") for _ in range(nctx)]
    streams = [torch.cuda.Stream() for _ in range(nctx)]
    for _ in range(2):
        for c,f,r,s in zip(ctxs,frames,regs,streams): c.run_batch_ptr(f.data_ptr(), B, r.data_ptr(), s.cuda_stream)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(steps):
        for c,f,r,s in zip(ctxs,frames,regs,streams): c.run_batch_ptr(f.data_ptr(), B, r.data_ptr(), s.cuda_stream)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/steps
    print(f"{nctx} stream(s) x batch {B}: {dt*1e3:.2f} ms per round, {nctx*B/dt:.1f} FPS")
    for c in ctxs: c.close()
run(1, 64); run(2, 32); run(2, 64); run(4, 16); run(1, 128)
