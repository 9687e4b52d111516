import sys, os, numpy as np
ROOT='/root/repo'
sys.path.insert(0, os.path.join(ROOT,'yolo-fpga-accelerator_amd')); sys.path.insert(0, os.path.join(ROOT,'tests'))
from yolo2_amd import hipdrv, synth
FULL = np.load(os.path.join(ROOT,'tests','golden','fullnet.npz'))
m = synth.SynthModel(seed=1)
f = synth.frames(7,1)
ctx = hipdrv.Yolo2Hip(0)
ctx.load_weights_fp32(m.weights_f32(), m.bias_f32())
r = ctx.run_batch_fp16_host(f)
want = FULL['f32/std/region_raw_f32'].reshape(425,13,13)
d = np.abs(r[0]-want)
print('max abs err', d.max(), 'mean abs', d.mean(), 'ref max', np.abs(want).max(), 'ref std', want.std())
print('corr', np.corrcoef(r[0].ravel(), want.ravel())[0,1])
