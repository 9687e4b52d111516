"""Drop-in proof: the reference's COMPLETE Linux application -- linux_app/src/*.c
compiled from its own sources with its own headers, minus the two FPGA driver files (yolo2_accel_linux.c,
dma_buffer_manager.c) -- linked against libyolo2_hip.so (oracle/Makefile target `ref_app`, build container only;
the prebuilt binary travels to the GPU box like oracle/_ref's libraries).  Its unchanged layer loop
(linux_app/src/yolo2_inference.c:763-910) drives the GPU through yolo2_execute_conv_layer /
yolo2_execute_maxpool_layer, 28 calls per frame, and must print the region tensor the compiled reference's CPU
path computed for the same image (tests/golden/refapp.npz)."""
import importlib.util
import os
import re
import subprocess

import numpy as np
import pytest

import orclib
from yolo2_amd import synth

pytestmark = pytest.mark.gpu
ROOT = orclib.ROOT
APP = os.path.join(ROOT, "oracle", "_ref", "yolo2_linux")
PKG = os.path.join(ROOT, "yolo-fpga-accelerator_amd")
REFAPP = np.load(os.path.join(ROOT, "tests", "golden", "refapp.npz"))


def _mg():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    return mg


def write_ppm(path, rgb):
    h, w, _ = rgb.shape
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (w, h))
        f.write(np.ascontiguousarray(rgb, dtype=np.uint8).tobytes())


@pytest.mark.parametrize("qset", ["std", "varq"])
def test_reference_linux_app_runs_unchanged_on_the_gpu(qset, tmp_path):
    if not os.path.exists(APP):
        pytest.skip("oracle/_ref/yolo2_linux is built in the build container only (make -C oracle ref_app)")
    mg = _mg()
    model = synth.SynthModel(seed=1, **mg.Q_SETS[qset])
    wdir = tmp_path / "weights"
    model.write_files(str(wdir), fp32=False, int16=True)
    img = tmp_path / "frame416.ppm"
    write_ppm(str(img), mg.refapp_image())
    env = dict(os.environ, YOLO2_VERBOSE="1")
    cmd = [APP, "-i", str(img), "-w", str(wdir), "-c", os.path.join(PKG, "config", "yolov2.cfg"),
           "-l", os.path.join(PKG, "config", "coco.names"), "-t", "0.05", "-v", "1"]
    r = subprocess.run(cmd, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "Accelerator driver initialized OK" in r.stdout and "DMA buffer manager initialized OK" in r.stdout
    m = re.search(r"Inference time: ([0-9.]+) ms", r.stdout)      # the line scripts/yolo2_report.py:685-729 parses
    assert m and float(m.group(1)) > 0
    # one frame = 23 conv + 5 maxpool calls into the driver tier, nothing else (reorg/route/region stay on its host)
    m2 = re.search(r"driver served (\d+) layer calls", r.stderr)
    assert m2 and int(m2.group(1)) == 28, r.stderr[-2000:]
    # its own dump of the dequantised region tensor (main.c:806-817, "%.9g" per line) == the reference CPU path's
    raw = np.loadtxt(str(tmp_path / "yolov2_region_raw_hw.txt"), dtype=np.float64)
    q = int(REFAPP[f"{qset}/final_q"])
    assert raw.size == 425 * 169
    ri = np.rint(raw * (1 << q)).astype(np.int64)
    assert np.all(np.abs(ri * 2.0 ** -q - raw) <= 1e-7 * np.maximum(1.0, np.abs(raw)))
    want = REFAPP[f"{qset}/region_raw_i16"].astype(np.int64)
    assert np.array_equal(ri, want), f"{int((ri != want).sum())} of {want.size} region values differ"
    assert os.path.exists(tmp_path / "yolov2_region_proc_hw.txt")
