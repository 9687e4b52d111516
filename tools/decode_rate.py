#!/usr/bin/env python3
"""The host-feed ceiling of the streaming frontend: how many images per second the reader stage of yolov2_detect can
decode on this host's cores, by image kind and thread count - the number that bounds `--devices 0,1,...` from the host side, since one
process shares ONE decode pool between its device lanes.  Decoding is host/y2_codec.cpp through libyolo2_host.so (y2h_decode_image,
which releases nothing Python-side: ctypes drops the GIL around the call), images are made here with PIL (synthetic content).
usage: python3 tools/decode_rate.py [seconds per point = 1.0]"""
import ctypes, io, os, sys, threading, time
import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H = ctypes.CDLL(os.path.join(ROOT, "yolo-fpga-accelerator_amd", "libyolo2_host.so"))
H.y2h_decode_image.restype = ctypes.c_long
H.y2h_decode_image.argtypes = [ctypes.c_char_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long]


def picture(w, h, seed):
    """camera-like content: smooth gradients + blocks + a little noise (a flat image decodes unrealistically fast)"""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([(x * 255 // w), (y * 255 // h), ((x + y) * 255 // (w + h))], -1).astype(np.float32)
    for _ in range(12):
        x0, y0 = int(rng.integers(0, w - 8)), int(rng.integers(0, h - 8))
        img[y0:y0 + int(rng.integers(8, h // 2)), x0:x0 + int(rng.integers(8, w // 2))] = rng.integers(0, 255, 3)
    img += rng.normal(0, 6, img.shape)
    return Image.fromarray(np.clip(img, 0, 255).astype(np.uint8))


def encode(kind, w, h):
    out = []
    for s in range(8):
        b = io.BytesIO(); im = picture(w, h, s)
        if kind == "png":
            im.save(b, "PNG")
        elif kind == "jpeg420":
            im.save(b, "JPEG", quality=90, subsampling=2)
        elif kind == "jpeg444":
            im.save(b, "JPEG", quality=90, subsampling=0)
        elif kind == "jpegprog":
            im.save(b, "JPEG", quality=90, subsampling=2, progressive=True)
        out.append(b.getvalue())
    return out


def rate(blobs, w, h, threads, seconds):
    stop = time.perf_counter() + seconds
    counts = [0] * threads
    def work(i):
        buf = (ctypes.c_ubyte * (w * h * 3))(); ww = ctypes.c_int(); hh = ctypes.c_int(); k = i
        while time.perf_counter() < stop:
            d = blobs[k % len(blobs)]; k += 1
            n = H.y2h_decode_image(d, len(d), ctypes.byref(ww), ctypes.byref(hh), buf, w * h * 3)
            assert n == w * h * 3, n
            counts[i] += 1
    ts = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    return sum(counts) / (time.perf_counter() - t0)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    cores = len(os.sched_getaffinity(0))
    print(f"# tools/decode_rate.py: host/y2_codec.cpp decode rate, images/s, {cores} cores available to this process")
    print(f"# {'image':28s} {'bytes':>8s} " + " ".join(f"{t:>3d} thr" .rjust(9) for t in (1, 4, 8, 16) if t <= 2 * cores))
    for kind, w, h in (("jpeg420", 416, 416), ("jpeg420", 640, 480), ("jpeg444", 640, 480), ("jpegprog", 640, 480),
                       ("jpeg420", 1280, 720), ("jpeg420", 1920, 1080), ("png", 640, 480)):
        blobs = encode(kind, w, h)
        rate(blobs, w, h, 4, 0.3)          # warm: tables, allocator arenas of the worker threads
        rs = [max(rate(blobs, w, h, t, seconds) for _ in range(2)) for t in (1, 4, 8, 16) if t <= 2 * cores]   # best of two: shared hosts are noisy
        print(f"  {kind + ' ' + str(w) + 'x' + str(h):28s} {len(blobs[0]):8d} " + " ".join(f"{r:9.0f}" for r in rs), flush=True)


if __name__ == "__main__":
    main()
