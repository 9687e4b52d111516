"""YOLOv2 layer table (config/yolov2.cfg of the reference) and work/byte accounting.

The table is what the reference's parser prints for config/yolov2.cfg
(src/core/yolo_net.cpp:218-291; 1x1 convs get padding size/2 = 0, src/core/yolo_layers.cpp:98).
`ord` is the conv ordinal that indexes the weight/bias/Q tables
(hls/models/yolov2/model_config.cpp:4-10, yolo2_model.cpp:299-340).
"""
from dataclasses import dataclass
from typing import List, Optional

CONV, MAXPOOL, ROUTE, REORG, REGION = "conv", "max", "route", "reorg", "region"
TN, TM = 4, 32  # hls/core/params.hpp (scripts/hw_params_gen.py:19-22)


@dataclass(frozen=True)
class Layer:
    idx: int
    type: str
    c: int = 0
    h: int = 0
    w: int = 0
    n: int = 0
    size: int = 0
    stride: int = 1
    pad: int = 0
    leaky: bool = False
    ord: Optional[int] = None

    @property
    def out_h(self):
        if self.type == CONV:
            return (self.h - self.size + 2 * self.pad) // self.stride + 1
        if self.type in (MAXPOOL, REORG):
            return self.h // 2
        return self.h

    @property
    def out_w(self):
        if self.type == CONV:
            return (self.w - self.size + 2 * self.pad) // self.stride + 1
        if self.type in (MAXPOOL, REORG):
            return self.w // 2
        return self.w

    @property
    def out_c(self):
        if self.type == CONV:
            return self.n
        if self.type == REORG:
            return self.c * 4
        return self.c


def _build() -> List[Layer]:
    L = []
    o = 0

    def conv(c, hw, n, k, leaky=True):
        nonlocal o
        L.append(Layer(len(L), CONV, c, hw, hw, n, k, 1, 1 if k == 3 else 0, leaky, o))
        o += 1

    def mp(c, hw):
        L.append(Layer(len(L), MAXPOOL, c, hw, hw, c, 2, 2, 0))

    conv(3, 416, 32, 3); mp(32, 416)
    conv(32, 208, 64, 3); mp(64, 208)
    conv(64, 104, 128, 3); conv(128, 104, 64, 1); conv(64, 104, 128, 3); mp(128, 104)
    conv(128, 52, 256, 3); conv(256, 52, 128, 1); conv(128, 52, 256, 3); mp(256, 52)
    conv(256, 26, 512, 3); conv(512, 26, 256, 1); conv(256, 26, 512, 3); conv(512, 26, 256, 1)
    conv(256, 26, 512, 3); mp(512, 26)
    conv(512, 13, 1024, 3); conv(1024, 13, 512, 1); conv(512, 13, 1024, 3); conv(1024, 13, 512, 1)
    conv(512, 13, 1024, 3); conv(1024, 13, 1024, 3); conv(1024, 13, 1024, 3)
    L.append(Layer(len(L), ROUTE, 512, 26, 26))                     # 25: layer 16
    conv(512, 26, 64, 1)                                            # 26
    L.append(Layer(len(L), REORG, 64, 26, 26, 256, 0, 2))           # 27
    L.append(Layer(len(L), ROUTE, 1280, 13, 13))                    # 28: concat(27, 24)
    conv(1280, 13, 1024, 3)                                         # 29
    conv(1024, 13, 425, 1, leaky=False)                             # 30
    L.append(Layer(len(L), REGION, 425, 13, 13))                    # 31
    return L


LAYERS: List[Layer] = _build()
CONVS: List[Layer] = [l for l in LAYERS if l.type == CONV]
WEIGHT_LEN = [l.n * l.c * l.size * l.size for l in CONVS]
BIAS_LEN = [l.n for l in CONVS]
N_WEIGHTS = sum(WEIGHT_LEN)          # 50,941,792
N_BIAS = sum(BIAS_LEN)               # 10,761
REGION_ELEMS = 425 * 13 * 13
ANCHORS = [0.57273, 0.677385, 1.87446, 2.06253, 3.33843, 5.47434, 7.88282, 3.52778, 9.77052, 9.16828]


def to_cfg() -> str:
    """A Darknet .cfg for this network with only the keys the parsers read (our
    host/y2_host.cpp and the reference's src/core/yolo_layers.cpp:90-117,312-326)."""
    out = ["[net]", "batch=1", "width=416", "height=416", "channels=3", ""]
    for l in LAYERS:
        if l.type == CONV:
            out += ["[convolutional]"] + (["batch_normalize=1"] if l.leaky else []) + \
                   [f"filters={l.n}", f"size={l.size}", "stride=1", "pad=1",
                    "activation=" + ("leaky" if l.leaky else "linear"), ""]
        elif l.type == MAXPOOL:
            out += ["[maxpool]", "size=2", "stride=2", ""]
        elif l.type == ROUTE:
            out += ["[route]", "layers=" + ("-9" if l.idx == 25 else "-1,-4"), ""]
        elif l.type == REORG:
            out += ["[reorg]", "stride=2", ""]
        elif l.type == REGION:
            out += ["[region]", "anchors = " + ", ".join(repr(a) for a in ANCHORS), "classes=80", "coords=4",
                    "num=5", "softmax=1", ""]
    return "\n".join(out)


def w8(w: int) -> int:
    return (w + 7) // 8 * 8


def macs_per_frame() -> int:
    return sum(l.n * l.c * l.size * l.size * l.out_h * l.out_w for l in CONVS)


def requant_steps_per_frame() -> int:
    """One requantise-and-saturate step per (4 input channels, tap, output) -- SURVEY.md 8(d)."""
    return sum(l.size * l.size * ((l.c + TN - 1) // TN) * l.n * l.out_h * l.out_w for l in CONVS)


def activation_elems_per_frame() -> int:
    """Layer-at-a-time model of SURVEY.md 8(d): each conv/maxpool/reorg reads its input once
    and writes its output once, logical (unpadded) sizes."""
    t = 0
    for l in LAYERS:
        if l.type in (CONV, MAXPOOL, REORG):
            t += l.c * l.h * l.w + l.out_c * l.out_h * l.out_w
    return t


def algorithmic_bytes_per_frame(batch: int, elem_bytes: int = 2) -> float:
    return activation_elems_per_frame() * elem_bytes + (N_WEIGHTS + N_BIAS) * elem_bytes / batch


assert N_WEIGHTS == 50941792 and N_BIAS == 10761
assert macs_per_frame() == 14732084224
assert requant_steps_per_frame() == 3695481088
