"""ResNeXt-101 (32x4d) body -- BUILD-SIDE EXTENSION: the reference vendors the
architecture in thirdparty/classification_models/models/resnext.py (conv_block :62-99, identity_block
:102-135, ResNeXt :138-241, repetitions (3,4,23,3) :248-252) but never wires it into `load_backbone`,
so there is no reference preprocessing / tap choice to mirror.  Choices made here: raw RGB 0..255 into
the model's own `bn_data` input BatchNorm (scale=False, :189-194; applied in the preprocess kernel
because it precedes the zero padding), taps C1=relu0, C2..C5 = last unit's relu of stages 1..4.
Layer names follow the vendored file; BatchNorm eps = 2e-5 (:44-54).  GroupConv2D (_common_blocks.py:13-76,
32 x [slice -> Conv2D -> concat]) stores its 32 kernels as one weight `<conv_name>2/kernel` [32,3,3,c,c]."""
import numpy as np

from .. import _lib, ops, packing
from ..keras_like import Conv2D, Layer, WeightSpec

BN_EPS = 2e-5
REPETITIONS = (3, 4, 23, 3)


class SlicedGroupConv2D(Conv2D):
    """thirdparty GroupConv2D: out[g*c+m] = sum_i conv3x3(x[g*c+i], K[g,:,:,i,m]) (+ folded BN + ReLU)."""

    def __init__(self, filters, groups=32, strides=(1, 1), fold_bn=None, activation=None, **kwargs):
        super().__init__(filters, (3, 3), strides=strides, padding=((1, 1), (1, 1)), activation=activation,
                         use_bias=False, fold_bn=fold_bn, **kwargs)
        self.groups = groups

    def build(self, input_shape):
        assert int(input_shape[-1]) == self.filters
        self.cin = self.filters
        c = self.filters // self.groups
        self.add_weight("kernel", (self.groups, 3, 3, c, c), "normal", stddev=float(np.sqrt(2.0 / (9 * c))))
        self.built = True
        H, W = input_shape[1], input_shape[2]
        if H is None or W is None:
            return (input_shape[0], None, None, self.filters)
        Ho, Wo, _, _ = packing.resolve_padding(H, W, 3, 3, self.strides[0], 1, self.padding)
        return (input_shape[0], Ho, Wo, self.filters)

    def folded(self, weights):
        k = self._get(weights, "kernel")                                   # [g,kh,kw,i,m]
        c = self.filters // self.groups
        # same layout the in-tree ResNeXt-50 uses: dw[kh,kw,g*c+i,m]
        dw = np.transpose(k, (1, 2, 0, 3, 4)).reshape(3, 3, self.filters, c)
        b = None
        if self.fold_bn:
            name, eps, scale = self.fold_bn[:3]
            g = np.asarray(weights[f"{name}/gamma"], np.float64) if scale else np.ones(self.filters)
            sc = g / np.sqrt(np.asarray(weights[f"{name}/moving_variance"], np.float64) + eps)
            out_idx = (np.arange(self.filters) // c)[:, None] * c + np.arange(c)[None, :]
            dw = (dw.astype(np.float64) * sc[out_idx][None, None]).astype(np.float32)
            b = (np.asarray(weights[f"{name}/beta"], np.float64) -
                 np.asarray(weights[f"{name}/moving_mean"], np.float64) * sc).astype(np.float32)
        return dw, b

    def _load_own(self, weights, device):
        import torch
        dw, b = self.folded(weights)
        self.c = self.filters // self.groups
        self.use_mfma4 = self.c <= 16 and self.filters % 64 == 0
        if self.use_mfma4:
            self.wgt4 = torch.from_numpy(packing.pack_grouped_mfma4(dw, self.groups)).to(device)
            self.bias4 = None if b is None else torch.from_numpy(np.ascontiguousarray(b, np.float32)).to(device)
            self.dev = True
        else:
            self.dev = ops.DeviceConv(packing.pack_grouped(dw, self.groups, b), device)
            # (half tensors with groups of 32: keras_like.grouped3x3 packs for the 32x32x16 grouped kernel on first use)
            self._k32, self._b32, self._dev32, self.wgt4, self.bias4 = (dw, b, device, None, None) if self.c == 32 and \
                self.filters % 64 == 0 else (None, None, None, None, None)

    def call(self, x, **kwargs):
        if self.dev is None:
            raise RuntimeError(f"layer '{self.name}' has no weights loaded")
        from ..keras_like import grouped3x3
        return grouped3x3(self, x)


class _Unit:
    def __init__(self, filters, stage, block, stride, conv_shortcut):
        base = f"stage{stage + 1}_unit{block + 1}_"
        rng = {1: (0.5, 1.5), 2: (0.5, 1.5), 3: (0.1, 0.3)}
        bn = lambda s: (f"{base}bn{s}", BN_EPS, True, rng[s])
        he = dict(kernel_initializer="he_normal")
        self.conv1 = Conv2D(filters, 1, use_bias=False, fold_bn=bn(1), activation='relu', name=base + "conv1", **he)
        self.conv2 = SlicedGroupConv2D(filters, 32, strides=stride, fold_bn=bn(2), activation='relu',
                                       name=base + "conv2")
        self.conv3 = Conv2D(filters * 2, 1, use_bias=False, fold_bn=bn(3), activation='relu',
                            name=base + "conv3", **he)
        self.shortcut = None
        if conv_shortcut:
            self.shortcut = Conv2D(filters * 2, 1, strides=stride, use_bias=False,
                                   fold_bn=(base + "sc_bn", BN_EPS, True, (0.5, 1.0)), name=base + "sc", **he)

    def layers(self):
        return [l for l in (self.conv1, self.conv2, self.conv3, self.shortcut) if l is not None]

    def build(self, shape):
        s = self.conv3.build(self.conv2.build(self.conv1.build(shape)))
        if self.shortcut is not None:
            self.shortcut.build(shape)
        return s

    def __call__(self, x):
        sc = self.shortcut(x) if self.shortcut is not None else x
        return self.conv3(self.conv2(self.conv1(x)), residual=sc)


class ResNeXt101(Layer):
    def __init__(self, repetitions=REPETITIONS, **kwargs):
        super().__init__(name=kwargs.pop("name", "resnext101_body"), **kwargs)
        self.conv0 = Conv2D(64, 7, strides=2, padding=((3, 3), (3, 3)), use_bias=False,
                            fold_bn=("bn0", BN_EPS, True), activation='relu', image_input=True,
                            kernel_initializer="he_normal", name="conv0")
        self.stages = []
        for stage, rep in enumerate(repetitions):
            filters = 128 * 2 ** stage
            units = []
            for block in range(rep):
                stride = 1 if (stage == 0 or block > 0) else 2
                units.append(_Unit(filters, stage, block, stride, conv_shortcut=(block == 0)))
            self.stages.append(units)

    def build(self, input_shape):
        self.add_weight_bn_data()
        s = self.conv0.build(input_shape)
        taps = {"C1": s}
        H, W = s[1], s[2]
        s = (s[0], None if H is None else (H + 2 - 3) // 2 + 1, None if W is None else (W + 2 - 3) // 2 + 1, s[3])
        for tap, units in zip(("C2", "C3", "C4", "C5"), self.stages):
            for u in units:
                s = u.build(s)
            taps[tap] = s
        self.built = True
        return taps

    def add_weight_bn_data(self):
        # input BatchNorm, scale=False: beta / moving stats over raw 0..255 RGB
        self._bn_data = {
            "bn_data/beta": WeightSpec((3,), "normal", stddev=0.1),
            "bn_data/moving_mean": WeightSpec((3,), "uniform", low=100.0, high=130.0),
            "bn_data/moving_variance": WeightSpec((3,), "uniform", low=3000.0, high=5000.0),
        }

    def children(self):
        return [self.conv0] + [l for st in self.stages for u in st for l in u.layers()]

    def weight_specs(self):
        out = dict(self._bn_data)
        for ch in self.children():
            out.update(ch.weight_specs())
        return out

    def input_affine(self, weights):
        """(mean, divisor, shift) realising bn_data inside the preprocess kernel."""
        mean = np.asarray(weights["bn_data/moving_mean"], np.float64)
        div = np.sqrt(np.asarray(weights["bn_data/moving_variance"], np.float64) + BN_EPS)
        return mean.astype(np.float32), div.astype(np.float32), np.asarray(weights["bn_data/beta"], np.float32)

    def call(self, x, wanted=("C3", "C4", "C5"), **kwargs):
        import torch
        half = ops.half_storage()            # fp16-storage mode: the body's tensors AND its taps are IEEE half
        taps = {}
        if half and "C1" not in wanted and self.conv0.dev is not None:
            x = ops.stem_pool_h(x, self.conv0.dev)        # stem + pool in one pass (csrc/stem_h.hip)
        elif ops.CONV_MATH in ("f32", "f32x3") and "C1" not in wanted and self.conv0.dev is not None:
            x = ops.stem_pool(x, self.conv0.dev)          # fp32-tensor twins (csrc/stem_f32.hip, stem_x3.hip): same bits as the pair below
        else:
            x = self.conv0(x, out_dtype=torch.float16 if half else None)
            taps["C1"] = x
            x = ops.maxpool3x3s2(x, pad=1)
        last = max(int(t[1]) for t in wanted)
        for tap, units in zip(("C2", "C3", "C4", "C5"), self.stages):
            for u in units:
                x = u(x)
            taps[tap] = x
            if int(tap[1]) >= last:
                break
        return taps
