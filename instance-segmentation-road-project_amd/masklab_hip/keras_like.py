"""A minimal Keras-`Layer`-shaped host framework over the HIP operators.

Mirrors the protocol the reference's layers rely on: `__init__(**hyperparams)`,
`call(inputs, **kwargs)`, `get_config()`, name-based registry, weight sharing by object identity.
Execution is eager: `layer(x)` enqueues kernels on the current stream.

Shapes are propagated symbolically by `build(input_shape)` (H/W may be None) so that weight
shapes are known -- and weights can be created / packed -- without touching a GPU.

Weight names: "<layer name>/<weight>" in Keras layouts.  Layers the reference names explicitly
keep that name; sub-layers Keras would auto-number get a deterministic hierarchical name.
"""
import collections
import re
import zlib

import numpy as np

from . import _lib, ops, packing

_name_counts = collections.defaultdict(int)


def clear_session():
    """tf.keras.backend.clear_session(): reset the automatic layer-name counters."""
    _name_counts.clear()


def _snake(name):
    s = re.sub("(.)([A-Z][a-z0-9]+)", r"\1_\2", name)
    return re.sub("([a-z])([A-Z])", r"\1_\2", s).lower()


def unique_name(cls_name):
    base = _snake(cls_name)
    n = _name_counts[base]
    _name_counts[base] += 1
    return base if n == 0 else f"{base}_{n}"


class WeightSpec:
    def __init__(self, shape, init, **kw):
        self.shape = tuple(int(s) for s in shape)
        self.init = init          # 'glorot_uniform' | 'zeros' | 'ones' | 'normal' | 'constant' | 'he_normal' ...
        self.kw = kw              # stddev= / value=

    def make(self, rng):
        shape = self.shape
        if self.init == "zeros":
            return np.zeros(shape, np.float32)
        if self.init == "ones":
            return np.ones(shape, np.float32)
        if self.init == "constant":
            return np.full(shape, self.kw["value"], np.float32)
        if self.init == "normal":
            return rng.normal(0.0, self.kw.get("stddev", 0.05), shape).astype(np.float32)
        if len(shape) >= 2:
            receptive = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
            fan_in, fan_out = shape[-2] * receptive, shape[-1] * receptive
            if self.kw.get("depthwise"):
                fan_in, fan_out = receptive, receptive * shape[-1]
        else:
            fan_in = fan_out = shape[0]
        if self.init == "glorot_uniform":
            lim = np.sqrt(6.0 / (fan_in + fan_out))
            return rng.uniform(-lim, lim, shape).astype(np.float32)
        if self.init == "glorot_normal":
            return rng.normal(0, np.sqrt(2.0 / (fan_in + fan_out)), shape).astype(np.float32)
        if self.init == "he_normal":
            return rng.normal(0, np.sqrt(2.0 / fan_in), shape).astype(np.float32)
        if self.init == "uniform":
            return rng.uniform(self.kw["low"], self.kw["high"], shape).astype(np.float32)
        raise ValueError(self.init)


class Layer:
    """Base class: Keras `Layer` surface (name, get_config, __call__ -> call)."""

    def __init__(self, name=None, trainable=True, dtype="float32", **kwargs):
        if kwargs:
            raise TypeError(f"{type(self).__name__}: unexpected keyword arguments {sorted(kwargs)}")
        self.name = name or unique_name(type(self).__name__)
        self.trainable = trainable
        self.built = False
        self._specs = collections.OrderedDict()   # weight name (relative) -> WeightSpec

    # -- Keras surface
    def __call__(self, inputs, **kwargs):
        return self.call(inputs, **kwargs)

    def call(self, inputs, **kwargs):
        raise NotImplementedError

    def get_config(self):
        return {"name": self.name, "trainable": self.trainable}

    @classmethod
    def from_config(cls, config):
        return cls(**config)

    # -- build / weights
    def build(self, input_shape):
        """Create weight specs; return the output shape.  Default: shape-preserving, no weights."""
        self.built = True
        return input_shape

    def children(self):
        return []

    def add_weight(self, name, shape, init, **kw):
        self._specs[name] = WeightSpec(shape, init, **kw)

    def weight_specs(self):
        """{full weight name: WeightSpec} of this layer and everything below it."""
        out = collections.OrderedDict()
        for k, v in self._specs.items():
            out[f"{self.name}/{k}"] = v
        for ch in self.children():
            out.update(ch.weight_specs())
        return out

    def load_weights(self, weights, device):
        """Pack + upload this layer's weights from a {name: ndarray} dict (recursive)."""
        self._load_own(weights, device)
        for ch in self.children():
            ch.load_weights(weights, device)

    def _load_own(self, weights, device):
        pass

    def _get(self, weights, key):
        full = f"{self.name}/{key}"
        if full not in weights:
            raise KeyError(f"weight '{full}' missing from the weight dict")
        arr = np.asarray(weights[full], np.float32)
        want = self._specs[key].shape
        if tuple(arr.shape) != want:
            raise ValueError(f"weight '{full}' has shape {arr.shape}, expected {want}")
        return arr


def seed_for(name, seed=0):
    return (zlib.crc32(name.encode()) + 7919 * seed) & 0xFFFFFFFF


def init_weights(specs, seed=0):
    """Deterministic per-name random init (independent of creation order)."""
    return {name: spec.make(np.random.default_rng(seed_for(name, seed))) for name, spec in specs.items()}


# =========================================================================== primitives
def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


class Conv2D(Layer):
    """tf.keras.layers.Conv2D on the MFMA implicit-GEMM kernel (csrc/conv_mfma.hip).
    `fold_bn=(bn_name, eps, scale)` declares an inference BatchNormalization that follows
    this conv in the reference graph; it is folded into the packed weights."""

    def __init__(self, filters, kernel_size, strides=(1, 1), padding="valid", dilation_rate=(1, 1),
                 activation=None, use_bias=True, kernel_initializer="glorot_uniform", kernel_stddev=None,
                 bias_initializer="zeros", bias_value=None, fold_bn=None, image_input=False, **kwargs):
        super().__init__(**kwargs)
        self.filters = int(filters)
        self.kernel_size = _pair(kernel_size)
        self.strides = _pair(strides)
        self.padding = padding
        self.dilation_rate = _pair(dilation_rate)
        self.activation = activation
        self.use_bias = use_bias
        self.kernel_initializer = kernel_initializer
        self.kernel_stddev = kernel_stddev
        self.bias_initializer = bias_initializer
        self.bias_value = bias_value
        self.fold_bn = fold_bn
        self.image_input = image_input   # input is the channel-padded NHWC4 image
        if self.strides[0] != self.strides[1] or self.dilation_rate[0] != self.dilation_rate[1]:
            raise NotImplementedError("anisotropic stride / dilation")
        self.dev = None

    def build(self, input_shape):
        cin = 3 if self.image_input else int(input_shape[-1])
        self.cin = cin
        kh, kw = self.kernel_size
        kw_init = {"stddev": self.kernel_stddev} if self.kernel_stddev is not None else {}
        self.add_weight("kernel", (kh, kw, cin, self.filters), self.kernel_initializer, **kw_init)
        if self.use_bias:
            if self.bias_value is not None:
                self.add_weight("bias", (self.filters,), "constant", value=self.bias_value)
            else:
                self.add_weight("bias", (self.filters,), self.bias_initializer)
        self.built = True
        H, W = input_shape[1], input_shape[2]
        if H is None or W is None:
            return (input_shape[0], None, None, self.filters)
        Ho, Wo, _, _ = packing.resolve_padding(H, W, kh, kw, self.strides[0], self.dilation_rate[0], self.padding)
        return (input_shape[0], Ho, Wo, self.filters)

    def bn_specs(self):
        """Weight specs of the folded BatchNormalization (owned by its own Keras name)."""
        if not self.fold_bn:
            return {}
        name, _eps, scale = self.fold_bn[:3]
        lo, hi = self.fold_bn[3] if len(self.fold_bn) > 3 else (0.5, 1.5)
        c = self.filters
        out = collections.OrderedDict()
        if scale:
            out[f"{name}/gamma"] = WeightSpec((c,), "uniform", low=lo, high=hi)
        out[f"{name}/beta"] = WeightSpec((c,), "normal", stddev=0.1)
        out[f"{name}/moving_mean"] = WeightSpec((c,), "normal", stddev=0.1)
        out[f"{name}/moving_variance"] = WeightSpec((c,), "uniform", low=0.5, high=1.5)
        return out

    def weight_specs(self):
        out = super().weight_specs()
        out.update(self.bn_specs())
        return out

    def folded(self, weights):
        """(kernel, bias) after BN folding, Keras layout -- numpy, no GPU."""
        k = self._get(weights, "kernel")
        b = self._get(weights, "bias") if self.use_bias else None
        if self.fold_bn:
            name, eps, scale = self.fold_bn[:3]
            g = np.asarray(weights[f"{name}/gamma"], np.float32) if scale else None
            k, b = packing.fold_bn(k, b, g, np.asarray(weights[f"{name}/beta"], np.float32),
                                   np.asarray(weights[f"{name}/moving_mean"], np.float32),
                                   np.asarray(weights[f"{name}/moving_variance"], np.float32), eps)
        return k, b

    def pack(self, weights):
        k, b = self.folded(weights)
        if self.image_input:
            return packing.pack_rowspan(k, b)
        return packing.pack_dense(k, b)

    def _load_own(self, weights, device):
        self.dev = ops.DeviceConv(self.pack(weights), device)

    def call(self, x, residual=None, out=None, out_coff=0, out_dtype=None, gn_partials=None, **kwargs):
        if self.dev is None:
            raise RuntimeError(f"layer '{self.name}' has no weights loaded")
        return ops.conv2d(x, self.dev, stride=self.strides[0], padding=self.padding,
                          dilation=self.dilation_rate[0], act=_lib.ACT_BY_NAME[self.activation],
                          residual=residual, out=out, out_coff=out_coff, out_dtype=out_dtype, gn_partials=gn_partials)

    def get_config(self):
        c = super().get_config()
        c.update(filters=self.filters, kernel_size=self.kernel_size, strides=self.strides, padding=self.padding,
                 dilation_rate=self.dilation_rate, activation=self.activation, use_bias=self.use_bias)
        return c


class GroupedConv2D(Conv2D):
    """ResNeXt's grouped 3x3 as the reference stores it: a DepthwiseConv2D kernel
    [3,3,filters,c] named '<name>/depthwise_kernel' (ResNext.py:214) whose SplitGroups /
    ReduceGroups / MergeGroups epilogue (:217-219) is folded into the weight re-layout."""

    def __init__(self, filters, groups, strides=(1, 1), padding=((1, 1), (1, 1)), activation=None,
                 fold_bn=None, **kwargs):
        super().__init__(filters, (3, 3), strides=strides, padding=padding, activation=activation,
                         use_bias=False, fold_bn=fold_bn, **kwargs)
        self.groups = groups

    def build(self, input_shape):
        assert int(input_shape[-1]) == self.filters
        self.cin = self.filters
        c = self.filters // self.groups
        # he-style init over the real fan-in (9*c) keeps activations O(1)
        self.add_weight("depthwise_kernel", (3, 3, self.filters, c), "normal",
                        stddev=float(np.sqrt(2.0 / (9 * c))))
        self.built = True
        H, W = input_shape[1], input_shape[2]
        if H is None or W is None:
            return (input_shape[0], None, None, self.filters)
        Ho, Wo, _, _ = packing.resolve_padding(H, W, 3, 3, self.strides[0], 1, self.padding)
        return (input_shape[0], Ho, Wo, self.filters)

    def folded(self, weights):
        k = self._get(weights, "depthwise_kernel")          # [3,3,filters(in), c(m)]
        b = None
        if self.fold_bn:
            name, eps, scale = self.fold_bn[:3]
            g = np.asarray(weights[f"{name}/gamma"], np.float32) if scale else None
            c = self.filters // self.groups
            # BN acts on OUTPUT channel g*c+m; kernel element [.., in=g*c+i, m] feeds output g*c+m
            sc = (np.ones(self.filters) if g is None else g.astype(np.float64)) / \
                np.sqrt(np.asarray(weights[f"{name}/moving_variance"], np.float64) + eps)
            grp = np.arange(self.filters) // c
            out_idx = grp[:, None] * c + np.arange(c)[None, :]          # [in, m] -> output channel
            k = (k.astype(np.float64) * sc[out_idx][None, None]).astype(np.float32)
            b = (np.asarray(weights[f"{name}/beta"], np.float64) -
                 np.asarray(weights[f"{name}/moving_mean"], np.float64) * sc).astype(np.float32)
        return k, b

    def pack(self, weights):
        """dense block-diagonal packing for the generic MFMA conv (kept as a cross-check path)"""
        k, b = self.folded(weights)
        return packing.pack_grouped(k, self.groups, b)

    def _load_own(self, weights, device):
        import torch
        k, b = self.folded(weights)
        self.c = self.filters // self.groups
        # small groups at stride 1 are HBM-bound: the 16-block 4x4x1 MFMA kernel has no padding waste;
        # c >= 16 (and the four stride-2 convs) run faster on the dense 32-wide MFMA tiles
        self.use_mfma4 = self.c <= 16 and self.filters % 64 == 0
        if self.use_mfma4:
            self.wgt4 = torch.from_numpy(packing.pack_grouped_mfma4(k, self.groups)).to(device)
            self.bias4 = None if b is None else torch.from_numpy(np.ascontiguousarray(b, np.float32)).to(device)
            self.dev = True
        else:
            self.dev = ops.DeviceConv(packing.pack_grouped(k, self.groups, b), device)
            # half tensors (fp16-storage mode) with groups of 32 run csrc/gconv_mfma4.hip's 32x32x16 form: the [C][9][c]
            # packing of the small-group kernels, made on first use
            self._k32, self._b32, self._dev32, self.wgt4, self.bias4 = (k, b, device, None, None) if self.c == 32 and \
                self.filters % 64 == 0 else (None, None, None, None, None)

    def call(self, x, **kwargs):
        if self.dev is None:
            raise RuntimeError(f"layer '{self.name}' has no weights loaded")
        return grouped3x3(self, x)


def grouped3x3(layer, x):
    """The grouped 3x3 of a ResNeXt block on whichever kernel fits its group width; in the fp16-storage mode (half
    `x`) the 32-channel groups of the last stage run on the 32x32x16 form of the grouped kernel (round 4; before: an fp32
    copy of the input through the dense kernel)."""
    import torch
    act = _lib.ACT_BY_NAME[layer.activation]
    if layer.use_mfma4:
        return ops.gconv3x3(x, layer.wgt4, layer.bias4, layer.c, stride=layer.strides[0], padding=layer.padding, act=act)
    if x.dtype == torch.float16 and getattr(layer, "_k32", None) is not None:
        if layer.wgt4 is None:
            layer.wgt4 = torch.from_numpy(packing.pack_grouped_mfma4(layer._k32, layer.groups)).to(layer._dev32)
            layer.bias4 = None if layer._b32 is None else torch.from_numpy(np.ascontiguousarray(layer._b32, np.float32)).to(layer._dev32)
        return ops.gconv3x3(x, layer.wgt4, layer.bias4, layer.c, stride=layer.strides[0], padding=layer.padding, act=act)
    if x.dtype == torch.float16:
        return ops.conv2d(ops.cast_h2f(x), layer.dev, stride=layer.strides[0], padding=layer.padding, act=act,
                          out_dtype=torch.float16)
    return ops.conv2d(x, layer.dev, stride=layer.strides[0], padding=layer.padding, act=act)


class DepthwiseConv2D(Layer):
    """3x3 DepthwiseConv2D (depth_multiplier 1) on the VALU depthwise kernel."""

    def __init__(self, kernel_size=(3, 3), strides=(1, 1), padding="valid", dilation_rate=(1, 1),
                 activation=None, use_bias=True, fold_bn=None, **kwargs):
        super().__init__(**kwargs)
        if _pair(kernel_size) != (3, 3):
            raise NotImplementedError("only 3x3 depthwise kernels are on the hot path")
        self.strides = _pair(strides)
        self.padding = padding
        self.dilation_rate = _pair(dilation_rate)
        self.activation = activation
        self.use_bias = use_bias
        self.fold_bn = fold_bn
        self.wgt = self.bias = None

    def build(self, input_shape):
        self.C = int(input_shape[-1])
        self.add_weight("depthwise_kernel", (3, 3, self.C, 1), "glorot_uniform", depthwise=True)
        if self.use_bias:
            self.add_weight("bias", (self.C,), "zeros")
        self.built = True
        H, W = input_shape[1], input_shape[2]
        if H is None or W is None:
            return (input_shape[0], None, None, self.C)
        Ho, Wo, _, _ = packing.resolve_padding(H, W, 3, 3, self.strides[0], self.dilation_rate[0], self.padding)
        return (input_shape[0], Ho, Wo, self.C)

    def weight_specs(self):
        out = super().weight_specs()
        if self.fold_bn:
            name, _eps, scale = self.fold_bn[:3]
            if scale:
                out[f"{name}/gamma"] = WeightSpec((self.C,), "uniform", low=0.5, high=1.5)
            out[f"{name}/beta"] = WeightSpec((self.C,), "normal", stddev=0.1)
            out[f"{name}/moving_mean"] = WeightSpec((self.C,), "normal", stddev=0.1)
            out[f"{name}/moving_variance"] = WeightSpec((self.C,), "uniform", low=0.5, high=1.5)
        return out

    def folded(self, weights):
        k = self._get(weights, "depthwise_kernel")
        b = self._get(weights, "bias") if self.use_bias else None
        if self.fold_bn:
            name, eps, scale = self.fold_bn[:3]
            g = np.asarray(weights[f"{name}/gamma"], np.float32) if scale else None
            k, b = packing.fold_bn(k, b, g, np.asarray(weights[f"{name}/beta"], np.float32),
                                   np.asarray(weights[f"{name}/moving_mean"], np.float32),
                                   np.asarray(weights[f"{name}/moving_variance"], np.float32), eps, depthwise=True)
        return k, b

    def _load_own(self, weights, device):
        import torch
        k, b = self.folded(weights)
        self.wgt = torch.from_numpy(packing.pack_depthwise(k)).to(device)
        self.bias = None if b is None else torch.from_numpy(np.ascontiguousarray(b)).to(device)

    def call(self, x, **kwargs):
        if self.wgt is None:
            raise RuntimeError(f"layer '{self.name}' has no weights loaded")
        return ops.dwconv3x3(x, self.wgt, self.bias, stride=self.strides[0], padding=self.padding,
                             dilation=self.dilation_rate[0], act=_lib.ACT_BY_NAME[self.activation])


class Conv2DTranspose(Layer):
    """Conv2DTranspose(filters,(2,2),(2,2),'same') = 1x1 MFMA GEMM (N=4*filters) + pixel shuffle."""

    def __init__(self, filters, kernel_size=(2, 2), strides=(2, 2), padding="same", activation=None,
                 kernel_stddev=None, **kwargs):
        super().__init__(**kwargs)
        if _pair(kernel_size) != (2, 2) or _pair(strides) != (2, 2):
            raise NotImplementedError("only the 2x2 stride-2 transposed conv is on the hot path")
        self.filters = int(filters)
        self.activation = activation
        self.kernel_stddev = kernel_stddev
        self.dev = None

    def build(self, input_shape):
        self.cin = int(input_shape[-1])
        kw = {"stddev": self.kernel_stddev} if self.kernel_stddev is not None else {}
        self.add_weight("kernel", (2, 2, self.filters, self.cin),
                        "normal" if self.kernel_stddev is not None else "glorot_uniform", **kw)
        self.add_weight("bias", (self.filters,), "zeros")
        self.built = True
        H, W = input_shape[1], input_shape[2]
        return (input_shape[0], None if H is None else 2 * H, None if W is None else 2 * W, self.filters)

    def pack(self, weights):
        return packing.pack_transpose2x2(self._get(weights, "kernel"), self._get(weights, "bias"))

    def _load_own(self, weights, device):
        self.dev = ops.DeviceConv(self.pack(weights), device)

    def call(self, x, **kwargs):
        return ops.conv2d(x, self.dev, act=_lib.ACT_BY_NAME[self.activation])


class Dense(Layer):
    """Dense without bias on a [B,1,1,C] map (SqueezeExcite, misc.py:34-40) = 1x1 conv.
    The MFMA conv reads 16-byte channel quads, so a width that is not a multiple of 4
    (e.g. 160 // 16 = 10 in the decoder's SqueezeExcite) is carried in a zero-padded buffer:
    the output tensor has ceil4(units) channels (extra ones 0) and the next Dense pads its kernel."""

    def __init__(self, units, activation=None, kernel_initializer="glorot_uniform", **kwargs):
        super().__init__(**kwargs)
        self.units = int(units)
        self.activation = activation
        self.kernel_initializer = kernel_initializer
        self.dev = None

    def build(self, input_shape):
        self.cin = int(input_shape[-1])
        self.add_weight("kernel", (self.cin, self.units), self.kernel_initializer)
        self.built = True
        return tuple(input_shape[:-1]) + (self.units,)

    def _load_own(self, weights, device):
        k = self._get(weights, "kernel")
        cin_pad = -(-self.cin // 4) * 4
        kp = np.zeros((1, 1, cin_pad, self.units), np.float32)
        kp[0, 0, :self.cin] = k
        self.dev = ops.DeviceConv(packing.pack_dense(kp, None), device)

    def call(self, x, **kwargs):
        import torch
        units_pad = -(-self.units // 4) * 4
        out = None
        if units_pad != self.units:
            out = torch.zeros(tuple(x.shape[:-1]) + (units_pad,), dtype=torch.float32, device=x.device)
        return ops.conv2d(x, self.dev, act=_lib.ACT_BY_NAME[self.activation], out=out)
