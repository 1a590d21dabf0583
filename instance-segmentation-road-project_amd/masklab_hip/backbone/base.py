"""Backbone loader + preprocess -- drop-in for reference engine/backbone/base.py
(BackBonePreProcess :22-84, BACKBONE_LAYERS :104-182, load_backbone :185-316).

Backbones on the hot path: 'resnext50' (in-tree in the reference) and 'mobilenet'
(tf.keras.applications.MobileNet v1).  Everything else raises NotImplementedError like the
reference does for unknown types.  BatchNormalization is folded into the conv weights at load
time; the 3-channel stems read a channel-padded NHWC4 image written by the preprocess kernel.
"""
from .. import ops
from ..keras_like import Conv2D, DepthwiseConv2D, GroupedConv2D, Layer
from ..layers.misc import Identity
from ..normalization import GroupNormalization

BACKBONE_LAYERS = {
    "resnext50": {"C1": 'conv1_relu', "C2": 'conv2_block3_out', "C3": "conv3_block4_out",
                  "C4": "conv4_block6_out", "C5": "conv5_block3_out"},
    # build-side extension: architecture from the vendored thirdparty model zoo
    "resnext101": {"C1": "relu0", "C2": "stage1_unit3_relu", "C3": "stage2_unit4_relu",
                   "C4": "stage3_unit23_relu", "C5": "stage4_unit3_relu"},
    "mobilenet": {"C1": "conv_pw_1_relu", "C2": "conv_pw_3_relu", "C3": "conv_pw_5_relu",
                  "C4": "conv_pw_11_relu", "C5": "conv_pw_13_relu"},
}


class BackBonePreProcess(Layer):
    """Image preprocess (reference :22-84): optional BGR flip, mean shift, normalisation mode
    0 none / 1 [0,1] / 2 [-1,1] / 3 standardisation.  Output is NHWC4 (4th channel zero) so the
    MFMA stem can read 16-byte pixels; `out_channels=3` gives the plain tensor."""

    def __init__(self, rgb=True, mean_shift=False, normalize=0, out_channels=4, **kwargs):
        self.rgb = rgb
        self.mean_shift = mean_shift
        self.normalize = normalize
        self.out_channels = out_channels
        super().__init__(**kwargs)
        self.mean = [123.68, 116.779, 103.939] if rgb else [103.939, 116.779, 123.68]
        self.std = [0.225, 0.224, 0.229] if rgb else [0.229, 0.224, 0.225]
        self.extra_affine = None      # (mean, divisor, shift) of a model-owned input BatchNorm (ResNeXt-101 bn_data)

    def build(self, input_shape):
        self.built = True
        return tuple(input_shape[:3]) + (self.out_channels,)

    def call(self, inputs, **kwargs):
        mean = self.mean if self.mean_shift else [0.0, 0.0, 0.0]
        if self.normalize == 1:
            div, shift = 255.0, 0.0
        elif self.normalize == 2:
            div, shift = 127.5, (0.0 if self.mean_shift else -1.0)
        elif self.normalize == 3:
            div, shift = [255.0 * s for s in self.std], 0.0          # x / 255 / std (:71-73)
        else:
            div, shift = 1.0, 0.0
        if self.extra_affine is not None:
            if self.normalize != 0 or self.mean_shift:
                raise NotImplementedError("an input BatchNorm is only combined with the raw (normalize=0) mode")
            mean, div, shift = self.extra_affine
        return ops.preprocess(inputs, flip=not self.rgb, mean=mean, divisor=div, shift=shift,
                              out_channels=self.out_channels)

    def get_config(self):
        config = super().get_config()
        config.update({"rgb": self.rgb, "mean_shift": self.mean_shift, "normalize": self.normalize})
        return config


class BackboneModel(Layer):
    """What `load_backbone` returns: callable images -> [features], with Keras-Model-like
    `.output_names` (C-taps ascending, then P6, P7 -- reference :287-314)."""

    def __init__(self, backbone_type, backbone_outputs, num_features, **kwargs):
        super().__init__(name=backbone_type, **kwargs)
        from .mobilenet import MobileNetV1
        from .resnext import ResNeXt50
        from .resnext101 import ResNeXt101
        bt = backbone_type.lower()
        self.backbone_type = bt
        self.backbone_outputs = tuple(backbone_outputs)
        self.num_features = num_features
        if bt == 'resnext50':
            self.preprocess = BackBonePreProcess(rgb=True, mean_shift=True, normalize=2)      # :215-217
            self.body = ResNeXt50()
            same = True
        elif bt == 'resnext101':
            self.preprocess = BackBonePreProcess(rgb=True, mean_shift=False, normalize=0)
            self.body = ResNeXt101()
            same = True
        elif bt == 'mobilenet':
            self.preprocess = BackBonePreProcess(rgb=False, mean_shift=False, normalize=2)    # :254-256
            self.body = MobileNetV1()
            same = False
        else:
            raise NotImplementedError(
                f"backbone_type must be one of {list(BACKBONE_LAYERS.keys())} on the MI355X path "
                f"(got '{backbone_type}')")
        self.taps = [k for k in ("C1", "C2", "C3", "C4", "C5") if k in self.backbone_outputs]
        if not self.taps:
            raise ValueError("backbone_outputs must name at least one C-level")
        self.identities = {k: Identity(name=k) for k in self.taps}
        # extra levels (reference :292-314): mobilenet pads ((0,1),(0,1)) + valid, others 'same'
        pad = 'same' if same else ((0, 1), (0, 1))
        self.p6_conv = Conv2D(num_features, (3, 3), strides=(2, 2), padding=pad, activation='relu', name='P6_conv')
        self.p6_norm = GroupNormalization(name='P6_norm')          # default groups=32 (Appendix B.2)
        self.p7_conv = Conv2D(num_features, (3, 3), strides=(2, 2), padding=pad, activation='relu', name='P7_conv')
        self.output_names = list(self.taps)
        if 'P6' in self.backbone_outputs:
            self.output_names.append('P6')
        if 'P7' in self.backbone_outputs:
            self.output_names.append('P7')
        self.input_names = ['images']

    def build(self, input_shape=(None, None, None, 3)):
        s = self.preprocess.build(input_shape)
        tap_shapes = self.body.build(s)
        shapes = [tap_shapes[k] for k in self.taps]
        s6 = self.p6_conv.build(shapes[-1])
        self.p6_norm.build(s6)
        s7 = self.p7_conv.build(s6)
        if 'P6' in self.backbone_outputs:
            shapes.append(s6)
        if 'P7' in self.backbone_outputs:
            shapes.append(s7)
        self.built = True
        self.output_shapes = shapes
        return shapes

    def children(self):
        return [self.body, self.p6_conv, self.p6_norm, self.p7_conv]

    def weight_specs(self):
        out = {}
        for ch in self.children():
            out.update(ch.weight_specs())
        return out

    def load_weights(self, weights, device):
        super().load_weights(weights, device)
        if hasattr(self.body, "input_affine"):
            self.preprocess.extra_affine = self.body.input_affine(weights)

    def call(self, images, **kwargs):
        x = self.preprocess(images)
        taps = self.body(x, wanted=self.taps)
        feats = [self.identities[k](taps[k]) for k in self.taps]
        # The extra levels are three small launches (16- and 4-tile convs, a 64-block GroupNorm: ~0.14 ms during which
        # most of the chip idles).  A caller that has other work for the main stream (the FPN chain) passes
        # `fork_stream` and joins `self.pending_stream` before it consumes P6 / P7.
        fork = kwargs.get("fork_stream")
        aux = fork("_extra_level_stream") if fork is not None else None
        self.pending_stream = aux
        import contextlib
        import torch
        with (torch.cuda.stream(aux) if aux is not None else contextlib.nullcontext()):
            p6 = self.p6_conv(feats[-1])
            g6 = self.p6_norm(p6)                              # new tensor: P6 itself stays intact
            p7 = self.p7_conv(g6)
        if 'P6' in self.backbone_outputs:
            feats.append(p6)                                   # exported P6 is PRE-norm (:308-309)
        if 'P7' in self.backbone_outputs:
            feats.append(p7)
        return feats

    def join_extra_levels(self, feats):
        """Make the current stream wait for the P6 / P7 chain started by call(fork_stream=...)."""
        import torch
        aux = getattr(self, "pending_stream", None)
        if aux is None:
            return
        cur = torch.cuda.current_stream()
        cur.wait_stream(aux)
        for name, t in zip(self.output_names, feats):
            if name in ("P6", "P7"):
                t.record_stream(cur)
        self.pending_stream = None


def load_backbone(backbone_type="resnet50", backbone_outputs=('C3', 'C4', 'C5', 'P6', 'P7'), num_features=256):
    """Same signature as reference engine/backbone/base.py:185-187."""
    if backbone_type.lower() not in BACKBONE_LAYERS:
        raise NotImplementedError(
            f"backbone_type must be one of {list(BACKBONE_LAYERS.keys())} (got '{backbone_type}'); the other "
            f"reference backbones need un-vendored keras_applications / efficientnet weights and are outside "
            f"the accelerated hot path")
    model = BackboneModel(backbone_type, backbone_outputs, num_features)
    model.build((None, None, None, 3))
    return model
