"""ResNeXt-50 (32x4d) body -- follows reference engine/backbone/ResNext.py:180-253 (block /
stack), :343-354 (stem) and :399-416 (stage plan), with the reference's Keras layer names so a
converted checkpoint loads by name.  Every BatchNormalization (eps 1.001e-5) is folded into the
preceding conv; ReLU / residual Add are conv-epilogue fusions; the grouped 3x3
(DepthwiseConv2D(depth_multiplier=c) + Split/Reduce/MergeGroups, :212-219) is one MFMA conv over
block-diagonal packed weights."""
from .. import ops
from ..keras_like import Conv2D, GroupedConv2D, Layer

BN_EPS = 1.001e-5


class _Block:
    def __init__(self, filters, stride, conv_shortcut, name, groups=32):
        # synthetic-init gamma ranges: the residual branch's last BN starts small so that random
        # weights keep activations O(1) through 16 residual blocks (real checkpoints override)
        rng = {0: (0.5, 1.0), 1: (0.5, 1.5), 2: (0.5, 1.5), 3: (0.1, 0.3)}
        bn = lambda s: (f"{name}_{s}_bn", BN_EPS, True, rng[s])
        he = dict(kernel_initializer="he_normal")
        self.shortcut = None
        if conv_shortcut:   # :199-203
            self.shortcut = Conv2D((64 // groups) * filters, 1, strides=stride, use_bias=False, fold_bn=bn(0),
                                   name=f"{name}_0_conv", **he)
        self.conv1 = Conv2D(filters, 1, use_bias=False, fold_bn=bn(1), activation='relu',
                            name=f"{name}_1_conv", **he)                                       # :207-210
        self.conv2 = GroupedConv2D(filters, groups, strides=stride, padding=((1, 1), (1, 1)), fold_bn=bn(2),
                                   activation='relu', name=f"{name}_2_conv")                   # :212-223
        self.conv3 = Conv2D((64 // groups) * filters, 1, use_bias=False, fold_bn=bn(3), activation='relu',
                            name=f"{name}_3_conv", **he)                                       # :225-231 (+add+relu)

    def layers(self):
        return [l for l in (self.shortcut, self.conv1, self.conv2, self.conv3) if l is not None]

    def build(self, shape):
        sc = self.shortcut.build(shape) if self.shortcut is not None else shape
        s = self.conv1.build(shape)
        s = self.conv2.build(s)
        s = self.conv3.build(s)
        assert tuple(s[1:]) == tuple(sc[1:]) or None in s, (s, sc)
        return s

    def __call__(self, x):
        sc = self.shortcut(x) if self.shortcut is not None else x
        y = self.conv1(x)
        y = self.conv2(y)
        return self.conv3(y, residual=sc)          # relu(shortcut + bn(conv)) fused in the epilogue


class ResNeXt50(Layer):
    STAGES = (("conv2", 128, 3, 1), ("conv3", 256, 4, 2), ("conv4", 512, 6, 2), ("conv5", 1024, 3, 2))  # :407-410

    def __init__(self, **kwargs):
        super().__init__(name=kwargs.pop("name", "resnext50_body"), **kwargs)
        # stem (:343-349): ZeroPadding2D(3) + Conv 7x7 s2 (no bias) + BN + ReLU on the NHWC4 image
        self.conv1 = Conv2D(64, 7, strides=2, padding=((3, 3), (3, 3)), use_bias=False,
                            fold_bn=("conv1_bn", BN_EPS, True), activation='relu', image_input=True,
                            kernel_initializer="he_normal", name="conv1_conv")
        self.stages = []
        for name, filters, blocks, stride1 in self.STAGES:                                     # :235-253
            stage = [_Block(filters, stride1, True, f"{name}_block1")]
            for i in range(2, blocks + 1):
                stage.append(_Block(filters, 1, False, f"{name}_block{i}"))
            self.stages.append(stage)

    def build(self, input_shape):
        s = self.conv1.build(input_shape)
        taps = {"C1": s}
        H, W = s[1], s[2]
        s = (s[0], None if H is None else (H + 2 - 3) // 2 + 1, None if W is None else (W + 2 - 3) // 2 + 1, s[3])
        for tap, stage in zip(("C2", "C3", "C4", "C5"), self.stages):
            for blk in stage:
                s = blk.build(s)
            taps[tap] = s
        self.built = True
        return taps

    def children(self):
        return [self.conv1] + [l for st in self.stages for blk in st for l in blk.layers()]

    def weight_specs(self):
        out = {}
        for ch in self.children():
            out.update(ch.weight_specs())
        return out

    def call(self, x, wanted=("C3", "C4", "C5"), **kwargs):
        import torch
        half = ops.half_storage()            # fp16-storage mode: the body's tensors AND its taps are IEEE half
        taps = {}
        if half and "C1" not in wanted and self.conv1.dev is not None:
            # fp16-storage mode and nobody asked for the un-pooled stem output: stem + pool in one pass (csrc/stem_h.hip)
            x = ops.stem_pool_h(x, self.conv1.dev)
        elif ops.CONV_MATH in ("f32", "f32x3") and "C1" not in wanted and self.conv1.dev is not None:
            x = ops.stem_pool(x, self.conv1.dev)               # fp32-tensor twins (csrc/stem_f32.hip, stem_x3.hip): same bits as the pair below
        else:
            x = self.conv1(x, out_dtype=torch.float16 if half else None)
            taps["C1"] = x
            x = ops.maxpool3x3s2(x, pad=1)                     # pool1_pad + pool1_pool (:351-352)
        last = max(int(t[1]) for t in wanted)
        for tap, stage in zip(("C2", "C3", "C4", "C5"), self.stages):
            for blk in stage:
                x = blk(x)
            taps[tap] = x
            if int(tap[1]) >= last:
                break
        return taps
