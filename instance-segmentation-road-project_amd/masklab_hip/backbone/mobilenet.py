"""MobileNet v1 (alpha=1.0, include_top=False) body.  The reference does not vendor it: it calls
tf.keras.applications.MobileNet (engine/backbone/base.py:253-258) and taps conv_pw_{1,3,5,11,13}_relu
(:161-167).  The architecture is restated from keras-applications (layer names kept):
conv1 = pad ((0,1),(0,1)) + 3x3 s2 'valid' (no bias) + BN(eps 1e-3) + ReLU6, then 13 blocks of
[depthwise 3x3 (stride-2 blocks: same one-sided pad + 'valid') + BN + ReLU6 + 1x1 + BN + ReLU6]."""
from ..keras_like import Conv2D, DepthwiseConv2D, Layer

BN_EPS = 1e-3
BLOCKS = [(64, 1), (128, 2), (128, 1), (256, 2), (256, 1), (512, 2)] + [(512, 1)] * 5 + [(1024, 2), (1024, 1)]
TAP_OF_BLOCK = {1: "C1", 3: "C2", 5: "C3", 11: "C4", 13: "C5"}


class MobileNetV1(Layer):
    def __init__(self, **kwargs):
        super().__init__(name=kwargs.pop("name", "mobilenet_body"), **kwargs)
        self.conv1 = Conv2D(32, 3, strides=2, padding=((0, 1), (0, 1)), use_bias=False,
                            fold_bn=("conv1_bn", BN_EPS, True), activation='relu6', image_input=True,
                            kernel_initializer="he_normal", name="conv1")
        self.blocks = []
        for i, (filters, stride) in enumerate(BLOCKS, start=1):
            pad = 'same' if stride == 1 else ((0, 1), (0, 1))
            dw = DepthwiseConv2D((3, 3), strides=stride, padding=pad, use_bias=False,
                                 fold_bn=(f"conv_dw_{i}_bn", BN_EPS, True), activation='relu6',
                                 name=f"conv_dw_{i}")
            pw = Conv2D(filters, 1, padding='same', use_bias=False, fold_bn=(f"conv_pw_{i}_bn", BN_EPS, True),
                        activation='relu6', kernel_initializer="he_normal", name=f"conv_pw_{i}")
            self.blocks.append((dw, pw))

    def build(self, input_shape):
        s = self.conv1.build(input_shape)
        taps = {}
        for i, (dw, pw) in enumerate(self.blocks, start=1):
            s = pw.build(dw.build(s))
            if i in TAP_OF_BLOCK:
                taps[TAP_OF_BLOCK[i]] = s
        self.built = True
        return taps

    def children(self):
        return [self.conv1] + [l for b in self.blocks for l in b]

    def weight_specs(self):
        out = {}
        for ch in self.children():
            out.update(ch.weight_specs())
        return out

    def call(self, x, wanted=("C3", "C4", "C5"), **kwargs):
        taps = {}
        x = self.conv1(x)
        for i, (dw, pw) in enumerate(self.blocks, start=1):
            x = pw(dw(x))
            if i in TAP_OF_BLOCK:
                taps[TAP_OF_BLOCK[i]] = x
        return taps
