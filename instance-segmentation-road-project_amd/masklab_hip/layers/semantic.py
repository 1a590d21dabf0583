"""Semantic-segmentation layers -- drop-ins for reference engine/layers/semantic.py:
AtrousSeparableConv2D :32-90, ASPPNetwork :93-168, SegmentationSubNet :178-246.
Every ReLU is fused into the GroupNormalization apply pass (or conv epilogue) and every
tf.concat is fused away: producers write their channel slice of the concat buffer directly."""
import torch

from .. import ops
from ..keras_like import Conv2D, DepthwiseConv2D, Layer
from ..normalization import GroupNormalization
from .detection import _TowerMixin


class AtrousSeparableConv2D(Layer):
    """depthwise 3x3 (dilated, 'same', no bias) -> GN -> ReLU -> 1x1 (no bias) -> GN -> ReLU
    (reference :32-90)."""

    def __init__(self, filters, dilation_rate=3, groups=16, **kwargs):
        prefix = kwargs.get('name', 'AtrousSeparableConv2d')
        super().__init__(**kwargs)
        self.groups = groups
        self.filters = filters
        self.dilation_rate = dilation_rate
        self.depth_conv2d = DepthwiseConv2D((3, 3), dilation_rate=dilation_rate, padding='same', use_bias=False,
                                            name=prefix + '_depthwise')
        self.point_conv2d = Conv2D(self.filters, (1, 1), use_bias=False, name=prefix + '_pointwise')
        self.depth_norm = GroupNormalization(groups=self.groups, name=prefix + '_depthwise_GN')
        self.point_norm = GroupNormalization(groups=self.groups, name=prefix + '_pointwise_GN')

    def build(self, input_shape):
        s = self.depth_conv2d.build(input_shape)
        s = self.depth_norm.build(s)
        s = self.point_conv2d.build(s)
        s = self.point_norm.build(s)
        self.built = True
        return s

    def children(self):
        return [self.depth_conv2d, self.depth_norm, self.point_conv2d, self.point_norm]

    def weight_specs(self):
        out = {}
        for ch in self.children():          # children carry the reference's explicit names
            out.update(ch.weight_specs())
        return out

    def call(self, inputs, out=None, out_coff=0, **kwargs):
        x = self.depth_conv2d(inputs)
        x = self.depth_norm(x, fuse_relu=True, inplace=True)
        x = self.point_conv2d(x)
        if out is None:
            return self.point_norm(x, fuse_relu=True, inplace=True)
        return ops.groupnorm_chunk(x, self.point_norm.gamma, self.point_norm.beta, self.groups,
                                   self.point_norm.epsilon, relu=True, out=out, out_coff=out_coff)

    def get_config(self):
        config = super().get_config()
        config.update({'filters': self.filters, 'dilation_rate': self.dilation_rate, 'groups': self.groups})
        return config


class ASPPNetwork(Layer):
    """Atrous Spatial Pyramid Pooling (reference :93-168): 1x1 branch, three atrous separable
    branches, image-pooling branch (ReLU, NO GroupNorm), concat (5*nf) -> 1x1 -> GN -> ReLU."""

    def __init__(self, num_features=256, atrous_rate=(6, 12, 18), groups=16, **kwargs):
        self.num_features = num_features
        self.atrous_rate = atrous_rate
        self.groups = groups
        super().__init__(**kwargs)
        self.aspp_1x1 = Conv2D(num_features, (1, 1), use_bias=False, name='aspp_1x1')
        self.aspp_1x1_gn = GroupNormalization(groups=groups, name='aspp_1x1_GN')
        self.aspp_branches = [AtrousSeparableConv2D(num_features, dilation_rate=rate, groups=groups,
                                                    name=f'aspp_{rate}') for rate in atrous_rate]
        self.aspp_pool_conv = Conv2D(num_features, (1, 1), activation='relu', use_bias=False, name='aspp_pool')
        self.concat_conv = Conv2D(num_features, (1, 1), use_bias=False, name='concat_projection')
        self.concat_gn = GroupNormalization(groups=groups, name='concat_projection_GN')

    def build(self, input_shape):
        s = self.aspp_1x1.build(input_shape)
        self.aspp_1x1_gn.build(s)
        for br in self.aspp_branches:
            br.build(input_shape)
        self.aspp_pool_conv.build((input_shape[0], 1, 1, input_shape[-1]))
        n_cat = self.num_features * (2 + len(self.aspp_branches))
        s = self.concat_conv.build(tuple(input_shape[:3]) + (n_cat,))
        self.built = True
        return self.concat_gn.build(s)

    def children(self):
        return [self.aspp_1x1, self.aspp_1x1_gn, *self.aspp_branches, self.aspp_pool_conv,
                self.concat_conv, self.concat_gn]

    def weight_specs(self):
        out = {}
        for ch in self.children():
            out.update(ch.weight_specs())
        return out

    def call(self, inputs, **kwargs):
        B, H, W, _ = inputs.shape
        nf = self.num_features
        cat = torch.empty((B, H, W, nf * (2 + len(self.aspp_branches))), dtype=inputs.dtype, device=inputs.device)
        x = self.aspp_1x1(inputs)
        ops.groupnorm_chunk(x, self.aspp_1x1_gn.gamma, self.aspp_1x1_gn.beta, self.groups,
                            self.aspp_1x1_gn.epsilon, relu=True, out=cat, out_coff=0)
        for i, br in enumerate(self.aspp_branches):
            br(inputs, out=cat, out_coff=nf * (1 + i))
        pool = ops.global_mean(inputs)                       # tf.reduce_mean(axis=(1,2)) (:149)
        pool = self.aspp_pool_conv(pool)                     # 1x1 + ReLU, no GN (:126-129)
        ops.resize_bilinear_ac(pool, H, W, out=cat, out_coff=nf * (1 + len(self.aspp_branches)))  # :152
        x = self.concat_conv(cat)
        return self.concat_gn(x, fuse_relu=True, inplace=True)

    def get_config(self):
        config = super().get_config()
        config.update({"num_features": self.num_features, "atrous_rate": self.atrous_rate, "groups": self.groups})
        return config


class SegmentationSubNet(Layer, _TowerMixin):
    """DeepLab V3+ decoder (reference :178-246): skip 1x1 -> GN -> ReLU ; upsample ASPP output
    (align_corners) ; concat ; depth x [Conv3x3+ReLU ; GN] ; Conv1x1 + sigmoid."""

    def __init__(self, num_depth=2, num_features=256, num_skip_features=48, num_classes=3,
                 use_separable_conv=False, expand_ratio=4., use_squeeze_excite=False, squeeze_ratio=16.,
                 groups=16, **kwargs):
        self.num_depth = num_depth
        self.num_features = num_features
        self.num_skip_features = num_skip_features
        self.num_classes = num_classes
        self.use_separable_conv = use_separable_conv
        self.expand_ratio = expand_ratio
        self.use_squeeze_excite = use_squeeze_excite
        self.squeeze_ratio = squeeze_ratio
        self.groups = groups
        super().__init__(**kwargs)
        self.skip_conv = Conv2D(self.num_skip_features, (1, 1), use_bias=False, name='skip_projection')
        self.skip_gn = GroupNormalization(groups=groups, name='skip_projection_GN')
        self.block = self._make_tower(self.name, num_depth, num_features, groups, use_separable_conv,
                                      expand_ratio, use_squeeze_excite, squeeze_ratio)
        self.output_layer = Conv2D(num_classes, (1, 1), activation='sigmoid', name=f'{self.name}/output')

    def build(self, input_shapes):
        dec_shape, skip_shape = input_shapes
        s = self.skip_conv.build(skip_shape)
        self.skip_gn.build(s)
        s = tuple(skip_shape[:3]) + (dec_shape[-1] + self.num_skip_features,)
        s = self._build_chain(self.block, s)
        self.built = True
        return self.output_layer.build(s)

    def children(self):
        return [self.skip_conv, self.skip_gn, *self.block, self.output_layer]

    def weight_specs(self):
        out = {}
        for ch in self.children():
            out.update(ch.weight_specs())
        return out

    def call(self, inputs, **kwargs):
        dec_input, skip_dec_input = inputs[0], inputs[1]
        B, H, W, _ = skip_dec_input.shape
        c_dec = dec_input.shape[-1]
        cat = torch.empty((B, H, W, c_dec + self.num_skip_features), dtype=dec_input.dtype,
                          device=dec_input.device)
        s = self.skip_conv(skip_dec_input)
        ops.groupnorm_chunk(s, self.skip_gn.gamma, self.skip_gn.beta, self.groups, self.skip_gn.epsilon,
                            relu=True, out=cat, out_coff=c_dec)
        ops.resize_bilinear_ac(dec_input, H, W, out=cat, out_coff=0)      # ResizeLike (:226) into the concat (:227)
        x = self._run_tower(self.block, cat)
        return self.output_layer(x)

    def get_config(self):
        config = super().get_config()
        config.update({
            "num_depth": self.num_depth, "num_features": self.num_features,
            "num_skip_features": self.num_skip_features, "num_classes": self.num_classes,
            "use_separable_conv": self.use_separable_conv, "expand_ratio": self.expand_ratio,
            "use_squeeze_excite": self.use_squeeze_excite, "squeeze_ratio": self.squeeze_ratio,
            "groups": self.groups})
        return config


class SemanticSmoothing(Layer):
    """Label-map post-processing (reference semantic.py:258-292): grey opening -- tf.nn.erosion2d then
    tf.nn.dilation2d with an all-zero kernel_size x kernel_size element, SAME -- times `weight`;
    kernel_size <= 0 only applies the weight.  One instance handles every channel of its input with the
    same (kernel_size, weight), like the reference's per-class tf.split; `smooth_classes` below runs the
    whole split -> smooth -> concat of retinamasklab.py:619-627 in one call."""

    def __init__(self, kernel_size=10, weight=1., **kwargs):
        self.kernel_size = kernel_size
        self.weight = weight
        super().__init__(**kwargs)

    def call(self, inputs, **kwargs):
        n_classes = int(inputs.shape[-1])
        return ops.semantic_smoothing(inputs, [self.kernel_size] * n_classes, [self.weight] * n_classes)

    @staticmethod
    def smooth_classes(seg_pred, kernel_sizes, weights):
        """tf.split(seg_pred, n) -> SemanticSmoothing(kernel, weight) per class -> tf.concat."""
        return ops.semantic_smoothing(seg_pred, list(kernel_sizes), list(weights))

    def get_config(self):
        config = super().get_config()
        config.update({"kernel_size": self.kernel_size, "weight": self.weight})
        return config
