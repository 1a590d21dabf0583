"""Misc layers of the hot path -- drop-ins for reference engine/layers/misc.py
(Identity :206-210, ResizeLike :296-319, SqueezeExcite :24-54, MoldBatch :213-293) and of the
deploy wrapper either side of it (DownSampleInput :143-154, UpSampleOutput :169-196) and the arithmetic
layers of the serving graph (CropAndPadMask :358-401, CrackToInstance :524-560, SummaryOutput :563-598,
IncludeMyRoad :601-626, CalculateInstanceSize :629-727)."""
import numpy as np
import torch

from .. import ops
from ..keras_like import Conv2D, Dense, DepthwiseConv2D, Layer
from ..normalization import GroupNormalization


class Identity(Layer):
    """Names a tensor (reference misc.py:206-210)."""

    def call(self, inputs, **kwargs):
        return inputs


class ReLU(Layer):
    """Placeholder for tf.keras.layers.ReLU: on this path every ReLU is fused into the producing
    kernel (conv epilogue or GroupNormalization apply pass); calling it standalone is unsupported."""

    def call(self, inputs, **kwargs):
        raise RuntimeError("ReLU is fused into the producer kernel on the MI355X path")


class ResizeLike(Layer):
    """Change the size of tensor(height & width) to target node (reference misc.py:296-319):
    tf.compat.v1.image.resize_bilinear(align_corners=True)."""

    def __init__(self, align_corners=True, **kwargs):
        super().__init__(**kwargs)
        if not align_corners:
            raise NotImplementedError("only align_corners=True is used on the hot path")
        self.align_corners = align_corners

    def call(self, inputs, **kwargs):
        target = kwargs.get('target')
        return ops.resize_bilinear_ac(inputs, int(target.shape[1]), int(target.shape[2]),
                                      add=kwargs.get('add'), out=kwargs.get('out'),
                                      out_coff=kwargs.get('out_coff', 0))

    def compute_output_shape(self, input_shapes):
        input_shape, target_shape = input_shapes
        return input_shape[0], target_shape[1], target_shape[2], input_shape[-1]

    def get_config(self):
        config = super().get_config()
        config.update({"align_corners": self.align_corners})
        return config


class SqueezeExcite(Layer):
    """SqueezeAndExcite (reference misc.py:24-54): GAP -> Dense(C//ratio, relu, no bias) ->
    Dense(C, sigmoid, no bias) -> channel scale.  The two Dense layers are 1x1 MFMA convs on the
    pooled [B,1,1,C] map; the scale is applied in place."""

    def __init__(self, ratio=16., **kwargs):
        super().__init__(**kwargs)
        self.ratio = ratio
        self.dense1 = self.dense2 = None

    def build(self, input_shape):
        n_channel = int(input_shape[-1])
        self.dense1 = Dense(int(n_channel // self.ratio), activation='relu', kernel_initializer='he_normal',
                            name=f"{self.name}/dense1")
        self.dense2 = Dense(n_channel, activation='sigmoid', kernel_initializer='glorot_normal',
                            name=f"{self.name}/dense2")
        s = self.dense1.build((input_shape[0], 1, 1, n_channel))
        self.dense2.build(s)
        self.built = True
        return input_shape

    def children(self):
        return [l for l in (self.dense1, self.dense2) if l is not None]

    def call(self, inputs, **kwargs):
        se = ops.global_mean(inputs)
        se = self.dense2(self.dense1(se))
        out = inputs.clone() if kwargs.get("keep_input", True) else inputs
        return ops.scale_channels_(out, se)

    def get_config(self):
        config = super().get_config()
        config.update({"ratio": self.ratio})
        return config


class MobileSeparableConv2D(Layer):
    """Separable Convolution Layer With Inverted Residual Block (reference misc.py:57-117):
    1x1 expand (no bias) -> GN -> ReLU -> depthwise 3x3 'same' (no bias) -> GN -> ReLU ->
    1x1 squeeze (no bias) -> GN -> inputs + x.  ReLUs are fused into the GroupNormalization apply pass."""

    def __init__(self, filters, kernel_size=(3, 3), expand_ratio=4., stride=1, groups=16, **kwargs):
        prefix = kwargs.get('name', 'SeparableConv2d')
        super().__init__(**kwargs)
        self.filters = filters
        self.kernel_size = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
        self.expand_ratio = expand_ratio
        self.stride = stride
        self.groups = groups
        self.expand_conv2d = Conv2D(int(self.expand_ratio * filters), (1, 1), use_bias=False,
                                    name=prefix + '_expand_conv')
        self.expand_norm = GroupNormalization(groups=self.groups, name=prefix + '_expand_GN')
        self.depth_conv2d = DepthwiseConv2D(self.kernel_size, (stride, stride), padding='same', use_bias=False,
                                            name=prefix + '_depthwise')
        self.depth_norm = GroupNormalization(groups=self.groups, name=prefix + '_depthwise_GN')
        self.squeeze_conv2d = Conv2D(filters, (1, 1), use_bias=False, name=prefix + '_squeeze_conv')
        self.squeeze_norm = GroupNormalization(groups=self.groups, name=prefix + '_squeeze_GN')

    def children(self):
        return [self.expand_conv2d, self.expand_norm, self.depth_conv2d, self.depth_norm,
                self.squeeze_conv2d, self.squeeze_norm]

    def weight_specs(self):
        out = {}
        for ch in self.children():
            out.update(ch.weight_specs())
        return out

    def build(self, input_shape):
        if int(input_shape[-1]) != self.filters or self.stride != 1:
            # the reference's Add() needs identical shapes (misc.py:105)
            raise ValueError(f"MobileSeparableConv2D: skip connection needs input channels == filters "
                             f"({input_shape[-1]} vs {self.filters}) and stride 1")
        s = self.expand_conv2d.build(input_shape)
        s = self.expand_norm.build(s)
        s = self.depth_conv2d.build(s)
        s = self.depth_norm.build(s)
        s = self.squeeze_conv2d.build(s)
        self.built = True
        return self.squeeze_norm.build(s)

    def call(self, inputs, **kwargs):
        x = self.expand_conv2d(inputs)
        x = self.expand_norm(x, fuse_relu=True, inplace=True)
        x = self.depth_conv2d(x)
        x = self.depth_norm(x, fuse_relu=True, inplace=True)
        x = self.squeeze_conv2d(x)
        x = self.squeeze_norm(x, inplace=True)
        return ops.add_(x, inputs)              # skip_connection (misc.py:105)

    def get_config(self):
        config = super().get_config()
        config.update({"filters": self.filters, "kernel_size": self.kernel_size, "expand_ratio": self.expand_ratio,
                       "stride": self.stride, "groups": self.groups})
        return config


class MoldBatch(Layer):
    """Group rows by image and pad with -1 (reference misc.py:213-293).  The fused kernels
    (detection proposal / RoI crop) emit molded tensors directly; this standalone layer covers
    the remaining call sites and is pure data movement (row scatter), done with torch indexing."""

    def __init__(self, max_batch_size=None, **kwargs):
        super().__init__(**kwargs)
        self.max_batch_size = max_batch_size

    def call(self, inputs, **kwargs):
        batch_indices = kwargs.get('batch_indices').to(torch.int64)
        batch_size = int(kwargs.get('batch_size'))
        if self.max_batch_size is not None and batch_size > 32:
            raise ValueError("MoldBatch partitions into 32 slots (reference misc.py:275): batch <= 32")
        counts = torch.bincount(batch_indices, minlength=batch_size)
        n = max(1, int(counts.max().item()) if counts.numel() else 1)
        out = torch.full((batch_size, n) + tuple(inputs.shape[1:]), -1.0, dtype=inputs.dtype, device=inputs.device)
        if inputs.shape[0]:
            start = torch.cumsum(counts, 0) - counts
            order = torch.argsort(batch_indices, stable=True)
            sorted_b = batch_indices[order]
            rank = torch.arange(inputs.shape[0], device=inputs.device) - start[sorted_b]
            out[sorted_b, rank] = inputs[order]
        return out

    def get_config(self):
        config = super().get_config()
        config.update({"max_batch_size": self.max_batch_size})
        return config


class DownSampleInput(Layer):
    """Resize the input to the model resolution, keeping the aspect ratio (reference misc.py:135-161):
    ratio = min(target_h / h, target_w / w) in float32, size = int32(ratio * h, ratio * w) (truncated),
    tf.compat.v1.image.resize_bilinear(align_corners=True) of the float32-cast image."""

    def __init__(self, target_size=(540, 960), **kwargs):
        self.target_size = target_size
        super().__init__(**kwargs)

    def output_size(self, input_h, input_w):
        f = np.float32
        ratio = min(f(self.target_size[0]) / f(input_h), f(self.target_size[1]) / f(input_w))
        return int(f(ratio) * f(input_h)), int(f(ratio) * f(input_w))      # tf.cast(float32 -> int32) truncates

    def call(self, inputs, **kwargs):
        oh, ow = self.output_size(int(inputs.shape[1]), int(inputs.shape[2]))
        return ops.resize_image_ac(inputs, oh, ow)

    def get_config(self):
        config = super().get_config()
        config.update({"target_size": self.target_size})
        return config


class UpSampleOutput(Layer):
    """Restore the outputs to the input resolution (reference misc.py:164-196).
    inputs = [roi_box, roi_mask, semantic_output], target= the original images.  Returns int32
    (roi_box, roi_mask, semantic_output): boxes scaled (cx, w by the HEIGHT ratio and cy, h by the
    width ratio -- the reference's own mix-up, misc.py:180-183 -- conf * 100), masks > 0.5, semantic map
    resized bilinear(align_corners=True) to the input size then > 0.5."""

    def call(self, inputs, **kwargs):
        target_node = kwargs.get('target')
        roi_box, roi_mask, semantic_output = inputs[0], inputs[1], inputs[2]
        f = np.float32
        src = (f(semantic_output.shape[1]), f(semantic_output.shape[2]))
        dst = (f(target_node.shape[1]), f(target_node.shape[2]))
        ratio_shape = (dst[0] / src[0], dst[1] / src[1])
        boxes = ops.upsample_boxes(roi_box, ratio_shape[0], ratio_shape[1])
        masks = ops.threshold_i32(roi_mask, 0.5)
        semantic = ops.resize_image_ac(semantic_output, int(target_node.shape[1]), int(target_node.shape[2]),
                                       threshold=0.5)
        return boxes, masks, semantic


# ----------------------------------------------------------------------------- serving post-processing
class CropAndPadMask(Layer):
    """Resize every kept instance mask to its box and pad it to the image (reference misc.py:358-401).
    inputs = [images, det_outs int32 [B,n,6], ins_outs int32 [B,n,h,w], ...] -> float32 [B,n,H,W]."""

    def call(self, inputs, **kwargs):
        images, det_outs, ins_outs = inputs[0], inputs[1], inputs[2]
        return ops.crop_pad_mask(det_outs.contiguous(), ins_outs.contiguous(), int(images.shape[1]), int(images.shape[2]))


class CrackToInstance(Layer):
    """Turn the crack channel of the semantic map into one pseudo instance (reference misc.py:524-560): the
    bounding box of ALL non-zero pixels of the batch, class id 5, conf = clip(100 * h * w, 0, 100)."""

    def __init__(self, crack_id=5, **kwargs):
        self.crack_id = crack_id
        super().__init__(**kwargs)

    def call(self, inputs, **kwargs):
        # the reference passes seg_outs[..., 2] ([B,H,W]); `channel=` reads that slice of [B,H,W,C] in place
        seg, channel = inputs, kwargs.get("channel")
        if seg.dim() == 3:
            seg, channel = seg[..., None].contiguous(), 0
        elif channel is None:
            raise ValueError("CrackToInstance: pass a [B,H,W] map, or a [B,H,W,C] map with channel=")
        box = ops.nonzero_bbox(seg.contiguous(), channel).cpu().tolist()          # one small host read
        ymin, xmin, ymax, xmax, any_ = box
        if not any_:
            ymin = xmin = ymax = xmax = 0                                       # tf.cond -> [[0, 0, 0]] (:534-536)
        height, width = ymax - ymin, xmax - xmin
        cy, cx = ymin + int(height / 2), xmin + int(width / 2)
        conf = min(max(100 * height * width, 0), 100)
        B = seg.shape[0]
        row = torch.tensor([cx, cy, width, height, 5, conf], dtype=torch.int32, device=seg.device)   # `* 5`, as written (:546)
        crack_det_outs = row[None, None, :].expand(B, 1, 6).contiguous()
        crack_seg_outs = seg[..., channel].to(torch.float32)[:, None].contiguous()                  # data movement
        return crack_det_outs, crack_seg_outs

    def get_config(self):
        config = super().get_config()
        config.update({'crack_id': self.crack_id})
        return config


class IncludeMyRoad(Layer):
    """Does an instance overlap the 'my road' class (reference misc.py:601-626)?  -> float32 [B,n] of 0 / 1."""

    def __init__(self, threshold=0.1, **kwargs):
        self.threshold = threshold
        super().__init__(**kwargs)

    def call(self, inputs, **kwargs):
        seg_outs, crop_ins_outs = inputs[0], inputs[1]
        return ops.instance_summary(seg_outs.contiguous(), crop_ins_outs.contiguous(), 1, 3.25, self.threshold)[..., 4]

    def get_config(self):
        config = super().get_config()
        config.update({'threshold': self.threshold})
        return config


class CalculateInstanceSize(Layer):
    """Metric size of every instance from the road-width regression (reference misc.py:629-727)
    -> float32 [B,n,3] = (instance_size, horizontal_size, vertical_size)."""

    def __init__(self, default_road_size=3.25, **kwargs):
        self.default_road_size = default_road_size
        super().__init__(**kwargs)

    def call(self, inputs, **kwargs):
        seg_outs, pad_ins_outs = inputs[0], inputs[1]
        s = ops.instance_summary(seg_outs.contiguous(), pad_ins_outs.contiguous(), 1, self.default_road_size, 0.1)
        return s[..., 1:4].contiguous()

    def get_config(self):
        config = super().get_config()
        config.update({"default_road_size": self.default_road_size})
        return config


class SummaryOutput(Layer):
    """The summary tensor of the serving model (reference misc.py:563-598): inputs = [det_outs int32 [B,n,6],
    seg_outs int32 [B,H,W,3], crop_ins_outs float32 [B,n,H,W]] -> float32 [B,n',11] =
    (class, cx, cy, w, h, conf, pixel count, instance size, horizontal size, vertical size, include_my_road);
    n' = n + 1 when the crack channel yields a pseudo instance."""

    def __init__(self, default_road_size=3.25, **kwargs):
        self.default_road_size = default_road_size
        super().__init__(**kwargs)

    def call(self, inputs, **kwargs):
        """inputs[2] = CropAndPadMask's float32 [B,n,H,W] output (the reference's wiring), or -- `from_rois=True` --
        that layer's own inputs' masks, int32 [B,n,mh,mw]: the canvases are then never materialised (3.4 GB at 8 x 100
        instances of 1024^2; ml_instance_summary_rois_f32 recomputes the pasted values, bit-identical numbers)."""
        det_outs, seg_outs, crop_ins_outs = inputs[0], inputs[1], inputs[2]
        crack_det_outs, crack_seg_outs = CrackToInstance()(seg_outs, channel=2)              # :575
        with_crack = bool((crack_det_outs[..., -1] > 0).all())                               # :577-583
        if kwargs.get("from_rois", False):
            s = ops.instance_summary_rois(seg_outs.contiguous(), det_outs.contiguous(), crop_ins_outs.contiguous(), 1,
                                          self.default_road_size, 0.1)
            if with_crack:                          # the crack pseudo-instance is one full-size map per image: as before
                s = torch.cat([s, ops.instance_summary(seg_outs.contiguous(), crack_seg_outs, 1, self.default_road_size,
                                                       0.1)], dim=1)
                det_outs = torch.cat([det_outs, crack_det_outs], dim=1)
        else:
            if with_crack:
                det_outs = torch.cat([det_outs, crack_det_outs], dim=1)
                crop_ins_outs = torch.cat([crop_ins_outs, crack_seg_outs], dim=1)
            s = ops.instance_summary(seg_outs.contiguous(), crop_ins_outs.contiguous(), 1, self.default_road_size, 0.1)
        d = det_outs.to(torch.float32)
        cx, cy, w, h, classes, conf = [d[..., i] for i in range(6)]
        return torch.stack([classes, cx, cy, w, h, conf, s[..., 0], s[..., 1], s[..., 2], s[..., 3], s[..., 4]], dim=-1)

    def get_config(self):
        config = super().get_config()
        config.update({'default_road_size': self.default_road_size})
        return config
