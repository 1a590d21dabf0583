"""Instance-segmentation layers -- drop-ins for reference engine/layers/instance.py:
MaskDistribute :32-74, PyramidRoiAlign :77-147, MaskSubNet :158-240."""
import torch

from .. import ops
from ..keras_like import Conv2D, Conv2DTranspose, Layer
from .detection import _TowerMixin


class MaskDistribute(Layer):
    """Pick the pyramid level of every box from its size (reference :32-74):
    k = clip(floor(log2((sqrt(w*h)+eps)/(base+eps))), 0, max_k); k = -1 for padded rows."""

    def __init__(self, max_k=2, base_size=64, **kwargs):
        self.max_k = max_k
        self.base_size = base_size
        super().__init__(**kwargs)

    def call(self, inputs, **kwargs):
        inputs = inputs.contiguous()
        _, _, _, kvals = ops.mask_distribute(inputs, self.max_k, self.base_size, has_k=False, want_k=True)
        # Concatenate([k, boxes], axis=-1) (:66): data movement of a [B,N,7] table, off the fused hot path
        # (InferenceModel consumes the level slots directly and never builds dist_boxes)
        return torch.cat([kvals[..., None], inputs], dim=-1)

    def get_config(self):
        config = super().get_config()
        config.update({"max_k": self.max_k, "base_size": self.base_size})
        return config


class PyramidRoiAlign(Layer):
    """Per level: crop_and_resize the boxes assigned to it and mold by image with -1 padding
    (reference :77-147).  NOTE: one bilinear sample per output cell (tf.image.crop_and_resize),
    boxes normalised by the IMAGE (H, W)."""

    def __init__(self, crop_size=(14, 14), max_batch_size=64, **kwargs):
        self.crop_size = crop_size
        self.max_batch_size = max_batch_size
        super().__init__(**kwargs)

    def distribute(self, n_levels, rows, has_k, base_size=1.0):
        """Enqueue the level assignment + per-level slot lists (no host read)."""
        if rows.shape[0] > 32 and self.max_batch_size is not None:
            raise ValueError("PyramidRoiAlign: MoldBatch supports at most 32 images per call")
        slots, lcounts, lmax, _ = ops.mask_distribute(rows, n_levels - 1, base_size, has_k=has_k)
        return slots, lcounts, lmax

    def crop_distributed(self, fmap_outputs, rows, image_hw, slots, lcounts, lmax):
        """One device->host read (`lmax`: the per-level maxima the distribute kernel wrote, L ints) sizes the
        molded outputs like the reference's dynamic shapes; callers enqueue independent work (the semantic
        head) before calling this."""
        B = rows.shape[0]
        n_l = [max(1, int(v)) for v in lmax.tolist()]     # the single sync
        total = sum(n_l)
        roi_boxes = torch.empty((B, total, 6), dtype=torch.float32, device=rows.device)
        roi_fmaps, off = [], 0
        for level, fmap in enumerate(fmap_outputs):
            roi_fmaps.append(ops.roi_crop_resize(fmap, rows, slots, lcounts, level, n_l[level],
                                                 tuple(self.crop_size), image_hw, roi_boxes, off))
            off += n_l[level]
        return roi_fmaps, roi_boxes

    def crop_capacity(self, fmap_outputs, rows, image_hw, slots, lcounts, lmax):
        """The same crops with NO host read: every level is launched at capacity (n_l = cap RoI slots per image) and the
        kernels skip the slots past the level's maximum, which they read from `lmax` on the device.
        -> (roi_fmaps per level [B,cap,ch,cw,C], roi_boxes_cap [B,L*cap,6], lives = [(lmax[l:l+1], cap)])."""
        B, cap = rows.shape[0], rows.shape[1]
        L = len(fmap_outputs)
        roi_boxes = torch.empty((B, L * cap, 6), dtype=torch.float32, device=rows.device)
        roi_fmaps, lives = [], []
        for level, fmap in enumerate(fmap_outputs):
            live = lmax[level:level + 1]
            roi_fmaps.append(ops.roi_crop_resize(fmap, rows, slots, lcounts, level, cap, tuple(self.crop_size), image_hw,
                                                 roi_boxes, level * cap, live=live))
            lives.append((live, cap))
        return roi_fmaps, roi_boxes, lives

    def crop_levels(self, fmap_outputs, rows, image_hw, has_k, base_size=1.0):
        """rows: [B,cap,6] proposals (has_k=False) or [B,cap,7] dist_boxes (has_k=True)."""
        slots, lcounts, lmax = self.distribute(len(fmap_outputs), rows, has_k, base_size)
        return self.crop_distributed(fmap_outputs, rows, image_hw, slots, lcounts, lmax)

    def call(self, inputs, **kwargs):
        fmap_outputs, dist_boxes, images = inputs[0], inputs[1], inputs[2]
        image_hw = (int(images.shape[1]), int(images.shape[2]))
        roi_fmaps, roi_boxes = self.crop_levels(fmap_outputs, dist_boxes.contiguous(), image_hw, has_k=True)
        return [roi_fmaps, roi_boxes]

    def get_config(self):
        config = super().get_config()
        config.update({"crop_size": self.crop_size, "max_batch_size": self.max_batch_size})
        return config


class MaskSubNet(Layer, _TowerMixin):
    """Mask Prediction Module In RetinaMask (reference :158-240): per level (own weights)
    depth x [Conv3x3+ReLU ; GN] ; Conv2DTranspose 2x2 s2 + ReLU ; Conv1x1 + sigmoid."""

    def __init__(self, num_blocks, num_classes, num_depth=4, num_features=256, use_separable_conv=False,
                 expand_ratio=4., use_squeeze_excite=False, squeeze_ratio=16., groups=16, **kwargs):
        self.num_blocks = num_blocks
        self.num_classes = num_classes
        self.num_depth = num_depth
        self.num_features = num_features
        self.use_separable_conv = use_separable_conv
        self.expand_ratio = expand_ratio
        self.use_squeeze_excite = use_squeeze_excite
        self.squeeze_ratio = squeeze_ratio
        self.groups = groups
        super().__init__(**kwargs)
        self.blocks = []
        for idx in range(self.num_blocks):
            prefix = f'{self.name}/block{idx}'
            block = self._make_tower(prefix, num_depth, num_features, groups, use_separable_conv, expand_ratio,
                                     use_squeeze_excite, squeeze_ratio)
            block.append(Conv2DTranspose(num_features, (2, 2), (2, 2), padding='same', activation='relu',
                                         kernel_stddev=0.01, name=f'{prefix}/deconv'))
            block.append(Conv2D(num_classes, (1, 1), padding='same', activation='sigmoid',
                                kernel_initializer='normal', kernel_stddev=0.01, name=f'{prefix}/output'))
            self.blocks.append(block)

    def build(self, input_shapes):
        if not isinstance(input_shapes, list):
            input_shapes = [input_shapes]
        out = None
        for block, shape in zip(self.blocks, input_shapes):
            s = (None,) + tuple(shape[2:]) if len(shape) == 5 else tuple(shape)
            out = self._build_chain(block, s)
        self.built = True
        return (input_shapes[0][0], None) + tuple(out[1:])

    def children(self):
        return [l for b in self.blocks for l in b]

    def load_weights(self, weights, device):
        super().load_weights(weights, device)
        # the 1x1 output kernels once more, in the lane order of the fused tail kernel (csrc/deconv_out.hip)
        from .. import packing
        self._tail_tables = []
        for block in self.blocks:
            deconv, out = block[-2], block[-1]
            p = deconv.dev.p
            ok = (p.span == p.span_pad and p.n_pad == p.cout and p.cout // 4 in (128, 256) and self.num_classes <= 32
                  and out.kernel_size == (1, 1) and out.strides == (1, 1))
            if not ok:
                self._tail_tables = None
                break
            k, b = out.folded(weights)
            table, bo, _ = packing.pack_out1x1_table(k, b)
            self._tail_tables.append((torch.from_numpy(table).to(device), torch.from_numpy(bo).to(device)))

    def _fused_tail(self, blocks, xs, shapes, roi_masks, per_roi, lives=None):
        """Conv2DTranspose + ReLU -> Conv2D 1x1 + sigmoid of every level in one launch, written straight into
        `roi_masks` (csrc/deconv_out.hip); False when the configuration is outside that kernel's shapes."""
        from .. import _lib
        if getattr(self, "_tail_tables", None) is None or ops.CONV_MATH == "f16":
            return False
        if any(x.shape[0] * x.shape[1] * x.shape[2] >= 1 << 24 for x in xs):
            return False
        if xs[0].dtype == torch.float16 and any(x.shape[-1] % 64 for x in xs):
            return False
        problems, off = [], 0
        for i, ((_, n), b, x) in enumerate(zip(shapes, blocks, xs)):
            table, bo = self._tail_tables[i]
            problems.append(dict(x=x.contiguous(), dc=b[-2].dev, wo_table=table, bo=bo, out=roi_masks,
                                 out_base=off * per_roi, rois_per_image=n,
                                 live=None if lives is None else lives[i][0]))
            off += n
        ops.deconv2x2_out1x1_multi(problems, self.num_classes, _lib.ACT_BY_NAME[blocks[0][-2].activation],
                                   _lib.ACT_BY_NAME[blocks[0][-1].activation])
        return True

    def capacity_supported(self, crop_size):
        """True when the whole head runs in the fixed-capacity form (plain towers + the fused tail kernel)."""
        plain = all(len(b) >= 2 and all(type(l).__name__ in ("Conv2D", "GroupNormalization") for l in b[:-2]) for b in self.blocks)
        return plain and getattr(self, "_tail_tables", None) is not None and crop_size[0] * crop_size[1] >= 2

    def call(self, inputs, lives=None, **kwargs):
        """lives (fixed-capacity form, PyramidRoiAlign.crop_capacity): per level (device int32 [1], cap) -- RoI slots past
        max(1, live) of an image do not exist; the result is then [B, L*cap, 2h, 2w, classes] with level l's RoIs at
        rows l*cap .. (molded by ops.mold_levels once the host knows the maxima)."""
        if not isinstance(inputs, list):
            inputs = [inputs]
        from .. import _lib
        blocks = self.blocks[:len(inputs)]
        shapes = [(h.shape[0], h.shape[1]) for h in inputs]
        # fold rois into the batch (:211-213); all levels advance together (multi-problem launches)
        xs = [h.reshape((h.shape[0] * h.shape[1],) + tuple(h.shape[2:])) for h in inputs]
        xs = self._run_towers_multi([b[:-2] for b in blocks], xs, lives=lives)
        if all(n > 0 for _, n in shapes):
            B = shapes[0][0]
            oh, ow, ncls = 2 * int(xs[0].shape[1]), 2 * int(xs[0].shape[2]), self.num_classes
            total = sum(n for _, n in shapes)
            roi_masks = torch.empty((B, total, oh, ow, ncls), dtype=torch.float32, device=xs[0].device)
            if self._fused_tail(blocks, xs, shapes, roi_masks, oh * ow * ncls, lives=lives):
                return roi_masks
        if lives is not None:
            raise NotImplementedError("MaskSubNet: the fixed-capacity form needs the fused tail kernel")
        if xs[0].dtype == torch.float16:          # (the pixel-shuffle epilogue of the unfused pair is fp32 only)
            xs = [ops.cast_h2f(x) for x in xs]
        xs = ops.conv2d_multi([dict(x=x, dc=b[-2].dev, act=_lib.ACT_BY_NAME[b[-2].activation])
                               for b, x in zip(blocks, xs)])                    # Conv2DTranspose + ReLU
        # unfold + Concatenate(axis=1) (:222-225) fused into the output convs: level l's [B*n_l, h, w, classes] maps
        # land at rows [off_l, off_l + n_l) of every image of roi_masks through a per-image strided view.  The
        # view's "image" is one RoI-level block: problem (b, level) would need B problems per level, so instead the
        # conv's batch axis is the IMAGE (n_l RoIs of one image are n_l*h rows of a [B, n_l*h, w, classes] map).
        B = shapes[0][0]
        oh, ow = int(xs[0].shape[1]), int(xs[0].shape[2])
        ncls = self.num_classes
        total = sum(n for _, n in shapes)
        roi_masks = torch.empty((B, total, oh, ow, ncls), dtype=torch.float32, device=xs[0].device)
        per_roi = oh * ow * ncls
        problems, off = [], 0
        for (_, n), b, x in zip(shapes, blocks, xs):
            xv = x.reshape(B, n * oh, ow, x.shape[3])          # a view: RoIs of one image stacked along H (1x1 conv)
            problems.append(dict(x=xv, dc=b[-1].dev, stride=1, padding=b[-1].padding,
                                 act=_lib.ACT_BY_NAME[b[-1].activation],
                                 out_view=(roi_masks, off * per_roi, ncls, total * per_roi)))
            off += n
        ops.conv2d_multi(problems)
        return roi_masks

    def get_config(self):
        config = super().get_config()
        config.update({
            "num_blocks": self.num_blocks, "num_classes": self.num_classes, "num_depth": self.num_depth,
            "num_features": self.num_features, "use_separable_conv": self.use_separable_conv,
            "expand_ratio": self.expand_ratio, "use_squeeze_excite": self.use_squeeze_excite,
            "squeeze_ratio": self.squeeze_ratio, "groups": self.groups})
        return config


class TrimInstances(Layer):
    """Drop the -1 padded RoIs and keep each RoI's own class channel (reference instance.py:258-286):
    rows with class != -1, in tf.where (row-major) order; mask = roi_masks[b, r, :, :, class];
    mold=True re-groups per image with -1 padding (MoldBatch) -> ([B,n,6], [B,n,h,w]), n = the
    largest per-image count; mold=False returns the flat rows ([R,6], [R,h,w])."""

    def __init__(self, mold=True, max_batch_size=64, **kwargs):
        self.mold = mold
        self.max_batch_size = max_batch_size
        super().__init__(**kwargs)
        self.last_counts = None

    def call(self, inputs, **kwargs):
        roi_boxes, roi_masks = inputs[0], inputs[1]
        boxes, masks, counts = ops.trim_instances(roi_boxes, roi_masks)
        self.last_counts = counts
        host_counts = counts.cpu()                       # the dynamic output shape needs the count on the host
        if self.mold:
            n = max(1, int(host_counts.max())) if host_counts.numel() else 1
            return boxes[:, :n].contiguous(), masks[:, :n].contiguous()
        keep = torch.arange(boxes.shape[1], device=boxes.device)[None, :] < counts[:, None]
        return boxes[keep], masks[keep]

    def get_config(self):
        config = super().get_config()
        config.update({"mold": self.mold, "max_batch_size": self.max_batch_size})
        return config
