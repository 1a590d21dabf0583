"""Detection layers -- drop-ins for reference engine/layers/detection.py:
FeaturePyramid :30-74, BoxRegressionSubNet :89-155, ClassificationSubNet :158-228,
PriorLayer :236-306, RestoreBoxes :309-344, NormalizeBoxes :347-375, DetectionProposal :435-578.
Same class names / constructor kwargs / get_config(); `call` runs the HIP kernels."""
import numpy as np
import torch

from .. import ops
from ..keras_like import Conv2D, Layer
from ..normalization import GroupNormalization
from ..prior import PriorBoxes
from .misc import MobileSeparableConv2D, SqueezeExcite


class FeaturePyramid(Layer):
    """Build Feature Pyramid Network (reference :30-74).  Top-down pass: lateral 1x1, bilinear
    (align_corners) upsample of the previous PRE-3x3 sum fused with the add, then the 3x3 'P{k}'."""

    def __init__(self, strides, num_features=256, **kwargs):
        self.strides = strides
        self.num_features = num_features
        super().__init__(**kwargs)
        self.blocks = []
        for layer_stride in sorted(self.strides, reverse=True):
            p_num = int(np.round(np.log2(layer_stride)))
            lateral = Conv2D(num_features, (1, 1), padding='same', name=f'{self.name}/C{p_num}_lateral')
            out = Conv2D(num_features, (3, 3), padding='same', name=f'{self.name}/P{p_num}')
            self.blocks.append([lateral, out])

    def build(self, input_shapes):
        outs = []
        for block, shape in zip(self.blocks, list(input_shapes)[::-1]):
            s = block[0].build(shape)
            outs.append(block[1].build(s))
        self.built = True
        return outs[::-1]

    def children(self):
        return [l for b in self.blocks for l in b]

    def call(self, inputs, **kwargs):
        # The top-down chain (lateral -> resize + add -> next lateral ...) is serial, but the 3x3 output conv of a
        # coarse level only reads its own merged map and is a 64- / 256-tile launch: with `fork_stream` those run on an
        # auxiliary stream beside the chain; the finest level's conv (1024 tiles) closes the chain on the main stream.
        fork = kwargs.get("fork_stream")
        aux = fork("_fpn_stream") if fork is not None and len(inputs) > 1 else None
        main = torch.cuda.current_stream() if aux is not None else None
        prev = None
        pyramid_outputs = []
        last = len(inputs) - 1
        for idx, head in enumerate(inputs[::-1]):
            block = self.blocks[idx]
            lateral = block[0](head)
            if prev is not None:
                # ResizeLike(prev -> lateral size) + Add, in place into `lateral` (reference :58-60)
                ops.resize_bilinear_ac(prev, lateral.shape[1], lateral.shape[2], add=lateral, out=lateral)
            prev = lateral
            if aux is not None and idx != last:
                aux.wait_stream(main)
                lateral.record_stream(aux)
                with torch.cuda.stream(aux):
                    pyramid_outputs.append(block[1](lateral))
            else:
                pyramid_outputs.append(block[1](lateral))
        if aux is not None:
            main.wait_stream(aux)
            for t in pyramid_outputs[:-1]:
                t.record_stream(main)
        return pyramid_outputs[::-1]

    def get_config(self):
        config = super().get_config()
        config.update({"strides": self.strides, "num_features": self.num_features})
        return config


class _TowerMixin:
    """depth x [optional SqueezeExcite ; Conv3x3+ReLU ; GroupNormalization] shared by every head."""

    def _make_tower(self, prefix, num_depth, num_features, groups, use_separable_conv, expand_ratio,
                    use_squeeze_excite, squeeze_ratio):
        block = []
        for i in range(num_depth):
            if use_squeeze_excite:
                block.append(SqueezeExcite(squeeze_ratio, name=f'{prefix}/se{i}'))
            if use_separable_conv:
                block.append(MobileSeparableConv2D(num_features, (3, 3), expand_ratio=expand_ratio,
                                                   name=f'{prefix}/sep{i}'))
            else:
                block.append(Conv2D(num_features, (3, 3), activation='relu', padding='same',
                                    kernel_initializer='normal', kernel_stddev=0.01, name=f'{prefix}/conv{i}'))
            block.append(GroupNormalization(groups, name=f'{prefix}/gn{i}'))
        return block

    @staticmethod
    def _conv_tiles(c, x):
        """128 x 128 tiles of conv `c` on input `x` (the library's split-K / narrow-tile decisions look at this count)."""
        from ..packing import resolve_padding
        Ho, Wo, _, _ = resolve_padding(int(x.shape[1]), int(x.shape[2]), c.kernel_size[0], c.kernel_size[1], c.strides[0],
                                       c.dilation_rate[0], c.padding)
        return -(-int(x.shape[0]) * Ho * Wo // 128) * (c.dev.p.n_pad // 128 if c.dev.p.n_pad >= 128 else 1), (int(x.shape[0]), Ho, Wo)

    @staticmethod
    def _run_tower(block, x):
        block_input = x                          # never modified in place (it is a caller's tensor)
        i = 0
        while i < len(block):
            layer = block[i]
            nxt = block[i + 1] if i + 1 < len(block) else None
            if isinstance(layer, Conv2D) and isinstance(nxt, GroupNormalization) and layer.dev is not None:
                # conv -> (ReLU) -> GroupNormalization: the conv's epilogue sums its tiles, no statistics pass (fp32, big maps)
                tiles, oshape = _TowerMixin._conv_tiles(layer, x)
                tpc = ops.gn_fusable(oshape, layer.filters, nxt.groups, layer.dev, tiles, x.dtype)
                if tpc:
                    part = torch.empty((tiles, 4, 2), dtype=torch.float64, device=x.device)
                    x = layer(x, gn_partials=part)
                    x = GroupNormalization.call_multi([nxt], [x], inplace=True, partials=[(part, tpc)])[0]
                    i += 2
                    continue
            i += 1
            if isinstance(layer, GroupNormalization):
                x = layer(x, inplace=True)      # conv output is a fresh tensor: normalise in place
            elif isinstance(layer, SqueezeExcite):
                x = layer(x, keep_input=(x is block_input))
            else:
                x = layer(x)
        return x

    @staticmethod
    def _run_towers_multi(blocks, xs, lives=None):
        """Depth-major execution of several un-shared towers (one per pyramid level / RoI level):
        the conv of depth i runs for ALL towers in one multi-problem launch, so the few-tile
        problems of the coarse levels ride along with the fine level instead of each being a
        latency-bound launch of its own.  Falls back to tower-major order when a tower holds
        anything but [Conv2D, GroupNormalization] pairs (SqueezeExcite)."""
        plain = all(len(b) % 2 == 0 and all(isinstance(b[2 * i], Conv2D) and
                                              isinstance(b[2 * i + 1], GroupNormalization)
                                              for i in range(len(b) // 2)) for b in blocks)
        if not plain or len({len(b) for b in blocks}) != 1:
            if lives is not None:
                raise NotImplementedError("fixed-capacity RoI batches: plain [Conv2D, GroupNormalization] towers only")
            return [_TowerMixin._run_tower(b, x) for b, x in zip(blocks, xs)]
        xs = list(xs)
        lives = lives if lives is not None else [None] * len(xs)
        no_lives = all(lv is None for lv in lives)
        for i in range(len(blocks[0]) // 2):
            convs = [b[2 * i] for b in blocks]
            norms = [b[2 * i + 1] for b in blocks]
            # levels whose GroupNorm chunks are whole conv tiles take their statistics from the conv's epilogue
            parts = [None] * len(xs)
            if no_lives:
                geo = [_TowerMixin._conv_tiles(c, x) for c, x in zip(convs, xs)]
                launch_tiles = sum(t for t, _ in geo)
                for k, (c, g, x, (t, oshape)) in enumerate(zip(convs, norms, xs, geo)):
                    tpc = ops.gn_fusable(oshape, c.filters, g.groups, c.dev, launch_tiles, x.dtype)
                    if tpc:
                        parts[k] = (torch.empty((t, 4, 2), dtype=torch.float64, device=x.device), tpc)
            xs = ops.conv2d_multi([dict(x=x, dc=c.dev, stride=c.strides[0], padding=c.padding,
                                        dilation=c.dilation_rate[0], act=ops._lib.ACT_BY_NAME[c.activation], live=lv,
                                        gn_partials=None if pt is None else pt[0])
                                   for c, x, lv, pt in zip(convs, xs, lives, parts)])
            xs = GroupNormalization.call_multi(norms, xs, inplace=True, lives=None if no_lives else lives,
                                               partials=None if all(pt is None for pt in parts) else parts)
        return xs

    @staticmethod
    def _build_chain(block, shape):
        for layer in block:
            shape = layer.build(shape)
        return shape


class BoxRegressionSubNet(Layer, _TowerMixin):
    """Box Regression Module in RetinaMask (reference :89-155); one un-shared block per level."""

    def __init__(self, num_blocks, num_depth=4, num_features=256, num_priors=9, use_separable_conv=False,
                 expand_ratio=4., use_squeeze_excite=False, squeeze_ratio=16., groups=16, **kwargs):
        self.num_blocks = num_blocks
        self.num_depth = num_depth
        self.num_features = num_features
        self.num_priors = num_priors
        self.use_separable_conv = use_separable_conv
        self.expand_ratio = expand_ratio
        self.use_squeeze_excite = use_squeeze_excite
        self.squeeze_ratio = squeeze_ratio
        self.groups = groups
        super().__init__(**kwargs)
        self.out_dim = 4
        self.blocks = []
        for idx in range(self.num_blocks):
            prefix = f'{self.name}/block{idx}'
            block = self._make_tower(prefix, num_depth, num_features, groups, use_separable_conv, expand_ratio,
                                     use_squeeze_excite, squeeze_ratio)
            block.append(self._output_conv(prefix))
            self.blocks.append(block)

    def _output_conv(self, prefix):
        return Conv2D(self.num_priors * 4, (3, 3), padding='same', kernel_initializer='normal',
                      kernel_stddev=0.01, name=f'{prefix}/output')

    def build(self, input_shapes):
        for block, shape in zip(self.blocks, input_shapes):
            self._build_chain(block, shape)
        self.built = True
        return (input_shapes[0][0], None, self.out_dim)

    def children(self):
        return [l for b in self.blocks for l in b]

    def call(self, inputs, **kwargs):
        # Reshape((-1, d)) + Concatenate(axis=1) (reference :138-140) are fused: every level's
        # output conv writes straight into its row range of the [B, A, d] prediction.
        d = self.out_dim
        B = inputs[0].shape[0]
        per_level = [int(x.shape[1]) * int(x.shape[2]) * self.num_priors for x in inputs]
        total = sum(per_level)
        pred = torch.empty((B, total, d), dtype=torch.float32, device=inputs[0].device)
        blocks = self.blocks[:len(inputs)]
        xs = self._run_towers_multi([b[:-1] for b in blocks], inputs)
        problems, off = [], 0
        for idx, x in enumerate(xs):
            out_conv = blocks[idx][-1]
            problems.append(dict(x=x, dc=out_conv.dev, stride=1, padding='same',
                                 act=ops._lib.ACT_BY_NAME[out_conv.activation],
                                 out_view=(pred, off * d, self.num_priors * d, total * d)))
            off += per_level[idx]
        ops.conv2d_multi(problems)
        return pred

    def get_config(self):
        config = super().get_config()
        config.update({
            "num_blocks": self.num_blocks, "num_depth": self.num_depth, "num_features": self.num_features,
            "num_priors": self.num_priors, "use_separable_conv": self.use_separable_conv,
            "expand_ratio": self.expand_ratio, "use_squeeze_excite": self.use_squeeze_excite,
            "squeeze_ratio": self.squeeze_ratio, 'groups': self.groups})
        return config


class ClassificationSubNet(BoxRegressionSubNet):
    """Classifcation Module in RetinaMask (reference :158-228): sigmoid outputs, bias -log(99)."""

    def __init__(self, num_blocks, num_classes, num_depth=4, num_features=256, num_priors=9,
                 use_separable_conv=False, expand_ratio=4., use_squeeze_excite=False, squeeze_ratio=16.,
                 groups=16, **kwargs):
        self.num_classes = num_classes
        super().__init__(num_blocks, num_depth=num_depth, num_features=num_features, num_priors=num_priors,
                         use_separable_conv=use_separable_conv, expand_ratio=expand_ratio,
                         use_squeeze_excite=use_squeeze_excite, squeeze_ratio=squeeze_ratio, groups=groups, **kwargs)
        self.out_dim = num_classes

    def _output_conv(self, prefix):
        return Conv2D(self.num_priors * self.num_classes, (3, 3), padding='same', activation='sigmoid',
                      kernel_initializer='normal', kernel_stddev=0.01,
                      bias_value=float(-np.log((1 - 0.01) / 0.01)), name=f'{prefix}/output')

    def get_config(self):
        config = super().get_config()
        config.update({"num_classes": self.num_classes})
        return config


class PriorLayer(Layer):
    """Prior boxes for the image size (reference :236-306).  The table depends only on (H, W), so it
    is built once on the host (PriorBoxes.anchors) and cached on the device as int32 [A,4]."""

    def __init__(self, prior, padding='same', **kwargs):
        if isinstance(prior, dict):
            self.prior = PriorBoxes(**prior)
        elif isinstance(prior, PriorBoxes):
            self.prior = prior
        else:
            raise ValueError('prior must be an instance of the PriorBoxes class.')
        self.padding = padding
        kwargs.update({"trainable": False})
        super().__init__(**kwargs)
        self._cache = {}

    def anchors(self, height, width, device):
        key = (int(height), int(width), str(device))
        if key not in self._cache:
            self._cache[key] = torch.from_numpy(self.prior.anchors(height, width, self.padding)).to(device)
        return self._cache[key]

    def call(self, inputs, **kwargs):
        B, H, W = inputs.shape[0], inputs.shape[1], inputs.shape[2]
        a = self.anchors(H, W, inputs.device)
        return a.unsqueeze(0).expand(B, -1, -1)     # K.repeat + transpose (reference :296-297) as a view

    def get_config(self):
        config = super().get_config()
        config.update({"prior": self.prior.config, 'padding': self.padding})
        return config


class RestoreBoxes(Layer):
    """(loc_pred, pr_boxes) -> (cx,cy,w,h) (reference :309-344)."""

    def call(self, inputs, **kwargs):
        loc_pred, pr_boxes = inputs[0], inputs[1]
        priors = pr_boxes[0] if pr_boxes.dim() == 3 else pr_boxes   # shared by the batch
        return ops.restore_boxes(loc_pred.contiguous(), priors.contiguous().to(torch.int32))


class NormalizeBoxes(Layer):
    """(cx,cy,w,h) -> normalised (y1,x1,y2,x2) (reference :347-375).  On the hot path this is
    fused into the NMS / RoI kernels; the standalone layer is tiny index arithmetic on <= B*N
    boxes and is evaluated with torch elementwise ops."""

    def call(self, inputs, **kwargs):
        shape = kwargs.get('shape', (1.0, 1.0))
        ih, iw = float(shape[0]), float(shape[1])
        cx, cy, w, h = inputs[..., 0], inputs[..., 1], inputs[..., 2], inputs[..., 3]
        return torch.stack([(cy - h / 2) / ih, (cx - w / 2) / iw, (cy + h / 2) / ih, (cx + w / 2) / iw], dim=-1)


class DetectionProposal(Layer):
    """Threshold -> per-(image,class) NMS -> per-image cross-class NMS -> [B,N,6] rows
    (cx,cy,w,h,class id,confidence), -1 padded (reference :435-578).  The device path is fixed
    capacity (N = nms_max_output_size); `call` trims to the reference's dynamic N = max(1, max count)
    with ONE tiny device->host read of the per-image counts."""

    def __init__(self, min_confidence=0.05, nms_iou_threshold=0.4, post_iou_threshold=0.65,
                 nms_max_output_size=1000, max_batch_size=64, **kwargs):
        self.min_confidence = min_confidence
        self.nms_iou_threshold = nms_iou_threshold
        self.post_iou_threshold = post_iou_threshold
        self.nms_max_output_size = nms_max_output_size
        self.max_batch_size = max_batch_size
        super().__init__(**kwargs)

    def propose_fixed(self, cls_pred, boxes, want_kept=False, want_payload=False):
        """-> proposed [B,cap,6] (-1 padded), counts [B] int32 (device), kept [B,cap,2] or None
        [, payload [B,cap*6+1]: the all-gather record, see parallel.all_gather_detections]."""
        if cls_pred.shape[0] > 32 and self.max_batch_size is not None:
            raise ValueError("DetectionProposal: MoldBatch supports at most 32 images per call "
                             "(reference misc.py:275); shard larger batches")
        return ops.detection_proposal(cls_pred.contiguous(), boxes.contiguous(), self.min_confidence,
                                      self.nms_iou_threshold, self.post_iou_threshold,
                                      self.nms_max_output_size, want_kept=want_kept, want_payload=want_payload)

    def call(self, inputs, **kwargs):
        cls_pred, boxes = inputs[0], inputs[1]
        proposed, counts, _ = self.propose_fixed(cls_pred, boxes)
        n = max(1, max(counts.tolist()))          # the dynamic N of the reference (MoldBatch): B ints read by the host
        return proposed[:, :n].contiguous()

    def get_config(self):
        config = super().get_config()
        config.update({
            "min_confidence": self.min_confidence, "nms_iou_threshold": self.nms_iou_threshold,
            "post_iou_threshold": self.post_iou_threshold, "nms_max_output_size": self.nms_max_output_size,
            "max_batch_size": self.max_batch_size})
        return config
