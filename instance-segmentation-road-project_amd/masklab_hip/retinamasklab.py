"""Model build API -- drop-in for the inference half of reference engine/retinamasklab.py:
build_backbone_network :19-37, build_detection_network :40-112, build_instance_network :115-157,
build_semantic_network :160-198, construct_inference_network :420-495, find_layer_name :646-649,
and the deploy wrapper of load_masklab_inference_model_from_h5 :598-643 (DeployModel below).
Same function names, argument meaning and return structure; the returned model runs eagerly on
the MI355X kernels.  Training-only pieces (construct_trainer_network, losses, dataset) are out of
scope of the accelerated path.
"""
import numpy as np
import torch

from . import backbone
from . import keras_like as K
from .config import ModelConfiguration
from .layers import (ASPPNetwork, BoxRegressionSubNet, ClassificationSubNet, CropAndPadMask, DetectionProposal,
                     DownSampleInput, SummaryOutput,
                     FeaturePyramid, MaskDistribute, MaskSubNet, PriorLayer, PyramidRoiAlign, ResizeLike,
                     RestoreBoxes, SegmentationSubNet, SemanticSmoothing, TrimInstances, UpSampleOutput)
from .prior import PriorBoxes

VERBOSE = False


def _say(*lines):
    if VERBOSE:
        print("\n".join(lines))


def build_backbone_network(configuration: ModelConfiguration):
    config = configuration.backbone
    model = backbone.load_backbone(config.backbone_type, config.backbone_outputs, config.num_features)
    _say("Backbone Network Summary", "------------------------",
         f"* Backbone Type : {config.backbone_type}", f"* Backbone outputs : {config.backbone_outputs}",
         f"* Num features of Backbone Additional Layers: {config.num_features}", "------------------------\n")
    return model


def build_detection_network(configuration: ModelConfiguration):
    config = configuration.detection
    num_backbone_outputs = len(configuration.backbone.backbone_outputs)
    num_classes = len(configuration.dataset.instance_labels)
    # the output block's name encodes its stride (= number of stride-2 layers), reference :46-48
    prior_strides = [2 ** int(output_name[-1]) for output_name in configuration.backbone.backbone_outputs]
    prior_sizes = [4 * stride for stride in prior_strides]
    prior = PriorBoxes(strides=prior_strides, sizes=prior_sizes, pr_scales=config.pr_scales,
                       pr_ratios=config.pr_ratios)
    prior_subnet = PriorLayer(prior)
    assert len(set(config.feature_pyramid_inputs) - {'C1', 'C2', 'C3', 'C4', 'C5', 'P6', 'P7'}) == 0, \
        "feature pyramid inputs must be drawn from C1~C5, P6, P7"
    feature_pyramid_strides = [2 ** int(name[1]) for name in config.feature_pyramid_inputs]
    fpn_subnet = FeaturePyramid(strides=feature_pyramid_strides, num_features=config.num_features)
    cls_subnet = ClassificationSubNet(num_blocks=num_backbone_outputs, num_classes=num_classes,
                                      num_depth=config.num_depth, num_features=config.num_features,
                                      num_priors=len(prior), use_separable_conv=config.use_separable_conv,
                                      expand_ratio=config.expand_ratio,
                                      use_squeeze_excite=config.use_squeeze_excite,
                                      squeeze_ratio=config.squeeze_ratio, groups=config.groups)
    # NB reference quirk kept (:95): the box tower's squeeze-excite flag is `use_separable_conv`
    loc_subnet = BoxRegressionSubNet(num_blocks=num_backbone_outputs, num_depth=config.num_depth,
                                     num_features=config.num_features, num_priors=len(prior),
                                     use_separable_conv=config.use_separable_conv,
                                     expand_ratio=config.expand_ratio,
                                     use_squeeze_excite=config.use_separable_conv,
                                     squeeze_ratio=config.squeeze_ratio, groups=config.groups)
    return prior_subnet, fpn_subnet, cls_subnet, loc_subnet


def build_instance_network(configuration: ModelConfiguration):
    num_classes = len(configuration.dataset.instance_labels)
    config = configuration.instance
    restore_subnet = RestoreBoxes()
    distribute_subnet = MaskDistribute(max_k=config.max_k, base_size=config.base_size)
    pyramid_roi_align = PyramidRoiAlign(crop_size=config.crop_size)
    # NB reference quirk kept (:139): expand_ratio receives `use_separable_conv`
    mask_subnet = MaskSubNet(num_blocks=config.max_k + 1, num_classes=num_classes, num_depth=config.num_depth,
                             num_features=config.num_features, use_separable_conv=config.use_separable_conv,
                             expand_ratio=config.use_separable_conv, use_squeeze_excite=config.use_squeeze_excite,
                             squeeze_ratio=config.squeeze_ratio, groups=config.groups)
    return restore_subnet, distribute_subnet, pyramid_roi_align, mask_subnet


def build_semantic_network(configuration: ModelConfiguration):
    config = configuration.semantic
    num_classes = len(configuration.dataset.semantic_labels)
    aspp_subnet = ASPPNetwork(num_features=config.num_aspp_features, atrous_rate=config.atrous_rate,
                              groups=config.atrous_groups)
    # NB reference quirk kept (:179): expand_ratio receives `use_separable_conv`
    seg_subnet = SegmentationSubNet(num_depth=config.num_depth, num_features=config.num_features,
                                    num_skip_features=config.num_skip_features, num_classes=num_classes,
                                    use_separable_conv=config.use_separable_conv,
                                    expand_ratio=config.use_separable_conv,
                                    use_squeeze_excite=config.use_squeeze_excite,
                                    squeeze_ratio=config.squeeze_ratio, groups=config.groups)
    return aspp_subnet, seg_subnet


class InferenceModel(K.Layer):
    """The Keras functional `Model(images -> [cls_pred, loc_pred, roi_boxes, roi_masks, seg_pred])`
    of reference :420-495, executed eagerly.  Output list shrinks like the reference when a head
    group is None."""

    def __init__(self, configuration, backbone_network, detection_networks=None, semantic_networks=None,
                 instance_networks=None, name='inference'):
        super().__init__(name=name)
        self.configuration = configuration
        self.backbone_network = backbone_network
        self.detection_networks = detection_networks
        self.semantic_networks = semantic_networks
        self.instance_networks = instance_networks if detection_networks is not None else None
        self.detection_proposal = None
        self.output_names = []
        if detection_networks is not None:
            self.output_names += ['cls_pred', 'loc_pred']
            if self.instance_networks is not None:
                det_config = configuration.detection
                # re-instantiated from config.detection.* every time (reference :459-466)
                self.detection_proposal = DetectionProposal(
                    min_confidence=det_config.min_confidence, nms_iou_threshold=det_config.nms_iou_threshold,
                    post_iou_threshold=det_config.post_iou_threshold,
                    nms_max_output_size=det_config.nms_max_output_size,
                    max_batch_size=configuration.train.inference_batch_size)
                self.output_names += ['roi_boxes', 'roi_masks']
        if semantic_networks is not None:
            self.output_names.append('seg_pred')
        self.layers = self._flat_layers()
        self.device = None
        self.last_detections = None
        self._use_graphs = False
        self._graphs = {}
        # Auxiliary HIP streams (semantic head, box tower, P6/P7, coarse FPN convs beside the main chain): True / False /
        # "auto" = on unless the forward is so small that it is bound by the host's launch rate, where the extra
        # stream hand-overs cost more than the overlap gives (1 x 512 x 512: 2.78 ms with, 2.45 ms without); under
        # hipGraph capture they are always worth it (the branches replay without host work: 1.94 ms).
        self.use_side_stream = "auto"
        self._side_stream = None
        # Stage 2 without a host read ("dynamic shapes without host sync"): RoI crops + mask head launched at
        # capacity, the kernels reading the per-level RoI maxima from device memory; the molded shapes are produced when
        # the outputs are handed over (ONE read, after the whole forward is enqueued).  True / False / "auto" = with
        # hipGraph replay (the whole forward is then ONE graph), where nothing but that read remains for the host.
        self.device_counts = "auto"
        self._capacity_held = 0          # bytes of capacity tensors already captured in this model's graphs (they count as free)
        self._build_shapes()

    # ---- structure
    def _flat_layers(self):
        layers = [self.backbone_network]
        for grp in (self.detection_networks, self.instance_networks, self.semantic_networks):
            if grp is not None:
                layers += list(grp)
        if self.detection_proposal is not None:
            layers.append(self.detection_proposal)
        return layers

    def get_layer(self, name):
        for l in self.layers:
            if l.name == name:
                return l
        raise ValueError(f"No such layer: {name}")

    def children(self):
        return self.layers

    def weight_specs(self):
        out = {}
        for l in self.layers:
            out.update(l.weight_specs())
        return out

    def _build_shapes(self):
        bb = self.backbone_network
        shapes = bb.output_shapes if bb.built else bb.build((None, None, None, 3))
        by_name = dict(zip(bb.output_names, shapes))
        if self.detection_networks is not None:
            det_config = self.configuration.detection
            _, fpn, cls, loc = self.detection_networks
            fpn_in = [by_name[n] for n in bb.output_names if n in det_config.feature_pyramid_inputs]
            rest = [by_name[n] for n in bb.output_names if n not in det_config.feature_pyramid_inputs]
            feats = fpn.build(fpn_in) + rest
            cls.build(feats)
            loc.build(feats)
            if self.instance_networks is not None:
                ins = self.configuration.instance
                mask = self.instance_networks[3]
                ch, cw = ins.crop_size
                mask.build([(None, None, ch, cw, f[-1]) for f in feats[:ins.max_k + 1]])
        if self.semantic_networks is not None:
            seg_config = self.configuration.semantic
            aspp, seg = self.semantic_networks
            a = aspp.build(by_name[seg_config.aspp_input_name])
            seg.build([a, by_name[seg_config.skip_input_name]])
        self.built = True

    # ---- weights
    def init_weights(self, seed=0):
        """Random weights from each layer's initializer (what a fresh Keras model has), keyed and
        seeded by weight name so the result is independent of construction order."""
        return K.init_weights(self.weight_specs(), seed)

    def _drop_graphs(self):
        """Forget the captured graphs (their private memory pools go with them) and let ops release the scratch buffers
        that only those captures kept alive."""
        from . import ops
        self._graphs = {}
        self._capacity_held = 0
        ops.graph_owner_released(self)

    def load_weights(self, weights, device="cuda"):
        self.device = torch.device(device)
        self._drop_graphs()                  # captured graphs hold the addresses of the tensors being replaced
        for l in self.layers:
            l.load_weights(weights, self.device)
        return self

    def reload_class_outputs(self, weights):
        """Re-pack and upload only the class towers' output convs (parity fixtures rescale those kernels to
        move the score distribution; everything else keeps its device copy)."""
        if self.detection_networks is None:
            raise RuntimeError("no detection networks")
        self._drop_graphs()
        for block in self.detection_networks[2].blocks:
            block[-1].load_weights(weights, self.device)

    # ---- forward (reference :431-491)
    # Stage 1 = everything up to the one host read (backbone, FPN, towers, box decode, DetectionProposal,
    # level assignment, semantic head); stage 2 = RoI crops + mask head, whose molded shapes depend on the
    # RoI counts the host reads in between (the reference's dynamic shapes).
    def _stage1(self, images, want_kept=False):
        cfg = self.configuration
        bb = self.backbone_network
        aux = self._aux_streams_wanted(images)
        fork_ok = aux and self.detection_networks is not None and hasattr(bb, "join_extra_levels")
        feats = bb(images, fork_stream=self._fork_stream) if fork_ok else bb(images)
        by_name = dict(zip(bb.output_names, feats))
        st = {"image_hw": (int(images.shape[1]), int(images.shape[2]))}

        def semantic_head():
            seg_config = cfg.semantic
            aspp_subnet, seg_subnet = self.semantic_networks
            aspp_outputs = aspp_subnet(by_name[seg_config.aspp_input_name])
            return seg_subnet([aspp_outputs, by_name[seg_config.skip_input_name]])

        # The semantic head only needs backbone taps.  With `use_side_stream` it goes to a second HIP stream, enqueued
        # once the towers are queued: it then runs beside the detection post-processing (40 + 8 blocks), through
        # the host's read of the RoI counts and its preparation of stage 2 (~0.2 ms during which the main stream is
        # empty), and beside the first mask-head launches.  Joined at the end of stage 2 (of stage 1 under capture).
        side = None

        def launch_semantic_on_side():
            from . import ops as _ops
            if self.semantic_networks is None or not aux or _ops.PROFILE is not None:
                return None
            main = torch.cuda.current_stream()
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream(device=self.device)
            self._side_stream.wait_stream(main)
            with torch.cuda.stream(self._side_stream):
                st["seg_pred"] = semantic_head()
            return self._side_stream

        if self.detection_networks is None:
            side = launch_semantic_on_side()
        if self.detection_networks is not None:
            det_config = cfg.detection
            prior_subnet, fpn_subnet, cls_subnet, loc_subnet = self.detection_networks
            pr_boxes = prior_subnet(images)
            fpn_inputs = [by_name[n] for n in bb.output_names if n in det_config.feature_pyramid_inputs]
            without_fpn = [by_name[n] for n in bb.output_names if n not in det_config.feature_pyramid_inputs]
            feature_outputs = (fpn_subnet(fpn_inputs, fork_stream=self._fork_stream) if fork_ok
                               else fpn_subnet(fpn_inputs)) + without_fpn
            if fork_ok:
                bb.join_extra_levels(feats)                    # P6 / P7 ran beside the FPN chain
            # The class and the box tower are independent chains of the same shape (5 levels x 4 convs, 1365 tiles per
            # launch = 2.67 rounds of the chip's 512 resident blocks): on two streams the last, partly filled round of
            # one launch is topped up by the other tower's blocks.  Same kernels, same inputs: bit-identical outputs.
            tower_stream = self._fork_stream("_tower_stream") if aux else None
            if tower_stream is not None:
                with torch.cuda.stream(tower_stream):
                    st["loc_pred"] = loc_subnet(feature_outputs)
                st["cls_pred"] = cls_subnet(feature_outputs)
                torch.cuda.current_stream().wait_stream(tower_stream)
                st["loc_pred"].record_stream(torch.cuda.current_stream())
            else:
                st["cls_pred"] = cls_subnet(feature_outputs)
                st["loc_pred"] = loc_subnet(feature_outputs)
            side = launch_semantic_on_side()
            if self.instance_networks is not None:
                restore_subnet, distribute_subnet, pyramid_roi_align, _ = self.instance_networks
                restored_boxes = restore_subnet([st["loc_pred"], pr_boxes])
                # fused DetectionProposal -> MaskDistribute -> PyramidRoiAlign in fixed capacity:
                # the only host read is the per-level RoI count that sizes the molded outputs.
                proposed, counts, kept, payload = self.detection_proposal.propose_fixed(
                    st["cls_pred"], restored_boxes, want_kept=want_kept, want_payload=True)
                n_levels = cfg.instance.max_k + 1
                if distribute_subnet.max_k + 1 != n_levels:
                    raise ValueError("MaskDistribute.max_k does not match config.instance.max_k")
                slots, lcounts, lmax = pyramid_roi_align.distribute(n_levels, proposed, has_k=False,
                                                                    base_size=distribute_subnet.base_size)
                st.update(proposed=proposed, counts=counts, kept=kept, payload=payload, boxes=restored_boxes,
                          slots=slots, lcounts=lcounts, lmax=lmax, roi_features=feature_outputs[:n_levels])
        if side is not None:
            st["side"] = side
            st["keepalive"] = by_name            # backbone taps the side stream still reads
            if torch.cuda.is_current_stream_capturing() and not getattr(self, "_capturing_whole", False):
                self._join_side(st)              # a captured stage 1 must be self-contained
        elif self.semantic_networks is not None:
            # independent of the instance branch and enqueued BEFORE the host reads the RoI counts, so the
            # GPU stays busy while the host waits (same stream, same results)
            st["seg_pred"] = semantic_head()
        return st

    def _aux_streams_wanted(self, images):
        mode = getattr(self, "use_side_stream", False)
        if mode != "auto":
            return bool(mode)
        if torch.cuda.is_current_stream_capturing():
            return True
        return int(images.shape[0]) * int(images.shape[1]) * int(images.shape[2]) > 512 * 512

    def _fork_stream(self, attr):
        """A lazily created auxiliary stream that starts behind everything enqueued on the current one (None under the
        per-launch profiling hook, whose event pairs assume one stream)."""
        from . import ops as _ops
        if _ops.PROFILE is not None:
            return None
        s = getattr(self, attr, None)
        if s is None:
            s = torch.cuda.Stream(device=self.device)
            setattr(self, attr, s)
        s.wait_stream(torch.cuda.current_stream())
        return s

    def _join_side(self, st):
        side = st.pop("side", None)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
            st["seg_pred"].record_stream(torch.cuda.current_stream())
            st.pop("keepalive", None)

    def _stage2(self, st):
        outputs = []
        if self.detection_networks is not None:
            outputs += [st["cls_pred"], st["loc_pred"]]
            if self.instance_networks is not None:
                _, _, pyramid_roi_align, mask_subnet = self.instance_networks
                roi_fmaps, roi_boxes = pyramid_roi_align.crop_distributed(
                    st["roi_features"], st["proposed"], st["image_hw"], st["slots"], st["lcounts"], st["lmax"])
                roi_masks = mask_subnet(roi_fmaps)
                outputs += [roi_boxes, roi_masks]
                self.last_detections = dict(proposed=st["proposed"], counts=st["counts"], kept=st["kept"],
                                            boxes=st["boxes"], payload=st["payload"],
                                            level_counts=st["lcounts"], level_max=st["lmax"])
        self._join_side(st)
        if self.semantic_networks is not None:
            outputs.append(st["seg_pred"])
        return outputs

    # ---- stage 2 at capacity: no host read inside the forward
    # The capacity form trades memory for the missing host read: every RoI level is sized for B x nms_max_output_size slots.
    # It is used while its tensors stay below CAPACITY_BYTES_LIMIT and a quarter of the device's free memory (each input
    # shape captured under hipGraph keeps its own set: graph-private pools are not shared), and never past
    # MAX_CAPACITY_ROIS slots per level.
    MAX_CAPACITY_ROIS = 8192
    CAPACITY_BYTES_LIMIT = 16 << 30

    def capacity_bytes(self, batch):
        """Estimate of what ONE fixed-capacity stage 2 holds: per RoI level the crops and every tower / deconv
        intermediate ([B*cap, ch, cw, C] each: under graph capture none of them is recycled), plus masks_cap
        [B, L*cap, 2ch, 2cw, classes] fp32 twice (at capacity, and the buffer its molded form is the front of) and the
        split-K workspace (per stream)."""
        from . import ops
        ins = self.configuration.instance
        pra, mask = self.instance_networks[2], self.instance_networks[3]
        cap = int(self.detection_proposal.nms_max_output_size)
        L = ins.max_k + 1
        ch, cw = pra.crop_size
        es = 2 if ops.half_storage() else 4
        slots = int(batch) * cap
        per_level = slots * ch * cw * mask.num_features * es * (2 + mask.num_depth)
        masks = slots * L * 4 * ch * cw * mask.num_classes * 4
        return L * per_level + 2 * masks + 2 * int(ops._lib.load().ml_conv2d_workspace_bytes())   # (masks: at capacity + molded)

    def _capacity_wanted(self, images):
        mode = getattr(self, "device_counts", "auto")
        on = self._use_graphs if mode == "auto" else bool(mode)
        if not on or self.instance_networks is None:
            return False
        pra, mask = self.instance_networks[2], self.instance_networks[3]
        cap = int(self.detection_proposal.nms_max_output_size)
        if not (mask.capacity_supported(tuple(pra.crop_size)) and int(images.shape[0]) * cap <= self.MAX_CAPACITY_ROIS
                and cap * pra.crop_size[0] * pra.crop_size[1] >= 128):
            return False
        need = self.capacity_bytes(int(images.shape[0]))
        limit = self.CAPACITY_BYTES_LIMIT
        if self.device is not None and self.device.type == "cuda":
            limit = min(limit, torch.cuda.mem_get_info(self.device)[0] // 4 + self._capacity_held)
        return need <= limit

    def _stage2_capacity(self, st):
        """RoI crops + mask head with every level at capacity (reference engine/layers/instance.py:115-134,211-233 +
        MoldBatch misc.py:231-286): kernels skip the RoI slots past the level maxima they read on the device.
        -> the forward's RAW result: fixed-shape tensors + `lmax` (what `_mold` needs the host to read)."""
        from . import ops
        _, _, pyramid_roi_align, mask_subnet = self.instance_networks
        roi_fmaps, boxes_cap, lives = pyramid_roi_align.crop_capacity(
            st["roi_features"], st["proposed"], st["image_hw"], st["slots"], st["lcounts"], st["lmax"])
        masks_cap = mask_subnet(roi_fmaps, lives=lives)
        self.last_detections = dict(proposed=st["proposed"], counts=st["counts"], kept=st["kept"], boxes=st["boxes"],
                                    payload=st["payload"], level_counts=st["lcounts"], level_max=st["lmax"])
        self._join_side(st)
        # MoldBatch + Concatenate(axis=1) at the front of two capacity buffers, sized on the device from `lmax`: part of the
        # captured forward, so that nothing is launched after it (round 4: the two molding launches behind the host's read
        # left the GPU idle for their launch latencies -- 0.25 ms of gaps per forward under a kernel trace at 1 x 512^2)
        cap = int(st["proposed"].shape[1])
        return dict(cls_pred=st["cls_pred"], loc_pred=st["loc_pred"], boxes_cap=boxes_cap, masks_cap=masks_cap,
                    boxes_molded=ops.mold_levels_dev(boxes_cap, st["lmax"], cap),
                    masks_molded=ops.mold_levels_dev(masks_cap, st["lmax"], cap),
                    lmax=st["lmax"], cap=cap, seg_pred=st.get("seg_pred"))

    def _mold(self, raw):
        """The ONE host read of the forward (the per-level RoI maxima, L ints) and the molded outputs it sizes."""
        from . import ops
        n_l = [min(max(1, int(v)), raw["cap"]) for v in raw["lmax"].tolist()]
        outputs = [raw["cls_pred"], raw["loc_pred"], ops.molded_front(raw["boxes_molded"], n_l),
                   ops.molded_front(raw["masks_molded"], n_l)]
        if self.semantic_networks is not None:
            outputs.append(raw["seg_pred"])
        return outputs

    # ---- hipGraph replay (launch-bound small batches: serving one image at a time)
    def enable_graphs(self, enabled=True):
        """Capture the forward (~300 kernel launches) into a hipGraph per input shape and replay it: one launch instead
        of hundreds, which is what a batch-1 forward is bound by.  With the fixed-capacity stage 2 (`device_counts`) the
        WHOLE forward is one graph; otherwise stage 1 is (stage 2's shapes then depend on a host read).  The tensors
        the graph returns are graph-owned buffers, valid until the next call (`outputs_graph_owned`).
        Memory: every captured input shape (x conv math x thresholds) keeps its own activations, capacity tensors
        (`capacity_bytes`) and split-K workspace for as long as the graph lives; `enable_graphs(False)`, `load_weights`
        and `reload_class_outputs` drop them all.  A server that sees many input shapes should bucket them."""
        self._use_graphs = bool(enabled)
        if not enabled:
            self._drop_graphs()
        return self

    @property
    def outputs_graph_owned(self):
        return bool(self._use_graphs)

    def _stage1_graphed(self, images, whole=False):
        from . import ops
        # kernel arguments are baked into a capture: everything they are computed from is part of the key
        dp = self.detection_proposal
        key = (tuple(images.shape), images.dtype, ops.CONV_MATH, bool(whole)) + (
            () if dp is None else (float(dp.min_confidence), float(dp.nms_iou_threshold), float(dp.post_iou_threshold),
                                   int(dp.nms_max_output_size)))
        entry = self._graphs.get(key)
        if entry is None:
            if ops.PROFILE is not None:
                raise RuntimeError("graph capture cannot run under the per-launch profiling hook")
            # warm-up: fills the anchor / workspace caches, sets kernel attributes
            if whole:
                self._stage2_capacity(self._stage1(images))
            else:
                self._join_side(self._stage1(images))
            torch.cuda.synchronize(self.device)
            static_in = images.clone()
            graph = torch.cuda.CUDAGraph()
            self._capturing_whole = bool(whole)          # (the side stream then joins at the end of stage 2)
            try:
                with torch.cuda.graph(graph):
                    st = self._stage1(static_in)
                    if whole:
                        st = self._stage2_capacity(st)
            finally:
                self._capturing_whole = False
            entry = self._graphs[key] = (graph, static_in, st)
            ops.graph_owner_registered(self)
            if whole:
                self._capacity_held += self.capacity_bytes(int(images.shape[0]))
        graph, static_in, st = entry
        static_in.copy_(images)
        graph.replay()
        return st

    def call(self, images, **kwargs):
        if not isinstance(images, torch.Tensor):
            images = torch.as_tensor(np.asarray(images))
        if self.device is None:
            raise RuntimeError("InferenceModel: call load_weights(weights, device) first")
        images = images.to(self.device).contiguous()
        want_kept = kwargs.get("want_kept", False)
        from . import ops
        graphs = getattr(self, "_use_graphs", False) and not want_kept and ops.PROFILE is None
        if not want_kept and self._capacity_wanted(images):
            # no host read inside the forward; with graphs the whole of it is one replay
            raw = self._stage1_graphed(images, whole=True) if graphs else self._stage2_capacity(self._stage1(images))
            if kwargs.get("defer", False):
                return DeferredOutputs(self, raw)
            return self._mold(raw)
        if graphs:
            st = self._stage1_graphed(images)
        else:
            st = self._stage1(images, want_kept=want_kept)
        return self._stage2(st)

    def predict(self, images, **kwargs):
        """Keras `Model.predict`: numpy in, list of numpy out."""
        outs = self.call(images, **kwargs)
        if isinstance(outs, DeferredOutputs):
            outs = outs.materialize()
        torch.cuda.synchronize(self.device)
        return [o.cpu().numpy() for o in outs]


class DeferredOutputs:
    """What `InferenceModel(images, defer=True)` returns on the fixed-capacity path: the forward is fully enqueued and
    NOTHING has been read by the host yet, so a caller that pipelines forwards (bench.py) never stalls; `materialize()`
    does the one read (per-level RoI maxima) and returns the molded output list, shaped exactly like the reference's
    (roi_boxes / roi_masks with N = sum of the level maxima, engine/layers/misc.py:231-286)."""

    def __init__(self, model, raw):
        self.model, self.raw = model, raw

    def materialize(self):
        return self.model._mold(self.raw)


def construct_inference_network(configuration: ModelConfiguration, backbone_network, detection_networks=None,
                                semantic_networks=None, instance_networks=None):
    """Same signature as reference :420-424."""
    return InferenceModel(configuration, backbone_network, detection_networks=detection_networks,
                          semantic_networks=semantic_networks, instance_networks=instance_networks)


def construct_masklab_networks(config: ModelConfiguration, with_trainer=False):
    """Reference :201-220 returns (trainer, inference); the training graph is outside the
    accelerated path, so the first element is None."""
    K.clear_session()
    backbone_network = build_backbone_network(config)
    detection_networks = build_detection_network(config)
    instance_networks = build_instance_network(config)
    semantic_networks = build_semantic_network(config)
    if with_trainer:
        raise NotImplementedError("construct_trainer_network is training-only (out of the hot-path scope)")
    inference = construct_inference_network(configuration=config, backbone_network=backbone_network,
                                            detection_networks=detection_networks,
                                            semantic_networks=semantic_networks,
                                            instance_networks=instance_networks)
    return None, inference


class DeployModel(K.Layer):
    """The non-serving `Model(images -> (detection, instance, semantic))` that reference
    load_masklab_inference_model_from_h5 assembles around the inference network (:598-643):
    DownSampleInput(config.postprocess.resolution) -> inference model -> TrimInstances(mold=True),
    per-class SemanticSmoothing + ResizeLike(target=downsampled) -> UpSampleOutput(target=images).
    All three outputs are int32: detection [B,n,6] (cx,cy,w,h,label,conf*100 at input resolution),
    instance [B,n,h,w] (0/1), semantic [B,H,W,classes] (0/1)."""

    def __init__(self, configuration, inference_model, name='inference'):
        super().__init__(name=name)
        if inference_model.output_names != ['cls_pred', 'loc_pred', 'roi_boxes', 'roi_masks', 'seg_pred']:
            raise ValueError("the deploy wrapper unpacks five outputs (reference :608): it needs the detection, "
                             "instance and semantic heads")
        post = configuration.postprocess
        if len(post.smoothing_kernel_sizes) != len(post.smoothing_weights):
            raise ValueError("postprocess.smoothing_kernel_sizes and smoothing_weights differ in length")
        self.configuration = configuration
        self.model = inference_model
        self.down_sample = DownSampleInput(post.resolution)
        self.trim = TrimInstances(mold=True)
        self.smoothing = [SemanticSmoothing(kernel_size=k, weight=w)
                          for k, w in zip(post.smoothing_kernel_sizes, post.smoothing_weights)]
        self.resize_like = ResizeLike()
        self.up_sample = UpSampleOutput()
        self.output_names = ['detection', 'instance', 'semantic']
        self.built = True

    def call(self, images, **kwargs):
        if not isinstance(images, torch.Tensor):
            images = torch.as_tensor(np.asarray(images))
        if self.model.device is None:
            raise RuntimeError("DeployModel: load weights into the inference model first")
        images = images.to(self.model.device).contiguous()
        downsampled = self.down_sample(images)                                     # :607
        _, _, box_pred, mask_pred, seg_pred = self.model(downsampled)              # :608
        detection_pred, instance_pred = self.trim([box_pred, mask_pred])           # :614-615
        n_split = len(self.smoothing)
        n_cls = int(seg_pred.shape[-1])
        if n_cls % n_split:
            raise ValueError(f"tf.split: {n_cls} semantic classes do not split into {n_split} parts (:619)")
        rep = n_cls // n_split
        post_semantics = SemanticSmoothing.smooth_classes(                         # :619-627 in one launch group
            seg_pred, [l.kernel_size for l in self.smoothing for _ in range(rep)],
            [l.weight for l in self.smoothing for _ in range(rep)])
        semantic_pred = self.resize_like(post_semantics, target=downsampled)       # :628
        return self.up_sample([detection_pred, instance_pred, semantic_pred], target=images)   # :634-635

    def predict(self, images, **kwargs):
        outs = self.call(images, **kwargs)
        torch.cuda.synchronize(self.model.device)
        return [o.cpu().numpy() for o in outs]


def construct_deploy_network(configuration: ModelConfiguration, inference_model):
    """Wrap an inference model like reference :598-643 (serving=False)."""
    return DeployModel(configuration, inference_model)


class ServingModel(K.Layer):
    """The arithmetic half of reference road_project/setup/serving.py:16-52 (`load_serving_model_from_h5`):
    deploy model -> CropAndPadMask -> SummaryOutput.  Returns the 'summarize' tensor float32 [B,n',11]
    (class, cx, cy, w, h, conf, pixel count, instance size, horizontal size, vertical size, include_my_road).
    The 'visualize' output (DrawBoxes / DrawInstance / DrawSegmentation / JPEG encode, :32-40) and the JPEG
    decode in front are image I/O, not arithmetic of this path, and are not built."""

    def __init__(self, configuration, deploy_model, name='serving'):
        super().__init__(name=name)
        self.deploy_model = deploy_model
        self.crop_and_pad = CropAndPadMask()
        self.summary = SummaryOutput(default_road_size=configuration.postprocess.default_road_size)
        self.output_names = ['summarize']
        self.built = True

    def call(self, images, **kwargs):
        if not isinstance(images, torch.Tensor):
            images = torch.as_tensor(np.asarray(images))
        images = images.to(self.deploy_model.model.device).contiguous()
        det_outs, ins_outs, seg_outs = self.deploy_model(images)                            # :27-29
        if kwargs.get("materialise_masks", False):                                          # the reference's literal wiring
            crop_and_pad_masks = self.crop_and_pad([images, det_outs, ins_outs, seg_outs])  # :30
            return self.summary([det_outs, seg_outs, crop_and_pad_masks])                   # :47-48
        # CropAndPadMask folded into SummaryOutput: same numbers bit for bit, no [B,n,H,W] tensor
        return self.summary([det_outs, seg_outs, ins_outs], from_rois=True)

    def predict(self, images, **kwargs):
        out = self.call(images, **kwargs)
        torch.cuda.synchronize(self.deploy_model.model.device)
        return out.cpu().numpy()


def construct_serving_network(configuration: ModelConfiguration, deploy_model):
    return ServingModel(configuration, deploy_model)


def load_masklab_inference_model_from_weights(weights, config: ModelConfiguration, device="cuda"):
    """Counterpart of reference load_masklab_inference_model_from_h5 (:498-643) for a name-keyed
    weight dict / .npz (see tools/convert_keras_h5.py for the h5 -> npz step): builds the networks
    from `config`, loads the weights and returns the deploy model."""
    if isinstance(weights, str):
        with np.load(weights) as z:
            weights = {k: z[k] for k in z.files}
    _, inference = construct_masklab_networks(config)
    inference.load_weights(weights, device)
    return construct_deploy_network(config, inference)


def load_masklab_inference_model_from_h5(save_path, config: ModelConfiguration, serving=False, device="cuda"):
    """Same name and arguments as reference engine/retinamasklab.py:498-643: the checkpoint at `save_path` -> the deploy
    model `images uint8 [B,H,W,3] -> (detection, instance, semantic)` (DownSampleInput -> inference network ->
    TrimInstances / SemanticSmoothing / ResizeLike -> UpSampleOutput).  `save_path`: a Keras .h5 as the reference
    writes it (needs `h5py`, imported lazily; the layer-name re-wiring of :515-586 becomes masklab_hip/checkpoint.py's
    name mapping) or the .npz tools/convert_keras_h5.py makes of it.  The networks are built from `config` (the
    reference rebuilds them from the loaded Keras model; `config.json` saved beside the weights, engine/train.py:31-32,
    is that configuration).  serving=True puts DecodeImageContent (JPEG bytes in, misc.py:328-337) in front -- image I/O,
    outside this library: decode on the host and call the serving=False model."""
    if serving:
        raise NotImplementedError("load_masklab_inference_model_from_h5(serving=True) wraps the model in "
                                  "DecodeImageContent (JPEG byte strings in): image decoding is outside the accelerated "
                                  "path -- decode on the host and use the serving=False model (or construct_serving_network "
                                  "for the summary outputs)")
    _, inference = construct_masklab_networks(config)
    path = str(save_path)
    if path.lower().endswith(".npz"):
        with np.load(path) as z:
            weights = {k: z[k] for k in z.files}
    else:
        from . import checkpoint
        specs = {k: tuple(v.shape) for k, v in inference.weight_specs().items()}
        weights = checkpoint.load_keras_h5(path, specs)
    inference.load_weights(weights, device)
    return construct_deploy_network(config, inference)


def find_layer_name(re_format, model):
    return sorted([layer.name for layer in model.layers if re_format.match(layer.name)])
