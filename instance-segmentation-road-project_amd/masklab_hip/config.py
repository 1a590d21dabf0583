"""ModelConfiguration -- the same attribute tree, field names and defaults as the reference's
engine/config.py:10-248 (it is part of the drop-in API: the builders read it verbatim).
Differences: groups are per-INSTANCE objects (the reference shares class-level singletons, so
two configs alias each other), and `get_arg_parser` parses tuples/bools properly.
"""
import argparse
import copy
import os

ROOT_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Group:
    def attrs(self):
        return [a for a in dir(self) if not a.startswith("_") and a != "attrs"]


class _PostProcess(_Group):
    resolution = (540, 960)
    min_confidence = 0.3
    nms_iou_threshold = 0.4
    post_iou_threshold = 0.6
    nms_max_output_size = 100
    smoothing_kernel_sizes = (0, 0, 0)
    smoothing_weights = (1., 1., 1.)
    instance_colors = [[192, 32, 128], [160, 96, 0], [96, 0, 128], [32, 96, 192], [96, 32, 128]]
    instance_alpha = 0.3
    semantic_colors = [[64, 0, 128], [128, 96, 0], [128, 192, 0]]
    semantic_alpha = 0.3
    default_road_size = 3.25


class _BackBone(_Group):
    backbone_type = 'resnet50'
    num_features = 128
    backbone_outputs = ('C3', 'C4', 'C5', 'P6', 'P7')


class _Detection(_Group):
    pr_scales = [2 ** 0, 2 ** (1 / 3), 2 ** (2 / 3)]
    pr_ratios = [1 / 3, 1 / 2, 1, 2, 3]
    feature_pyramid_inputs = ('C3', 'C4', 'C5')
    num_features = 128
    num_depth = 4
    use_separable_conv = False
    expand_ratio = 4.
    use_squeeze_excite = False
    squeeze_ratio = 16
    groups = 16
    min_confidence = 0.5
    nms_iou_threshold = 0.4
    post_iou_threshold = 0.6
    nms_max_output_size = 100


class _Instance(_Group):
    max_k = 2
    base_size = 36
    crop_size = (14, 14)
    num_features = 128
    num_depth = 4
    use_separable_conv = False
    expand_ratio = 4.
    use_squeeze_excite = False
    squeeze_ratio = 16
    groups = 16


class _Semantic(_Group):
    num_aspp_features = 128
    atrous_rate = (6, 12, 18)
    atrous_groups = 16
    skip_input_name = 'C3'
    aspp_input_name = 'C5'
    num_features = 128
    num_skip_features = 32
    num_depth = 4
    use_separable_conv = False
    expand_ratio = 4.
    use_squeeze_excite = False
    squeeze_ratio = 16
    groups = 16


class _Loss(_Group):
    cls_loss_weight = 300
    cls_loss_alpha = 0.25
    cls_loss_gamma = 2.
    box_loss_weight = 1.
    box_loss_momentum = .9
    box_loss_beta = .11
    box_loss_use_adjust = True
    mask_loss_weight = 1e-2
    mask_loss_label_smoothing = 0.
    seg_loss_weight = .5
    seg_loss_label_smoothing = 0.
    min_confidence = 5e-2
    nms_iou_threshold = 0.6
    post_iou_threshold = 0.8
    nms_max_output_size = 100


class _Dataset(_Group):
    train_cases = []
    valid_cases = []
    min_area = 1000.0
    instance_labels = ('car', 'bump', 'manhole', 'steel', 'pothole')
    semantic_labels = ('other_road', 'my_road', 'crack')
    except_semantic_labels = ('car',)
    data_dir = os.path.join(ROOT_DIR, "datasets/")


class _Train(_Group):
    save_dir = os.path.join(ROOT_DIR, "logs/")
    gpu_count = 2
    use_multiprocessing = True
    batch_size = 8
    max_batch_size = 32
    inference_batch_size = 1
    scale_ratio = (0.4, 0.6)
    train_head_tune = True
    train_head_level = 'C5'
    train_head_tune_epoch = 10
    head_base_lr = 1e-4
    head_max_lr = 1e-3
    head_step_size = 700
    train_waist_tune = True
    train_waist_level = 'C2'
    train_waist_tune_epoch = 10
    waist_base_lr = 1e-4
    waist_max_lr = 1e-3
    waist_step_size = 700
    train_all = True
    train_all_epoch = 30
    all_base_lr = 1e-5
    all_max_lr = 1e-4
    all_step_size = 700


_GROUPS = (("postprocess", _PostProcess), ("backbone", _BackBone), ("detection", _Detection),
           ("instance", _Instance), ("semantic", _Semantic), ("loss", _Loss), ("dataset", _Dataset),
           ("train", _Train))


class ModelConfiguration:
    _PostProcess, _BackBone, _Detection, _Instance = _PostProcess, _BackBone, _Detection, _Instance
    _Semantic, _Loss, _Dataset, _Train = _Semantic, _Loss, _Dataset, _Train

    def __init__(self):
        for name, cls in _GROUPS:
            grp = cls()
            for a in grp.attrs():                     # per-instance copies of mutable defaults
                setattr(grp, a, copy.deepcopy(getattr(cls, a)))
            setattr(self, name, grp)

    def __dir__(self):
        return [name for name, _ in _GROUPS]

    def to_dict(self):
        return {g: {a: getattr(getattr(self, g), a) for a in getattr(self, g).attrs()} for g in dir(self)}

    def from_dict(self, config_dict):
        for attr_group, attr_dict in config_dict.items():
            for key, value in attr_dict.items():
                setattr(getattr(self, attr_group), key, value)

    def update(self, attr_group, key, value):
        setattr(getattr(self, attr_group), key, value)

    def get_arg_parser(self, default_config=None, argv=None):
        default_config = default_config or self
        parser = argparse.ArgumentParser()

        def conv(default):
            if isinstance(default, bool):
                return lambda s: s.lower() in ("1", "true", "yes")
            return type(default)

        for g in dir(self):
            grp = getattr(default_config, g)
            for a in grp.attrs():
                default = getattr(grp, a)
                if isinstance(default, (list, tuple)):
                    elem = conv(default[0]) if len(default) and not isinstance(default[0], (list, tuple)) else str
                    parser.add_argument(f"-{g}.{a}", required=False, nargs='+', default=default, type=elem)
                else:
                    parser.add_argument(f"-{g}.{a}", required=False, default=default, type=conv(default))
        return parser.parse_args(argv)
