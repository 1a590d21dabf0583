"""masklab_hip -- MI355X-native (gfx950) implementation of the MaskLab inference hot path of
craftsangjae/instance-segmentation-road-project.  Python host code mirrors the reference's
`engine` package (same class / function names); all arithmetic runs in hand-written HIP kernels
behind the C ABI of include/masklab_hip.h.  PyTorch supplies device memory and streams only."""
from .config import ModelConfiguration
from .keras_like import clear_session
from .normalization import GroupNormalization
from .prior import PriorBoxes

__version__ = "0.1.0"


def get_custom_objects():
    """Name -> class registry, like the dict reference engine/__init__.py:17-73 fills."""
    from . import layers as L
    from .backbone import BackBonePreProcess
    names = ["RestoreBoxes", "PriorLayer", "Identity", "FeaturePyramid", "ClassificationSubNet",
             "BoxRegressionSubNet", "MaskSubNet", "NormalizeBoxes", "DetectionProposal", "MoldBatch",
             "MaskDistribute", "PyramidRoiAlign", "ResizeLike", "AtrousSeparableConv2D", "ASPPNetwork",
             "SegmentationSubNet", "SqueezeExcite", "MobileSeparableConv2D", "DownSampleInput", "UpSampleOutput",
             "TrimInstances", "SemanticSmoothing", "CropAndPadMask", "CrackToInstance", "SummaryOutput",
             "IncludeMyRoad", "CalculateInstanceSize"]
    reg = {n: getattr(L, n) for n in names}
    reg["BackBonePreProcess"] = BackBonePreProcess
    reg["GroupNormalization"] = GroupNormalization
    return reg
