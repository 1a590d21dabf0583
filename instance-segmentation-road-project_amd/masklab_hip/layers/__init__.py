"""Layer registry of the hot path (mirrors reference engine/layers/__init__.py:5-8)."""
from .detection import *  # noqa: F401,F403
from .detection import (BoxRegressionSubNet, ClassificationSubNet, DetectionProposal, FeaturePyramid,
                        NormalizeBoxes, PriorLayer, RestoreBoxes)
from .instance import MaskDistribute, MaskSubNet, PyramidRoiAlign, TrimInstances
from .misc import (CalculateInstanceSize, CrackToInstance, CropAndPadMask, DownSampleInput, Identity, IncludeMyRoad,
                   MobileSeparableConv2D, MoldBatch, ReLU, ResizeLike, SqueezeExcite, SummaryOutput, UpSampleOutput)
from .semantic import ASPPNetwork, AtrousSeparableConv2D, SegmentationSubNet, SemanticSmoothing
