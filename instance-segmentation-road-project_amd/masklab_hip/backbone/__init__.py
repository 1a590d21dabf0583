"""Backbones on the accelerated path (mirrors reference engine/backbone/__init__.py:8-11)."""
from .base import BACKBONE_LAYERS, BackBonePreProcess, BackboneModel, load_backbone
from .mobilenet import MobileNetV1
from .resnext import ResNeXt50
from .resnext101 import ResNeXt101
