"""Drop-in for the reference's `models` package (resnet/models/__init__.py:1-5): every lowercase callable is
an `--arch` choice for `resnet/train.py:21-26,158` (`import mrla_amd.models as models`)."""
from .resnet import MRLA_Bottleneck, ResNet_mrlal, resnet50_mrlal, resnet101_mrlal  # noqa: F401

__all__ = ["ResNet_mrlal", "resnet50_mrlal", "resnet101_mrlal"]
