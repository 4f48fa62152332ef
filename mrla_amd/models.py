"""Drop-in for the reference's `models` package (resnet/models/__init__.py:1-5): every lowercase callable is
an `--arch` choice for `resnet/train.py:21-26,158` (`import mrla_amd.models as models`)."""
from .resnet import (MRLA_BasicBlock, MRLA_Bottleneck, MRLA_Bottleneck_base, ResNet_mrlab, ResNet_mrlal,  # noqa: F401
                     resnet18_mrlal, resnet34_mrlal, resnet50_mrlab, resnet50_mrlal, resnet101_mrlab, resnet101_mrlal)

# resnet18_mrlal / resnet34_mrlal: build-side BasicBlock extensions (the reference defines none)
__all__ = ["ResNet_mrlal", "resnet50_mrlal", "resnet101_mrlal", "ResNet_mrlab", "resnet50_mrlab", "resnet101_mrlab",
           "resnet18_mrlal", "resnet34_mrlal"]
