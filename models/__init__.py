"""Top-level `models` package: with the repository root on PYTHONPATH the reference's driver works unedited --
`resnet/train.py:21` does `import models` and takes every lowercase callable of `models.__dict__` as an `--arch`
choice (`:24-26,158`).  Pure re-export of mrla_amd.models (the reference's own resnet/models/__init__.py:1-5 is the
same kind of file)."""
from mrla_amd.models import *  # noqa: F401,F403
from mrla_amd.models import __all__  # noqa: F401
