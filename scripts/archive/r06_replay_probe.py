"""One-off: the resnet101_mrlab replay test with the row ranges off / on (which of the two runs is non-finite, and where)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mrla_amd import _lib as L
from tests import test_graph_replay_gpu as T
lib = L.load()
for mode in (1, 0):
    lib.mrla_tuning_row_ranges(mode)
    try:
        T.test_replayed_bf16_step_equals_eager_steps("resnet101_mrlab")
        print("mode", mode, "PASSED", flush=True)
    except AssertionError as e:
        print("mode", mode, "FAILED", str(e)[:300], flush=True)
