"""mrla_amd -- MI355X-native MRLA (multi-head recurrent layer attention) hot path.

The compute path is libmrla_hip.so (hand-written HIP for gfx950, C ABI in include/mrla_hip.h); this
package is the host-side mirror of the reference's nn.Module surface on top of it.
"""


def __getattr__(name):
    # (lazy: importing the package must not import torch-heavy modules or load the library)
    if name in ("graphed_step", "GraphedStep", "replay_matches_eager", "capture_step", "GraphReplayMismatch"):
        from . import graphs
        return getattr(graphs, name)
    raise AttributeError(f"module 'mrla_amd' has no attribute {name!r}")
