"""mrla_amd -- MI355X-native MRLA (multi-head recurrent layer attention) hot path.

The compute path is libmrla_hip.so (hand-written HIP for gfx950, C ABI in include/mrla_hip.h); this
package is the host-side mirror of the reference's nn.Module surface on top of it.
"""
