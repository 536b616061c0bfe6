"""MI355X-native GMR-1 PHY receive path (FCCH sweep, pi/4-CxPSK demod, L1 Viterbi chain).

The directory name carries a hyphen, so it is imported by path:
``from __graft_entry__ import load_package; pkg = load_package()``.

Sub-modules
-----------
synth   seeded synthetic signal generator (host, numpy)
build   hipcc build of csrc/ into libgmr1_hip.so (gfx950)
api     ctypes mirror of the C ABI declared in include/ (the product entry points)
shard   partitioning of bursts / ARFCNs over ranks, IQ scatter and record gather (torch.distributed)
"""
from . import synth  # noqa: F401
from . import build  # noqa: F401
from . import api  # noqa: F401
from . import shard  # noqa: F401
