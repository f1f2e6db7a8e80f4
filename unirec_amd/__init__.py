"""unirec_amd -- MI355X-native (gfx950) implementation of UniRec's nested Q-Former +
Qwen3/LoRA hot path behind the reference's own Python class boundary.

Only what the path needs lives here: ``csrc/`` (HIP kernels + the C ABI of
``include/unirec_hip.h``), the ctypes binding (``_lib``/``hip``) and the host-side mirrors of the
reference classes.  There is no CPU fallback: importing the compute modules without the built
library raises.
"""
import os as _os

# Kernel arguments in device memory instead of host-coherent memory: every workgroup of a 6144-tile GEMM launch starts by
# fetching its 230-byte argument block (entry -> first DMA is 2200 cycles of a 52 k-cycle block).  The HIP runtime reads the
# variable when it initialises, i.e. at the first GPU call of the process: +0.6 % on the joint step in alternating same-box
# runs.  setdefault: an explicit user setting wins.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

__version__ = "0.1.0"
