"""unirec_amd -- MI355X-native (gfx950) implementation of UniRec's nested Q-Former +
Qwen3/LoRA hot path behind the reference's own Python class boundary.

Only what the path needs lives here: ``csrc/`` (HIP kernels + the C ABI of
``include/unirec_hip.h``), the ctypes binding (``_lib``/``hip``) and the host-side mirrors of the
reference classes.  There is no CPU fallback: importing the compute modules without the built
library raises.
"""
__version__ = "0.1.0"
