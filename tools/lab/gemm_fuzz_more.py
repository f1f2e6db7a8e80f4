import sys, os
sys.path.insert(0, os.getcwd())
import tests.test_gpu_gemm_persistent as T
import random
kinds = ["plain", "bias", "residual", "bias_residual", "lora", "lora_residual", "drop", "gelu_out", "gelu_grad", "swiglu_bwd", "swiglu_bwd_drop", "drop3"]
# re-run the seeded test body with other seeds by monkeypatching the kind selection: seed s -> kind s % 12, rng seeded by s
import types
src = T.test_persistent_equals_generic_on_random_configurations
bad = 0
for s in range(12, 72):
    # emulate: the test indexes kinds[seed]; wrap
    class L(list):
        def __getitem__(self, i): return list.__getitem__(self, i % 12)
    g = src.__globals__
    code = src.__code__
    try:
        # call original function with seed % 12 for the kind but a different rng: patch random.Random to offset
        real = random.Random
        class R(real):
            def __init__(self, x=None): real.__init__(self, (x or 0) + 7919 * s)
        random.Random = R
        src(s % 12)
    except AssertionError as e:
        bad += 1; print("FAIL seed", s, e)
    finally:
        random.Random = real
print("done, failures:", bad)
