"""CPU oracle for the UniRec hot path (TEST INFRASTRUCTURE ONLY).

This package is a from-scratch fp32 restatement of the reference algorithm for
the nested Q-Former + Qwen3/LoRA joint path.  It exists to *check* the HIP
product path and to provide the timed CPU baseline leg of ``bench.py``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it.  Nothing under ``unirec_amd/`` imports it; the product path
fails loudly when the HIP extension is missing instead of falling back here.

Pinning status:
  * Q-Former (item + user), heads, QFormerLoss, eval metrics, token injection,
    mean-pool, InfoNCE, MRR rank: pinned against the reference's own Python
    (imported from /root/reference in the build container through
    tests/golden/make_golden.py) -> tests/golden/*.npz.
  * Qwen3 decoder math: algorithm lives in third-party `transformers`
    (unpinned by the reference, README.md:60-63; installed here 5.15.0);
    pinned against the installed Qwen3Model, fixtures in tests/golden/.
  * LoRA: `peft` (unpinned, not installed) -> PARITY UNPINNED; restated from the
    published LoRA definition, self-consistency checked by weight merging.
"""
