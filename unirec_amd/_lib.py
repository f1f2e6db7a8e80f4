"""ctypes binding of libunirec_hip.so (the C ABI declared in include/unirec_hip.h).

Fails loudly: there is no CPU / eager fallback for the product path.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# UNIREC_HIP_LIB selects another build of the SAME library (kernel A/B experiments); there is still no fallback.
LIB_PATH = os.environ.get("UNIREC_HIP_LIB") or os.path.join(_HERE, "lib", "libunirec_hip.so")
ABI_VERSION = 12

c_void_p, c_int, c_i64, c_u64, c_float = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint64, ctypes.c_float


class GemmArgs(ctypes.Structure):
    """Mirror of ur_gemm_args (include/unirec_hip.h)."""
    _fields_ = [("R", c_void_p), ("ldr", c_i64), ("r_kcontig", c_int),
                ("S", c_void_p), ("lds", c_i64), ("s_kcontig", c_int),
                ("K", c_int),
                ("R2", c_void_p), ("ldr2", c_i64), ("S2", c_void_p), ("lds2", c_i64), ("K2", c_int),
                ("C", c_void_p), ("ldc", c_i64), ("c_f32", c_int),
                ("M", c_int), ("N", c_int),
                ("alpha", c_float),
                ("bias", c_void_p),
                ("residual", c_void_p), ("ldres", c_i64),
                ("gelu_out", c_void_p), ("ldg", c_i64),
                ("gelu_grad_aux", c_void_p), ("ldaux", c_i64),
                ("split_k", c_int),
                ("drop_bits", c_void_p), ("drop_bits_ld", c_i64), ("drop_bits_stride", c_i64),
                ("drop_rank", c_int), ("drop_p", c_float),
                ("swiglu_gu", c_void_p), ("swiglu_ldgu", c_i64), ("swiglu_dgu", c_void_p), ("swiglu_lddgu", c_i64),
                ("swiglu_I", c_int),
                ("swiglu_gate", c_void_p), ("swiglu_ldgate", c_i64), ("swiglu_act", c_void_p), ("swiglu_ldact", c_i64),
                ("qkr_q", c_void_p), ("qkr_ldq", c_i64), ("qkr_k", c_void_p), ("qkr_ldk", c_i64), ("qkr_v", c_void_p), ("qkr_ldv", c_i64),
                ("qkr_rstd", c_void_p),
                ("qkr_qw", c_void_p), ("qkr_kw", c_void_p), ("qkr_cos", c_void_p), ("qkr_sin", c_void_p),
                ("qkr_S", c_int), ("qkr_nq_cols", c_int), ("qkr_nk_cols", c_int), ("qkr_eps", c_float),
                ("swp_act", c_void_p), ("swp_ldact", c_i64), ("swp_I", c_int)]


class LoraArgs(ctypes.Structure):
    """Mirror of ur_lora_args."""
    _fields_ = [("X", c_void_p), ("ldx", c_i64), ("M", c_int),
                ("nad", c_int), ("rank", c_int), ("shared", c_int),
                ("col0", c_int * 4), ("width", c_int * 4),
                ("drop_bits", c_void_p), ("bits_ld", c_i64), ("bits_stride", c_i64),
                ("alpha", c_float),
                ("U", c_void_p * 4), ("ldu", c_i64 * 4),
                ("P", c_void_p), ("ldp", c_i64),
                ("V", c_void_p), ("ldv", c_i64),
                ("G", c_void_p), ("g_transposed", c_int),
                ("drop_bits_t", c_void_p), ("bits_t_ld", c_i64), ("bits_t_stride", c_i64)]


class AttnArgs(ctypes.Structure):
    """Mirror of ur_attn_args."""
    _fields_ = [("q", c_void_p), ("k", c_void_p), ("v", c_void_p), ("o", c_void_p), ("stats", c_void_p),
                ("ldq", c_i64), ("ldk", c_i64), ("ldv", c_i64), ("ldo", c_i64),
                ("key_mask", c_void_p),
                ("B", c_int), ("Sq", c_int), ("Sk", c_int), ("nq", c_int), ("nkv", c_int), ("head_dim", c_int),
                ("causal", c_int), ("scale", c_float), ("dropout_p", c_float), ("seed", c_u64), ("drop_batch0", c_i64)]


class AttnBwdArgs(ctypes.Structure):
    """Mirror of ur_attn_bwd_args."""
    _fields_ = [("dout", c_void_p), ("dq", c_void_p), ("dk", c_void_p), ("dv", c_void_p),
                ("lddo", c_i64), ("lddq", c_i64), ("lddk", c_i64), ("lddv", c_i64), ("delta", c_void_p),
                ("rope_q_raw", c_void_p), ("rope_ldraw", c_i64), ("rope_q_weight", c_void_p), ("rope_cos", c_void_p),
                ("rope_sin", c_void_p), ("rope_eps", c_float), ("rope_dq_raw", c_void_p), ("rope_lddraw", c_i64),
                ("rope_rstd", c_void_p), ("rope_rstd_ld", c_i64), ("rope_rstd_h0", c_int),
                ("rope_k", c_void_p), ("rope_ldk", c_i64), ("rope_k_weight", c_void_p), ("rope_rstd_hk0", c_int),
                ("rope_dk_raw", c_void_p), ("rope_lddkraw", c_i64),
                ("kv_colsum", c_void_p), ("kv_colsum_ws", c_void_p)]


# name -> (restype, argtypes).  Every symbol include/unirec_hip.h declares must appear here
# (tests/test_cabi.py cross-checks the header against this table and the built library).
SIGNATURES = {
    "ur_version": (c_int, []),
    "ur_last_error": (ctypes.c_char_p, []),
    "ur_comm_unique_id": (c_int, [c_void_p]),
    "ur_comm_init": (c_int, [ctypes.POINTER(c_void_p), c_int, c_int, c_void_p, c_int]),
    "ur_comm_allreduce_async": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_void_p]),
    "ur_comm_ticket": (c_i64, [c_void_p]),
    "ur_comm_wait_ticket": (c_int, [c_void_p, c_i64, c_void_p]),
    "ur_comm_wait": (c_int, [c_void_p, c_void_p]),
    "ur_comm_destroy": (c_int, [c_void_p]),
    "ur_gemm_workspace_bytes": (c_i64, [ctypes.POINTER(GemmArgs)]),
    "ur_gemm": (c_int, [ctypes.POINTER(GemmArgs), c_void_p, c_i64, c_void_p]),
    "ur_gemm_grouped_workspace_bytes": (c_i64, [ctypes.POINTER(GemmArgs), c_int]),
    "ur_gemm_grouped": (c_int, [ctypes.POINTER(GemmArgs), c_int, c_void_p, c_i64, c_void_p]),
    "ur_gemm_persistent_mode": (c_int, [c_int]),
    "ur_attn_mode": (c_int, [c_int, c_int]),
    "ur_gemm_qkrope_supported": (c_int, [ctypes.POINTER(GemmArgs)]),
    "ur_gemm_swiglu_paired_supported": (c_int, [ctypes.POINTER(GemmArgs)]),
    "ur_qkrope_perm": (c_int, [c_int]),
    "ur_lora_bits_ld": (c_i64, [c_int]),
    "ur_lora_dropout_bits": (c_int, [c_u64, c_float, c_int, c_int, c_int, c_void_p, c_i64, c_i64, c_i64, c_void_p]),
    "ur_lora_bits_t_ld": (c_i64, [c_int]),
    "ur_lora_bits_transpose": (c_int, [c_void_p, c_i64, c_i64, c_int, c_int, c_int, c_void_p, c_i64, c_i64, c_void_p]),
    "ur_lora_project": (c_int, [ctypes.POINTER(LoraArgs), c_void_p]),
    "ur_swiglu_lora_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, ctypes.POINTER(LoraArgs), c_void_p]),
    "ur_rmsnorm_lora_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, ctypes.POINTER(LoraArgs), c_void_p]),
    "ur_lora_reduce_workspace_bytes": (c_i64, [ctypes.POINTER(LoraArgs)]),
    "ur_lora_reduce": (c_int, [ctypes.POINTER(LoraArgs), c_void_p, c_i64, c_void_p]),
    "ur_lora_bgrad_workspace_bytes": (c_i64, [ctypes.POINTER(LoraArgs)]),
    "ur_lora_bgrad": (c_int, [ctypes.POINTER(LoraArgs), c_void_p, c_i64, c_void_p]),
    "ur_layernorm_fwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_int, c_int, c_float, c_float, c_u64, c_float, c_u64, c_i64, c_void_p]),
    "ur_layernorm_bwd_workspace_bytes": (c_i64, [c_int]),
    "ur_layernorm_bwd_reduce": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ur_layernorm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_int, c_int, c_float, c_u64, c_float, c_u64, c_i64, c_void_p, c_i64, c_void_p]),
    "ur_batch_reduce": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_i64, c_void_p]),
    "ur_batch_reduce_workspace_bytes": (c_i64, [c_int, c_int, c_int]),
    "ur_rmsnorm_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "ur_rmsnorm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "ur_attn_fwd": (c_int, [ctypes.POINTER(AttnArgs), c_void_p]),
    "ur_attn_bwd": (c_int, [ctypes.POINTER(AttnArgs), ctypes.POINTER(AttnBwdArgs), c_void_p]),
    "ur_attn_bwd_workspace_floats": (c_i64, [c_int, c_int, c_int]),
    "ur_attn_bwd_kv_colsum_floats": (c_i64, [ctypes.POINTER(AttnArgs)]),
    "ur_dropout_keep": (c_int, [c_u64, c_float, c_u64, c_i64, c_void_p, c_void_p]),
    "ur_attn_dropout_keep": (c_int, [c_u64, c_float, c_u64, c_i64, c_int, c_void_p, c_void_p]),
    "ur_rope_table": (c_int, [c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "ur_qknorm_rope_fwd": (c_int, [c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_int,
                                   c_int, c_int, c_int, c_float, c_void_p]),
    "ur_qknorm_rope_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64,
                                   c_i64, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "ur_qknorm_rope_bwd_roped": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_i64, c_i64, c_int, c_int, c_int, c_int, c_void_p]),
    "ur_qknorm_rope_bwd_roped_k": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_i64,
                                           c_i64, c_int, c_int, c_int, c_void_p]),
    "ur_embed_inject_fwd": (c_int, [c_void_p, c_i64, c_void_p, c_void_p, c_i64, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ur_inject_bwd": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ur_user_sequence_assemble": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_u64,
                                          c_i64, c_void_p]),
    "ur_mean_pool_workspace_bytes": (c_i64, [c_int, c_int]),
    "ur_mean_pool_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_i64, c_void_p]),
    "ur_mean_pool_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ur_cosine_scores": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ur_infonce_workspace_bytes": (c_i64, [c_int, c_int, c_int]),
    "ur_infonce_fwd_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_void_p,
                                   c_void_p, c_int, c_int, c_int, c_void_p, c_i64, c_void_p]),
    "ur_gather_rows": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_i64, c_i64, c_i64, c_void_p]),
    "ur_catalog_scores": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_i64, c_int, c_void_p]),
    "ur_rank_of_index": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_i64, c_void_p]),
    "ur_context_mlp1": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_void_p]),
    "ur_mrr_rank": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "ur_topk": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_i64, c_void_p]),
    "ur_heads_workspace_bytes": (c_i64, [c_int, c_int]),
    "ur_field_projection_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "ur_field_projection_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                        c_void_p, c_i64, c_void_p]),
    "ur_recon_stats": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_void_p, c_i64, c_void_p]),
    "ur_recon_grad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_i64, c_int, c_void_p]),
    "ur_triplet_margin": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_float, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                  c_i64, c_void_p]),
    "ur_mse_loss": (c_int, [c_void_p, c_void_p, c_i64, c_float, c_void_p, c_void_p, c_void_p, c_i64, c_void_p]),
    "ur_cast_f32_to_bf16": (c_int, [c_void_p, c_void_p, c_i64, c_void_p]),
    "ur_cast_bf16_to_f32": (c_int, [c_void_p, c_void_p, c_i64, c_void_p]),
    "ur_transpose_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "ur_transpose_bf16_batched": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "ur_add_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_void_p]),
    "ur_gelu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_void_p]),
    "ur_swiglu_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "ur_swiglu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "ur_adamw_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_float, c_float, c_float, c_float, c_float,
                              c_int, c_float, c_void_p]),
}

_lib = None


class UniRecHipError(RuntimeError):
    pass


def load():
    """Load the shared library (once) and bind every declared symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise UniRecHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C unirec_amd/csrc`).  The UniRec MI355X path has no CPU fallback.")
    # torch first: its bundled HIP runtime must be the process's only one.  The library's libamdhip64 dependency then resolves
    # to the already-loaded copy; loaded the other way round (build() before anything imported torch) the process ends up with
    # two runtimes and the library's launches fail with "no ROCm-capable device is detected".
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here == header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    v = lib.ur_version()
    if v != ABI_VERSION:
        raise UniRecHipError(f"libunirec_hip.so ABI version {v} != expected {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().ur_last_error().decode("utf-8", "replace")
        raise UniRecHipError(f"{what} failed (rc={rc}): {msg}")
