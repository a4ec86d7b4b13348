"""ctypes binding of libidgrec.so (C ABI: include/idgrec.h).

Nothing here computes: every function marshals pointers and sizes and raises RuntimeError
with idg_last_error() when the library reports a failure.  There is deliberately no
fallback: if the shared library is missing the import of this module fails, and device
entry points fail on a machine without a gfx950 GPU.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# IDG_LIB_PATH: another build of the same ABI (A/B timing of two library versions inside one GPU session)
LIB_PATH = os.environ.get("IDG_LIB_PATH") or os.path.join(_HERE, "lib", "libidgrec.so")

c_i64p = C.POINTER(C.c_int64)
c_i32p = C.POINTER(C.c_int32)
c_u32p = C.POINTER(C.c_uint32)
c_f32p = C.POINTER(C.c_float)
c_vp = C.c_void_p

# name -> (restype, argtypes); kept in one table so tests can check it against the header
PROTOTYPES = {
    "idg_version": (C.c_int, []),
    "idg_last_error": (C.c_char_p, []),
    "idg_device_count": (C.c_int, []),
    "idg_rng_create": (C.c_int, [C.c_uint32, C.POINTER(c_vp)]),
    "idg_rng_destroy": (C.c_int, [c_vp]),
    "idg_rng_get_state": (C.c_int, [c_vp, c_u32p, C.POINTER(C.c_int32)]),
    "idg_rng_set_state": (C.c_int, [c_vp, c_u32p, C.c_int32]),
    "idg_rng_bytes": (C.c_int, [c_vp, C.c_int64, c_vp]),
    "idg_rng_randint": (C.c_int, [c_vp, C.c_int64, C.c_int64, c_i64p]),
    "idg_sample_epoch": (C.c_int, [c_vp, c_i64p, c_i64p, C.c_int64, c_i64p, c_i32p, C.c_int64, C.c_int64,
                                   c_i64p, c_i64p]),
    "idg_shuffle_perm": (C.c_int, [c_vp, C.c_int64, c_i64p]),
    "idg_py_random_sample": (C.c_int, [c_vp, C.c_int64, C.c_int64, C.c_int, c_i64p]),
    "idg_ratings_open": (C.c_int, [C.c_char_p, C.POINTER(c_vp), c_i64p, c_i64p, c_i64p, c_i64p]),
    "idg_ratings_read": (C.c_int, [c_vp, c_i64p, c_i64p, c_i64p, c_i64p]),
    "idg_ratings_destroy": (C.c_int, [c_vp]),
    "idg_build_norm_adj": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, c_i64p, c_i64p, C.c_int,
                                     C.POINTER(C.c_double), c_i64p, c_i64p, c_i32p, c_f32p]),
    "idg_graph_create": (C.c_int, [C.c_int, C.c_int64, C.c_int64, C.c_int64, c_i64p, c_i32p, c_f32p, C.c_uint32,
                                   C.c_int64, C.POINTER(c_vp)]),
    "idg_graph_create_from_device": (C.c_int, [C.c_int, C.c_int64, C.c_int64, C.c_int64, c_vp, c_vp, c_vp, C.c_uint32,
                                               C.c_int64, c_vp, C.POINTER(c_vp)]),
    "idg_graph_destroy": (C.c_int, [c_vp]),
    "idg_graph_live_units_bytes": (C.c_size_t, [c_vp, C.c_int64]),
    "idg_graph_live_units": (C.c_int, [c_vp, c_vp, c_vp, C.c_int64, c_vp]),
    "idg_graph_bind_live_units": (C.c_int, [c_vp, c_vp, c_vp, C.c_int64]),
    "idg_graph_forget_live_units": (C.c_int, [c_vp, c_vp]),
    "idg_graph_live_units_check": (C.c_int, [c_vp, c_vp]),
    "idg_graph_revalued_copy": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, C.POINTER(c_vp)]),
    "idg_subgraph_values_f32": (C.c_int, [C.c_int64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "idg_graph_remask": (C.c_int, [c_vp, c_vp, C.c_float, C.c_float, C.c_uint64, C.c_uint64, C.c_int, c_vp]),
    "idg_graph_masked_copy": (C.c_int, [c_vp, C.c_float, C.c_float, C.c_uint64, C.c_uint64, C.c_int, c_vp, C.POINTER(c_vp)]),
    "idg_graph_info": (C.c_int, [c_vp, c_i64p]),
    "idg_graph_forget_units_ws": (C.c_int, [c_vp]),
    "idg_graph_compact_inputs_bytes": (C.c_size_t, [c_vp]),
    "idg_graph_compact_inputs": (C.c_int, [c_vp, c_vp, c_vp, c_vp]),
    "idg_graph_long_rows": (C.c_int, [c_vp, c_i64p, c_i64p, c_i64p]),
    "idg_spmm_workspace_bytes": (C.c_size_t, [c_vp, C.c_int64]),
    "idg_spmm_f32": (C.c_int, [c_vp, c_vp, C.c_int64, c_vp, C.c_int64, c_vp, C.c_int64, c_vp, c_vp]),
    "idg_spmm_ex_f32": (C.c_int, [c_vp, c_vp, C.c_int64, c_vp, c_vp, c_vp, c_vp, C.c_int64, C.c_float, C.c_int,
                                  c_vp, c_vp, C.c_int64, c_vp, c_vp]),
    "idg_spmm_epi_f32": (C.c_int, [c_vp, c_vp, C.c_int64, C.c_int64, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "idg_rows_gather2_f32": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, C.c_int64, C.c_int64, c_vp]),
    "idg_rows_scatter_f32": (C.c_int, [c_vp, c_vp, c_vp, C.c_int64, C.c_int64, c_vp]),
    "idg_rows_chain_store2_f32": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, C.c_int64, C.c_int64, c_vp]),
    "idg_rows_layer_mean_f32": (C.c_int, [c_vp, c_vp, C.c_int64, c_vp, c_vp, c_vp, c_vp, C.c_float, C.c_int64, c_vp]),
    "idg_rows_layer_mean_n_f32": (C.c_int, [c_vp, c_vp, C.c_int64, c_vp, C.c_int, c_vp, C.c_float, C.c_int64, c_vp]),
    "idg_flags_compact_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "idg_flags_compact_f32": (C.c_int, [c_vp, C.c_int64, c_vp, C.c_int64, c_vp, c_vp, c_vp]),
    "idg_grad_tail_adam_f32": (C.c_int, [c_vp, c_vp, c_vp, c_vp, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_float, C.c_int,
                                         c_vp, c_vp, c_vp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int64, c_vp]),
    "idg_shard_prepare": (C.c_int, [c_vp]),
    "idg_lincomb_f32": (C.c_int, [c_vp, c_vp, C.c_float, c_vp, C.c_float, C.c_int64, c_vp]),
    "idg_rows_gather_f32": (C.c_int, [c_vp, c_vp, c_vp, C.c_int64, C.c_int64, c_vp]),
    "idg_rows_chain_add_f32": (C.c_int, [c_vp, c_vp, c_vp, c_vp, C.c_int64, C.c_int64, c_vp]),
    "idg_spmm_noise_f32": (C.c_int, [c_vp, c_vp, C.c_int64, c_vp, C.c_int64, c_vp, C.c_int64, C.c_float, C.c_uint64, C.c_uint64,
                                     c_vp, c_vp]),
    "idg_perturb_f32": (C.c_int, [c_vp, c_vp, C.c_int64, C.c_int64, c_vp, C.c_float, C.c_uint64, C.c_uint64, c_vp]),
    "idg_propagate_views_workspace_bytes": (C.c_size_t, [c_vp, C.c_int64, C.c_int]),
    "idg_propagate_views_f32": (C.c_int, [c_vp, c_vp, C.c_int, C.c_int64, C.c_float, C.c_int, C.POINTER(C.c_uint64),
                                          C.POINTER(C.c_uint64), c_vp, C.POINTER(C.c_void_p), c_vp, c_vp, c_vp]),
    "idg_propagate_workspace_bytes": (C.c_size_t, [c_vp, C.c_int64]),
    "idg_propagate_mean_f32": (C.c_int, [c_vp, c_vp, c_vp, c_vp, C.c_int, C.c_int, C.c_int64, c_vp, c_vp]),
    "idg_propagate_mean_noise_f32": (C.c_int, [c_vp, c_vp, c_vp, c_vp, C.c_int, C.c_int, C.c_int64, C.c_float, C.c_uint64,
                                               C.c_uint64, c_vp, c_vp]),
    "idg_propagate_mean_bwd_f32": (C.c_int, [c_vp, c_vp, c_vp, c_vp, C.c_int, C.c_int, C.c_int64, C.c_int, c_vp,
                                             c_vp]),
    "idg_propagate_mean_bwd_adam_f32": (C.c_int, [c_vp, c_vp, c_vp, c_vp, C.c_int, C.c_int, C.c_int64, C.c_int, c_vp, c_vp,
                                                  c_vp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int64,
                                                  c_vp, c_vp]),
    "idg_propagate_mean_bwd_adam_fields_f32": (C.c_int, [c_vp, c_vp, c_vp, c_vp, C.c_int, C.c_int, C.c_int64, C.c_int, c_vp,
                                                         c_vp, c_vp, C.c_double, C.c_double, C.c_double, C.c_double,
                                                         C.c_int64, c_vp, c_vp]),
    "idg_propagate_mean_fields_f32": (C.c_int, [c_vp, c_vp, c_vp, c_vp, C.c_int, C.c_int, C.c_int64, c_vp, c_vp]),
    "idg_graph_expand_rows": (C.c_int, [c_vp, c_vp, c_vp, c_vp]),
    "idg_graph_flag_cols": (C.c_int, [c_vp, c_vp, c_vp, c_vp]),
    "idg_graph_mark_cols": (C.c_int, [c_vp, c_vp, c_vp, c_vp]),
    "idg_linear_wgrad_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int64]),
    "idg_linear_wgrad_f32": (C.c_int, [c_vp, C.c_int64, c_vp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, c_vp, C.c_int, c_vp,
                                       c_vp]),
    "idg_ngcf_transform_f32": (C.c_int, [c_vp, c_vp, c_vp, c_vp, C.c_int64, C.c_int64, C.c_int64, c_vp, c_vp, c_vp]),
    "idg_ngcf_transform_bwd_f32": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, C.c_int64, C.c_int64, C.c_int64, c_vp, c_vp, c_vp]),
    "idg_ngcf_tail_f32": (C.c_int, [c_vp, c_vp, c_vp, c_vp, C.c_int64, C.c_int64, C.c_float, C.c_float, C.c_uint64, C.c_uint64,
                                    c_vp, c_vp, c_vp]),
    "idg_ngcf_tail_bwd_f32": (C.c_int, [c_vp, c_vp, c_vp, C.c_int64, C.c_int64, C.c_float, C.c_float, C.c_uint64, C.c_uint64,
                                        c_vp, c_vp]),
    "idg_infonce_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int64]),
    "idg_infonce_cross_f32": (C.c_int, [c_vp, C.c_int64, C.c_int64, c_vp, c_vp, C.c_int64, C.c_int64, C.c_float, c_vp, c_vp,
                                        C.c_float, c_vp, c_vp]),
    "idg_rows_tanh_bwd_f32": (C.c_int, [c_vp, c_vp, c_vp, C.c_int64, C.c_int64, c_vp, c_vp]),
    "idg_infonce_plan": (C.c_int, [c_vp, c_vp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, c_vp, c_vp]),
    "idg_infonce_cross_ex_f32": (C.c_int, [c_vp, C.c_int64, C.c_int64, c_vp, c_vp, C.c_int64, C.c_int64, C.c_float, c_vp, c_vp,
                                           C.c_float, C.c_int, c_vp, c_vp]),
    "idg_infonce_pair_f32": (C.c_int, [c_vp, c_vp, C.c_int64, C.c_int64, c_vp, c_vp, C.c_int64, C.c_int64, C.c_int, C.c_float,
                                       c_vp, c_vp, c_vp, C.c_float, C.c_int, c_vp, c_vp]),
    "idg_ngcf_wgrad_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64]),
    "idg_ngcf_wgrad_f32": (C.c_int, [c_vp, c_vp, c_vp, C.c_int64, C.c_int64, C.c_int64, c_vp, c_vp, c_vp]),
    "idg_ngcf_tail_ex_f32": (C.c_int, [c_vp, c_vp, c_vp, c_vp, C.c_int64, C.c_int64, C.c_float, C.c_float, C.c_uint64, C.c_uint64,
                                       c_vp, c_vp, C.c_int64, c_vp]),
    "idg_ngcf_tail_bwd_ex_f32": (C.c_int, [c_vp, c_vp, c_vp, C.c_int64, c_vp, C.c_int64, C.c_int64, C.c_float, C.c_float,
                                           C.c_uint64, C.c_uint64, c_vp, c_vp]),
    "idg_ngcf_layer_fwd_f32": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, C.c_int64, C.c_int64, C.c_float, C.c_float, C.c_uint64,
                                         C.c_uint64, c_vp, c_vp, C.c_int64, c_vp]),
    "idg_ngcf_layer_bwd_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "idg_ngcf_layer_bwd_f32": (C.c_int, [c_vp, c_vp, c_vp, C.c_int64, c_vp, c_vp, c_vp, c_vp, c_vp, C.c_int64, C.c_int64, C.c_float,
                                         C.c_float, C.c_uint64, C.c_uint64, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "idg_colsum_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "idg_colsum_f32": (C.c_int, [c_vp, C.c_int64, C.c_int64, C.c_int64, c_vp, C.c_int, c_vp, c_vp]),
    "idg_copy_cols_f32": (C.c_int, [c_vp, C.c_int64, c_vp, C.c_int64, C.c_int64, C.c_int64, c_vp]),
    "idg_rows_add2_f32": (C.c_int, [c_vp, C.c_int64, c_vp, C.c_int64, c_vp, C.c_int64, c_vp, C.c_int64, C.c_int64, c_vp]),
    "idg_bpr_fused_ex_f32": (C.c_int, [c_vp, C.c_int64, c_vp, C.c_int64, C.c_int64, C.c_int64, c_vp, c_vp, c_vp, C.c_int64,
                                       C.c_float, C.c_int, c_vp, c_vp, c_vp, C.c_int, c_vp, c_vp, c_vp]),
    "idg_bpr_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64]),
    "idg_bpr_fused_f32": (C.c_int, [c_vp, c_vp, C.c_int64, C.c_int64, c_vp, c_vp, c_vp, C.c_int64, C.c_int64,
                                    C.c_float, c_vp, c_vp, c_vp, C.c_int, c_vp, c_vp, c_vp]),
    "idg_bpr_touch_rows": (C.c_int, [c_vp, c_vp, c_vp, C.c_int64, C.c_int64, c_vp, c_vp]),
    "idg_bitmap_clear": (C.c_int, [c_vp, C.c_int64, c_vp]),
    "idg_event_create": (C.c_int, [C.POINTER(c_vp)]),
    "idg_event_destroy": (C.c_int, [c_vp]),
    "idg_event_record": (C.c_int, [c_vp, c_vp]),
    "idg_stream_wait_event": (C.c_int, [c_vp, c_vp]),
    "idg_event_query": (C.c_int, [c_vp, C.POINTER(C.c_int)]),
    "idg_event_synchronize": (C.c_int, [c_vp]),
    "idg_comm_load": (C.c_int, [C.c_char_p]),
    "idg_comm_rccl_version": (C.c_int, [C.POINTER(C.c_int)]),
    "idg_comm_unique_id": (C.c_int, [c_vp]),
    "idg_comm_create": (C.c_int, [C.c_int, C.c_int, c_vp, C.c_int, C.POINTER(c_vp)]),
    "idg_comm_destroy": (C.c_int, [c_vp]),
    "idg_allreduce_f32": (C.c_int, [c_vp, c_vp, C.c_int64, C.c_int, c_vp]),
    "idg_allgather_f32": (C.c_int, [c_vp, c_vp, c_vp, C.c_int64, c_vp]),
    "idg_reduce_scatter_f32": (C.c_int, [c_vp, c_vp, c_vp, C.c_int64, c_vp]),
    "idg_bpr_rows_message_floats": (C.c_size_t, [C.c_int64, C.c_int64]),
    "idg_bpr_pack_rows_f32": (C.c_int, [c_vp, C.c_int64, C.c_int64, c_vp, c_vp, c_vp, c_vp, C.c_int64, c_vp]),
    "idg_bpr_unpack_rows_f32": (C.c_int, [c_vp, C.c_int, C.c_int64, C.c_int64, C.c_int64, c_vp, C.c_float, c_vp, c_vp,
                                          c_vp, C.c_int, c_vp, c_vp]),
    "idg_bpr_plan_f32": (C.c_int, [c_vp, c_vp, c_vp, C.c_int64, C.c_int64, C.c_int64, c_vp, c_vp]),
    "idg_bpr_plan_rows_f32": (C.c_int, [c_vp, c_vp, c_vp, C.c_int64, C.c_int64, C.c_int64, c_vp, c_vp, c_vp]),
    "idg_bpr_forward_f32": (C.c_int, [c_vp, c_vp, C.c_int64, C.c_int64, c_vp, c_vp, c_vp, C.c_int64, C.c_int64,
                                      C.c_float, c_vp, c_vp, c_vp]),
    "idg_bpr_backward_f32": (C.c_int, [c_vp, c_vp, C.c_int64, C.c_int64, c_vp, c_vp, c_vp, C.c_int64, C.c_int64,
                                       C.c_float, c_vp, c_vp, c_vp, C.c_int, c_vp, c_vp, c_vp]),
    "idg_adam_step_f32": (C.c_int, [c_vp, c_vp, c_vp, c_vp, C.c_int64, C.c_double, C.c_double, C.c_double,
                                    C.c_double, C.c_int64, c_vp]),
    "idg_score_dense_f32": (C.c_int, [c_vp, c_vp, c_vp, C.c_int64, C.c_int64, C.c_int64, C.c_int, c_vp, c_vp]),
    "idg_score_topk_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int64, C.c_int]),
    "idg_score_topk_info": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_int, c_vp, C.POINTER(C.c_int64), c_vp]),
    "idg_pack24_f32": (C.c_int, [c_vp, c_vp, C.c_int64, c_vp]),
    "idg_unpack24_f32": (C.c_int, [c_vp, c_vp, C.c_int64, c_vp]),
    "idg_reduce24_f32": (C.c_int, [c_vp, C.c_int, C.c_int64, c_vp, c_vp, c_vp]),
    "idg_alltoall_f32": (C.c_int, [c_vp, c_vp, c_vp, C.c_int64, C.c_int, c_vp]),
    "idg_reduce_blocks_f32": (C.c_int, [c_vp, C.c_int, C.c_int64, c_vp, c_vp]),
    "idg_score_topk_option": (C.c_int, [C.c_int, C.c_int64, C.POINTER(C.c_int64)]),
    "idg_score_topk_candidate_counts": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_int, c_vp, c_vp, c_vp]),
    "idg_score_topk_f32": (C.c_int, [c_vp, c_vp, c_vp, C.c_int64, C.c_int64, C.c_int64, c_vp, c_vp, C.c_int,
                                     C.c_int, c_vp, c_vp, c_vp, c_vp]),
    "idg_step_create": (C.c_int, [c_vp, C.POINTER(c_vp)]),
    "idg_step_destroy": (C.c_int, [c_vp]),
    "idg_step_prefetch": (C.c_int, [c_vp, c_vp, c_vp, c_vp, C.c_int64, C.c_uint64, c_vp]),
    "idg_step_run_f32": (C.c_int, [c_vp, c_vp, c_vp, c_vp, C.c_int64, c_vp, c_vp, c_vp, C.c_int64, C.c_uint64, C.c_uint64, c_vp, C.c_int64,
                                   C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, c_vp]),
    "idg_step_last_bitmap": (C.c_int, [c_vp, C.POINTER(c_vp)]),
    "idg_step_synchronize": (C.c_int, [c_vp]),
    "idg_step_stats": (C.c_int, [c_vp, C.POINTER(C.c_int64)]),
    "idg_adam_rows_f32": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_double,
                                    C.c_double, C.c_int64, c_vp]),
}

IDG_ADAM_DISCARD_GRAD = 2  # OR-ed into `accumulate` of idg_propagate_mean_bwd_adam(_fields)_f32: do not write the gradient back
IDG_SSL_PLANNED = 2  # OR-ed into idg_infonce_pair_f32's `dedup`: the id lists are in the workspace already (idg_infonce_plan)
IDG_BPR_TOUCHED_PRESET = 4  # OR-ed into `deterministic`: the touched bitmap already holds the batch's rows
IDG_BPR_PLANNED = 2  # `deterministic` value: the sorted scatter plan is already in the workspace (idg_bpr_plan_f32)
IDG_GRAPH_SYMMETRIC = 1
IDG_GRAPH_EXACT_ORDER = 2


IDG_STEP_SLOTS, IDG_STEP_STORE_GRAD, IDG_STEP_PACED = 3, 1, 2
# idg_score_topk_option: which -> (index, default)
IDG_TOPK_OPTS = {"form": (0, -1), "collect": (1, 1), "floor": (2, 1), "wgs": (3, 0), "chunks": (4, 0), "fallback_permille": (5, 20)}
IDG_TOPK_OPT_RESET, IDG_TOPK_OPT_KEEP = -1, -(1 << 63)


class StepDesc(C.Structure):
    """idg_step_desc (include/idgrec.h): every buffer of the one-call training step."""
    _fields_ = [("graph", c_vp), ("num_users", C.c_int64), ("n", C.c_int64), ("d", C.c_int64), ("n_layers", C.c_int),
                ("include_layer0", C.c_int), ("reg_lambda", C.c_float), ("params", c_vp), ("grad", c_vp), ("final_panel", c_vp),
                ("g_final", c_vp), ("exp_avg", c_vp), ("exp_avg_sq", c_vp), ("prop_ws", c_vp), ("batch_capacity", C.c_int64),
                ("slot_bitmap", c_vp * 3), ("slot_units", c_vp * 3), ("slot_bpr_ws", c_vp * 3), ("side_stream", c_vp),
                ("flags", C.c_int)]


class ShardPrep(C.Structure):
    """idg_shard_prep (include/idgrec.h): one global batch's index-only preparation for the sharded step."""
    _fields_ = [("own_users", c_vp), ("n_own", C.c_int64), ("pos", c_vp), ("neg", c_vp), ("guest_ids", c_vp),
                ("B", C.c_int64), ("B_cap", C.c_int64), ("users_bits", c_vp), ("n_local_users", C.c_int64),
                ("items_bits", c_vp), ("n_items_padded", C.c_int64), ("scatter_bits", c_vp), ("n_panel_rows", C.c_int64),
                ("user_graph", c_vp), ("user_units", c_vp), ("n_slices", C.c_int), ("slice_graphs", C.POINTER(c_vp)),
                ("slice_row0", C.POINTER(C.c_int64)), ("slice_units", C.POINTER(c_vp)), ("plan_ws", c_vp),
                ("main_stream", c_vp), ("side_stream", c_vp), ("ev_fork", c_vp), ("ev_rows", c_vp), ("ev_plan", c_vp)]


class Epilogue(C.Structure):
    """idg_epilogue (include/idgrec.h): every epilogue option of idg_spmm_epi_f32."""
    _fields_ = [("Y", c_vp), ("addend", c_vp), ("sum_in", c_vp), ("sum_in2", c_vp), ("sum_in3", c_vp), ("sum_out", c_vp),
                ("ldy", C.c_int64), ("div", C.c_float), ("accumulate", C.c_int), ("mask", c_vp),
                ("adam_param", c_vp), ("adam_exp_avg", c_vp), ("adam_exp_avg_sq", c_vp),
                ("adam_lr", C.c_double), ("adam_beta1", C.c_double), ("adam_beta2", C.c_double), ("adam_eps", C.c_double),
                ("adam_step", C.c_int64), ("adam_discard_grad", C.c_int),
                ("act", C.c_int), ("act_src", c_vp), ("act_rows", C.c_int64), ("y24", c_vp)]

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "libidgrec.so is not built (%s). Run `python id-grec_amd/build.py` (needs hipcc); "
        "there is no pure-Python or CPU substitute for it." % LIB_PATH)

# libidgrec.so is handed torch's streams and device pointers, so both must sit on ONE HIP
# runtime: torch bundles its own libamdhip64 (same SONAME as /opt/rocm's) and whichever copy
# is mapped first wins.  Import torch first so the library binds to torch's runtime.
try:
    import torch as _torch  # noqa: F401
except ImportError:  # host-only use (sampler / parser / adjacency) works without torch
    _torch = None

ACT_TANH, ACT_TANH_BWD = 1, 2  # idg_epilogue.act
ABI_VERSION = 140  # include/idgrec.h IDG_VERSION the prototype table above was written against

lib = C.CDLL(LIB_PATH)
lib.idg_version.restype = C.c_int
if lib.idg_version() != ABI_VERSION:
    raise ImportError("%s reports ABI version %d, this binding expects %d: the library is stale — rebuild it with "
                      "`python id-grec_amd/build.py`." % (LIB_PATH, lib.idg_version(), ABI_VERSION))
for _name, (_res, _args) in PROTOTYPES.items():
    if os.environ.get("IDG_LIB_PATH") and not hasattr(lib, _name):
        continue  # an A/B build of an earlier revision of the same ABI: entry points added since are simply absent
    _fn = getattr(lib, _name)
    _fn.restype = _res
    _fn.argtypes = _args


class IdgError(RuntimeError):
    def __init__(self, code, where):
        msg = lib.idg_last_error()
        msg = msg.decode("utf-8", "replace") if msg else ""
        super().__init__("%s failed (%d): %s" % (where, code, msg))
        self.code = code


def check(code, where):
    if code != 0:
        raise IdgError(code, where)


def device_count():
    return int(lib.idg_device_count())


def np_ptr(arr, ctype):
    """Pointer to a C-contiguous numpy array's buffer (the caller keeps `arr` alive)."""
    return arr.ctypes.data_as(C.POINTER(ctype))
