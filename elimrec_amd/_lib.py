"""ctypes binding of libelimrec_hip.so (include/elimrec_hip.h).

This is the stub INTEGRATION.md shows a maintainer of the reference: every entry point of the
C ABI with its argument types, loaded once, failing LOUDLY when the HIP library is missing --
there is no CPU fallback anywhere in this package.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libelimrec_hip.so")

c_i32, c_i64, c_f32, c_u32, c_u64 = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_uint32, ctypes.c_uint64
c_ptr, c_size = ctypes.c_void_p, ctypes.c_size_t

class CsrSplit(ctypes.Structure):
    """struct elimrec_csr_split (include/elimrec_hip.h)."""
    _fields_ = [("long_threshold", ctypes.c_int32), ("n_long", ctypes.c_int32), ("n_seg", ctypes.c_int32),
                ("d_long_rows", ctypes.c_void_p), ("d_long_seg_ptr", ctypes.c_void_p),
                ("d_seg_bounds", ctypes.c_void_p), ("d_partials", ctypes.c_void_p), ("d_seg_row", ctypes.c_void_p),
                ("d_row_order", ctypes.c_void_p), ("d_tickets", ctypes.c_void_p), ("d_row_items", ctypes.c_void_p),
                ("n_row_items", ctypes.c_int32)]


class CsrDesc(ctypes.Structure):
    """struct elimrec_csr (include/elimrec_hip.h)."""
    _fields_ = [("n_rows", ctypes.c_int64), ("d_rowptr", ctypes.c_void_p), ("d_col", ctypes.c_void_p),
                ("d_val", ctypes.c_void_p), ("split", CsrSplit)]


class LinearDesc(ctypes.Structure):
    """struct elimrec_linear_desc."""
    _fields_ = [("d_A", ctypes.c_void_p), ("lda", ctypes.c_int64), ("d_W", ctypes.c_void_p), ("ldw", ctypes.c_int64),
                ("d_bias", ctypes.c_void_p), ("d_C", ctypes.c_void_p), ("ldc", ctypes.c_int64), ("M", ctypes.c_int64),
                ("N", ctypes.c_int32), ("K", ctypes.c_int32), ("d_rowscale", ctypes.c_void_p), ("d_add", ctypes.c_void_p),
                ("ldadd", ctypes.c_int64), ("d_row_index", ctypes.c_void_p), ("d_row_range", ctypes.c_void_p),
                ("act", ctypes.c_int32)]


class LinearBwdDesc(ctypes.Structure):
    """struct elimrec_linear_bwd_desc."""
    _fields_ = [("d_A", ctypes.c_void_p), ("lda", ctypes.c_int64), ("d_B", ctypes.c_void_p), ("ldb", ctypes.c_int64),
                ("d_row_index", ctypes.c_void_p), ("d_range", ctypes.c_void_p), ("R", ctypes.c_int64),
                ("n1", ctypes.c_int32), ("n2", ctypes.c_int32), ("d_out", ctypes.c_void_p), ("ldo", ctypes.c_int64),
                ("d_colsum", ctypes.c_void_p), ("accumulate", ctypes.c_int32), ("d_colsum_weight", ctypes.c_void_p)]


class SellDesc(ctypes.Structure):
    """struct elimrec_sell."""
    _fields_ = [("n_rows", ctypes.c_int64), ("n_src", ctypes.c_int64), ("n_items", ctypes.c_int32),
                ("n_seg_items", ctypes.c_int32), ("n_seg", ctypes.c_int32), ("n_long", ctypes.c_int32),
                ("d_item_dst", ctypes.c_void_p), ("d_item_len", ctypes.c_void_p), ("d_blk_off", ctypes.c_void_p),
                ("d_col", ctypes.c_void_p), ("d_val", ctypes.c_void_p), ("d_long_rows", ctypes.c_void_p),
                ("d_long_seg_ptr", ctypes.c_void_p), ("d_long_index", ctypes.c_void_p), ("d_rowptr", ctypes.c_void_p),
                ("d_csr_col", ctypes.c_void_p), ("d_csr_val", ctypes.c_void_p), ("d_item_long", ctypes.c_void_p),
                ("tiered", ctypes.c_int32), ("n_w1", ctypes.c_int32), ("n_w4", ctypes.c_int32),
                ("tile_groups", ctypes.c_int32), ("n_t4", ctypes.c_int32), ("n_t1", ctypes.c_int32), ("n_tseg", ctypes.c_int32),
                ("n_tfin", ctypes.c_int32), ("tile_kmax", ctypes.c_int32), ("d_tile_off", ctypes.c_void_p), ("d_tile_len", ctypes.c_void_p),
                ("d_tile_dst", ctypes.c_void_p), ("d_tile_long", ctypes.c_void_p), ("d_tile_col", ctypes.c_void_p),
                ("d_tile_val", ctypes.c_void_p)]


class AdamJob(ctypes.Structure):
    """struct elimrec_adam_job."""
    _fields_ = [("d_p_in", ctypes.c_void_p), ("d_p_out", ctypes.c_void_p), ("d_g", ctypes.c_void_p),
                ("d_m", ctypes.c_void_p), ("d_v", ctypes.c_void_p), ("d_copy_dst", ctypes.c_void_p), ("n", ctypes.c_int64),
                ("step", ctypes.c_int64)]


PROGRAM_MAX_ARGS = 32


class HeadRows(ctypes.Structure):
    """struct elimrec_head_rows."""
    _fields_ = [("A", ctypes.POINTER(SellDesc)), ("ns", ctypes.c_int32), ("w", ctypes.c_int32), ("L", ctypes.c_int32),
                ("U", ctypes.c_int64), ("layers", ctypes.c_void_p * 9), ("d_long", ctypes.c_void_p),
                ("d_narrow_out", ctypes.c_void_p), ("ld_narrow_out", ctypes.c_int64)]


class HeadSrc16(ctypes.Structure):
    """struct elimrec_head_src16."""
    _fields_ = [("d_table", ctypes.c_void_p), ("row_elems", ctypes.c_int64), ("dtype", ctypes.c_int32),
                ("d_S_out", ctypes.c_void_p), ("ld_S_out", ctypes.c_int64), ("d_c_out", ctypes.c_void_p)]


class ProgramOp(ctypes.Structure):
    """struct elimrec_op."""
    _fields_ = [("kind", ctypes.c_int32), ("fn", ctypes.c_int32), ("args", ctypes.c_uint64 * PROGRAM_MAX_ARGS)]


class ProgramPatch(ctypes.Structure):
    """struct elimrec_patch."""
    _fields_ = [("op", ctypes.c_int32), ("arg", ctypes.c_int32), ("value", ctypes.c_uint64)]


c_sell = ctypes.POINTER(SellDesc)
c_split = ctypes.POINTER(CsrSplit)
c_csr = ctypes.POINTER(CsrDesc)

# name -> (restype, argtypes); order and types follow include/elimrec_hip.h exactly.
SIGNATURES = {
    "elimrec_abi_version": (c_i32, []),
    "elimrec_last_error": (ctypes.c_char_p, []),
    "elimrec_linear_fwd": (c_i32, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_i64, c_i32, c_i32, c_ptr]),
    "elimrec_linear_fwd_batched": (c_i32, [ctypes.POINTER(LinearDesc), c_i32, c_ptr]),
    "elimrec_linear_bwd_w_batched_workspace": (c_size, [ctypes.POINTER(LinearBwdDesc), c_i32]),
    "elimrec_linear_bwd_w_batched": (c_i32, [ctypes.POINTER(LinearBwdDesc), c_i32, c_ptr, c_size, c_ptr]),
    "elimrec_linear_bwd_w_batched_merge": (c_i32, [ctypes.POINTER(LinearBwdDesc), c_i32, c_ptr, c_size, c_ptr, c_ptr, c_i32, c_i64,
                                                   c_i64, c_i64, c_i32, c_i32, c_i32, c_ptr, c_ptr, c_ptr, c_i32, c_ptr]),
    "elimrec_linear_bwd_w_reduce": (c_i32, [ctypes.POINTER(LinearBwdDesc), c_i32, c_ptr, c_size, c_ptr]),
    "elimrec_linear_bwd_w_workspace": (c_size, [c_i64, c_i32, c_i32]),
    "elimrec_linear_bwd_w": (c_i32, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_i32, c_i32, c_ptr, c_i64,
                                     c_ptr, c_i32, c_ptr, c_size, c_ptr]),
    "elimrec_assemble_x0": (c_i32, [c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i32, c_i32, c_ptr]),
    "elimrec_spmm_hop": (c_i32, [c_ptr, c_ptr, c_ptr, c_i64, c_i32, c_split, c_ptr, c_ptr, c_ptr, c_ptr, c_f32, c_ptr]),
    "elimrec_propagate": (c_i32, [c_ptr, c_ptr, c_ptr, c_i64, c_i32, c_split, c_i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "elimrec_bipartite_workspace": (c_size, [c_i64, c_i64, c_i32, c_i32]),
    "elimrec_propagate_bipartite": (c_i32, [c_csr, c_csr, c_i64, c_i64, c_i32, c_i32, c_i32, c_ptr, c_ptr, c_ptr, c_ptr,
                                            c_ptr, c_size, c_ptr]),
    "elimrec_propagate_bipartite_bwd": (c_i32, [c_csr, c_csr, c_i64, c_i64, c_i32, c_i32, c_i32, c_ptr, c_ptr, c_ptr,
                                                c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "elimrec_ticket_fixup": (c_i32, []),
    "elimrec_set_ticket_fixup": (None, [c_i32]),
    "elimrec_concurrency": (c_i32, []),
    "elimrec_set_concurrency": (None, [c_i32]),
    "elimrec_folded_workspace": (c_size, [c_i64, c_i32]),
    "elimrec_propagate_folded": (c_i32, [c_csr, c_i64, c_i64, c_i32, c_i32, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_size, c_ptr]),
    "elimrec_source_rows_split": (c_i32, [c_ptr, c_ptr, c_i64, c_i32, c_i32, c_i32, c_ptr, c_ptr]),
    "elimrec_propagate_folded_bwd": (c_i32, [c_csr, c_i64, c_i64, c_i32, c_i32, c_i32, c_ptr, c_ptr, c_ptr, c_i64, c_ptr,
                                             c_ptr, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "elimrec_block_spmm": (c_i32, [c_csr, c_i32, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_f32, c_ptr]),
    "elimrec_blocksum_rows": (c_i32, [c_ptr, c_ptr, c_ptr, c_i64, c_i32, c_i32, c_i32, c_ptr, c_ptr]),
    "elimrec_triplet_rows": (c_i32, [c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_i64, c_i32, c_ptr, c_i64, c_ptr]),
    "elimrec_triplet_rows_checked": (c_i32, [c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_ptr]),
    "elimrec_batch_plan": (c_i32, [c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_ptr, c_ptr,
                                   c_size, c_ptr]),
    "elimrec_gather_rows": (c_i32, [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_i32, c_ptr, c_i64, c_ptr]),
    "elimrec_copy_cols": (c_i32, [c_ptr, c_i64, c_ptr, c_i64, c_i64, c_i32, c_ptr]),
    "elimrec_bpr_head": (c_i32, [c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_i32, c_i32, c_i32,
                                 ctypes.POINTER(c_f32), c_ptr, c_ptr, c_ptr, c_ptr]),
    "elimrec_bpr_head_rows": (c_i32, [c_ptr, c_i64, c_ptr, c_i32, c_i32, c_i32, ctypes.POINTER(c_f32), c_ptr, c_ptr, c_ptr]),
    "elimrec_bpr_head_rows_sum": (c_i32, [c_ptr, c_i64, c_ptr, c_i32, c_i32, c_i32, ctypes.POINTER(c_f32), c_ptr, c_ptr, c_ptr, c_ptr,
                                          c_ptr]),
    "elimrec_bpr_head_rows_sum_pub": (c_i32, [c_ptr, c_i64, c_ptr, c_i32, c_i32, c_i32, ctypes.POINTER(c_f32), c_ptr, c_ptr, c_ptr, c_ptr,
                                              c_ptr, c_ptr]),
    "elimrec_loss_pub_create": (c_i32, [c_i32, ctypes.POINTER(c_ptr)]),
    "elimrec_loss_pub_destroy": (c_i32, [c_ptr]),
    "elimrec_loss_pub_issued": (c_u32, [c_ptr]),
    "elimrec_loss_pub_wait": (c_i32, [c_ptr, c_u32, ctypes.c_double, ctypes.POINTER(c_f32)]),
    "elimrec_sum": (c_i32, [c_ptr, c_i64, c_ptr, c_ptr]),
    "elimrec_segment_reduce_workspace": (c_size, [c_i64]),
    "elimrec_segment_plan_workspace": (c_size, [c_i64]),
    "elimrec_segment_plan": (c_i32, [c_ptr, c_i64, c_i32, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "elimrec_segment_apply_head_bwd": (c_i32, [c_ptr, c_i64, c_i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_size, c_i64, c_i32,
                                               c_i32, c_i32, ctypes.POINTER(c_i32), c_ptr, c_ptr, ctypes.POINTER(c_ptr), c_ptr,
                                               c_ptr]),
    "elimrec_segment_apply_head_bwd_packed": (c_i32, [c_ptr, c_i64, c_i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_size, c_i64, c_i32,
                                               c_i32, c_i32, ctypes.POINTER(c_i32), c_ptr, c_ptr, ctypes.POINTER(c_ptr), c_ptr,
                                               c_ptr, c_ptr]),
    "elimrec_segment_apply_head_bwd_sources": (c_i32, [c_ptr, c_i64, c_i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_size, c_i64, c_i32,
                                                c_i32, c_i32, ctypes.POINTER(c_i32), c_ptr, c_ptr, ctypes.POINTER(c_ptr), c_ptr,
                                                c_ptr, c_i64, c_i32, c_i32, c_ptr, c_ptr, c_ptr]),
    "elimrec_segment_apply_head_bwd_split": (c_i32, [c_ptr, c_i64, c_i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_size, c_i64, c_i32,
                                              c_i32, c_i32, ctypes.POINTER(c_i32), c_ptr, c_ptr, ctypes.POINTER(c_ptr), c_ptr,
                                              c_ptr, c_i64, c_i32, c_ptr, c_ptr]),
    "elimrec_segment_apply": (c_i32, [c_ptr, c_i64, c_i32, c_ptr, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "elimrec_segment_reduce_rows": (c_i32, [c_ptr, c_ptr, c_i64, c_i32, c_i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "elimrec_head_bwd_input": (c_i32, [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_i64, c_i32, c_i32, c_i32,
                                       ctypes.POINTER(c_i32), c_ptr, c_ptr, ctypes.POINTER(c_ptr), c_f32, c_ptr, c_i64,
                                       c_i32, c_ptr, c_ptr]),
    "elimrec_embed_grad": (c_i32, [c_ptr, c_i64, c_i64, c_i32, c_i32, c_ptr, c_ptr, c_ptr]),
    "elimrec_adam_step": (c_i32, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_f32, c_f32, c_f32, c_f32, c_f32, c_i64, c_ptr]),
    "elimrec_score_workspace": (c_size, [c_i32, c_i64, c_i32]),
    "elimrec_score_workspace2": (c_size, [c_i32, c_i64, c_i64, c_i32, c_i32]),
    "elimrec_score_workspace_topk": (c_size, [c_i32, c_i64, c_i64, c_i32, c_i32]),
    "elimrec_score_workspace_for": (c_size, [c_i32, c_i64, c_i64, c_i32, c_i32, c_i32, c_i32]),
    "elimrec_row_sqnorms": (c_i32, [c_ptr, c_i64, c_i64, c_i32, c_i32, c_ptr, c_ptr]),
    "elimrec_score_topk": (c_i32, [c_ptr, c_i64, c_i64, c_i64, c_ptr, c_i32, c_i32, c_i32, c_u32, c_i32, c_i32,
                                   c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i32, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "elimrec_score_topk_shard": (c_i32, [c_ptr, c_i64, c_i64, c_i64, c_ptr, c_i32, c_i32, c_i32, c_u32, c_i32, c_i32,
                                         c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i32, c_ptr, c_ptr, c_ptr, c_size, c_i32, c_ptr, c_i64,
                                         c_i64, c_ptr]),
    "elimrec_topk_merge": (c_i32, [c_ptr, c_ptr, c_i32, c_i32, c_i32, c_ptr, c_ptr, c_ptr]),
    "elimrec_rank_metrics": (c_i32, [c_ptr, c_i32, c_i32, c_ptr, c_ptr, ctypes.POINTER(c_i32), c_i32, c_ptr, c_ptr]),
    "elimrec_slab_partials_bytes": (c_size, [c_sell, c_i32, c_i32]),
    "elimrec_slab_hop": (c_i32, [c_sell, c_i32, c_i32, c_i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_f32, c_ptr, c_size, c_i32,
                                 c_ptr]),
    "elimrec_topk_reference_order": (c_i32, [c_ptr, c_i64, c_i64, c_i64, c_i32, c_ptr]),
    "elimrec_topk_reference_order_device": (c_i32, [c_ptr, c_i64, c_i64, c_i64, c_i32, c_ptr, c_ptr, c_ptr]),
    "elimrec_score_topk_ordered": (c_i32, [c_ptr, c_i64, c_i64, c_i64, c_ptr, c_i32, c_i32, c_i32, c_u32, c_i32, c_i32,
                                           c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i32, c_ptr, c_ptr, c_ptr, c_size, c_i32, c_ptr]),
    "elimrec_plan_workspace": (c_size, [c_i64, c_i64, c_i64]),
    "elimrec_plan_tile_count": (c_i64, [c_i64, c_i64, c_i64, c_i64, c_i32]),
    "elimrec_plan_rows": (c_i32, [c_ptr, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "elimrec_plan_tiles": (c_i32, [c_ptr, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_i64, c_i64,
                                   c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "elimrec_plan_scatter": (c_i32, [c_i64, c_i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "elimrec_slab_sweep_lds_rows": (c_size, [c_i32]),
    "elimrec_slab_sweep_hop": (c_i32, [c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32,
                                       c_ptr, c_ptr, c_ptr, c_ptr, c_f32, c_ptr]),
    "elimrec_slab_sweep_hop_adam": (c_i32, [c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32,
                                            c_ptr, c_ptr, c_ptr, c_ptr, c_f32, c_ptr, c_ptr, c_ptr, c_ptr, c_f32, c_f32, c_f32, c_f32, c_f32,
                                            c_i64, c_ptr]),
    "elimrec_slab_hop_bwd_w": (c_i32, [c_sell, c_i32, c_i32, c_i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_f32, c_ptr, c_size, c_i32,
                                       ctypes.POINTER(LinearBwdDesc), c_i32, c_ptr, c_size, c_i32, c_ptr]),
    "elimrec_slab_source_bits": (c_i32, [c_sell, c_i32, c_i32, c_i32, c_ptr, c_ptr, c_size, c_ptr]),
    "elimrec_slab_hop_adam": (c_i32, [c_sell, c_i32, c_i32, c_i32, c_ptr, c_ptr, c_ptr, c_ptr, c_f32, c_ptr, c_size, c_ptr, c_ptr,
                                      c_ptr, c_ptr, c_f32, c_f32, c_f32, c_f32, c_f32, c_i64, ctypes.POINTER(AdamJob), c_i32, c_ptr, c_i64,
                                      c_ptr, c_ptr]),
    "elimrec_slab_rows": (c_i32, [c_sell, c_i32, c_i32, c_i32, c_i64, ctypes.POINTER(c_ptr), c_ptr, c_ptr, c_ptr, c_i64, c_i32,
                                  c_ptr, c_i64, c_ptr, c_i64, c_i32, c_ptr]),
    "elimrec_slab_from_rows": (c_i32, [c_ptr, c_i64, c_i64, c_i64, c_i32, c_i32, c_ptr, c_ptr]),
    "elimrec_slab_to_rows": (c_i32, [c_ptr, c_i64, c_i32, c_i32, c_ptr, c_i64, c_i64, c_ptr]),
    "elimrec_slab_merge_rows": (c_i32, [c_ptr, c_ptr, c_i32, c_i64, c_i64, c_i64, c_i32, c_i32, c_i32, c_ptr, c_ptr, c_ptr, c_ptr]),
    "elimrec_adam_step_out": (c_i32, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_f32, c_f32, c_f32, c_f32, c_f32, c_i64,
                                      c_ptr]),
    "elimrec_head_pack_floats": (c_size, [c_i32, ctypes.POINTER(c_i32)]),
    "elimrec_head_pack_bwd_offset": (c_size, [c_i32, ctypes.POINTER(c_i32)]),
    "elimrec_head_fwd_fused": (c_i32, [c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i32, ctypes.POINTER(c_ptr),
                                       ctypes.POINTER(c_i64), ctypes.POINTER(c_i32), ctypes.POINTER(c_ptr), ctypes.POINTER(c_ptr),
                                       c_ptr, c_ptr, c_ptr, c_ptr, ctypes.POINTER(c_ptr), ctypes.POINTER(c_ptr), c_ptr, c_size,
                                       c_ptr, c_i64, c_ptr, c_i64, c_i32, c_i32, c_ptr]),
    "elimrec_head_fwd_fused_peers": (c_i32, [c_ptr, c_ptr, c_i64, c_ptr, c_i32, c_i64, c_ptr, c_i32, ctypes.POINTER(c_ptr),
                                             ctypes.POINTER(c_i64), ctypes.POINTER(c_i32), ctypes.POINTER(c_ptr), ctypes.POINTER(c_ptr),
                                             c_ptr, c_ptr, c_ptr, c_ptr, ctypes.POINTER(c_ptr), ctypes.POINTER(c_ptr), c_ptr, c_size,
                                             c_ptr, c_i64, c_ptr, c_i64, c_i32, c_i32, c_ptr]),
    "elimrec_head_fwd_fused_rows": (c_i32, [ctypes.POINTER(HeadRows), c_ptr, c_ptr, c_i64, c_ptr, c_i32, ctypes.POINTER(c_ptr),
                                            ctypes.POINTER(c_i64), ctypes.POINTER(c_i32), ctypes.POINTER(c_ptr), ctypes.POINTER(c_ptr),
                                            c_ptr, c_ptr, c_ptr, c_ptr, ctypes.POINTER(c_ptr), ctypes.POINTER(c_ptr), c_ptr, c_size,
                                            c_ptr, c_i64, c_ptr, c_i64, c_i32, c_ptr]),
    "elimrec_head_fwd_fused_src16": (c_i32, [ctypes.POINTER(HeadSrc16), c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i32,
                                             ctypes.POINTER(c_i32), ctypes.POINTER(c_ptr), ctypes.POINTER(c_ptr), c_ptr, c_ptr, c_ptr, c_ptr,
                                             ctypes.POINTER(c_ptr), ctypes.POINTER(c_ptr), c_ptr, c_size, c_ptr, c_i64, c_ptr, c_i64,
                                             c_i32, c_i32, c_ptr]),
    "elimrec_score_range_violations": (c_i32, [ctypes.POINTER(c_i64), c_i32, c_ptr]),
    "elimrec_score_range_check": (c_i32, [c_ptr, c_ptr, c_i32, c_i32, c_i32, c_i32, c_ptr, c_ptr]),
    "elimrec_score_set_math": (None, [c_i32]),
    "elimrec_score_get_math": (c_i32, []),
    "elimrec_score_set_bf16x3": (None, [c_i32]),
    "elimrec_score_get_bf16x3": (c_i32, []),
    "elimrec_build_adj_workspace": (c_size, [c_i64, c_i64, c_i32]),
    "elimrec_build_adj": (c_i32, [c_ptr, c_ptr, c_i64, c_i64, c_i64, c_i32, c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_size,
                                  c_ptr]),
    "elimrec_adam_multi": (c_i32, [ctypes.POINTER(AdamJob), c_i32, c_f32, c_f32, c_f32, c_f32, c_f32, c_ptr]),
    "elimrec_lookup_counts": (c_i32, [c_ptr, c_i32, c_i64, c_i64, c_i64, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64), c_ptr, c_ptr]),
    "elimrec_lookup_pack": (c_i32, [c_ptr, c_i32, c_i64, c_i64, c_i64, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64), c_i32, c_ptr, c_i64,
                                    c_ptr, c_ptr, c_ptr]),
    "elimrec_lookup_unpack": (c_i32, [c_ptr, c_i32, c_i64, c_i64, c_i64, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64), c_i32, c_ptr, c_i64,
                                      c_i32, c_i32, c_i32, c_ptr, c_i64, c_ptr, c_ptr]),
    "elimrec_wide_from_master": (c_i32, [c_ptr, c_i64, c_i64, c_i32, c_i32, c_ptr, c_ptr]),
    "elimrec_wide_rows": (c_i32, [ctypes.POINTER(c_ptr), c_i32, c_i64, c_i32, c_i32, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr]),
    "elimrec_wide_grad": (c_i32, [c_ptr, c_i64, c_i64, c_i32, c_i32, c_f32, c_ptr, c_ptr]),
    "elimrec_rows_bitmap": (c_i32, [c_ptr, c_i32, c_i64, c_i64, c_ptr, c_ptr]),
    "elimrec_peer_cols_to_rows": (c_i32, [c_ptr, c_i32, c_i64, c_i32, c_ptr, c_i64, c_ptr, c_i64, c_ptr]),
    "elimrec_program_fn_count": (c_i32, []),
    "elimrec_program_fn_name": (ctypes.c_char_p, [c_i32]),
    "elimrec_program_fn_args": (c_i32, [c_i32]),
    "elimrec_program_create": (c_i32, [ctypes.POINTER(ProgramOp), c_i32, ctypes.POINTER(c_ptr)]),
    "elimrec_program_create_scoped": (c_i32, [ctypes.POINTER(ProgramOp), c_i32, c_i32, ctypes.POINTER(c_ptr)]),
    "elimrec_program_run": (c_i32, [c_ptr, ctypes.POINTER(ProgramPatch), c_i32]),
    "elimrec_program_destroy": (c_i32, [c_ptr]),
    "elimrec_comm_unique_id": (c_i32, [c_ptr]),
    "elimrec_comm_create": (c_i32, [c_ptr, c_i32, c_i32, ctypes.POINTER(c_ptr)]),
    "elimrec_comm_nranks": (c_i32, [c_ptr, ctypes.POINTER(c_i32)]),
    "elimrec_comm_destroy": (c_i32, [c_ptr]),
    "elimrec_comm_all_gather": (c_i32, [c_ptr, c_ptr, c_ptr, c_i64, c_ptr]),
    "elimrec_comm_all_reduce_f32": (c_i32, [c_ptr, c_ptr, c_i64, c_ptr]),
    "elimrec_comm_all_to_all": (c_i32, [c_ptr, c_ptr, c_ptr, c_i64, c_ptr]),
    "elimrec_comm_all_to_all_v": (c_i32, [c_ptr, c_ptr, c_ptr, ctypes.POINTER(c_i64), c_ptr]),
    "elimrec_sample_triplets": (c_i32, [c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_u64, c_u64, c_ptr, c_ptr, c_ptr, c_ptr]),
}

_lib = None


class HipLibraryError(RuntimeError):
    pass


def load():
    """dlopen the HIP library and bind every symbol. Raises HipLibraryError if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            "elimrec_amd: %s is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C elimrec_amd/csrc`). There is no CPU fallback for the hot path." % LIB_PATH)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise HipLibraryError("elimrec_amd: cannot load %s: %s" % (LIB_PATH, e))
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise HipLibraryError("elimrec_amd: %s does not export %s (stale build?)" % (LIB_PATH, name))
        fn.restype = res
        fn.argtypes = args
    if lib.elimrec_abi_version() != 1:
        raise HipLibraryError("elimrec_amd: ABI version mismatch")
    _lib = _Recording(lib)
    return _lib


class _Recording(object):
    """The bound library with an optional call recorder: while `record(list)` is active every entry point that
    returns an error code appends (function, converted arguments) to the list before it runs. A recorded list is a
    fixed sequence of launches on fixed buffers; `replay` re-issues it without the Python that built the
    arguments (elimrec_amd.model uses it for the step's regions -- the host side of a 25-launch step otherwise
    costs as much as the GPU side)."""

    def __init__(self, lib):
        self._cdll = lib
        self._rec = None
        self._trace = None          # program.py: every call of a traced step, in issue order, with stream hand-overs in between
        for name, (res, _) in SIGNATURES.items():
            fn = getattr(lib, name)
            plain = name in ("elimrec_abi_version", "elimrec_program_fn_count", "elimrec_program_fn_args", "elimrec_comm_unique_id",
                             "elimrec_score_get_math", "elimrec_score_get_bf16x3",
                             "elimrec_comm_create", "elimrec_comm_destroy", "elimrec_comm_nranks") or name.startswith("elimrec_program_")
            setattr(self, name, self._wrap(fn, name) if res is c_i32 and not plain else fn)

    def _wrap(self, fn, name):
        def call(*args):
            if self._rec is not None:
                self._rec.append((fn, args))
            if self._trace is not None:
                self._trace.append(("call", name, fn, args))
            return fn(*args)
        return call


def record(calls):
    """Start (calls = list) or stop (calls = None) recording the C-ABI calls issued through load()."""
    load()._rec = calls


def replay(calls, what):
    for fn, args in calls:
        rc = fn(*args)
        if rc:
            check(rc, what)


def check(rc, what):
    if rc != 0:
        msg = load().elimrec_last_error()
        raise RuntimeError("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else "?"))
