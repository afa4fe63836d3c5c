"""Kernel -> family table of libreidgan_hip.so, one row per __global__ kernel of reid-gan_amd/csrc/*.hip.

`FAMILY[kernel]` is the accounting family used by the profile tools (tools/pmc_traffic.py, tools/prof_summary.py): the family
of the rg::ProfScope the kernel is launched under (rg_common.h `enum Family`), except that the kernels which exist only to
finish or feed a convolution launch — split-K finishers, the folded-BatchNorm weight-gradient finisher, filter re-layouts /
folds, fp8 quantisation — are booked on the convolution family they serve, because `roofline.traffic` is "HBM bytes the conv
family moves per launch, everything it needs included".  tests/test_kernel_families.py fails when a kernel of the sources is
missing here, so a new kernel cannot silently land in "other" (round 2: conv3x3_halo / bn_fold_wgrad / *_k1 / smallc did).
"""

CONV = "conv"            # fp32 implicit-GEMM family (FAM_CONV_FWD / _DGRAD / _WGRAD + its helpers)
CONV_F8 = "conv_f8"

FAMILY = {
    # conv_igemm.hip
    "conv_fwd_kernel": CONV, "conv_dgrad_kernel": CONV, "conv_wgrad_kernel": CONV, "conv3x3_halo_kernel": CONV,
    "conv1x1_dma_kernel": CONV, "wgrad1x1_dma_kernel": CONV, "splitk_reduce_fold_kernel": CONV, "conv_splitk_finish_strided_kernel": CONV,
    "conv_fwd_pl_kernel": CONV, "conv_dgrad_pl_kernel": CONV, "conv_wgrad_pl_kernel": CONV,      # conv_planes.h
    "conv_dgrad_smallc_kernel": CONV, "conv_dgrad_smallc_px_kernel": CONV, "conv_fwd_k1_kernel": CONV,
    "conv_wgrad_k1_kernel": CONV, "conv_splitk_finish_kernel": CONV, "splitk_reduce_kernel": CONV,
    "conv_splitk_finish_vec_kernel": CONV, "splitk_reduce_vec_kernel": CONV, "conv_wgrad_k1_px4_kernel": CONV,
    "weights_to_krsc_kernel": CONV, "weights_to_krsc_multi_kernel": CONV, "weights_to_ck_kernel": CONV,
    # norm.hip: conv helpers of the folded (frozen-statistics) BatchNorm
    "bn_fold_wgrad_kernel": CONV, "fold_filters_multi_kernel": CONV, "bn_fold_kernel": CONV,
    # conv_f8.hip
    "conv_f8_kernel": CONV_F8, "f8_amax_kernel": CONV_F8, "f8_quantize_dual_kernel": CONV_F8,
    "f8_quantize_transpose_kernel": CONV_F8, "f8_roll_kernel": CONV_F8, "f8_splitk_reduce_kernel": CONV_F8, "f8_splitk_reduce_vec_kernel": CONV_F8,
    # norm.hip
    "act_bwd_sum_kernel": "norm", "bn_apply_fwd_kernel": "norm", "bn_bwd_apply_kernel": "norm",
    "bn_bwd_reduce_finalize_kernel": "norm", "bn_bwd_reduce_partial_kernel": "norm", "bn_eval_bwd_fused_kernel": "norm",
    "bn_stats_finalize_kernel": "norm", "bn_stats_partial_kernel": "norm", "bn_train_bwd_fused_kernel": "norm",
    "bn_train_fwd_fused_kernel": "norm", "instnorm_bwd_kernel": "norm", "instnorm_fwd_kernel": "norm",
    "bn_train_fwd_reg_kernel": "norm", "channel_sum_small_kernel": "norm", "bn_train_bwd_reg_kernel": "norm", "instnorm_fwd_reg_kernel": "norm", "instnorm_bwd_reg_kernel": "norm",
    "rows_sum_pair_kernel": "norm", "scale_rows_kernel": "norm", "sum_slices_kernel": "norm",
    "bn_stats_from_partials_kernel": "norm", "bn_bwd_from_partials_kernel": "norm",
    # optim.hip
    "adam_advance_kernel": "optim", "adam_dev_kernel": "optim", "adam_kernel": "optim", "sgd_kernel": "optim",
    "u64_add_kernel": "optim",
    # pool.hip
    "gap_bwd_kernel": "pool", "gap_fwd_kernel": "pool", "gem_bwd_kernel": "pool", "gem_fwd_kernel": "pool",
    "maxpool_bwd_3x3s2_kernel": "pool", "maxpool_bwd_kernel": "pool", "maxpool_fwd_kernel": "pool", "sum_all_kernel": "pool",
    # eltwise.hip
    "act_bwd_kernel": "eltwise", "act_fwd_kernel": "eltwise", "axpby_kernel": "eltwise", "copy_channels_kernel": "eltwise",
    "dropout_kernel": "eltwise", "fill_kernel": "eltwise", "spin_kernel": "eltwise", "l2norm_rows_bwd_kernel": "eltwise",
    "l2norm_rows_fwd_kernel": "eltwise", "l2norm_channels_fwd_kernel": "eltwise", "l2norm_channels_bwd_kernel": "eltwise", "mix_rows_bwd_kernel": "eltwise", "mix_rows_fwd_kernel": "eltwise",
    "pair_cat_kernel": "eltwise", "sub_square_bwd_kernel": "eltwise", "sub_square_fwd_kernel": "eltwise",
    # loss.hip
    "affine_relu_mean_bwd_kernel": "loss", "affine_relu_mean_partial_kernel": "loss", "bce_bwd_kernel": "loss",
    "bce_partial_kernel": "loss", "finalize_sum_kernel": "loss", "grad_penalty_rows_kernel": "loss", "l1_bwd_kernel": "loss", "l1_rows_fwd_kernel": "loss", "mse_const_rows_fwd_kernel": "loss", "mse_const_rows_bwd_kernel": "loss", "l1_rows_bwd_kernel": "loss",
    "l1_finalize_kernel": "loss", "l1_partial_kernel": "loss", "mse_const_bwd_kernel": "loss",
    "mse_const_partial_kernel": "loss", "softmax_ce_bwd_kernel": "loss", "softmax_ce_fwd_kernel": "loss",
    "wsum_bwd_kernel": "loss", "wsum_kernel": "loss",
    # cm.hip
    "cm_update_hard_kernel": "cm", "cm_update_kernel": "cm", "normalize_listed_rows_kernel": "cm",
    # bgemm.hip (attention), gan_extra.hip, resize.hip, retrieval.hip, datagen.hip: rg::FAM_MISC
    "bgemm_kernel": "misc", "softmax_rows_bwd_kernel": "misc", "softmax_rows_fwd_kernel": "misc",
    "attn_fwd_kernel": "misc", "attn_bwd_kernel": "misc",
    "avgpool_bwd_kernel": "misc", "avgpool_fwd_kernel": "misc", "reflect_pad_bwd_kernel": "misc",
    "reflect_pad_fwd_kernel": "misc", "reflect_pad_fwd_vec_kernel": "misc", "reflect_pad_bwd_vec_kernel": "misc", "scale_by_device_scalar_kernel": "misc", "spectral_norm_bwd_apply_kernel": "misc",
    "spectral_norm_bwd_kernel": "misc", "spectral_norm_dot_partial_kernel": "misc", "spectral_norm_power_kernel": "misc",
    "spectral_norm_power_multi_kernel": "misc", "spectral_norm_scale_multi_kernel": "misc",
    "bicubic_norm_bwd_kernel": "misc", "bicubic_norm_fwd_kernel": "misc",
    "add_outer_terms_kernel": "misc", "row_sqsum_kernel": "misc", "segment_mean_kernel": "misc", "topk_rows_kernel": "misc",
    "erase_rects_kernel": "misc", "flip_pad_crop_kernel": "misc", "pose_maps_kernel": "misc",
}


def base_name(kernel):
    """'conv_fwd_kernel<128, 128, 2, 2, 1, true>' / 'void (anonymous namespace)::conv_fwd_kernel<...>(ConvP)' -> 'conv_fwd_kernel'"""
    k = kernel.strip().strip('"')
    if k.startswith("void "):
        k = k[5:]
    k = k.replace("(anonymous namespace)::", "")
    for sep in ("<", "("):
        i = k.find(sep)
        if i >= 0:
            k = k[:i]
    return k.strip()


def family_of(kernel):
    """accounting family of a kernel name as rocprofv3 prints it; runtime / ATen kernels are 'other'"""
    return FAMILY.get(base_name(kernel), "other")


def source_kernels(csrc_dir):
    """every __global__ kernel name defined in csrc/*.hip and the kernel headers they include (csrc/*.h)"""
    import glob
    import os
    import re
    names = set()
    for f in sorted(glob.glob(os.path.join(csrc_dir, "*.hip")) + glob.glob(os.path.join(csrc_dir, "*.h"))):
        t = open(f).read()
        for m in re.finditer(r"__global__", t):
            mm = re.search(r"\bvoid\s+(\w+)\s*\(", t[m.end():m.end() + 400])
            if mm:
                names.add(mm.group(1))
    return names
