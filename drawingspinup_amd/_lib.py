"""ctypes binding of libdsu_hip.so (C ABI in include/dsu_hip.h).

There is no CPU fallback: if the shared library is missing, or a kernel is asked to run
on a non-GPU tensor, the call raises.  `python -m drawingspinup_amd.build` (or
`__graft_entry__.build()`) produces the library in-tree.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# DSU_HIP_LIB=<path>: load a variant built by `python -m drawingspinup_amd.build --variant ...`
# (A/B measurements); unset = the in-tree library.
LIB_PATH = os.environ.get("DSU_HIP_LIB") or os.path.join(_HERE, "libdsu_hip.so")
MAX_LEVELS = 16

c_i32, c_i64, c_u32, c_f32, c_vp = C.c_int32, C.c_int64, C.c_uint32, C.c_float, C.c_void_p


class HashGridCfg(C.Structure):
    _fields_ = [("n_levels", c_u32), ("n_features", c_u32), ("log2_hashmap_size", c_u32),
                ("base_resolution", c_u32), ("per_level_scale", C.c_double)]


class HashGridLevels(C.Structure):
    _fields_ = [("offsets", c_u32 * (MAX_LEVELS + 1)), ("resolution", c_u32 * MAX_LEVELS),
                ("scale", c_f32 * MAX_LEVELS), ("hashed", c_u32 * MAX_LEVELS)]


class SdfMlp(C.Structure):
    _fields_ = [("w0", c_vp), ("b0", c_vp), ("w1", c_vp), ("b1", c_vp)]


class TexMlp(C.Structure):
    _fields_ = [("w0", c_vp), ("b0", c_vp), ("w1", c_vp), ("b1", c_vp), ("w2", c_vp),
                ("b2", c_vp)]


class PartialReduce(C.Structure):                     # dsu_partial_reduce
    _fields_ = [("partials", c_vp), ("map", c_vp), ("base", c_vp), ("nblocks", c_i32),
                ("stride", c_i32), ("n", c_i32)]


class OccRefreshArgs(C.Structure):                    # dsu_occgrid_refresh_args
    _fields_ = [("occs", c_vp), ("binary", c_vp), ("res", c_i32), ("all_cells", c_i32),
                ("aabb", C.POINTER(c_f32)), ("seed", C.c_uint64), ("step", c_i64),
                ("grid", C.POINTER(HashGridCfg)), ("table_img", c_vp), ("mlp", C.POINTER(SdfMlp)),
                ("inv_s", c_vp), ("radius", c_f32), ("render_step_size", c_f32),
                ("ema_decay", c_f32), ("occ_thre", c_f32), ("active_levels", c_u32),
                ("inj_count", c_i32), ("inj_cells", c_vp), ("inj_rand", c_vp), ("thre_out", c_vp),
                ("workspace", c_vp), ("workspace_bytes", c_i64), ("cells_out", c_vp),
                ("rand_out", c_vp)]


class NormCfg(C.Structure):
    _fields_ = [("batch", c_i32), ("channels", c_i32), ("hw", c_i32), ("instance", c_i32),
                ("act", c_i32), ("stat_updates", c_i32), ("eps", c_f32), ("momentum", c_f32)]


class RayLossCfg(C.Structure):
    _fields_ = [("rgb_p_ratio", C.c_double), ("normal_p_ratio", C.c_double),
                ("mask_p_ratio", C.c_double), ("lambda_rgb_mse", c_f32),
                ("lambda_rgb_l1", c_f32), ("lambda_normal", c_f32), ("lambda_mask", c_f32),
                ("geo_aware", c_i32), ("reserved", c_i32)]


# name -> argtypes; every function returns int.  Kept in one table so that the CPU test
# "the library exports every symbol the header declares" can walk it.
P = c_vp
class AdamwTensor(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p),
                ("n", C.c_int64), ("lr", C.c_float), ("bias_correction1", C.c_float),
                ("bias_correction2_sqrt", C.c_float), ("reserved", C.c_float)]


ADAMW_MAX_TENSORS = 24


class NsrDriverCfg(C.Structure):                     # dsu_nsr_driver_cfg
    _fields_ = [("grid", HashGridCfg), ("radius", c_f32), ("render_step_size", c_f32),
                ("cap_points", c_i32), ("cap_rays", c_i32), ("n_random", c_i32),
                ("sort_bits", c_i32), ("dynamic_ray_sampling", c_i32),
                ("train_num_samples", c_i32),
                ("c2w", c_vp), ("origins", c_vp), ("directions", c_vp), ("images", c_vp),
                ("normals", c_vp), ("masks", c_vp), ("view_weights", c_vp),
                ("V", c_i32), ("H", c_i32), ("W", c_i32), ("image_channels", c_i32),
                ("w0_v", c_vp), ("w0_g", c_vp), ("b0", c_vp), ("w1_v", c_vp), ("w1_g", c_vp),
                ("b1", c_vp), ("tex", c_vp * 6), ("variance", c_vp),
                ("ray_loss", RayLossCfg),
                ("lambda_eikonal", c_f32), ("lambda_sparsity", c_f32), ("sparsity_scale", c_f32),
                ("lambda_smooth", c_f32),
                ("beta1", c_f32), ("beta2", c_f32), ("adam_eps", c_f32), ("weight_decay", c_f32),
                ("seed", C.c_uint64), ("workspace", c_vp), ("workspace_bytes", c_i64)]


class NsrStepArgs(C.Structure):                      # dsu_nsr_step_args
    _fields_ = [("step", c_i64), ("n_rays", c_i32), ("prefetch_next", c_i32),
                ("active_levels", c_u32), ("eps", c_f32), ("cos_anneal_ratio", c_f32),
                ("lr_geometry", c_f32), ("lr_texture", c_f32), ("lr_variance", c_f32),
                ("adam_step", c_i32), ("randomized", c_i32), ("refresh_effective", c_i32),
                ("occ_res", c_i32), ("occ_binary", c_vp), ("table_img", c_vp),
                ("table_grad", c_vp),
                ("inj_index", c_vp), ("inj_x", c_vp), ("inj_y", c_vp), ("inj_jitter", c_vp),
                ("inj_pts_random", c_vp), ("inj_perturb", c_vp),
                ("inj_rays", c_vp), ("inj_rgb", c_vp), ("inj_normal", c_vp), ("inj_mask", c_vp),
                ("inj_cosines", c_vp), ("inj_view_weights", c_vp),
                ("table_p", c_vp), ("table_m", c_vp), ("table_v", c_vp), ("table_n", c_i64),
                ("table_lr", c_f32), ("table_bc1", c_f32), ("table_bc2_sqrt", c_f32),
                ("table_eps", c_f32), ("table_wd", c_f32),
                ("out_n_samples", c_i32), ("out_max_count", c_i32), ("out_next_n_rays", c_i32),
                ("terms_out", c_vp)]


_PROTOS = {
    "dsu_abi_version": [],
    "dsu_hashgrid_make_levels": [C.POINTER(HashGridCfg), C.POINTER(HashGridLevels)],
    "dsu_hashgrid_encode_fwd": [C.POINTER(HashGridCfg), P, P, c_i64, c_u32, P, P],
    "dsu_hashgrid_encode_bwd": [C.POINTER(HashGridCfg), P, P, c_i64, c_u32, P, P],
    "dsu_sdf_fwd": [C.POINTER(HashGridCfg), P, C.POINTER(SdfMlp), P, c_i64, c_f32, c_u32, c_u32,
                    P, P],
    "dsu_sdf_fd_fwd": [C.POINTER(HashGridCfg), P, C.POINTER(SdfMlp), P, c_i64, c_f32, c_f32,
                       c_u32, P, P, P, P, P],
    "dsu_sdf_fd_bwd": [C.POINTER(HashGridCfg), P, C.POINTER(SdfMlp), P, c_i64, c_f32, c_f32,
                       c_u32, P, P, P, P, P, P, P, P, P, P, c_i64, P],
    "dsu_sdf_fd_bwd_workspace_bytes": [C.POINTER(HashGridCfg), c_i64],
    "dsu_sdf_fd_fwd_cached": [C.POINTER(HashGridCfg), P, C.POINTER(SdfMlp), P, c_i64, c_f32, c_f32,
                              c_u32, P, P, P, P, P, P],
    "dsu_sdf_fd_bwd_cached": [C.POINTER(HashGridCfg), P, C.POINTER(SdfMlp), P, c_i64, c_f32, c_f32,
                              c_u32, P, P, P, P, P, P, P, P, P, P, c_i64, P, P],
    "dsu_sdf_fd_enc_cache_bytes": [c_i64, c_u32],
    "dsu_sdf_fd_fwd_sorted": [C.POINTER(HashGridCfg), P, C.POINTER(SdfMlp), P, P, c_i64, c_f32,
                              c_f32, c_u32, P, P, P, P, P, P],
    "dsu_sdf_fd_bwd_sorted": [C.POINTER(HashGridCfg), P, C.POINTER(SdfMlp), P, P, c_i64, c_f32,
                              c_f32, c_u32, P, P, P, P, P, P, P, P, P, P, c_i64, P, P],
    "dsu_sdf_fd_bwd_sorted_mid": [C.POINTER(HashGridCfg), P, C.POINTER(SdfMlp), P, P, c_i64, c_f32,
                                  c_f32, c_u32, P, P, P, P, P, P, P, P, P, P, c_i64, P, P, P],
    "dsu_sdf_fd_bwd_sorted_fold": [C.POINTER(HashGridCfg), P, C.POINTER(SdfMlp), P, P, c_i64, c_f32,
                                   c_f32, c_u32, P, P, P, P, P, P, P, P, P, P, c_i64, P, P,
                                   C.POINTER(PartialReduce), P],
    "dsu_texture_partial_map": [P],
    "dsu_texture_bwd_shaded_partials": [C.POINTER(TexMlp), P, P, P, P, P, c_i64, c_i64, P, P, P, c_i64,
                                        C.POINTER(PartialReduce), P],
    "dsu_texture_bwd_shaded_partials_m": [C.POINTER(TexMlp), P, P, P, P, P, c_i64, c_i64, P, P, P, P, c_i64,
                                          C.POINTER(PartialReduce), P],
    "dsu_texture_fwd_shaded_m": [C.POINTER(TexMlp), P, P, c_i64, P, P, P, P],
    "dsu_occgrid_refresh_workspace_bytes": [c_i32],
    "dsu_occgrid_refresh": [C.POINTER(OccRefreshArgs), P],
    "dsu_nsr_driver_occ_refresh": [P, C.POINTER(OccRefreshArgs), P],
    "dsu_inpaint_telea_u8c3": [P, P, c_i32, c_i32, c_i32, P],
    "dsu_table_adamw": [P, P, P, P, P, c_i64, c_f32, c_f32, c_f32, c_f32, c_f32, c_f32, c_f32, P],
    "dsu_table_decay": [P, P, c_i64, c_f32, P],
    "dsu_adamw_multi": [P, c_i32, c_f32, c_f32, c_f32, c_f32, P],
    "dsu_smooth_iterate": [P, c_i64, P, P, C.c_double, c_i32, P, P, P],
    "dsu_smooth_energy": [P, c_i64, P, P, P, P],
    "dsu_smooth_energy_partials": [],
    "dsu_set_onewave_grid_cap": [c_i32],
    "dsu_set_scatter_grid_cap": [c_i32],
    "dsu_set_nsr_side_stream_priority": [c_i32],
    "dsu_set_nsr_side_stream_pooling": [c_i32],
    "dsu_volume_band_distance_workspace_bytes": [c_i32, c_i32, c_i32],
    "dsu_volume_band_distance": [P, c_i32, c_i32, c_i32, c_i32, P, P, P, P, P, c_i64, P],
    "dsu_mc_cube_index": [P, c_i32, c_i32, c_i32, C.c_double, P, P],
    "dsu_sdf_fwd_lattice": [P, P, P, P, c_i32, c_i32, c_i32, P, P, c_f32, c_u32, P, P],
    "dsu_zgrid_count": [P, c_i64, c_f32, c_f32, c_f32, c_i32, P, P],
    "dsu_zgrid_fill": [P, c_i64, c_f32, c_f32, c_f32, c_i32, P, P, P, P],
    "dsu_zray_cast": [P, P, c_i64, c_f32, c_f32, c_f32, c_i32, P, P, P, c_i64, c_i32, P, P, P, P, P,
                      P, P],
    "dsu_raster_mask": [P, c_i64, c_f32, c_i32, P, P],
    "dsu_erode_ellipse_u8": [P, c_i32, c_i32, c_i32, P, P],
    "dsu_point_bin_count": [P, c_i64, c_f32, c_f32, c_f32, c_i32, P, P],
    "dsu_point_bin_fill": [P, c_i64, c_f32, c_f32, c_f32, c_i32, P, P, P, P],
    "dsu_knn8_blend": [P, c_i64, P, P, c_i64, c_f32, c_f32, c_f32, c_i32, P, P, P, P],
    "dsu_ab_switches": [],
    "dsu_umbrella_implicit_solve": [P, P, c_i64, C.c_double, P, P, P, c_i32, P],
    "dsu_mesh_decimate_quadric": [P, c_i64, P, c_i64, c_i64, C.c_double, c_i32, P, P, P, P],
    "dsu_mesh_decimate_quadric_q": [P, c_i64, P, c_i64, c_i64, C.c_double, c_i32, P, P, P, P, P],
    "dsu_mesh_decimate_parallel_workspace_bytes": [c_i64, c_i64],
    "dsu_mesh_decimate_parallel": [P, c_i64, P, c_i64, c_i64, c_i64, C.c_double, c_i32, c_i32, P, P, P, P,
                                   c_i64, P],
    "dsu_distance_transform_l2_5x5": [P, c_i32, c_i32, P],
    "dsu_skeletonize_lee_2d": [P, c_i32, c_i32, P],
    "dsu_nsr_draws": [C.c_uint64, c_i64, c_i32, c_i32, c_i32, c_i32, P, P, P, P, c_i32, P, P, P],
    "dsu_nsr_driver_workspace_bytes": [C.POINTER(NsrDriverCfg)],
    "dsu_nsr_driver_create": [C.POINTER(NsrDriverCfg), C.POINTER(c_vp)],
    "dsu_nsr_driver_destroy": [c_vp],
    "dsu_nsr_driver_step": [c_vp, C.POINTER(NsrStepArgs), P],
    "dsu_nsr_driver_terms": [c_vp],
    "dsu_nsr_driver_adam_moments": [c_vp, c_i32],
    "dsu_nsr_driver_sync": [c_vp],
    "dsu_nsr_driver_timing": [c_vp, c_i32],
    "dsu_nsr_driver_timing_read": [c_vp, c_i32, C.POINTER(c_i64), C.POINTER(C.c_double),
                                   C.POINTER(C.c_double)],
    "dsu_nsr_driver_timing_flops": [c_vp, c_i32, C.POINTER(C.c_double)],
    "dsu_spatial_sort": [P, c_i64, c_f32, c_i32, P, P, P, c_i64, P],
    "dsu_spatial_sort_workspace_bytes": [c_i64, c_i32],
    "dsu_ray_aabb": [P, P, c_i64, P, P, c_f32, P, P, P],
    "dsu_ray_march_count": [P, P, P, P, c_i64, P, P, c_i32, c_f32, P, P],
    "dsu_ray_march_fill": [P, P, P, P, c_i64, P, P, c_i32, c_f32, P, P, P, P, P],
    "dsu_ray_march_scratch": [P, P, P, P, c_i64, P, P, c_i32, c_f32, c_i32, P, P, P, P],
    "dsu_ray_compact": [P, P, c_i32, P, P, c_i64, P, P, P, P],
    "dsu_ray_compact_points": [P, P, c_i32, P, P, c_i64, P, P, P, P, P, P],
    "dsu_ray_compact_points_cap": [P, P, c_i32, P, P, c_i64, P, P, P, P, P, c_i64, P],
    "dsu_points_tail": [P, c_i64, P, P, P, c_i64, c_f32, P],
    "dsu_spatial_sort_dev": [P, c_i64, P, c_i64, c_f32, c_i32, P, P, P, c_i64, P],
    "dsu_ray_offsets": [P, c_i64, P, P, P],
    "dsu_ortho_ray_batch": [P, P, P, c_i64, P, P, P, P, c_i32, P, P, P, c_i32, c_i32, P, P, P, P, P,
                            P, P],
    "dsu_ortho_ray_batch_split": [P, P, P, c_i64, P, P, P, P, c_i32, P, P, P, c_i32, c_i32, P, P, P, P, P,
                            P, P, P, P],
    "dsu_ray_losses": [P, P, P, P, P, P, c_i32, C.POINTER(RayLossCfg), P, P, P],
    "dsu_sample_losses": [P, P, c_i64, c_i64, c_f32, c_f32, c_f32, c_f32, c_i32, P, P, P, P],
    "dsu_texture_fwd": [C.POINTER(TexMlp), P, c_i64, P, P],
    "dsu_texture_bwd_workspace_bytes": [c_i64],
    "dsu_texture_bwd": [C.POINTER(TexMlp), P, P, P, c_i64, P, P, P, P, P, P, P, P, c_i64, P],
    "dsu_texture_fwd_shaded": [C.POINTER(TexMlp), P, P, c_i64, P, P, P],
    "dsu_texture_bwd_shaded": [C.POINTER(TexMlp), P, P, P, P, P, c_i64, c_i64, P, P, P, P, P, P, P, P,
                               P, c_i64, P],
    "dsu_weights_from_alpha_fwd": [P, P, P, c_i64, P, P],
    "dsu_weights_from_alpha_bwd": [P, P, P, P, P, c_i64, P, P],
    "dsu_accumulate_fwd": [P, P, c_i32, P, P, c_i64, P, P],
    "dsu_neus_composite_fwd": [P, P, P, P, P, P, P, P, c_i64, P, c_f32, P, P, P, P],
    "dsu_neus_composite_bwd": [P, P, P, P, P, P, P, P, c_i64, P, c_f32, P, P, P, P, P, P, P,
                               P, P],
    "dsu_shade_prep_fwd": [P, P, c_i64, P, P, P],
    "dsu_shade_prep_bwd": [P, P, P, c_i64, P, P, P],
    "dsu_shade_prep_bwd_tail": [P, P, P, c_i64, c_i64, P, P, P],
    "dsu_occgrid_ema": [P, P, P, c_i64, c_f32, P, P],
    "dsu_occgrid_binarize": [P, c_i64, c_f32, P, P],
    "dsu_ric_offsets": [c_i32, c_i32, P, P],
    "dsu_deform_conv3x3_fwd": [P, P, c_i64, P, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, P, P,
                               c_i32, P, P, P],
    "dsu_mv_attention_fwd": [P, P, P, P, P, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32,
                             C.POINTER(c_i64), C.POINTER(c_i64), C.POINTER(c_i64),
                             C.POINTER(c_i64), c_f32, P],
    "dsu_conv2d_nhwc_f16_fwd": [P, P, P, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32,
                                c_i32, P, P, P, P],
    "dsu_conv2d_nhwc_f16_fwd_ws": [P, P, P, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32,
                                   c_i32, P, P, P, c_i32, P, c_i64, P],
    "dsu_conv2d_nhwc_f16_fwd_fx": [P, P, P, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32,
                                   c_i32, P, P, P, c_i32, P, c_i64, P, c_i64, P],
    "dsu_gemm_f16_fwd_fx": [P, P, P, c_i64, c_i32, c_i32, P, P, c_i32, c_i32, P, c_i64, P, c_i64, P],
    "dsu_conv2d_nhwc_f16_split_k": [c_i32] * 9,
    "dsu_conv2d_nhwc_f16_workspace_bytes": [c_i32] * 9,
    "dsu_gemm_f16_split_k": [c_i64, c_i32, c_i32],
    "dsu_gemm_f16_workspace_bytes": [c_i64, c_i32, c_i32],
    "dsu_gemm_f16_fwd": [P, P, P, c_i64, c_i32, c_i32, P, P, c_i32, c_i32, P, c_i64, P],
    "dsu_gemm_geglu_fwd": [P, P, P, c_i64, c_i32, c_i32, P, P],
    "dsu_groupnorm_nhwc_f16": [P, P, P, c_i32, c_i32, c_i32, c_i32, c_f32, c_i32, P, P, P],
    "dsu_layernorm_f16": [P, P, P, c_i64, c_i32, c_f32, P, P],
    "dsu_geglu_f16": [P, c_i64, c_i32, P, P],
    "dsu_deform_tap_table_bytes": [c_i32, c_i32],
    "dsu_deform_tap_table": [P, c_i32, c_i32, P, P],
    "dsu_conv2d_wgrad_workspace_bytes": [c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32],
    "dsu_conv2d_wgrad": [P, P, P, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, P, P,
                         c_i32, P],
    "dsu_deform_conv3x3_dgrad_gather": [P, P, P, P, c_i64, c_i32, P, P],
    "dsu_norm_train_fwd": [C.POINTER(NormCfg), P, P, P, P, P, P, P, P, P],
    "dsu_norm_train_bwd": [C.POINTER(NormCfg), P, P, P, P, P, P, P, P, P, P],
    "dsu_channel_sum": [P, c_i32, c_i32, c_i32, P, P],
    "dsu_act_fwd": [P, P, c_i64, c_i32, P],
    "dsu_act_bwd": [P, P, P, c_i64, c_i32, P],
    "dsu_maxpool2_fwd": [P, P, c_i64, c_i32, c_i32, P],
    "dsu_maxpool2_bwd": [P, P, P, c_i64, c_i32, c_i32, P],
    "dsu_upsample2_fwd": [P, P, c_i64, c_i32, c_i32, P],
    "dsu_upsample2_bwd": [P, P, c_i64, c_i32, c_i32, P],
    "dsu_pair_loss": [P, P, c_f32, c_i64, c_i32, c_f32, P, P, P],
    "dsu_conv_x3_packed_elems": [c_i32, c_i32, c_i32],
    "dsu_conv_x3_pack_weights": [P, c_i32, c_i32, c_i32, P, P, P],
    "dsu_deform_conv3x3_fwd_x3": [P, P, c_i64, P, P, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32,
                                  P, P, c_i32, P, P, P],
    "dsu_conv2d_fwd_x3": [P, P, P, P, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32,
                          c_i32, P, P, c_i32, P, P, P],
    "dsu_conv_f32p_pack_weights": [P, c_i32, c_i32, c_i32, P, P],
    "dsu_deform_conv3x3_fwd_f32p": [P, P, c_i64, P, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32,
                                    P, P, c_i32, P, P, P],
    "dsu_conv2d_fwd_f32p": [P, P, P, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32,
                            P, P, c_i32, P, P, P],
    "dsu_conv2d_fwd": [P, P, P, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32,
                       P, P, c_i32, P, P, P],
}

# return types that are neither an error code nor a byte count
_RESTYPES = {"dsu_nsr_driver_destroy": None, "dsu_nsr_driver_terms": c_vp,
             "dsu_nsr_driver_adam_moments": c_vp,
             "dsu_conv_x3_packed_elems": C.c_int64}

_lib = None


class DsuError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle.  Raises if the library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DsuError(
                f"{LIB_PATH} is missing: the HIP extension is not built "
                "(run `python -m drawingspinup_amd.build`). There is no CPU fallback.")
        h = C.CDLL(LIB_PATH)
        h.dsu_strerror.restype = C.c_char_p
        h.dsu_strerror.argtypes = [C.c_int]
        for name, args in _PROTOS.items():
            fn = getattr(h, name)  # AttributeError here = header/library mismatch
            fn.restype = _RESTYPES.get(name, C.c_int64 if name.endswith("_bytes") else C.c_int)
            fn.argtypes = args
        _lib = h
    return _lib


def check(rc, what):
    if rc != 0:
        raise DsuError(f"{what} failed: {lib().dsu_strerror(rc).decode()} ({rc})")


def ptr(t, dtype=None):
    """Device pointer of a contiguous CUDA(HIP) tensor; None -> NULL."""
    if t is None:
        return None
    if not t.is_cuda:
        raise DsuError("libdsu_hip kernels need device tensors (no CPU fallback)")
    if not t.is_contiguous():
        raise DsuError("libdsu_hip kernels need contiguous tensors")
    if dtype is not None and t.dtype != dtype:
        raise DsuError(f"expected dtype {dtype}, got {t.dtype}")
    return C.c_void_p(t.data_ptr())


def publish_sync(device=None):
    """Before a lazily built device tensor goes into a process-wide cache: wait for the stream that
    produced it.  Several drawings may be in flight on one GPU (one Python thread + one stream each,
    bench.py --inflight): another thread's stream must not read a cached table ahead of the copies /
    kernels that fill it."""
    if torch.cuda.is_available() and torch.cuda.is_initialized() and (device is None or torch.device(device).type == "cuda"):
        torch.cuda.current_stream(device).synchronize()


def stream():
    """Raw handle of torch's current stream on the current device.  (The public
    torch.cuda.current_stream() wrapper costs ~9 us per call: 0.17 ms of a 2 ms optimisation
    step that launches ~100 kernels.)"""
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def hashgrid_levels(cfg: HashGridCfg):
    """Host-only level table (works without a GPU)."""
    lv = HashGridLevels()
    check(lib().dsu_hashgrid_make_levels(C.byref(cfg), C.byref(lv)), "dsu_hashgrid_make_levels")
    n = cfg.n_levels
    return {
        "offsets": [lv.offsets[i] for i in range(n + 1)],
        "resolution": [lv.resolution[i] for i in range(n)],
        "scale": [lv.scale[i] for i in range(n)],
        "hashed": [lv.hashed[i] for i in range(n)],
    }
