"""ctypes binding of ``libastrild_hip.so`` (the C-ABI in ``include/astrild_hip.h``).

There is NO CPU fallback: if the shared library is missing or fails to load
the product path raises.  Build it with ``python -c "import __graft_entry__ as
g; g.build()"`` or ``make -C astrild_amd/csrc``.
"""
import ctypes as ct
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ASTRILD_HIP_LIB: load another build of the same ABI (perf experiments: scripts/build_variants.sh)
LIB_PATH = os.environ.get("ASTRILD_HIP_LIB") or os.path.join(_HERE, "libastrild_hip.so")

F32, F64 = 0, 1
WIN = {"ngp": 0, "nnb": 0, "nearest": 0, "cic": 1, "tsc": 2}
FFT_R2C, FFT_C2R, FFT_C2C_FWD, FFT_C2C_INV = 0, 1, 2, 3
ERR_WORKSPACE = -4
SUM_PARTS = 1024          # AST_SUM_PARTS
BIN = {"integer": 0, "float64": 1}      # AST_BIN_*


class AstrildHipError(RuntimeError):
    pass


_vp, _sz, _i, _d, _u64 = ct.c_void_p, ct.c_size_t, ct.c_int, ct.c_double, ct.c_uint64

# name -> (restype, argtypes); mirrors include/astrild_hip.h one to one
SIGNATURES = {
    "ast_version": (_i, []),
    "ast_last_error": (ct.c_char_p, []),
    "ast_profile_enable": (_i, [_i]),
    "ast_profile_report": (_i, [ct.c_char_p, _sz]),
    "ast_fill": (_i, [_vp, _i, _sz, _d, _vp]),
    "ast_divide": (_i, [_vp, _i, _sz, _d, _vp]),
    "ast_synth_lattice_particles": (_i, [_vp, _i, _sz, _sz, _i, _d, _d, _u64, _u64, _vp]),
    "ast_synth_clustered_particles": (_i, [_vp, _i, _sz, _i, _d, _d, _u64, _i, _d, _i, _vp]),
    "ast_paint_occupancy_probe_bytes": (_sz, [_i]),
    "ast_paint_occupancy_probe": (_i, [_vp, _i, _sz, _i, _d, _d, _sz, _vp, _sz, _vp, _vp]),
    "ast_ngp_assign": (_i, [_vp, _vp, _vp, _vp, _i, _sz, _i, _vp, _vp, _vp, _vp]),
    "ast_paint": (_i, [_i, _i, _vp, _vp, _sz, _i, _d, _d, _i, _i, _vp, _vp, _d, _vp]),
    "ast_paint_tiled_workspace_bytes": (_sz, [_i, _i, _sz, _i, _i, _i]),
    "ast_paint_tiled": (_i, [_i, _i, _vp, _vp, _sz, _i, _d, _d, _i, _i, _vp, _vp, _sz, _vp, _i, _d, _d, _i, _i, _d, _vp]),
    "ast_paint_tiled_stage": (_i, [_i, _i, _vp, _vp, _sz, _i, _d, _d, _i, _i, _vp, _vp, _sz, _vp, _i, _d, _d, _i, _i, _d, _i, _i, _i, _i, _i, _vp]),
    "ast_deflection_to_shear": (_i, [_vp, _vp, _i, _d, _vp, _vp, _vp]),
    "ast_paint_tile_rows": (_i, [_i]),
    "ast_paint_tile_row_planes": (_i, []),
    "ast_paint_tiled_list_stats": (_i, [_vp, _i, _i, _sz, _i, _i, _i, _vp, _vp]),
    "ast_paint_order_probe": (_i, [_vp, _i, _sz, _i, _d, _d, _i, _vp, _vp]),
    "ast_route_count": (_i, [_vp, _i, _sz, _i, _d, _i, _i, _vp, _vp]),
    "ast_route_scatter": (_i, [_vp, _vp, _i, _sz, _i, _d, _i, _i, _vp, _vp, _vp, _vp]),
    "ast_accumulate": (_i, [_vp, _vp, _i, _sz, _vp]),
    "ast_stream_copy": (_i, [_vp, _vp, _sz, _i, _vp]),
    "ast_fft_plan_create": (_i, [ct.POINTER(_vp), _i, _i, _i, ct.POINTER(_sz), _sz, _d, _i]),
    "ast_fft_plan_create_strided_1d": (_i, [ct.POINTER(_vp), _i, _i, _sz, _sz, _sz, _sz, _d]),
    "ast_fft_plan_create_general": (_i, [ct.POINTER(_vp), _i, _i, _i, ct.POINTER(_sz), ct.POINTER(_sz),
                                        ct.POINTER(_sz), _sz, _sz, _sz, _d, _i]),
    "ast_fft_plan_work_bytes": (_sz, [_vp]),
    "ast_fft_exec": (_i, [_vp, _vp, _vp, _vp]),
    "ast_fft_plan_destroy": (_i, [_vp]),
    "ast_fft_tile_supported": (_i, [_i, _sz]),
    "ast_fft_tile_c2c": (_i, [_vp, _i, _sz, _sz, _sz, _sz, _sz, _d, _vp]),
    "ast_fft_tile_c2c_packed": (_i, [_vp, _vp, _i, _sz, _sz, _sz, _sz, _i, _i, _vp, _d, _vp]),
    "ast_fft_tile_rows_r2c": (_i, [_vp, _vp, _i, _sz, _sz, _sz, _sz, _d, _vp]),
    "ast_fft_tile_rows_r2c_slab_halo": (_i, [_vp, _vp, _i, _sz, _sz, _sz, _d, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "ast_fft_tile_r2c_3d": (_i, [_vp, _vp, _i, _sz, _d, _vp]),
    "ast_fft_tile_c2r_3d": (_i, [_vp, _vp, _vp, _i, _sz, _i, _i, _d, _vp]),
    "ast_fft_tile_power_scratch_bytes": (_sz, [_sz]),
    "ast_fft_tile_power_3d": (_i, [_vp, _vp, _sz, _i, _sz, _d, _d, _i, _i, _vp, _vp]),
    "ast_fft_tile_isqrt_table": (_i, [_vp, _i, _vp]),
    "ast_fft_tile_power_3d_halo": (_i, [_vp, _vp, _i, _vp, _sz, _i, _sz, _d, _d, _i, _i, _vp, _vp]),
    "ast_fft_tile_block_power_scratch_bytes": (_sz, [_sz, _sz]),
    "ast_fft_tile_block_power": (_i, [_vp, _vp, _sz, _i, _sz, _sz, _sz, _sz, _d, _d, _i, _i, _vp, _vp]),
    "ast_fft_tile_disc_layout": (_i, [_sz, _i, ct.POINTER(ct.c_uint), ct.POINTER(ct.c_ubyte), ct.POINTER(_i)]),
    "ast_fft_tile_disc_table": (_i, [_sz, _i, ct.POINTER(_i), _sz]),
    "ast_fft_tile_c2c_disc": (_i, [_vp, _vp, _i, _sz, _sz, _sz, _i, _i, _vp, _d, _vp]),
    "ast_fft_tile_disc_power_scratch_bytes": (_sz, [_sz, _i]),
    "ast_fft_tile_disc_block_power": (_i, [_vp, _vp, _sz, _i, _sz, _i, _i, _d, _d, _i, _i, _vp, _vp]),
    "ast_comm_unique_id": (_i, [_vp, _sz]),
    "ast_comm_init": (_i, [ct.POINTER(_vp), _i, _i, _vp, _sz]),
    "ast_comm_destroy": (_i, [_vp]),
    "ast_slab_transpose": (_i, [_vp, _vp, ct.POINTER(_sz), ct.POINTER(_sz), _vp, ct.POINTER(_sz), ct.POINTER(_sz), _i, _vp]),
    "ast_comm_allreduce_sum": (_i, [_vp, _vp, _sz, _vp]),
    "ast_fft_tile_c2r_triangles_scratch_bytes": (_sz, []),
    "ast_fft_tile_c2r_triangles": (_i, [ct.POINTER(_vp), ct.POINTER(_i), _i, _i, _sz, _sz, _d, _vp, _i, _vp, _vp, _vp]),
    "ast_lowk_work_bytes": (_sz, [_sz, _sz]),
    "ast_lowk_mode_count": (_i, []),
    "ast_lowk_shell_count": (_i, []),
    "ast_fft_tile_c2r_3d_batch": (_i, [_vp, ct.POINTER(_vp), ct.POINTER(_vp), _i, _sz, ct.POINTER(_i), ct.POINTER(_i), _i, _d, _i, _sz, _vp]),
    "ast_fft_tile_rows_r2c_lowz": (_i, [_vp, _vp, _i, _sz, _sz, _sz, _sz, _d, _vp, _vp]),
    "ast_lowk_modes_from_z": (_i, [_sz, _sz, _sz, _i, _vp, _vp, _sz, _vp]),
    "ast_lowk_modes": (_i, [_vp, _i, _sz, _sz, _sz, _i, _vp, _vp, _sz, _vp]),
    "ast_lowk_shell_sums": (_i, [_vp, _sz, _d, _i, _vp, _vp]),
    "ast_paint_tiled_halo": (_i, [_vp, _i, _i, _sz, _i, _i, _i, ct.POINTER(ct.c_void_p)]),
    "ast_power_bin_1d": (_i, [_vp, _vp, _i, _i, _d, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp]),
    "ast_interlace_compensate": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "ast_shell_filter": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "ast_shell_mask_real": (_i, [_vp, _i, _i, _i, _vp]),
    "ast_half_real_to_full": (_i, [_vp, _vp, _i, _vp]),
    "ast_triple_product_sum": (_i, [_vp, _vp, _vp, _i, _sz, _vp, _vp]),
    "ast_triple_product_sums_scratch_bytes": (_sz, []),
    "ast_triple_product_sums": (_i, [_vp, _i, _i, _sz, _vp, _i, _vp, _vp, _vp]),
    "ast_slab_pack": (_i, [_vp, _vp, _i, _sz, _sz, _sz, _i, _vp]),
    "ast_slab_unpack": (_i, [_vp, _vp, _i, _sz, _sz, _sz, _i, _vp]),
    "ast_kappa_stack": (_i, [_vp, _vp, _vp, _i, _sz, _i, _vp, _i, _vp]),
    "ast_lens_plan_create": (_i, [ct.POINTER(_vp), _i, _d]),
    "ast_lens_plan_destroy": (_i, [_vp]),
    "ast_fft64_supported": (_i, [_sz]),
    "ast_fft64_power_scratch_bytes": (_sz, [_sz]),
    "ast_fft64_r2c_3d": (_i, [_vp, _vp, _sz, _d, _vp]),
    "ast_fft64_power_3d": (_i, [_vp, _vp, _sz, _sz, _d, _i, _vp, _vp]),
    "ast_fft64_power_3d_f32": (_i, [_vp, _vp, _sz, _sz, _d, _i, _vp, _vp]),
    "ast_fft32_big_supported": (_i, [_sz]),
    "ast_fft32_big_power_scratch_bytes": (_sz, [_sz]),
    "ast_fft32_big_power_3d": (_i, [_vp, _vp, _sz, _sz, _d, _i, _d, _vp, _vp]),
    "ast_fft32_big_power_3d_halo": (_i, [_vp, _vp, _i, _vp, _sz, _sz, _d, _i, _d, _vp, _vp]),
    "ast_fft64_power_3d_halo": (_i, [_vp, _vp, _i, _vp, _sz, _sz, _d, _i, _vp, _vp]),
    "ast_lens_cols_supported": (_i, [_sz]),
    "ast_lens_rows_supported": (_i, [_sz]),
    "ast_lens_rows_forward": (_i, [_vp, _sz, _vp, _sz, _vp]),
    "ast_lens_rows_forward_full": (_i, [_vp, _sz, _sz, _vp, _sz, _vp]),
    "ast_lens_rows_inverse": (_i, [_vp, _sz, _sz, _d, _vp, _vp]),
    "ast_lens_cols_forward": (_i, [_vp, _sz, _sz, _sz, _sz, _vp]),
    "ast_lens_cols_inverse": (_i, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, _vp]),
    "ast_lens_cols_convolve": (_i, [_vp, _sz, _sz, _sz, _sz, _vp, _vp, _i, _sz, _vp]),
    "ast_kappa_to_alphas": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "ast_kappa_to_phi": (_i, [_vp, _vp, _vp, _vp]),
    "kappa0_to_alphas": (None, [_vp, _i, _d, _vp, _vp]),
    "kappa0_to_phi": (None, [_vp, _i, _d, _vp]),
    "ast_smooth_plan_create": (_i, [ct.POINTER(_vp), _i]),
    "ast_smooth_plan_destroy": (_i, [_vp]),
    "ast_gaussian_smooth": (_i, [_vp, _vp, _d, _i, _vp]),
    "ast_minmax": (_i, [_vp, _i, _sz, _vp, _vp]),
    "ast_sum": (_i, [_vp, _i, _sz, _vp, _vp]),
    "ast_flat_power_bin": (_i, [_vp, _vp, _i, _d, _vp, _i, _vp, _vp, _vp]),
    "ast_ring_filter_2d": (_i, [_vp, _vp, _i, _d, _d, _d, _vp]),
    "ast_peak_find": (_i, [_vp, _i, _i, _d, _d, _sz, _vp, _vp, _vp, _vp]),
    "ast_order_statistics": (_i, [_vp, _i, _sz, ct.POINTER(_sz), _i, ct.POINTER(_d), _vp, _vp]),
    "ast_histogram": (_i, [_vp, _i, _sz, _d, _d, _i, _vp, _vp]),
    "ast_histogram_auto": (_i, [_vp, _i, _sz, _i, _vp, _vp, _vp]),
    "ast_add": (_i, [_vp, _vp, _vp, _i, _sz, _vp]),
    "ast_nfw_paint": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _d, _i, _i, _d, _i, _vp, _i, _vp]),
    "ast_add_patch": (_i, [_vp, _i, _vp, _i, _i, _i, _vp]),
    "ast_dgd_filter": (_i, [_vp, _vp, _vp, _i, _d, _d, _i, _i, _vp]),
    "ast_hann_apodize": (_i, [_vp, _vp, _i, _vp]),
    "ast_zoom_linear": (_i, [_vp, _i, _vp, _i, _vp]),
    "ast_gaussian_filter_order": (_i, [_vp, _vp, _vp, _sz, _i, _d, _i, _i, _i, _vp]),
    "ast_convolve2d": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "ast_aperture_photometry": (_i, [_vp, _vp, _vp, _i, _d, _vp]),
}

_lib = None


def lib():
    """The loaded library (cached).  Raises if it is not built — no fallback."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise AstrildHipError(
                f"{LIB_PATH} not found: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()'). "
                "astrild_amd has no CPU fallback."
            )
        # torch ships its own libamdhip64 / librocfft; load it FIRST so this library
        # binds to the same HIP runtime instance that owns the tensors' memory and
        # streams (two runtimes in one process do not share devices or pointers).
        import torch  # noqa: F401
        handle = ct.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)     # AttributeError = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().ast_last_error().decode(errors="replace")
        raise AstrildHipError(f"{what or 'libastrild_hip'} failed (code {rc}): {msg}")
    return rc
