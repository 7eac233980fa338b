"""ctypes binding of libfregrid_hip.so (C ABI: include/fregrid_hip.h).

The library is built in-tree by ``make -C fre-nctools_amd/csrc`` (or ``__graft_entry__.build()``).
Loading fails loudly when it is missing -- there is no CPU fallback in the product path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class FregridHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libfregrid_hip error {code}: {msg}")
        self.code = code


def lib_path():
    # FREGRID_HIP_LIB: another build of the same library (same-box A/B timing of kernel variants, scripts/); never a fallback
    return os.environ.get("FREGRID_HIP_LIB") or os.path.join(_HERE, "libfregrid_hip.so")


# every symbol include/fregrid_hip.h declares (tests/test_cpu_capi_and_oracle.py::test_library_exports_every_declared_symbol)
EXPORTS = [
    "get_maxxgrid", "get_grid_area", "create_xgrid_2dx2d_order1", "create_xgrid_2dx2d_order2", "conserve_interp",
    "clip_2dx2d", "poly_area", "poly_ctrlon", "poly_ctrlat", "fix_lon", "pimod",
    "fg_clip_2dx2d_batch", "fg_poly_op_batch",
    "get_maxxgrid_", "get_grid_area_", "create_xgrid_2dx2d_order1_", "create_xgrid_2dx2d_order2_",
    "fg_last_error", "fg_device_count", "fg_plan_create", "fg_plan_create_dev", "fg_plan_create_empty",
    "fg_plan_destroy", "fg_plan_set_stream", "fg_pool_release", "fg_plan_nxgrid", "fg_plan_ncells_in",
    "fg_plan_cell_sums_dev", "fg_plan_copy_cell_sums", "fg_plan_accumulate_cell_sums", "fg_plan_accumulate_cell_sums_async", "fg_dev_gather_f64", "fg_dev_scatter_f64", "fg_plan_finalize", "fg_plan_get_xgrid", "fg_plan_get_polygons", "fg_plan_create_polylist", "fg_plan_get_cell_struct",
    "fg_plan_get_cell_area", "fg_plan_set_xgrid", "fg_plan_apply", "fg_plan_apply_interleaved", "fg_plan_apply_records", "fg_plan_apply_ex", "fg_plan_mono_begin",
    "fg_plan_mono_minmax_dev", "fg_plan_mono_copy_minmax", "fg_plan_mono_end",
    "fg_plan_create_great_circle", "fg_plan_create_great_circle_dev", "fg_latlon2xyz", "create_xgrid_great_circle",
    "create_xgrid_great_circle_", "get_grid_great_circle_area", "get_grid_great_circle_area_", "clip_2dx2d_great_circle",
    "great_circle_area", "fg_gc_clip_batch", "conserve_interp_great_circle", "fg_sincos_batch",
    "create_xgrid_1dx2d_order1", "create_xgrid_1dx2d_order2", "create_xgrid_2dx1d_order1", "create_xgrid_2dx1d_order2",
    "create_xgrid_1dx2d_order1_", "create_xgrid_1dx2d_order2_", "create_xgrid_2dx1d_order1_", "create_xgrid_2dx1d_order2_",
    "clip", "box_ctrlat", "box_ctrlon", "get_grid_area_no_adjust", "get_grid_area_no_adjust_", "fg_plan_stream", "fg_plan_sync",
    "fg_c2l_create", "fg_c2l_destroy", "fg_c2l_ncells", "fg_c2l_halo_size", "fg_c2l_set_stream", "fg_c2l_sync",
    "fg_c2l_get_centres", "fg_c2l_fill_halo", "fg_c2l_gradient", "fg_c2l_gradient_records", "fg_c2l_records", "fg_c2l_grid_info", "fg_find_contacts", "fg_halo_map",
    "fg_gnomonic_ed_grid", "fg_tripolar_corners", "fg_remap_write", "fg_remap_write_interp", "fg_remap_read_size", "fg_remap_read", "fg_remap_last_error",
    "fg_plan_trim", "fg_plan_ncells_out", "fg_plan_order", "fg_plan_device", "fg_dev_alloc", "fg_dev_free", "fg_dev_upload", "fg_dev_download",
    "fg_nc_open", "fg_nc_create", "fg_nc_def_dim", "fg_nc_def_var", "fg_nc_put_att_text", "fg_nc_put_att_double", "fg_nc_enddef",
    "fg_nc_inq_ndims", "fg_nc_inq_nvars", "fg_nc_inq_numrecs", "fg_nc_inq_dimid", "fg_nc_inq_dim", "fg_nc_inq_varid", "fg_nc_inq_var",
    "fg_nc_get_att_double", "fg_nc_get_att_text", "fg_nc_get_vara", "fg_nc_get_vara_double", "fg_nc_put_vara", "fg_nc_put_vara_double",
    "fg_nc_close", "fg_nc_last_error", "fg_dev_widen", "fg_dev_narrow", "fg_sweep_create", "fg_sweep_run", "fg_sweep_destroy", "fg_host_alloc", "fg_host_free",
    "fg_plan_stats", "fg_set_search_mode", "fg_set_search_chunks", "fg_set_search_cull", "fg_set_search_finalize", "fg_set_search_rect", "fg_set_search_frame", "fg_set_apply_xcd", "fg_set_apply_vec", "fg_set_apply_ep", "fg_set_gc_split", "fg_set_profiling", "fg_plan_phase_ms", "fg_gnomonic_ed_corners", "fg_latlon_corners",
]


class ApplyOpts(C.Structure):
    """fg_apply_opts (include/fregrid_hip.h); pointer members are device addresses."""
    _fields_ = [("has_missing", C.c_int), ("missing", C.c_double), ("weight", C.c_void_p),
                ("cell_methods_sum", C.c_int), ("field_area", C.c_void_p), ("area_missing", C.c_double),
                ("cell_area_in", C.c_void_p), ("cell_area_out", C.c_void_p), ("monotonic", C.c_int)]


def lib():
    """Load (once) and return the ctypes handle with argument types declared."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: build the HIP extension first (make -C fre-nctools_amd/csrc, or "
            "python -c 'import __graft_entry__ as g; g.build()').  There is no CPU fallback.")
    # PyTorch-ROCm bundles its own libamdhip64 (same SONAME as /opt/rocm's).  Two HIP runtimes in one
    # process break device discovery, so when torch is installed load it FIRST: the dynamic loader then
    # resolves our libamdhip64.so.7 dependency to the copy torch already mapped (one runtime, shared
    # device pointers and streams).  Without torch (plain C callers) the system runtime is used.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path)
    dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p
    dpp = C.POINTER(dp)
    cip = C.POINTER(C.c_int)

    L.get_maxxgrid.restype = C.c_int
    L.get_grid_area.argtypes = [cip, cip, dp, dp, dp]
    L.get_grid_area.restype = None
    L.create_xgrid_2dx2d_order1.argtypes = [cip] * 4 + [dp] * 5 + [ip] * 4 + [dp]
    L.create_xgrid_2dx2d_order1.restype = C.c_int
    L.create_xgrid_2dx2d_order2.argtypes = [cip] * 4 + [dp] * 5 + [ip] * 4 + [dp] * 3
    L.create_xgrid_2dx2d_order2.restype = C.c_int
    L.conserve_interp.argtypes = [C.c_int] * 4 + [dp] * 7
    L.conserve_interp.restype = None

    L.clip_2dx2d.argtypes = [dp, dp, C.c_int, dp, dp, C.c_int, dp, dp]
    L.clip_2dx2d.restype = C.c_int
    L.poly_area.argtypes = [dp, dp, C.c_int]
    L.poly_area.restype = C.c_double
    L.poly_ctrlon.argtypes = [dp, dp, C.c_int, C.c_double]
    L.poly_ctrlon.restype = C.c_double
    L.poly_ctrlat.argtypes = [dp, dp, C.c_int]
    L.poly_ctrlat.restype = C.c_double
    L.fix_lon.argtypes = [dp, dp, C.c_int, C.c_double]
    L.fix_lon.restype = C.c_int
    L.fg_clip_2dx2d_batch.argtypes = [C.c_int, dp, dp, ip, dp, dp, ip, dp, dp, ip]
    L.fg_clip_2dx2d_batch.restype = C.c_int
    L.fg_poly_op_batch.argtypes = [C.c_int, C.c_int, dp, dp, ip, dp, dp]
    L.fg_poly_op_batch.restype = C.c_int

    L.fg_last_error.restype = C.c_char_p
    L.fg_device_count.restype = C.c_int
    L.fg_plan_create.argtypes = [C.c_int, C.c_int, ip, ip, dpp, dpp, dpp, C.c_int, C.c_int, dp, dp, C.c_int, C.POINTER(vp)]
    L.fg_plan_create.restype = C.c_long
    L.fg_plan_create_dev.argtypes = [C.c_int, C.c_int, ip, ip, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp),
                                     C.c_int, C.c_int, vp, vp, C.c_double, C.c_double, C.c_int, vp, C.c_int,
                                     C.POINTER(vp)]
    L.fg_plan_create_dev.restype = C.c_long
    L.fg_plan_create_empty.argtypes = [C.c_int, C.c_int, ip, ip, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.fg_plan_create_empty.restype = C.c_int
    L.fg_plan_destroy.argtypes = [vp]
    L.fg_plan_destroy.restype = None
    L.fg_plan_set_stream.argtypes = [vp, vp]
    L.fg_plan_set_stream.restype = C.c_int
    L.fg_pool_release.restype = None
    L.fg_plan_nxgrid.argtypes = [vp]
    L.fg_plan_nxgrid.restype = C.c_long
    L.fg_plan_ncells_in.argtypes = [vp]
    L.fg_plan_ncells_in.restype = C.c_long
    L.fg_plan_cell_sums_dev.argtypes = [vp]
    L.fg_plan_cell_sums_dev.restype = vp
    L.fg_plan_copy_cell_sums.argtypes = [vp, vp]
    L.fg_plan_copy_cell_sums.restype = C.c_int
    L.fg_plan_accumulate_cell_sums.argtypes = [vp, vp, vp, C.c_int]
    L.fg_plan_accumulate_cell_sums.restype = C.c_int
    L.fg_plan_accumulate_cell_sums_async.argtypes = [vp, vp, vp, C.c_int]
    L.fg_plan_accumulate_cell_sums_async.restype = C.c_int
    L.fg_dev_alloc.argtypes = [C.c_size_t, C.c_int]
    L.fg_dev_alloc.restype = vp
    L.fg_dev_free.argtypes = [vp]
    L.fg_dev_free.restype = None
    L.fg_dev_upload.argtypes = [vp, vp, C.c_size_t]
    L.fg_dev_upload.restype = C.c_int
    L.fg_dev_download.argtypes = [vp, vp, C.c_size_t]
    L.fg_dev_download.restype = C.c_int
    for f in (L.fg_dev_gather_f64, L.fg_dev_scatter_f64):
        f.argtypes = [vp, vp, vp, C.c_long]
        f.restype = C.c_int
    L.fg_plan_finalize.argtypes = [vp, vp]
    L.fg_plan_finalize.restype = C.c_int
    L.fg_plan_get_xgrid.argtypes = [vp] + [ip] * 5 + [dp] * 3
    L.fg_plan_get_xgrid.restype = C.c_int
    L.fg_plan_get_polygons.argtypes = [vp, C.c_int, ip, dp, dp, dp]
    L.fg_plan_get_polygons.restype = C.c_int
    L.fg_plan_create_polylist.argtypes = [C.c_int, C.c_int, ip, dp, dp, dp, dp, C.c_int, C.c_int, dp, dp, C.c_int, C.POINTER(vp)]
    L.fg_plan_create_polylist.restype = C.c_long
    L.fg_plan_get_cell_struct.argtypes = [vp, C.c_int] + [dp] * 5 + [ip] + [dp] * 2
    L.fg_plan_get_cell_struct.restype = C.c_int
    L.fg_plan_get_cell_area.argtypes = [vp, dp, dp]
    L.fg_plan_get_cell_area.restype = C.c_int
    L.fg_plan_set_xgrid.argtypes = [vp, C.c_long] + [ip] * 5 + [dp] * 3
    L.fg_plan_set_xgrid.restype = C.c_int
    L.fg_plan_apply.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_double, C.c_int, vp, dp]
    L.fg_plan_apply.restype = C.c_int
    L.fg_plan_apply_interleaved.argtypes = [vp, C.c_int, vp, vp, vp, vp, dp]
    L.fg_plan_apply_interleaved.restype = C.c_int
    L.fg_plan_apply_records.argtypes = [vp, C.c_int, vp, vp, dp]
    L.fg_plan_apply_records.restype = C.c_int
    ao = C.POINTER(ApplyOpts)
    L.fg_plan_apply_ex.argtypes = [vp, ao, vp, vp, vp, vp, C.c_int, vp, dp]
    L.fg_plan_apply_ex.restype = C.c_int
    L.fg_plan_mono_begin.argtypes = [vp, ao, vp, vp, vp, vp]
    L.fg_plan_mono_begin.restype = C.c_int
    L.fg_plan_mono_minmax_dev.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    L.fg_plan_mono_minmax_dev.restype = C.c_int
    L.fg_plan_mono_copy_minmax.argtypes = [vp, C.c_int, vp, vp]
    L.fg_plan_mono_copy_minmax.restype = C.c_int
    L.fg_plan_mono_end.argtypes = [vp, ao, vp, vp, dp]
    L.fg_plan_mono_end.restype = C.c_int
    L.fg_plan_create_great_circle.argtypes = [C.c_int, ip, ip, dpp, dpp, dpp, C.c_int, C.c_int, dp, dp, C.c_int, C.POINTER(vp)]
    L.fg_plan_create_great_circle.restype = C.c_long
    L.fg_plan_create_great_circle_dev.argtypes = [C.c_int, ip, ip, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp),
                                                  C.c_int, C.c_int, vp, vp, vp, C.c_double, C.c_double, C.c_int, vp, C.c_int,
                                                  C.POINTER(vp)]
    L.fg_plan_create_great_circle_dev.restype = C.c_long
    L.fg_latlon2xyz.argtypes = [C.c_long, dp, dp, dp, dp, dp]
    L.fg_latlon2xyz.restype = None
    cip = C.POINTER(C.c_int)
    L.create_xgrid_great_circle.argtypes = [cip] * 4 + [dp] * 5 + [ip] * 4 + [dp] * 3
    L.create_xgrid_great_circle.restype = C.c_int
    L.get_grid_great_circle_area.argtypes = [cip, cip, dp, dp, dp]
    L.get_grid_great_circle_area.restype = None
    L.clip_2dx2d_great_circle.argtypes = [dp, dp, dp, C.c_int, dp, dp, dp, C.c_int, dp, dp, dp]
    L.clip_2dx2d_great_circle.restype = C.c_int
    L.great_circle_area.argtypes = [C.c_int, dp, dp, dp]
    L.great_circle_area.restype = C.c_double
    L.conserve_interp_great_circle.argtypes = [C.c_int] * 4 + [dp] * 7
    L.conserve_interp_great_circle.restype = None
    for nm in ("create_xgrid_1dx2d_order1", "create_xgrid_2dx1d_order1"):
        getattr(L, nm).argtypes = [cip] * 4 + [dp] * 5 + [ip] * 4 + [dp]
        getattr(L, nm).restype = C.c_int
    for nm in ("create_xgrid_1dx2d_order2", "create_xgrid_2dx1d_order2"):
        getattr(L, nm).argtypes = [cip] * 4 + [dp] * 5 + [ip] * 4 + [dp] * 3
        getattr(L, nm).restype = C.c_int
    L.clip.argtypes = [dp, dp, C.c_int] + [C.c_double] * 4 + [dp, dp]
    L.clip.restype = C.c_int
    L.box_ctrlat.argtypes = [C.c_double] * 4
    L.box_ctrlat.restype = C.c_double
    L.box_ctrlon.argtypes = [C.c_double] * 5
    L.box_ctrlon.restype = C.c_double
    L.get_grid_area_no_adjust.argtypes = [cip, cip, dp, dp, dp]
    L.get_grid_area_no_adjust.restype = None
    L.fg_sincos_batch.argtypes = [C.c_long, dp, dp, dp, C.c_int]
    L.fg_sincos_batch.restype = C.c_int
    L.fg_gc_clip_batch.argtypes = [C.c_int, dp, dp, dp, ip, dp, C.c_int]
    L.fg_gc_clip_batch.restype = C.c_int
    L.fg_plan_stream.argtypes = [vp]
    L.fg_plan_stream.restype = vp
    L.fg_plan_sync.argtypes = [vp]
    L.fg_plan_sync.restype = C.c_int
    L.fg_plan_stats.argtypes = [vp, C.POINTER(C.c_long), C.c_int]
    L.fg_plan_stats.restype = C.c_int
    L.fg_set_search_mode.argtypes = [C.c_int]
    L.fg_set_search_mode.restype = None
    L.fg_set_search_chunks.argtypes = [C.c_int]
    L.fg_set_search_chunks.restype = None
    for fn in ("fg_plan_ncells_out",):
        getattr(L, fn).argtypes = [vp]; getattr(L, fn).restype = C.c_long
    for fn in ("fg_plan_order", "fg_plan_device"):
        getattr(L, fn).argtypes = [vp]; getattr(L, fn).restype = C.c_int
    lp, cp = C.POINTER(C.c_long), C.c_char_p
    L.fg_nc_open.argtypes = [cp, C.POINTER(vp)]; L.fg_nc_open.restype = C.c_int
    L.fg_nc_create.argtypes = [cp, C.c_int, C.POINTER(vp)]; L.fg_nc_create.restype = C.c_int
    L.fg_nc_def_dim.argtypes = [vp, cp, C.c_long]; L.fg_nc_def_dim.restype = C.c_int
    L.fg_nc_def_var.argtypes = [vp, cp, C.c_int, C.c_int, ip]; L.fg_nc_def_var.restype = C.c_int
    L.fg_nc_put_att_text.argtypes = [vp, C.c_int, cp, cp]; L.fg_nc_put_att_text.restype = C.c_int
    L.fg_nc_put_att_double.argtypes = [vp, C.c_int, cp, C.c_int, C.c_int, dp]; L.fg_nc_put_att_double.restype = C.c_int
    L.fg_nc_enddef.argtypes = [vp]; L.fg_nc_enddef.restype = C.c_int
    L.fg_nc_inq_ndims.argtypes = [vp]; L.fg_nc_inq_ndims.restype = C.c_int
    L.fg_nc_inq_nvars.argtypes = [vp]; L.fg_nc_inq_nvars.restype = C.c_int
    L.fg_nc_inq_numrecs.argtypes = [vp]; L.fg_nc_inq_numrecs.restype = C.c_long
    L.fg_nc_inq_dimid.argtypes = [vp, cp]; L.fg_nc_inq_dimid.restype = C.c_int
    L.fg_nc_inq_dim.argtypes = [vp, C.c_int, cp, C.c_int, lp]; L.fg_nc_inq_dim.restype = C.c_int
    L.fg_nc_inq_varid.argtypes = [vp, cp]; L.fg_nc_inq_varid.restype = C.c_int
    L.fg_nc_inq_var.argtypes = [vp, C.c_int, cp, C.c_int, ip, ip, ip, lp]; L.fg_nc_inq_var.restype = C.c_int
    L.fg_nc_get_att_double.argtypes = [vp, C.c_int, cp, dp, C.c_int]; L.fg_nc_get_att_double.restype = C.c_int
    L.fg_nc_get_att_text.argtypes = [vp, C.c_int, cp, cp, C.c_int]; L.fg_nc_get_att_text.restype = C.c_int
    for fn in ("fg_nc_get_vara", "fg_nc_get_vara_double", "fg_nc_put_vara", "fg_nc_put_vara_double"):
        getattr(L, fn).argtypes = [vp, C.c_int, lp, lp, vp]; getattr(L, fn).restype = C.c_int
    L.fg_nc_close.argtypes = [vp]; L.fg_nc_close.restype = C.c_int
    L.fg_nc_last_error.restype = cp
    L.fg_sweep_create.argtypes = [C.c_int, C.POINTER(vp), vp, C.c_int, C.c_int, C.POINTER(vp)]; L.fg_sweep_create.restype = C.c_int
    L.fg_sweep_run.argtypes = [vp, vp, C.c_long, C.c_double, C.c_double, C.c_double, C.POINTER(vp)]; L.fg_sweep_run.restype = C.c_int
    L.fg_sweep_destroy.argtypes = [vp]; L.fg_sweep_destroy.restype = None
    L.fg_host_alloc.argtypes = [C.c_size_t]; L.fg_host_alloc.restype = vp
    L.fg_host_free.argtypes = [vp]; L.fg_host_free.restype = None
    L.fg_set_profiling.argtypes = [C.c_int]
    L.fg_set_profiling.restype = None
    L.fg_plan_phase_ms.argtypes = [vp, C.POINTER(C.c_float), C.c_int]
    L.fg_plan_phase_ms.restype = C.c_int
    lp = C.POINTER(C.c_long)
    L.fg_c2l_create.argtypes = [C.c_int, ip, ip, dpp, dpp, dpp, dpp, C.c_int] + [ip] * 10 + [C.c_int, C.POINTER(vp)]
    L.fg_c2l_create.restype = C.c_int
    L.fg_c2l_destroy.argtypes = [vp]
    L.fg_c2l_destroy.restype = None
    L.fg_c2l_ncells.argtypes = [vp]
    L.fg_c2l_ncells.restype = C.c_long
    L.fg_c2l_halo_size.argtypes = [vp]
    L.fg_c2l_halo_size.restype = C.c_long
    L.fg_c2l_set_stream.argtypes = [vp, vp]
    L.fg_c2l_set_stream.restype = C.c_int
    L.fg_c2l_sync.argtypes = [vp]
    L.fg_c2l_sync.restype = C.c_int
    L.fg_c2l_get_centres.argtypes = [vp, dp, dp]
    L.fg_c2l_get_centres.restype = C.c_int
    L.fg_c2l_fill_halo.argtypes = [vp, vp, vp, C.c_int]
    L.fg_c2l_fill_halo.restype = C.c_int
    L.fg_c2l_gradient.argtypes = [vp, vp, C.c_int, C.c_int, C.c_double, vp, vp, vp]
    L.fg_c2l_gradient.restype = C.c_int
    L.fg_c2l_gradient_records.argtypes = [vp, vp, C.c_int, vp]
    L.fg_c2l_gradient_records.restype = C.c_int
    L.fg_c2l_records.argtypes = [vp, vp, C.c_int, vp]
    L.fg_c2l_records.restype = C.c_int
    L.fg_c2l_grid_info.argtypes = [C.c_int, C.c_int] + [dp] * 15
    L.fg_c2l_grid_info.restype = C.c_int
    L.fg_find_contacts.argtypes = [C.c_int, ip, ip, dpp, dpp, C.c_int] + [ip] * 10
    L.fg_find_contacts.restype = C.c_int
    L.fg_halo_map.argtypes = [C.c_int, ip, ip, C.c_int] + [ip] * 10 + [lp, ip]
    L.fg_halo_map.restype = C.c_int
    L.fg_tripolar_corners.argtypes = [C.c_int, C.c_int] + [C.c_double] * 5 + [dp, dp]
    L.fg_tripolar_corners.restype = C.c_int
    L.fg_gnomonic_ed_grid.argtypes = [C.c_int, C.c_double, C.c_int, dp, dp, dp, dp]
    L.fg_gnomonic_ed_grid.restype = C.c_int
    L.fg_remap_write.argtypes = [C.c_char_p, C.c_int, C.c_long, ip, ip, ip, dp, dp]
    L.fg_remap_write.restype = C.c_int
    L.fg_remap_write_interp.argtypes = [C.c_char_p, C.c_int, C.c_long] + [ip] * 5 + [dp] * 3 + [C.c_int, C.c_int]
    L.fg_remap_write_interp.restype = C.c_int
    L.fg_remap_read_size.argtypes = [C.c_char_p]
    L.fg_remap_read_size.restype = C.c_long
    L.fg_remap_read.argtypes = [C.c_char_p, C.c_int, C.c_long] + [ip] * 5 + [dp] * 3
    L.fg_remap_read.restype = C.c_int
    L.fg_remap_last_error.restype = C.c_char_p
    L.fg_gnomonic_ed_corners.argtypes = [C.c_int, C.c_double, C.c_int, dp, dp]
    L.fg_gnomonic_ed_corners.restype = C.c_int
    L.fg_latlon_corners.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, dp, dp]
    L.fg_latlon_corners.restype = C.c_int
    _LIB = L
    return L


def last_error():
    return lib().fg_last_error().decode("utf-8", "replace")


def check(code):
    if code < 0:
        raise FregridHipError(code, last_error())
    return code


def require_gpu():
    n = lib().fg_device_count()
    if n < 1:
        raise FregridHipError(-2, "no HIP device visible (" + last_error() + "): the regrid hot path runs on an "
                              "MI355X-class GPU only; there is no CPU fallback")
    return n
