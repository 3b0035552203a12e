"""ctypes binding of libpcp.so (C ABI declared in include/pcp.h).

The library is the only compute backend: when it is missing or a call fails,
an exception is raised -- there is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PCP_LIB_PATH") or os.path.join(_HERE, "libpcp.so")

_lib = None


class PcpError(RuntimeError):
    pass


class VoxelFrame(C.Structure):
    """pcp_voxel_frame of include/pcp.h: what one fed frame did to the voxel map."""
    _fields_ = [(n, C.c_int) for n in ("serial", "stored_before", "stored", "unknown", "created", "staged", "key_bits", "flags")]


def _ptr(arr, ctype):
    return arr.ctypes.data_as(C.POINTER(ctype))


# name -> (restype, argtypes); every symbol include/pcp.h declares
_f32p, _f64p, _u32p, _i32p = (C.POINTER(C.c_float), C.POINTER(C.c_double),
                              C.POINTER(C.c_uint32), C.POINTER(C.c_int))
_ctx = C.c_void_p
SIGNATURES = {
    "pcp_last_error": (C.c_char_p, []),
    "pcp_version": (C.c_int, []),
    "pcp_device_count": (C.c_int, [_i32p]),
    "pcp_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(_ctx)]),
    "pcp_destroy": (C.c_int, [_ctx]),
    "pcp_set_stream": (C.c_int, [_ctx, C.c_void_p]),
    "pcp_synchronize": (C.c_int, [_ctx]),
    "pcp_set_target": (C.c_int, [_ctx, _f32p, C.c_int, _u32p, C.c_int, _f32p]),
    "pcp_reset_hints": (C.c_int, [_ctx]),
    "pcp_normals": (C.c_int, [_ctx, _f32p]),
    "pcp_nn": (C.c_int, [_ctx, _f32p, C.c_int, _u32p, _f32p, _f32p]),
    "pcp_nn_staged": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "pcp_ls_staged": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p]),
    "pcp_nn_kernel_ms": (C.c_int, [_ctx, _f32p]),
    "pcp_ls_point": (C.c_int, [_ctx, _f32p, C.c_int, C.c_int, _f32p]),
    "pcp_ls_plane": (C.c_int, [_ctx, _f32p, C.c_int, _f32p]),
    "pcp_ls_tiles": (C.c_int, [_ctx, _f32p, C.c_int, C.c_int, C.c_int, _f32p]),
    "pcp_hb_sum": (C.c_int, [_ctx, _f32p, C.c_int, _f32p]),
    "pcp_hb_sum_dev": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "pcp_hb_sum_static": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_void_p]),
    "pcp_icp_run": (C.c_int, [_ctx, _f64p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                              _f32p, _f32p, _i32p, _i32p, _f64p]),
    "pcp_upload_source": (C.c_int, [_ctx, _f64p, C.c_int]),
    "pcp_upload_source_soa": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "pcp_upload_source_dev": (C.c_int, [_ctx, C.c_void_p, C.c_int]),
    "pcp_icp_launch": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_float, C.c_int]),
    "pcp_icp_fetch": (C.c_int, [_ctx, _f32p, _f32p, _i32p, _i32p, _f64p]),
    "pcp_icp_fetch_pose": (C.c_int, [_ctx, _f64p, _f64p]),
    "pcp_pose_from_steps": (C.c_int, [_ctx, _f32p, C.c_int, _f64p, _f64p]),
    "pcp_prepare_resident": (C.c_int, [_ctx, C.c_int]),
    "pcp_voxel_index_upload": (C.c_int, [_ctx, _i32p, C.c_int]),
    "pcp_voxel_data_upload": (C.c_int, [_ctx, _i32p, C.c_int, C.c_int]),
    "pcp_voxel_data_upload_rows": (C.c_int, [_ctx, _i32p, C.c_int, C.c_int, C.c_int]),
    "pcp_voxel_data_read": (C.c_int, [_ctx, C.c_int, C.c_int, _i32p]),
    "pcp_voxelize": (C.c_int, [_ctx, _f32p, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, _i32p, C.c_int, _f32p]),
    "pcp_voxelize_xyz": (C.c_int, [_ctx, _f32p, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, _i32p, C.c_int, _f32p]),
    "pcp_voxelize_rows": (C.c_int, [_ctx, _f32p, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, _i32p, C.c_int, _f32p]),
    "pcp_voxel_stage": (C.c_int, [_ctx, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _i32p, _i32p]),
    "pcp_voxel_stage_resident": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _i32p, _i32p]),
    "pcp_voxel_stat_read": (C.c_int, [_ctx, C.c_int, _f32p]),
    "pcp_voxel_store_enable": (C.c_int, [_ctx, C.c_int]),
    "pcp_voxel_staged_points": (C.c_int, [_ctx, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), _i32p]),
    "pcp_voxel_log_begin": (C.c_int, [_ctx, C.c_int, C.c_size_t, C.c_void_p]),
    "pcp_voxel_log_end": (C.c_int, [_ctx, C.c_int]),
    "pcp_voxel_log_take": (C.c_int, [_ctx, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), _i32p]),
    "pcp_grid_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "pcp_grid_destroy": (C.c_int, [C.c_void_p]),
    "pcp_grid_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pcp_grid_synchronize": (C.c_int, [C.c_void_p]),
    "pcp_grid_build": (C.c_int, [C.c_void_p, _f64p, C.c_size_t, C.c_double, _f32p]),
    "pcp_grid_append": (C.c_int, [C.c_void_p, _f64p, C.c_size_t, _f32p, C.c_longlong]),
    "pcp_grid_append_slab": (C.c_int, [C.c_void_p, _f64p, C.c_size_t, _f32p, C.c_double, C.c_ulonglong, C.c_ulonglong, C.c_longlong,
                                       C.POINTER(C.c_size_t)]),
    "pcp_grid_collect_begin": (C.c_int, [C.c_void_p]),
    "pcp_grid_collect_end": (C.c_int, [C.c_void_p]),
    "pcp_grid_defer_appends": (C.c_int, [C.c_void_p, C.c_int]),
    "pcp_grid_append_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_longlong]),
    "pcp_grid_append_slab_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_double, C.c_ulonglong, C.c_ulonglong,
                                           C.c_longlong, C.POINTER(C.c_size_t)]),
    "pcp_grid_partition": (C.c_int, [C.c_int, _f64p, C.c_size_t, C.c_double, C.c_int, C.POINTER(C.c_ulonglong)]),
    "pcp_grid_build_slab": (C.c_int, [C.c_void_p, _f64p, C.c_size_t, C.c_double, _f32p, C.c_double, C.c_ulonglong, C.c_ulonglong, C.POINTER(C.c_size_t)]),
    "pcp_grid_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), _i32p, _i32p, _f32p, _f32p]),
    "pcp_grid_sorted_ids": (C.c_int, [C.c_void_p, C.POINTER(C.c_longlong), _f64p]),
    "pcp_grid_estimate_normals": (C.c_int, [C.c_void_p, C.c_double, C.c_int, _f64p, _f32p, C.POINTER(C.c_uint)]),
    "pcp_grid_source": (C.c_int, [C.c_void_p, _f64p, C.c_size_t]),
    "pcp_grid_source_rewind": (C.c_int, [C.c_void_p]),
    "pcp_grid_source_read": (C.c_int, [C.c_void_p, _f64p]),
    "pcp_grid_source_apply": (C.c_int, [C.c_void_p, _f64p, _f64p, C.c_int]),
    "pcp_grid_query_resident": (C.c_int, [C.c_void_p, C.c_double, C.POINTER(C.c_longlong), _f64p, _f32p]),
    "pcp_grid_query": (C.c_int, [C.c_void_p, _f64p, C.c_size_t, C.c_double, C.POINTER(C.c_longlong), _f64p]),
    "pcp_grid_kd_step": (C.c_int, [C.c_void_p, C.c_double, _f64p, _f64p, _f64p, C.POINTER(C.c_ulonglong)]),
    "pcp_grid_nn_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_longlong), _f64p, C.POINTER(C.c_ubyte)]),
    "pcp_grid_ls": (C.c_int, [C.c_void_p, C.c_int, C.c_double, _f32p]),
    "pcp_grid_icp_run": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_float, C.c_int, _f32p, _f32p, _i32p, _i32p]),
    "pcp_grid_set_owned_keys": (C.c_int, [C.c_void_p, C.c_double, C.c_ulonglong, C.c_ulonglong]),
    "pcp_grid_cell_keys": (C.c_int, [C.c_double, _f64p, C.c_size_t, C.POINTER(C.c_ulonglong)]),
    "pcp_grid_icp_reset": (C.c_int, [C.c_void_p]),
    "pcp_grid_ls_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_void_p]),
    "pcp_grid_pose_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int]),
    "pcp_grid_icp_finish": (C.c_int, [C.c_void_p]),
    "pcp_grid_icp_fetch": (C.c_int, [C.c_void_p, _f32p, _f32p, _i32p, _i32p, _i32p]),
    "pcp_comm_unique_id": (C.c_int, [C.c_char_p]),
    "pcp_comm_init": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_void_p)]),
    "pcp_comm_destroy": (C.c_int, [C.c_void_p]),
    "pcp_comm_abort": (C.c_int, [C.c_void_p]),
    "pcp_comm_info": (C.c_int, [C.c_void_p, _i32p, _i32p]),
    "pcp_comm_allreduce_tile": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "pcp_grid_allreduce_tile": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "pcp_grid_icp_run_comm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_float, C.c_int, _f32p, _f32p,
                                        _i32p, _i32p]),
    "pcp_set_target_dev": (C.c_int, [_ctx, C.c_void_p, C.c_int, _u32p, C.c_int, _f32p]),
    "pcp_frame_create": (C.c_int, [C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]),
    "pcp_frame_destroy": (C.c_int, [C.c_void_p]),
    "pcp_frame_decode": (C.c_int, [C.c_void_p, _f32p, C.c_size_t, C.c_float, _f64p, C.POINTER(C.c_size_t)]),
    "pcp_frame_upload": (C.c_int, [C.c_void_p, _f64p, C.c_size_t]),
    "pcp_frame_upload_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "pcp_frame_move": (C.c_int, [C.c_void_p, _f64p, _f64p]),
    "pcp_frame_scan_lines": (C.c_int, [C.c_void_p, _f64p, C.c_int, _u32p, _i32p]),
    "pcp_frame_read": (C.c_int, [C.c_void_p, _f64p, _f32p]),
    "pcp_frame_points_dev": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "pcp_frame_padded_dev": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "pcp_voxel_insert_unknown": (C.c_int, [_ctx, C.c_int, _i32p, _i32p]),
    "pcp_voxel_add": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _i32p]),
    "pcp_voxel_add_dev": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_void_p, _i32p]),
    "pcp_voxel_add_wait": (C.c_int, [_ctx, C.c_int, C.c_void_p]),
    "pcp_voxel_add_unknown_read": (C.c_int, [_ctx, _i32p, C.c_int, _i32p]),
    "pcp_voxel_staged_read": (C.c_int, [_ctx, C.c_int, C.c_int, _i32p]),
    "pcp_voxel_store_read": (C.c_int, [_ctx, C.c_longlong, C.c_int, _f32p]),
    "pcp_voxel_reset": (C.c_int, [_ctx]),
    "pcp_voxel_reserve": (C.c_int, [_ctx, C.c_int, C.c_int]),
    "pcp_voxel_index_read": (C.c_int, [_ctx, C.c_int, _i32p]),
    "pcp_voxel_counts_read": (C.c_int, [_ctx, C.c_int, C.c_int, _i32p]),
    "pcp_batch_create": (C.c_int, [C.POINTER(_ctx), C.c_int, C.POINTER(C.c_void_p)]),
    "pcp_batch_destroy": (C.c_int, [C.c_void_p]),
    "pcp_batch_prepare": (C.c_int, [C.c_void_p, C.c_int]),
    "pcp_batch_icp_launch": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int]),
    "pcp_batch_synchronize": (C.c_int, [C.c_void_p]),
    "pcp_batch_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "pcp_batch_profile_read": (C.c_int, [C.c_void_p, _f32p, _i32p, C.POINTER(C.c_longlong)]),
    "pcp_seq_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "pcp_seq_destroy": (C.c_int, [C.c_void_p]),
    "pcp_seq_submit": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), _i32p, C.c_int, C.c_float, _f64p, _f64p, C.c_int, C.c_int,
                                 C.c_float, C.c_int, _i32p]),
    "pcp_seq_submit_placed": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), _i32p, C.c_int, C.c_float, _f64p, _f64p, _f64p, C.c_int,
                                        C.c_int, C.c_float, C.c_int, _i32p]),
    "pcp_seq_collect": (C.c_int, [C.c_void_p, C.c_int, _f32p, C.c_int, _i32p, _i32p, _i32p]),
    "pcp_seq_collect_poses": (C.c_int, [C.c_void_p, C.c_int, _f64p, _f64p, _f32p, C.c_int, _i32p, _i32p, _i32p]),
    "pcp_seq_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "pcp_seq_profile_read": (C.c_int, [C.c_void_p, _f32p, _i32p, C.POINTER(C.c_longlong)]),
    "pcp_seq_profile_read_launches": (C.c_int, [C.c_void_p, _f32p, C.c_int, _i32p]),
    "pcp_chain_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "pcp_chain_destroy": (C.c_int, [C.c_void_p]),
    "pcp_chain_reset": (C.c_int, [C.c_void_p]),
    "pcp_chain_submit": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), _i32p, C.c_int, C.c_float, _f64p, _f64p, C.c_int, C.c_int,
                                   C.c_float, C.c_int, C.c_int, _ctx, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   _i32p, _i32p, _i32p]),
    "pcp_chain_collect": (C.c_int, [C.c_void_p, C.c_int, _f64p, _f64p, _f64p, _f64p, _f32p, C.c_int, _i32p, _i32p, _f32p]),
    "pcp_chain_read_refined": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _f64p, C.c_int]),
    "pcp_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "pcp_host_free": (C.c_int, [C.c_void_p]),
    "pcp_ring_stats": (C.c_int, [_ctx, _f32p]),
    "pcp_profile_enable": (C.c_int, [_ctx, C.c_int]),
    "pcp_profile_read": (C.c_int, [_ctx, _f32p, _i32p]),
    "pcp_timer_start": (C.c_int, [_ctx]),
    "pcp_timer_stop": (C.c_int, [_ctx, _f32p]),
}


def _share_torch_hip_runtime():
    """PyTorch wheels bundle their own libamdhip64 (soname libamdhip64.so.7, the one libpcp.so asks
    for).  A process that ends up with two HIP runtimes -- libpcp.so's from /opt/rocm loaded first,
    torch's afterwards -- finds no GPU in the second one, so when torch is installed its copy is
    loaded first and serves both."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    path = os.path.join(libdir, "libamdhip64.so")
    if os.path.exists(path):
        C.CDLL(path, mode=C.RTLD_GLOBAL)
    # the same for RCCL: pcp_comm_* binds it at run time and is pointed at PyTorch's copy, so that a process
    # using both torch.distributed and the library's own communicator holds one RCCL
    rccl = os.path.join(libdir, "librccl.so")
    if os.path.exists(rccl):
        os.environ.setdefault("PCP_RCCL_PATH", rccl)


def load():
    """Loads libpcp.so once; raises PcpError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PcpError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    _share_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header and library disagree
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise PcpError(load().pcp_last_error().decode("utf-8", "replace"))


def f32p(a):
    return None if a is None else _ptr(a, C.c_float)


def f64p(a):
    return None if a is None else _ptr(a, C.c_double)


def u32p(a):
    return None if a is None else _ptr(a, C.c_uint32)


def i32p(a):
    return None if a is None else _ptr(a, C.c_int)


def as_f32c(a):
    return np.ascontiguousarray(a, dtype=np.float32)
