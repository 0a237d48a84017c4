"""ctypes binding of include/frequensee.h (libfrequensee.so, HIP/gfx950).

There is no fallback: if the shared library is missing or a HIP device is unavailable every compute
call raises.  The product never imports oracle/.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfrequensee.so")

MAX_BANDS = 8
NO_MATERIAL = 0xFFFF
MAX_DEPTH = 64

OK = 0
ERR_INVALID_ARGUMENT = 1
ERR_NO_DEVICE = 2
ERR_HIP = 3
ERR_NOT_COMMITTED = 4
ERR_BAD_HANDLE = 5
ERR_SIZE_MISMATCH = 6
ERR_OUT_OF_MEMORY = 7

FLAG_FIXED_NORM_1000 = 1
FLAG_FLUSH_BEFORE_RECONSTRUCT = 2
FLAG_COSINE_SAMPLING = 4
FLAG_DETERMINISTIC = 8
FLAG_ALL_CONNECTIONS = 16
FLAG_MIS_BALANCE = 32
FLAG_MATERIAL_LOBES = 64
FLAG_ACCUMULATE_ENERGY = 128
FLAG_DOUBLE_POSITIONS = 256
NO_OBJECT = 0xFFFFFFFF

# every symbol include/frequensee.h declares (tests check the library exports all of them)
EXPORTS = [
    "fs_config_default", "fs_params_default", "fs_abi_version", "fs_context_create", "fs_context_destroy",
    "fs_last_error", "fs_context_advice", "fs_scene_set_triangles", "fs_scene_set_materials", "fs_scene_commit", "fs_source_create",
    "fs_source_destroy", "fs_source_set_position", "fs_listener_set_position", "fs_source_set_object", "fs_listener_set_object", "fs_compute_energy_response",
    "fs_compute_energy_response_async", "fs_compute_energy_response_batch_async", "fs_energy_device_ptr", "fs_reconstruct_impulse_response",
    "fs_reconstruct_impulse_response_async", "fs_reconstruct_impulse_response_batch_async", "fs_update_sources", "fs_synchronize", "fs_get_impulse_response", "fs_get_impulse_response_sequence",
    "fs_copy_impulse_response", "fs_copy_band_impulse_response", "fs_get_energy_buffer", "fs_flush_energy_buffer",
    "fs_add_energy_at_delay", "fs_update_energy_buffer", "fs_num_bins", "fs_num_samples", "fs_trace_rays",
    "fs_set_profiling", "fs_set_profiling_interval", "fs_get_stats", "fs_reset_stats", "fs_get_pipeline_counters", "fs_get_streams",
    "fs_sound_params_default", "fs_scene_set_objects", "fs_update_sound", "fs_get_occlusion_attenuation",
    "fs_save_array_to_file", "fs_load_float_array", "fs_save_impulse_response",
    "fs_reverb_init", "fs_reverb_process", "fs_reverb_release",
    "fs_apply_material_fd", "fs_energy_handoff", "fs_scene_update_triangles", "fs_scene_refit", "fs_set_impulse_response",
    "fs_scene_commit_fast", "fs_comm_unique_id", "fs_comm_init", "fs_comm_attach", "fs_comm_detach", "fs_comm_info", "fs_comm_enable_oneshot", "fs_shard_range",
    "fs_peers_init", "fs_peers_detach", "fs_gather_energy", "fs_gather_energy_async", "fs_set_pipelining", "fs_set_walk_stages", "fs_set_frames_per_launch", "fs_submit", "fs_scene_commit_progressive", "fs_scene_refine_pending", "fs_scene_refine_wait",
]
COMM_ID_BYTES = 128
ERR_COMM = 8
ERR_OVERFLOW = 9
REVERB_LITERAL_TAIL = 1


class SoundParams(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("raycasts_per_tick", C.c_int32),
        ("seed", C.c_uint64),
        ("raycast_bounces", C.c_int32),
        ("raycast_distance", C.c_float),
        ("simulated_duration", C.c_float),
        ("listener_radius", C.c_float),
    ]


class SoundResult(C.Structure):
    _fields_ = [
        ("total_energy", C.c_float),
        ("occlusion_attenuation", C.c_float),
        ("direct_energy_sum", C.c_float),
        ("rays_reaching_listener", C.c_uint32),
        ("direct_hits", C.c_uint32),
        ("traces", C.c_uint64),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class Config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("device", C.c_int32),
        ("num_bands", C.c_int32),
        ("sample_rate", C.c_int32),
        ("num_channels", C.c_int32),
        ("simulated_duration", C.c_float),
        ("bin_duration", C.c_float),
        ("rank", C.c_int32),
        ("world_size", C.c_int32),
        ("stream", C.c_void_p),
    ]


class Params(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("flags", C.c_uint32),
        ("seed", C.c_uint64),
        ("num_rays", C.c_uint32),
        ("depth", C.c_int32),
        ("russian_roulette", C.c_int32),
        ("rr_prob", C.c_float),
        ("max_trace_dist", C.c_float),
        ("surface_offset", C.c_float),
        ("connect_pullback", C.c_float),
        ("dist_divisor", C.c_float),
        ("min_seg", C.c_float),
        ("prob_exponent", C.c_float),
        ("energy_clamp", C.c_float),
        ("energy_gain", C.c_float),
        ("sound_speed", C.c_float),
        ("air_absorption", C.c_float * MAX_BANDS),
        ("samples_per_bin", C.c_int32),
        ("listener_radius", C.c_float),
        ("source_radius", C.c_float),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("frames", C.c_uint64),
        ("rays", C.c_uint64),
        ("pairs", C.c_uint64),
        ("walk_kernel_ms_sum", C.c_double),
        ("walk_kernel_ms_last", C.c_double),
        ("connect_kernel_ms_sum", C.c_double),
        ("reconstruct_ms_sum", C.c_double),
        ("timed_frames", C.c_uint64),
        ("timed_connects", C.c_uint64),
        ("timed_reconstructs", C.c_uint64),
        ("bvh_nodes", C.c_uint32),
        ("triangles", C.c_uint32),
        ("bvh_stack_need", C.c_uint32),
        ("bvh_depth", C.c_uint32),
        ("scene_bytes", C.c_uint64),
        ("segments", C.c_uint64),
        ("connections_tested", C.c_uint64),
        ("deposits", C.c_uint64),
        ("walk_node_fetches", C.c_uint64),
        ("walk_tri_fetches", C.c_uint64),
        ("any_node_fetches", C.c_uint64),
        ("any_tri_fetches", C.c_uint64),
        ("node_request_insts", C.c_uint64),
        ("node_request_lanes", C.c_uint64),
        ("node_request_distinct", C.c_uint64),
        ("planned_segments", C.c_uint64),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class PipelineCounters(C.Structure):
    """fs_pipeline_counters (include/frequensee.h)"""
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("reserved", C.c_uint32),
        ("fused_launches", C.c_uint64),
        ("flushes", C.c_uint64),
        ("flushed_frames", C.c_uint64),
        ("host_waits", C.c_uint64),
        ("host_wait_us", C.c_uint64),
        ("stream_waits_enqueued", C.c_uint64),
        ("stream_waits_skipped", C.c_uint64),
        ("tail_stream_ops", C.c_uint64),
        ("owed_on_tail", C.c_uint64),
        ("publishes_by_word", C.c_uint64),
        ("publishes_by_event", C.c_uint64),
        ("lane_launches", C.c_uint64),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k not in ("struct_size", "reserved")}


class FrequenSeeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"frequensee status {code}: {msg}")
        self.code = code


_lib = None


def load():
    """Load libfrequensee.so; raises if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            f"{LIB_PATH} not found: build it with `python __graft_entry__.py build` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    # Host requirement of the library (fs_context_advice): 16 hardware queues, decided before the
    # HIP runtime initialises — this binding is the host here.  An explicit setting of the process wins.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    lib = C.CDLL(LIB_PATH)
    vp, i32, u16p, f32p = C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p
    sig = {
        "fs_config_default": (None, [C.POINTER(Config)]),
        "fs_params_default": (None, [C.POINTER(Params)]),
        "fs_abi_version": (C.c_int, []),
        "fs_context_create": (C.c_int, [C.POINTER(Config), C.POINTER(vp)]),
        "fs_context_destroy": (C.c_int, [vp]),
        "fs_last_error": (C.c_char_p, [vp]),
        "fs_context_advice": (C.c_char_p, [vp]),
        "fs_scene_set_triangles": (C.c_int, [vp, f32p, u16p, i32]),
        "fs_scene_set_materials": (C.c_int, [vp, f32p, f32p, f32p, i32, i32]),
        "fs_scene_commit": (C.c_int, [vp]),
        "fs_scene_commit_fast": (C.c_int, [vp]),
        "fs_source_create": (C.c_int, [vp, C.POINTER(i32)]),
        "fs_source_destroy": (C.c_int, [vp, i32]),
        "fs_source_set_position": (C.c_int, [vp, i32, C.POINTER(C.c_float)]),
        "fs_listener_set_position": (C.c_int, [vp, C.POINTER(C.c_float)]),
        "fs_source_set_object": (C.c_int, [vp, i32, C.c_uint32]),
        "fs_listener_set_object": (C.c_int, [vp, C.c_uint32]),
        "fs_compute_energy_response": (C.c_int, [vp, i32, C.POINTER(Params), f32p]),
        "fs_compute_energy_response_async": (C.c_int, [vp, i32, C.POINTER(Params)]),
        "fs_compute_energy_response_batch_async": (C.c_int, [vp, C.POINTER(C.c_int32), i32, C.POINTER(Params)]),
        "fs_energy_device_ptr": (C.c_int, [vp, i32, C.POINTER(vp), C.POINTER(C.c_size_t)]),
        "fs_scene_update_triangles": (C.c_int, [vp, i32, i32, f32p]),
        "fs_scene_refit": (C.c_int, [vp]),
        "fs_set_impulse_response": (C.c_int, [vp, i32, f32p, i32]),
        "fs_energy_handoff": (C.c_int, [vp, i32, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(vp)]),
        "fs_reconstruct_impulse_response": (C.c_int, [vp, i32, C.POINTER(Params)]),
        "fs_reconstruct_impulse_response_async": (C.c_int, [vp, i32, C.POINTER(Params)]),
        "fs_reconstruct_impulse_response_batch_async": (C.c_int, [vp, C.POINTER(C.c_int32), i32, C.POINTER(Params)]),
        "fs_update_sources": (C.c_int, [vp, C.POINTER(C.c_int32), i32, C.POINTER(Params)]),
        "fs_synchronize": (C.c_int, [vp]),
        "fs_get_impulse_response": (C.c_int, [vp, i32, i32, C.POINTER(C.POINTER(C.c_float)), C.POINTER(i32)]),
        "fs_copy_impulse_response": (C.c_int, [vp, i32, i32, f32p, i32]),
        "fs_get_impulse_response_sequence": (C.c_int, [vp, i32, C.POINTER(C.c_uint64)]),
        "fs_copy_band_impulse_response": (C.c_int, [vp, i32, i32, f32p, i32]),
        "fs_get_energy_buffer": (C.c_int, [vp, i32, f32p, i32]),
        "fs_flush_energy_buffer": (C.c_int, [vp, i32]),
        "fs_add_energy_at_delay": (C.c_int, [vp, i32, i32, C.c_float, C.c_float]),
        "fs_update_energy_buffer": (C.c_int, [vp, i32, f32p, i32]),
        "fs_num_bins": (C.c_int, [vp]),
        "fs_num_samples": (C.c_int, [vp]),
        "fs_trace_rays": (C.c_int, [vp, f32p, f32p, f32p, i32, i32, vp, f32p, vp, f32p]),
        "fs_set_profiling": (C.c_int, [vp, i32]),
        "fs_set_profiling_interval": (C.c_int, [vp, i32]),
        "fs_get_stats": (C.c_int, [vp, C.POINTER(Stats)]),
        "fs_reset_stats": (C.c_int, [vp]),
        "fs_get_pipeline_counters": (C.c_int, [vp, C.POINTER(PipelineCounters)]),
        "fs_get_streams": (C.c_int, [vp, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
        "fs_sound_params_default": (None, [C.POINTER(SoundParams)]),
        "fs_scene_set_objects": (C.c_int, [vp, vp, i32]),
        "fs_update_sound": (C.c_int, [vp, i32, C.POINTER(SoundParams), C.POINTER(SoundResult)]),
        "fs_get_occlusion_attenuation": (C.c_int, [vp, i32, C.POINTER(C.c_float)]),
        "fs_save_array_to_file": (C.c_int, [f32p, i32, C.c_char_p]),
        "fs_load_float_array": (C.c_int, [C.c_char_p, f32p, i32, C.POINTER(i32)]),
        "fs_save_impulse_response": (C.c_int, [vp, i32, i32, C.c_char_p]),
        "fs_reverb_init": (C.c_int, [vp, i32, i32]),
        "fs_reverb_process": (C.c_int, [vp, i32, f32p, f32p, i32, C.c_uint32]),
        "fs_reverb_release": (C.c_int, [vp, i32]),
        "fs_apply_material_fd": (C.c_int, [vp, f32p, i32, f32p, f32p, f32p, i32, f32p, f32p, f32p]),
        "fs_comm_unique_id": (C.c_int, [vp, C.c_size_t]),
        "fs_comm_init": (C.c_int, [vp, vp, C.c_size_t]),
        "fs_comm_attach": (C.c_int, [vp, vp]),
        "fs_comm_detach": (C.c_int, [vp]),
        "fs_comm_info": (C.c_int, [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
        "fs_comm_enable_oneshot": (C.c_int, [vp]),
        "fs_shard_range": (C.c_int, [C.c_uint32, i32, i32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
        "fs_peers_init": (C.c_int, [vp, vp, C.c_size_t, i32, i32]),
        "fs_peers_detach": (C.c_int, [vp]),
        "fs_set_pipelining": (C.c_int, [vp, i32]),
        "fs_set_walk_stages": (C.c_int, [vp, vp, i32]),
        "fs_set_frames_per_launch": (C.c_int, [vp, i32]),
        "fs_submit": (C.c_int, [vp]),
        "fs_scene_commit_progressive": (C.c_int, [vp]),
        "fs_scene_refine_pending": (C.c_int, [vp, C.POINTER(i32)]),
        "fs_scene_refine_wait": (C.c_int, [vp]),
        "fs_gather_energy": (C.c_int, [vp, i32, f32p, i32]),
        "fs_gather_energy_async": (C.c_int, [vp, i32, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def default_config(**kw) -> Config:
    c = Config()
    load().fs_config_default(C.byref(c))
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def default_sound_params(**kw) -> SoundParams:
    p = SoundParams()
    load().fs_sound_params_default(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def default_params(**kw) -> Params:
    p = Params()
    load().fs_params_default(C.byref(p))
    for k, v in kw.items():
        if k == "air_absorption":
            for i, x in enumerate(v):
                p.air_absorption[i] = float(x)
        else:
            setattr(p, k, v)
    return p


def save_array_to_file(array, path):
    """SaveArrayToFile (FSAC.cpp:492-505): one SanitizeFloat'ed value per line"""
    import numpy as np
    a = np.ascontiguousarray(array, dtype=np.float32).reshape(-1)
    rc = load().fs_save_array_to_file(a.ctypes.data, a.shape[0], str(path).encode())
    if rc != OK:
        raise FrequenSeeError(rc, f"cannot write {path}")


def load_float_array(path):
    """LoadFloatArray (FSAC.cpp:454-490)"""
    import numpy as np
    n = C.c_int32()
    lib = load()
    rc = lib.fs_load_float_array(str(path).encode(), None, 0, C.byref(n))
    if rc != OK:
        raise FrequenSeeError(rc, f"cannot read {path}")
    out = np.zeros(n.value, dtype=np.float32)
    lib.fs_load_float_array(str(path).encode(), out.ctypes.data, out.shape[0], C.byref(n))
    return out
