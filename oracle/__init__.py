"""ctypes loader for the CPU oracle (oracle/fs_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (audio-pathtracer_amd/) never does.  See fs_oracle.h for provenance and the
"parity unpinned" statement.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
MAX_BANDS = 8
NO_MATERIAL = 0xFFFF

FLAG_FIXED_NORM_1000 = 1
FLAG_FLUSH_BEFORE_RECONSTRUCT = 2
FLAG_COSINE_SAMPLING = 4
FLAG_BRUTE_FORCE = 8
FLAG_ALL_CONNECTIONS = 16
FLAG_MIS_BALANCE = 32
FLAG_MATERIAL_LOBES = 64
FLAG_ACCUMULATE_ENERGY = 128
LOBE_DIFFUSE, LOBE_SPECULAR, LOBE_TRANSMIT = 0, 1, 2
LOBE_SHIFT = 16


class Params(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64),
        ("num_pairs", C.c_uint32),
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
        ("flags", C.c_uint32),
        ("listener_radius", C.c_float),
        ("source_radius", C.c_float),
        ("source_object", C.c_uint32),
        ("listener_object", C.c_uint32),
    ]


class Counters(C.Structure):
    _fields_ = [
        ("node_visits", C.c_uint64),
        ("tri_tests", C.c_uint64),
        ("closest_rays", C.c_uint64),
        ("any_rays", C.c_uint64),
        ("connected", C.c_uint64),
        ("deposits", C.c_uint64),
        ("path_nodes", C.c_uint64),
        ("any_node_visits", C.c_uint64),
        ("any_tri_tests", C.c_uint64),
    ]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}

    def add(self, other):
        for k, _ in self._fields_:
            setattr(self, k, getattr(self, k) + getattr(other, k))


class SoundParams(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64),
        ("raycasts_per_tick", C.c_int32),
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


class Node(C.Structure):
    _fields_ = [
        ("pos", C.c_float * 3),
        ("normal", C.c_float * 3),
        ("material", C.c_uint32),
        ("prob", C.c_float),
    ]


def build(native: bool = False, outdir: str | None = None) -> str:
    """Compile the oracle.  native=True builds -O3 -march=native into outdir (CPU-baseline timing)."""
    if native:
        outdir = outdir or "/tmp"
        os.makedirs(outdir, exist_ok=True)
        subprocess.run(["make", "-s", "-C", _HERE, "native", f"OUT={outdir}"], check=True)
        return os.path.join(outdir, "libfs_oracle_native.so")
    subprocess.run(["make", "-s", "-C", _HERE], check=True)
    return os.path.join(_HERE, "libfs_oracle.so")


_f3 = C.POINTER(C.c_float)


def _bind(lib):
    lib.fso_params_default.argtypes = [C.POINTER(Params)]
    lib.fso_scene_create.restype = C.c_void_p
    lib.fso_scene_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32]
    lib.fso_scene_destroy.argtypes = [C.c_void_p]
    lib.fso_scene_num_nodes.argtypes = [C.c_void_p]
    lib.fso_scene_num_nodes.restype = C.c_int32
    lib.fso_trace_closest.argtypes = [C.c_void_p, _f3, _f3, C.c_float, C.c_int32, C.POINTER(C.c_float),
                                      C.POINTER(C.c_int32), _f3, C.POINTER(Counters)]
    lib.fso_trace_closest.restype = C.c_int32
    lib.fso_trace_any.argtypes = [C.c_void_p, _f3, _f3, C.c_float, C.c_int32, C.POINTER(Counters)]
    lib.fso_trace_any.restype = C.c_int32
    lib.fso_philox4x32_10.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.fso_u01.argtypes = [C.c_uint32]
    lib.fso_u01.restype = C.c_float
    lib.fso_sincos2pi.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.fso_sample_sphere.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), _f3]
    lib.fso_sample_cone.argtypes = [_f3, C.c_float, C.c_float, C.c_int32, _f3]
    lib.fso_generate_path.argtypes = [C.c_void_p, C.POINTER(Params), C.c_uint32, C.c_uint32, _f3,
                                      C.POINTER(Node), C.c_int32, C.POINTER(Counters)]
    lib.fso_generate_path.restype = C.c_int32
    lib.fso_connect.argtypes = [C.c_void_p, C.POINTER(Params), C.POINTER(Node), C.POINTER(Node), C.POINTER(Counters)]
    lib.fso_connect.restype = C.c_int32
    lib.fso_connect_ep.argtypes = [C.c_void_p, C.POINTER(Params), C.POINTER(Node), C.POINTER(Node), _f3, _f3, C.POINTER(Counters)]
    lib.fso_connect_ep.restype = C.c_int32
    lib.fso_evaluate_path.argtypes = [C.c_void_p, C.POINTER(Params), C.POINTER(Node), C.c_int32,
                                      C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.fso_scene_set_lobes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.fso_scene_set_lobes.restype = None
    lib.fso_scene_lobe_table.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    lib.fso_scene_lobe_table.restype = None
    lib.fso_mis_weight.argtypes = [C.POINTER(Node), C.c_int32, C.c_int32, C.c_int32]
    lib.fso_mis_weight.restype = C.c_double
    lib.fso_add_energy_at_delay.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_float]
    lib.fso_add_energy_at_delay.restype = C.c_int32
    lib.fso_num_bins.argtypes = [C.c_float, C.c_float]
    lib.fso_num_bins.restype = C.c_int32
    lib.fso_num_samples.argtypes = [C.c_float, C.c_int32]
    lib.fso_num_samples.restype = C.c_int32
    lib.fso_samples_per_bin.argtypes = [C.c_float, C.c_int32]
    lib.fso_samples_per_bin.restype = C.c_int32
    lib.fso_compute_energy.argtypes = [C.c_void_p, C.POINTER(Params), _f3, _f3, C.c_uint32, C.c_uint32, C.c_int32,
                                       C.c_void_p, C.c_void_p, C.POINTER(Counters)]
    lib.fso_compute_energy_mt.argtypes = [C.c_void_p, C.POINTER(Params), _f3, _f3, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32,
                                          C.c_void_p, C.c_void_p, C.POINTER(Counters)]
    lib.fso_compute_energy_mt.restype = C.c_int32
    lib.fso_reconstruct.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_int32, C.c_void_p]
    lib.fso_sanitize_float.argtypes = [C.c_double, C.c_char_p, C.c_int32]
    lib.fso_sanitize_float.restype = C.c_int32
    lib.fso_save_array_to_file.argtypes = [C.c_void_p, C.c_int32, C.c_char_p]
    lib.fso_save_array_to_file.restype = C.c_int32
    lib.fso_load_float_array.argtypes = [C.c_char_p, C.c_void_p, C.c_int32]
    lib.fso_load_float_array.restype = C.c_int32
    lib.fso_scene_set_objects.argtypes = [C.c_void_p, C.c_void_p]
    lib.fso_sound_params_default.argtypes = [C.POINTER(SoundParams)]
    lib.fso_legacy_direction.argtypes = [C.c_uint64, C.c_uint32, _f3]
    lib.fso_update_sound.argtypes = [C.c_void_p, C.POINTER(SoundParams), _f3, _f3, C.POINTER(SoundResult)]
    return lib


_lib = None
_lock = threading.Lock()


def load(path: str | None = None):
    """Load (building if needed) the portable oracle library."""
    global _lib
    if path is not None:
        return _bind(C.CDLL(path))
    with _lock:
        if _lib is None:
            so = os.path.join(_HERE, "libfs_oracle.so")
            if not os.path.exists(so):
                build()
            _lib = _bind(C.CDLL(so))
    return _lib


def load_dpos():
    """The double-position build (-DFSO_DOUBLE_POSITIONS, oracle/Makefile target dpos): node positions, hit points and
    segment lengths in double like the reference's FVector.  Only Scene.compute_energy* may be used with it (the Node
    layout differs); it exists to MEASURE the deviation of the float positions (tests/test_double_positions.py)."""
    so = os.path.join(_HERE, "libfs_oracle_dpos.so")
    if not os.path.exists(so):
        build()
    return _bind(C.CDLL(so))


def _vec3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def default_params(**kw) -> Params:
    p = Params()
    load().fso_params_default(C.byref(p))
    for k, v in kw.items():
        if k == "air_absorption":
            for i, x in enumerate(v):
                p.air_absorption[i] = float(x)
        else:
            setattr(p, k, v)
    return p


class Scene:
    """Triangle soup + per-band material table (what RegisterGeometry + UAcousticMaterial provide)."""

    def __init__(self, tri_xyz, mat_id, absorption, lib=None, transmission=None, scattering=None):
        self.lib = lib or load()
        self.tri = np.ascontiguousarray(tri_xyz, dtype=np.float32).reshape(-1, 3, 3)
        self.mat = np.ascontiguousarray(mat_id, dtype=np.uint16).reshape(-1)
        self.absorption = np.ascontiguousarray(absorption, dtype=np.float32)
        if self.absorption.ndim == 1:
            self.absorption = self.absorption.reshape(-1, 1)
        self.M, self.B = self.absorption.shape
        self.T = self.tri.shape[0]
        assert self.mat.shape[0] == self.T
        self.h = self.lib.fso_scene_create(self.tri.ctypes.data, self.mat.ctypes.data, self.T,
                                           self.absorption.ctypes.data, self.M, self.B)
        if not self.h:
            raise ValueError("fso_scene_create failed")
        if transmission is not None or scattering is not None:
            self.set_lobes(transmission, scattering)

    def set_lobes(self, transmission=None, scattering=None):
        """UAcousticMaterial::Transmission / Scattering, [M][B] each (None = 0 / 1): tables of FLAG_MATERIAL_LOBES."""
        arrs = []
        for a in (transmission, scattering):
            if a is None:
                arrs.append(None)
            else:
                a = np.ascontiguousarray(a, dtype=np.float32).reshape(self.M, self.B)
                arrs.append(a)
        self._lobe_inputs = arrs        # keep alive during the call
        self.lib.fso_scene_set_lobes(self.h, None if arrs[0] is None else arrs[0].ctypes.data,
                                     None if arrs[1] is None else arrs[1].ctypes.data)

    def lobe_table(self, m):
        gains = np.zeros((3, self.B), np.float32)
        prob = np.zeros(3, np.float32)
        self.lib.fso_scene_lobe_table(self.h, m, gains.ctypes.data, prob.ctypes.data)
        return gains, prob

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.fso_scene_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # --- engine line-trace contract -------------------------------------------------------
    def trace_closest(self, o, d, tmax=1e6, brute=False, counters=None):
        t = C.c_float()
        tri = C.c_int32(-1)
        n = (C.c_float * 3)()
        hit = self.lib.fso_trace_closest(self.h, _vec3(o), _vec3(d), tmax, int(brute), C.byref(t), C.byref(tri), n,
                                         C.byref(counters) if counters is not None else None)
        return bool(hit), float(t.value), int(tri.value), np.array(list(n), dtype=np.float32)

    def trace_any(self, o, d, tmax, brute=False):
        return bool(self.lib.fso_trace_any(self.h, _vec3(o), _vec3(d), tmax, int(brute), None))

    # --- path pieces --------------------------------------------------------------------------
    def generate_path(self, params, pair, side, start, max_nodes=130):
        nodes = (Node * max_nodes)()
        n = self.lib.fso_generate_path(self.h, C.byref(params), pair, side, _vec3(start), nodes, max_nodes, None)
        return [nodes[i] for i in range(n)]

    def connect(self, params, f, b):
        return bool(self.lib.fso_connect(self.h, C.byref(params), C.byref(f), C.byref(b), None))

    def evaluate_path(self, params, nodes):
        arr = (Node * len(nodes))(*nodes)
        gains = (C.c_float * MAX_BANDS)()
        delay = C.c_float()
        self.lib.fso_evaluate_path(self.h, C.byref(params), arr, len(nodes), gains, C.byref(delay))
        return np.array(list(gains)[: self.B], dtype=np.float32), float(delay.value)

    # --- frame --------------------------------------------------------------------------------
    def compute_energy(self, params, src, lis, pair_begin=0, pair_end=None, num_bins=1000, want_f64=True, into=None):
        """UpdateSource up to the deposit (ARTS.cpp:128-173) for pairs [pair_begin, pair_end).
        into = (e32, e64) of an earlier frame: with FLAG_ACCUMULATE_ENERGY the deposits add to them (HEAD's behaviour)."""
        if pair_end is None:
            pair_end = params.num_pairs
        if into is not None:
            e32, e64 = into
            assert e32.dtype == np.float32 and e32.flags.c_contiguous and e64.dtype == np.float64 and e64.flags.c_contiguous
        else:
            e32 = np.zeros((self.B, num_bins), dtype=np.float32)
            e64 = np.zeros((self.B, num_bins), dtype=np.float64) if want_f64 else None
        c = Counters()
        self.lib.fso_compute_energy(self.h, C.byref(params), _vec3(src), _vec3(lis), pair_begin, pair_end, num_bins,
                                    e32.ctypes.data, e64.ctypes.data if want_f64 else None, C.byref(c))
        return e32, e64, c

    def compute_energy_mt(self, params, src, lis, threads, pair_begin=0, pair_end=None, num_bins=1000, pool=None):
        """All-cores CPU baseline: static partition of the pair range, private histograms, final sum.
        ctypes releases the GIL during the foreign call, so plain Python threads run in parallel.
        pool: a ThreadPoolExecutor the caller keeps (bench.py: the threads exist before the clock starts)."""
        if pair_end is None:
            pair_end = params.num_pairs
        n = pair_end - pair_begin
        cuts = [pair_begin + (n * i) // threads for i in range(threads + 1)]

        def work(i):
            return self.compute_energy(params, src, lis, cuts[i], cuts[i + 1], num_bins, want_f64=True)

        if pool is not None:
            parts = list(pool.map(work, range(threads)))
        else:
            with ThreadPoolExecutor(max_workers=threads) as ex:
                parts = list(ex.map(work, range(threads)))
        e64 = sum(p[1] for p in parts)
        c = Counters()
        for p in parts:
            c.add(p[2])
        return e64.astype(np.float32), e64, c

    def compute_energy_native_mt(self, params, src, lis, threads, pair_begin=0, pair_end=None, num_bins=1000):
        """All-cores CPU baseline inside the C library (fso_compute_energy_mt): pthreads with a static partition of the pair
        range and private double histograms — no interpreter between the threads.  Returns (e32, e64, counters, threads used)."""
        if pair_end is None:
            pair_end = params.num_pairs
        e32 = np.zeros((self.B, num_bins), dtype=np.float32)
        e64 = np.zeros((self.B, num_bins), dtype=np.float64)
        c = Counters()
        used = self.lib.fso_compute_energy_mt(self.h, C.byref(params), _vec3(src), _vec3(lis), pair_begin, pair_end, num_bins,
                                              int(threads), e32.ctypes.data, e64.ctypes.data, C.byref(c))
        return e32, e64, c, int(used)

    def set_objects(self, object_ids):
        """actor id per triangle (None = every triangle its own actor)"""
        if object_ids is None:
            self.lib.fso_scene_set_objects(self.h, None)
        else:
            self._obj = np.ascontiguousarray(object_ids, dtype=np.uint32)
            assert self._obj.shape[0] == self.T
            self.lib.fso_scene_set_objects(self.h, self._obj.ctypes.data)

    def update_sound(self, src, lis, **kw):
        """legacy forward tracer UpdateSound (FSAC.cpp:283-306)"""
        p = SoundParams()
        self.lib.fso_sound_params_default(C.byref(p))
        for k, v in kw.items():
            setattr(p, k, v)
        r = SoundResult()
        self.lib.fso_update_sound(self.h, C.byref(p), _vec3(src), _vec3(lis), C.byref(r))
        return r.as_dict()


def reconstruct(energy_row, sample_rate=48000, bin_duration=0.001, num_samples=48000, samples_per_bin=0, lib=None):
    """ReconstructImpulseResponse (FSAC.cpp:320-380) for one energy row."""
    lib = lib or load()
    e = np.ascontiguousarray(energy_row, dtype=np.float32)
    out = np.zeros(num_samples, dtype=np.float32)
    lib.fso_reconstruct(e.ctypes.data, e.shape[0], sample_rate, bin_duration, num_samples, samples_per_bin,
                        out.ctypes.data)
    return out


def mis_weight(nodes, s, depth, lib=None):
    """Balance-heuristic weight of strategy s for the connected path `nodes` (list of Node), depth cap `depth`."""
    lib = lib or load()
    arr = (Node * len(nodes))(*nodes)
    return float(lib.fso_mis_weight(arr, len(nodes), s, depth))


def make_node(pos, normal=(0.0, 0.0, 0.0), material=NO_MATERIAL, prob=1.0):
    nd = Node()
    nd.pos[:] = [float(x) for x in pos]
    nd.normal[:] = [float(x) for x in normal]
    nd.material = material
    nd.prob = prob
    return nd


def add_energy_at_delay(buf, delay, e, bin_size_ms=1, lib=None):
    lib = lib or load()
    assert buf.dtype == np.float32 and buf.flags.c_contiguous
    return lib.fso_add_energy_at_delay(buf.ctypes.data, buf.shape[0], bin_size_ms, delay, e)


# ---- row f2: reverb oracle built around the reference's own KissFFT (oracle/_ref, see oracle/Makefile) ----------
_REF_SO = os.path.join(_HERE, "_ref", "libfs_reverb_ref.so")
_KISS_DIR = "/root/reference/Plugins/FrequenSee/Source/FrequenSee/Private/FrequenSeeFFTConvolver/KissFFT"


def build_ref() -> str | None:
    """Compile oracle/_ref from the reference's KissFFT sources where they lie (only possible where
    /root/reference exists; the GPU box uses the prebuilt .so)."""
    if os.path.isdir(_KISS_DIR):
        subprocess.run(["make", "-s", "-C", _HERE, "_ref"], check=True)
    return _REF_SO if os.path.exists(_REF_SO) else None


class ReverbRef:
    """FFrequenSeeAudioReverbPlugin::ProcessSourceAudio restated around the reference's KissFFT."""

    def __init__(self, sample_rate=48000, simulated_duration=1.0, frame_size=1024):
        so = _REF_SO if os.path.exists(_REF_SO) else build_ref()
        if not so:
            raise FileNotFoundError("oracle/_ref/libfs_reverb_ref.so missing and /root/reference not present")
        lib = C.CDLL(so)
        lib.fso_reverb_create.restype = C.c_void_p
        lib.fso_reverb_create.argtypes = [C.c_int32, C.c_float, C.c_int32]
        lib.fso_reverb_destroy.argtypes = [C.c_void_p]
        lib.fso_reverb_fft_size.argtypes = [C.c_void_p]
        lib.fso_reverb_fft_size.restype = C.c_int32
        lib.fso_reverb_process.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
        self.lib = lib
        self.frame = frame_size
        self.h = lib.fso_reverb_create(sample_rate, simulated_duration, frame_size)
        self.fft_size = lib.fso_reverb_fft_size(self.h)

    def process(self, ir_l, ir_r, audio_interleaved, apply_reverb=True, literal_tail=False):
        a = np.ascontiguousarray(audio_interleaved, dtype=np.float32).reshape(-1)
        il = np.ascontiguousarray(ir_l, dtype=np.float32)
        ir = np.ascontiguousarray(ir_r, dtype=np.float32)
        out = np.zeros_like(a)
        self.lib.fso_reverb_process(self.h, il.ctypes.data, ir.ctypes.data, a.ctypes.data, out.ctypes.data,
                                    int(apply_reverb), int(literal_tail))
        return out

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.fso_reverb_destroy(self.h)
                self.h = None
        except Exception:
            pass


def apply_material_fd(in_buffer, absorption, transmission, scattering):
    """UMaterialAcousticProcessor::ApplyMaterialFD restated around the reference's KissFFT (oracle/_ref).
    Returns (specular, diffuse, transmitted), or None where the reference returns empty outputs
    (response curves not N/2+1 long, MAP.cpp:20-26)."""
    so = _REF_SO if os.path.exists(_REF_SO) else build_ref()
    if not so:
        raise FileNotFoundError("oracle/_ref/libfs_reverb_ref.so missing and /root/reference not present")
    lib = C.CDLL(so)
    vp = C.c_void_p
    lib.fso_apply_material_fd.argtypes = [vp, C.c_int32, vp, vp, vp, C.c_int32, vp, vp, vp]
    lib.fso_apply_material_fd.restype = C.c_int32
    x = np.ascontiguousarray(in_buffer, dtype=np.float32).reshape(-1)
    a = np.ascontiguousarray(absorption, dtype=np.float32).reshape(-1)
    t = np.ascontiguousarray(transmission, dtype=np.float32).reshape(-1)
    s = np.ascontiguousarray(scattering, dtype=np.float32).reshape(-1)
    if not (a.size == t.size == s.size):
        return None
    outs = [np.zeros(x.size, dtype=np.float32) for _ in range(3)]
    rc = lib.fso_apply_material_fd(x.ctypes.data, x.size, a.ctypes.data, t.ctypes.data, s.ctypes.data, a.size,
                                   *[o.ctypes.data for o in outs])
    return None if rc != 0 else tuple(outs)


def apply_material_fd_numpy(in_buffer, absorption, transmission, scattering):
    """The same block filter in float64 numpy (independent of any FFT library of the reference): a second
    opinion for the tolerance of the fp32 transforms."""
    x = np.asarray(in_buffer, dtype=np.float64).reshape(-1)
    L = x.size
    N = 1
    while N < L:
        N <<= 1
    a = np.asarray(absorption, dtype=np.float32)
    t = np.asarray(transmission, dtype=np.float32).copy()
    s = np.asarray(scattering, dtype=np.float32)
    refl = np.float32(1.0) - a
    over = (refl + t) > np.float32(1.0)
    t[over] = (np.float32(1.0) - refl)[over]
    X = np.fft.rfft(x, N)
    gains = (refl * (np.float32(1.0) - s), refl * s, t)
    return tuple(np.fft.irfft(X * g.astype(np.float64), N)[:L] for g in gains)
