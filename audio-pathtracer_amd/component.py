"""Host-side mirror of the reference's plugin interface for the BDPT path, over the C ABI.

  AudioRayTracingSubsystem  <->  UAudioRayTracingSubsystem   (Public/AudioRayTracingSubsystem.h:86-196)
  FrequenSeeAudioComponent  <->  UFrequenSeeAudioComponent   (Public/FrequenSeeAudioComponent.h:20-154)

Same method names, argument meaning and error behaviour (check() aborts become exceptions), so the
parity tests read like tests of the reference.  All compute happens in libfrequensee.so (HIP).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi
from ._capi import FrequenSeeError


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


class Context:
    """Thin RAII wrapper of fs_context (one HIP device, one stream)."""

    def __init__(self, num_bands=1, device=0, rank=0, world_size=1, stream=None, **cfg):
        self.lib = _capi.load()
        c = _capi.default_config(num_bands=num_bands, device=device, rank=rank, world_size=world_size, **cfg)
        if stream is not None:
            c.stream = C.c_void_p(int(stream))
        self.cfg = c
        h = C.c_void_p()
        rc = self.lib.fs_context_create(C.byref(c), C.byref(h))
        self.h = h
        if rc != _capi.OK:
            msg = self.lib.fs_last_error(h).decode() if h else "fs_context_create failed"
            if h:
                self.lib.fs_context_destroy(h)
                self.h = None
            raise FrequenSeeError(rc, msg)
        self.num_bands = num_bands
        self.num_bins = self.lib.fs_num_bins(self.h)
        self.num_samples = self.lib.fs_num_samples(self.h)

    def check(self, rc):
        if rc != _capi.OK:
            raise FrequenSeeError(rc, self.lib.fs_last_error(self.h).decode())

    def advice(self):
        """fs_context_advice: what fs_context_create found worth telling the host about its environment ('' = nothing)"""
        return self.lib.fs_context_advice(self.h).decode()

    def close(self):
        if getattr(self, "h", None):
            self.lib.fs_context_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- multi-GPU: the library's own RCCL communicator (include/frequensee.h) ----
    @staticmethod
    def comm_unique_id() -> bytes:
        """rank 0: the 128-byte id every rank passes to comm_init (ship it with any host-side transport)"""
        buf = (C.c_char * _capi.COMM_ID_BYTES)()
        rc = _capi.load().fs_comm_unique_id(buf, _capi.COMM_ID_BYTES)
        if rc != _capi.OK:
            raise FrequenSeeError(rc, "fs_comm_unique_id failed (librccl not loadable?)")
        return bytes(buf)

    def comm_init(self, unique_id: bytes):
        """collective over the world_size ranks: from now on every frame's energy buffer is summed over the ranks on
        the tail stream, and set_scene lets rank 0 build the acceleration structure for all"""
        if len(unique_id) != _capi.COMM_ID_BYTES:
            raise ValueError("unique id must be 128 bytes")
        buf = (C.c_char * _capi.COMM_ID_BYTES).from_buffer_copy(unique_id)
        self.check(self.lib.fs_comm_init(self.h, buf, _capi.COMM_ID_BYTES))

    def comm_attach(self, nccl_comm: int):
        self.check(self.lib.fs_comm_attach(self.h, C.c_void_p(int(nccl_comm))))

    def comm_enable_oneshot(self):
        """collective: the energy buffer's sum over the ranks as one peer-write exchange (HIP IPC mailboxes) instead of ncclAllReduce"""
        self.check(self.lib.fs_comm_enable_oneshot(self.h))

    def comm_info(self):
        """fs_comm_info: (ranks of the attached communicator as RCCL counts them, this rank, collective kind 0 / 1 / 2)"""
        n, r, k = C.c_int32(), C.c_int32(), C.c_int32()
        self.check(self.lib.fs_comm_info(self.h, C.byref(n), C.byref(r), C.byref(k)))
        return int(n.value), int(r.value), int(k.value)

    def comm_detach(self):
        self.check(self.lib.fs_comm_detach(self.h))

    # cfg5: independent sources, one per GPU — a peer communicator for the optional all-gather
    def peers_init(self, unique_id: bytes, rank: int, world_size: int):
        """collective over the processes that each own a source; frames are not sharded or reduced by it"""
        if len(unique_id) != _capi.COMM_ID_BYTES:
            raise ValueError("unique id must be 128 bytes")
        buf = (C.c_char * _capi.COMM_ID_BYTES).from_buffer_copy(unique_id)
        self.check(self.lib.fs_peers_init(self.h, buf, _capi.COMM_ID_BYTES, int(rank), int(world_size)))
        self._peers = int(world_size)

    def peers_detach(self):
        self.check(self.lib.fs_peers_detach(self.h))
        self._peers = 0

    def gather_energy(self, src: int) -> np.ndarray:
        """collective: [world_size][B][bins] — entry r is the current frame's histogram of the source rank r passed"""
        n = int(getattr(self, "_peers", 0))
        out = np.empty((max(n, 1), self.num_bands, self.num_bins), np.float32)
        self.check(self.lib.fs_gather_energy(self.h, int(src), out.ctypes.data_as(C.POINTER(C.c_float)), out.size))
        return out

    def set_pipelining(self, depth):
        """0 / False: off; 1 / True: a frame's connect pass is held back and launched with the next frame's walk; 2: its walk
        is held back as well — one launch plans frame f, walks f - 1 and connects f - 2 (include/frequensee.h)"""
        self.check(self.lib.fs_set_pipelining(self.h, int(depth)))

    def set_frames_per_launch(self, n: int):
        """fs_set_frames_per_launch: with pipelining on, plain frames wait until n have come and share one launch (every frame
        keeps its seed, energy buffer and recorded reconstruct: the results are those of n single frames)"""
        self.check(self.lib.fs_set_frames_per_launch(self.h, int(n)))

    def set_walk_stages(self, bounds):
        """fs_set_walk_stages: the steps at which a pipelined depth = 0 walk moves on to the next launch ([] = such frames are not held)"""
        arr = np.asarray(list(bounds), dtype=np.int32)
        self.check(self.lib.fs_set_walk_stages(self.h, arr.ctypes.data_as(C.c_void_p) if arr.size else None, int(arr.size)))

    def submit(self):
        self.check(self.lib.fs_submit(self.h))

    def gather_energy_async(self, src: int):
        """enqueue only: (device pointer, bytes) of the gathered [world_size][B][bins] histograms, in tail-stream order"""
        d, b = C.c_void_p(), C.c_size_t()
        self.check(self.lib.fs_gather_energy_async(self.h, int(src), C.byref(d), C.byref(b)))
        return d.value, b.value

    # ---- scene ----
    def set_scene(self, triangles, material_ids, absorption, transmission=None, scattering=None, object_ids=None, fast=False):
        """fast = True: the acceleration structure is built on the device (fs_scene_commit_fast); fast = "progressive": that
        tree now and the host's SAH tree as soon as a background thread has built it (fs_scene_commit_progressive)"""
        tri = np.ascontiguousarray(triangles, dtype=np.float32).reshape(-1, 3, 3)
        mat = np.ascontiguousarray(material_ids, dtype=np.uint16).reshape(-1)
        if mat.shape[0] != tri.shape[0]:
            raise ValueError("material_ids length != number of triangles")
        ab = np.ascontiguousarray(absorption, dtype=np.float32)
        if ab.ndim == 1:
            ab = ab.reshape(-1, 1)
        opt = []
        for a in (transmission, scattering):
            opt.append(None if a is None else np.ascontiguousarray(a, dtype=np.float32).reshape(ab.shape))
        self.check(self.lib.fs_scene_set_triangles(self.h, tri.ctypes.data, mat.ctypes.data, tri.shape[0]))
        self.check(self.lib.fs_scene_set_materials(self.h, ab.ctypes.data,
                                                   opt[0].ctypes.data if opt[0] is not None else None,
                                                   opt[1].ctypes.data if opt[1] is not None else None,
                                                   ab.shape[0], ab.shape[1]))
        if object_ids is not None:   # actor per triangle (legacy tracer's AddIgnoredActor / pass-through logic)
            obj = np.ascontiguousarray(object_ids, dtype=np.uint32).reshape(-1)
            self.check(self.lib.fs_scene_set_objects(self.h, obj.ctypes.data, obj.shape[0]))
        else:
            self.check(self.lib.fs_scene_set_objects(self.h, None, tri.shape[0]))
        commit = {False: self.lib.fs_scene_commit, True: self.lib.fs_scene_commit_fast,
                  "progressive": self.lib.fs_scene_commit_progressive}[fast]
        self.check(commit(self.h))

    def refine_pending(self) -> bool:
        """fast="progressive": is the background SAH build still outstanding (the swap happens at the next trace after it)"""
        v = C.c_int32(0)
        self.check(self.lib.fs_scene_refine_pending(self.h, C.byref(v)))
        return bool(v.value)

    def refine_wait(self):
        self.check(self.lib.fs_scene_refine_wait(self.h))

    def update_triangles(self, first, triangles):
        """move committed triangles [first, first + n) (row f4: dynamic props); the refit runs before the next trace"""
        t = np.ascontiguousarray(triangles, dtype=np.float32).reshape(-1, 3, 3)
        self.check(self.lib.fs_scene_update_triangles(self.h, int(first), t.shape[0], t.ctypes.data))

    def refit(self):
        self.check(self.lib.fs_scene_refit(self.h))

    def set_listener(self, xyz):
        self.check(self.lib.fs_listener_set_position(self.h, _f3(xyz)))

    def create_source(self, xyz=None) -> int:
        s = C.c_int32(-1)
        self.check(self.lib.fs_source_create(self.h, C.byref(s)))
        if xyz is not None:
            self.set_source_position(s.value, xyz)
        return s.value

    def destroy_source(self, src):
        self.check(self.lib.fs_source_destroy(self.h, src))

    def set_source_position(self, src, xyz):
        self.check(self.lib.fs_source_set_position(self.h, src, _f3(xyz)))

    # ---- hot path ----
    def compute_energy_response(self, src, params, want_host=True):
        out = np.empty((self.num_bands, self.num_bins), dtype=np.float32) if want_host else None
        self.check(self.lib.fs_compute_energy_response(self.h, src, C.byref(params),
                                                       out.ctypes.data if want_host else None))
        return out

    def compute_energy_response_async(self, src, params):
        self.check(self.lib.fs_compute_energy_response_async(self.h, src, C.byref(params)))

    def set_source_object(self, src, object_id=_capi.NO_OBJECT):
        """the actor the source belongs to (ids of set_objects): its own walks ignore it (AddIgnoredActor, ARTS.cpp:322-327)"""
        self.check(self.lib.fs_source_set_object(self.h, int(src), int(object_id)))

    def set_listener_object(self, object_id=_capi.NO_OBJECT):
        self.check(self.lib.fs_listener_set_object(self.h, int(object_id)))

    def compute_energy_response_batch_async(self, sources, params):
        """several sources in one traced frame (UpdateSource over ActiveSources): same result per source as separate calls"""
        arr = (C.c_int32 * len(sources))(*[int(x) for x in sources])
        self.check(self.lib.fs_compute_energy_response_batch_async(self.h, arr, len(sources), C.byref(params)))

    def reconstruct_impulse_response_batch_async(self, sources, params=None):
        """the tick's reconstructs as one launch with one completion event (same result per source as separate calls)"""
        arr = (C.c_int32 * len(sources))(*[int(x) for x in sources])
        self.check(self.lib.fs_reconstruct_impulse_response_batch_async(self.h, arr, len(sources), C.byref(params) if params else None))

    def update_sources(self, sources, params=None):
        """UpdateSources (ARTS.cpp:100-126): trace + reconstruct every listed source, return when every IR is published"""
        arr = (C.c_int32 * len(sources))(*[int(x) for x in sources])
        self.check(self.lib.fs_update_sources(self.h, arr, len(sources), C.byref(params) if params else None))

    def reconstruct_impulse_response(self, src, params=None):
        self.check(self.lib.fs_reconstruct_impulse_response(self.h, src, C.byref(params) if params else None))

    def reconstruct_impulse_response_async(self, src, params=None):
        self.check(self.lib.fs_reconstruct_impulse_response_async(self.h, src, C.byref(params) if params else None))

    def synchronize(self):
        self.check(self.lib.fs_synchronize(self.h))

    def energy_device_ptr(self, src):
        p, n = C.c_void_p(), C.c_size_t()
        self.check(self.lib.fs_energy_device_ptr(self.h, src, C.byref(p), C.byref(n)))
        return p.value, n.value

    def energy_handoff(self, src):
        """(device pointer, bytes, tail stream handle): the current frame's energy buffer, handed over to the
        tail stream on which the caller's all-reduce and the reconstruct run (include/frequensee.h)"""
        p, n, st = C.c_void_p(), C.c_size_t(), C.c_void_p()
        self.check(self.lib.fs_energy_handoff(self.h, src, C.byref(p), C.byref(n), C.byref(st)))
        return p.value, n.value, st.value

    def impulse_response(self, src, channel=0):
        """copy of the published channel IR (GetImpulseResponse()[channel])"""
        out = np.empty(self.num_samples, dtype=np.float32)
        self.check(self.lib.fs_copy_impulse_response(self.h, src, channel, out.ctypes.data, out.shape[0]))
        return out

    def set_impulse_response(self, src, ir):
        """install an IR of the caller's own (GetImpulseResponse() is a mutable reference, FSAC.h:113)"""
        a = np.ascontiguousarray(ir, dtype=np.float32).reshape(-1)
        self.check(self.lib.fs_set_impulse_response(self.h, src, a.ctypes.data, a.shape[0]))

    def impulse_response_view(self, src, channel=0):
        """zero-copy view of the published front buffer (valid until the second-next publish)"""
        p, n = C.POINTER(C.c_float)(), C.c_int32()
        self.check(self.lib.fs_get_impulse_response(self.h, src, channel, C.byref(p), C.byref(n)))
        return np.ctypeslib.as_array(p, shape=(n.value,))

    def impulse_response_sequence(self, src):
        """publishes of this source completed so far (the front buffer holds publish number `this` or a newer one)"""
        n = C.c_uint64()
        self.check(self.lib.fs_get_impulse_response_sequence(self.h, src, C.byref(n)))
        return int(n.value)

    def band_impulse_response(self, src, band):
        out = np.empty(self.num_samples, dtype=np.float32)
        self.check(self.lib.fs_copy_band_impulse_response(self.h, src, band, out.ctypes.data, out.shape[0]))
        return out

    def update_energy_buffer(self, src, values):
        """UpdateEnergyBuffer (FSAC.h:81-85): install a whole [bands][bins] histogram as the source's current frame"""
        v = np.ascontiguousarray(values, dtype=np.float32)
        self.check(self.lib.fs_update_energy_buffer(self.h, int(src), v.ctypes.data, v.size))

    def energy_buffer(self, src):
        out = np.empty((self.num_bands, self.num_bins), dtype=np.float32)
        self.check(self.lib.fs_get_energy_buffer(self.h, src, out.ctypes.data, out.size))
        return out

    def trace_rays(self, origins, dirs, tmax, any_hit=False):
        """any_hit: False / 0 closest hit, True / 1 any hit, 2 / 3 / 4 closest hit by the cooperative traversal (1 / 2 / 4 rays per wave)"""
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
        n = o.shape[0]
        tm = np.ascontiguousarray(np.broadcast_to(np.asarray(tmax, dtype=np.float32), (n,)))
        hit = np.zeros(n, dtype=np.int32)
        t = np.zeros(n, dtype=np.float32)
        tri = np.full(n, -1, dtype=np.int32)
        nrm = np.zeros((n, 3), dtype=np.float32)
        self.check(self.lib.fs_trace_rays(self.h, o.ctypes.data, d.ctypes.data, tm.ctypes.data, n, int(any_hit),
                                          hit.ctypes.data, t.ctypes.data, tri.ctypes.data, nrm.ctypes.data))
        return hit.astype(bool), t, tri, nrm

    def update_sound(self, src, params=None):
        """legacy forward tracer UpdateSound (FSAC.cpp:283-306); returns the fs_sound_result fields"""
        r = _capi.SoundResult()
        self.check(self.lib.fs_update_sound(self.h, src, C.byref(params) if params is not None else None, C.byref(r)))
        return r.as_dict()

    def occlusion_attenuation(self, src):
        v = C.c_float()
        self.check(self.lib.fs_get_occlusion_attenuation(self.h, src, C.byref(v)))
        return float(v.value)

    # ---- row f2: reverb plugin convolution ----
    def reverb_init(self, src, frame_size=1024):
        self.check(self.lib.fs_reverb_init(self.h, src, frame_size))
        self._rev_frame = getattr(self, "_rev_frame", {})
        self._rev_frame[src] = frame_size

    def reverb_process(self, src, audio_interleaved, apply_reverb=True, literal_tail=False):
        """ProcessSourceAudio (RVB.cpp:118-170): interleaved stereo block in -> convolved block out"""
        frame = self._rev_frame[src]
        a = np.ascontiguousarray(audio_interleaved, dtype=np.float32).reshape(-1)
        if a.shape[0] != 2 * frame:
            raise ValueError("audio block must hold frame_size * 2 interleaved samples")
        out = np.empty_like(a)
        self.check(self.lib.fs_reverb_process(self.h, src, a.ctypes.data, out.ctypes.data, int(apply_reverb),
                                              _capi.REVERB_LITERAL_TAIL if literal_tail else 0))
        return out

    def reverb_release(self, src):
        self.check(self.lib.fs_reverb_release(self.h, src))

    def apply_material_fd(self, in_buffer, absorption, transmission, scattering):
        """UMaterialAcousticProcessor::ApplyMaterialFD (MAP.cpp:8-107) -> (specular, diffuse, transmitted)"""
        x = np.ascontiguousarray(in_buffer, dtype=np.float32).reshape(-1)
        a = np.ascontiguousarray(absorption, dtype=np.float32).reshape(-1)
        t = np.ascontiguousarray(transmission, dtype=np.float32).reshape(-1)
        sc = np.ascontiguousarray(scattering, dtype=np.float32).reshape(-1)
        if not (a.size == t.size == sc.size):
            raise _capi.FrequenSeeError(_capi.ERR_SIZE_MISMATCH, "response curves differ in length")
        outs = [np.zeros(x.size, dtype=np.float32) for _ in range(3)]
        self.check(self.lib.fs_apply_material_fd(self.h, x.ctypes.data, x.size, a.ctypes.data, t.ctypes.data,
                                                 sc.ctypes.data, a.size, *[o.ctypes.data for o in outs]))
        return tuple(outs)

    def set_profiling(self, level=2, interval=1):
        """0 off, 1 = HIP events around the dominant (walk) kernel only (every `interval`-th frame), 2 = every kernel,
        3 = 2 + the kernels count the records they fetch"""
        self.check(self.lib.fs_set_profiling(self.h, int(level)))
        self.check(self.lib.fs_set_profiling_interval(self.h, int(interval)))

    def stats(self):
        s = _capi.Stats()
        self.check(self.lib.fs_get_stats(self.h, C.byref(s)))
        return s.as_dict()

    def reset_stats(self):
        self.check(self.lib.fs_reset_stats(self.h))

    def streams(self):
        """fs_get_streams: (compute stream, tail stream) as integer hipStream_t handles"""
        a, b = C.c_void_p(), C.c_void_p()
        self.check(self.lib.fs_get_streams(self.h, C.byref(a), C.byref(b)))
        return int(a.value or 0), int(b.value or 0)

    def pipeline_counters(self):
        """fs_get_pipeline_counters: the producer side's host counters since the context was created"""
        c = _capi.PipelineCounters()
        c.struct_size = C.sizeof(_capi.PipelineCounters)
        self.check(self.lib.fs_get_pipeline_counters(self.h, C.byref(c)))
        return c.as_dict()


class FrequenSeeAudioComponent:
    """UFrequenSeeAudioComponent's energy/IR surface (FSAC.h:69-91, 112-113, 133-143)."""

    def __init__(self, location=(0.0, 0.0, 0.0)):
        self._location = np.asarray(location, dtype=np.float32)
        self._subsys = None
        self._src = None
        self.bApplyReverb = True  # FSAC.h:60

    # OnRegister / OnUnregister (FSAC.cpp:42-64): auto-hook into the subsystem
    def OnRegister(self, subsystem: "AudioRayTracingSubsystem"):
        subsystem.RegisterSource(self)

    def OnUnregister(self):
        if self._subsys is not None:
            self._subsys.UnRegisterSource(self)

    def _ctx(self) -> Context:
        if self._subsys is None:
            raise RuntimeError("component is not registered with an AudioRayTracingSubsystem")
        return self._subsys.ctx

    @property
    def NumBins(self):
        return self._ctx().num_bins

    @property
    def NumSamples(self):
        return self._ctx().num_samples

    def GetComponentLocation(self):
        return self._location.copy()

    def SetComponentLocation(self, xyz):
        self._location = np.asarray(xyz, dtype=np.float32)
        if self._subsys is not None:
            self._ctx().set_source_position(self._src, self._location)

    @property
    def EnergyBuffer(self):
        """[NumBins] for one band (the reference), [bands][NumBins] otherwise"""
        e = self._ctx().energy_buffer(self._src)
        return e[0] if e.shape[0] == 1 else e

    def FlushEnergyBuffer(self):  # FSAC.h:76-79
        c = self._ctx()
        c.check(c.lib.fs_flush_energy_buffer(c.h, self._src))

    def UpdateEnergyBuffer(self, NewEnergyValues):  # FSAC.h:81-85 (check -> exception)
        c = self._ctx()
        v = np.ascontiguousarray(NewEnergyValues, dtype=np.float32)
        c.check(c.lib.fs_update_energy_buffer(c.h, self._src, v.ctypes.data, v.size))

    def AddEnergyAtDelay(self, DelaySeconds, EnergyValue, Band=0):  # FSAC.h:87-91
        c = self._ctx()
        c.check(c.lib.fs_add_energy_at_delay(c.h, self._src, Band, float(DelaySeconds), float(EnergyValue)))

    def ReconstructImpulseResponse(self, params=None):  # FSAC.cpp:320-380
        self._ctx().reconstruct_impulse_response(self._src, params)

    def GetImpulseResponse(self):  # FSAC.h:113 -> [NumChannels][NumSamples]
        c = self._ctx()
        return [c.impulse_response_view(self._src, ch) for ch in range(c.cfg.num_channels)]

    def SetImpulseResponse(self, ir):
        """write through the mutable reference GetImpulseResponse() returns (FSAC.h:113)"""
        self._ctx().set_impulse_response(self._src, ir)

    def GenerateDummyImpulseResponse(self):  # FSAC.cpp:408-452 as it ends up: a delta at samples 0 and N-1
        ir = np.zeros(self.NumSamples, np.float32)
        ir[0] = 1.0
        ir[-1] = 1.0
        self.SetImpulseResponse(ir)
        return ir

    def GetBandImpulseResponse(self, band):
        return self._ctx().band_impulse_response(self._src, band)

    # legacy per-frame forward tracer (TickComponent -> UpdateSound, FSAC.cpp:103-109, 283-306)
    RaycastsPerTick = 1500      # FSAC.h:39
    RaycastBounces = 10         # FSAC.h:42
    RaycastDistance = 5000.0    # FSAC.h:45

    def UpdateSound(self, seed=0x5EED, listener_radius=34.0):
        self._subsys._commit()
        p = _capi.default_sound_params(raycasts_per_tick=self.RaycastsPerTick, raycast_bounces=self.RaycastBounces,
                                       raycast_distance=self.RaycastDistance, seed=seed,
                                       listener_radius=listener_radius)
        return self._ctx().update_sound(self._src, p)

    def GetOcclusionAttenuation(self):  # FSAC.h:112
        return self._ctx().occlusion_attenuation(self._src)


class AudioRayTracingSubsystem:
    """UAudioRayTracingSubsystem's registries and per-source update (ARTS.h:86-196, ARTS.cpp:45-195)."""

    USED_RAY_COUNT = 1000  # ARTS.h:176

    def __init__(self, num_bands=1, device=0, rank=0, world_size=1, stream=None, comm_id=None):
        """comm_id: the 128 bytes of Context.comm_unique_id() (made by rank 0, shipped to all ranks): the subsystem of
        a sharded run (world_size > 1) then sums every frame's energy buffer over the ranks inside the library"""
        self.ctx = Context(num_bands=num_bands, device=device, rank=rank, world_size=world_size, stream=stream)
        if comm_id is not None:
            self.ctx.comm_init(comm_id)
        self.ActiveSources = []
        self._geom = []          # registered (triangles, material_ids, actor ids)
        self._next_actor = 0
        self._materials = None
        self._dirty = True
        self._committed = False   # the first commit builds the tree on the host (SAH), later registration changes on the device
        self.params = _capi.default_params()

    def Deinitialize(self):
        self.ctx.close()

    # RegisterGeometry / UnregisterGeometry (ARTS.h:99-100): a "component" is a triangle set + material ids
    def RegisterGeometry(self, triangles, material_ids, object_ids=None):
        """one UAcousticGeometryComponent = one actor, unless per-triangle actor ids are given"""
        tri = np.asarray(triangles, dtype=np.float32).reshape(-1, 3, 3)
        if object_ids is None:
            obj = np.full(tri.shape[0], self._next_actor, dtype=np.uint32)
            self._next_actor += 1
        else:
            obj = np.asarray(object_ids, dtype=np.uint32).reshape(-1)
            self._next_actor = max(self._next_actor, int(obj.max(initial=0)) + 1)
        comp = (tri, np.asarray(material_ids, dtype=np.uint16).reshape(-1), obj)
        self._geom.append(comp)
        self._dirty = True
        return comp

    def UnregisterGeometry(self, comp):
        self._geom = [g for g in self._geom if g is not comp]
        self._dirty = True

    def SetMaterials(self, absorption, transmission=None, scattering=None):
        self._materials = (absorption, transmission, scattering)
        self._dirty = True

    def _commit(self):
        if not self._dirty:
            return
        if self._geom:
            tri = np.concatenate([g[0] for g in self._geom], axis=0)
            mat = np.concatenate([g[1] for g in self._geom], axis=0)
            obj = np.concatenate([g[2] for g in self._geom], axis=0)
        else:
            tri, mat, obj = np.zeros((0, 3, 3), np.float32), np.zeros((0,), np.uint16), None
        ab, tr, sc = self._materials if self._materials is not None else (
            np.zeros((0, self.ctx.num_bands), np.float32), None, None)
        # the first commit builds the SAH tree on the host; a registration change during play takes the device-built tree at
        # once and gets the SAH tree swapped in when the library's background thread has built it
        self.ctx.set_scene(tri, mat, ab, tr, sc, object_ids=obj, fast="progressive" if self._committed else False)
        self._dirty = False
        self._committed = True

    def RebuildQuality(self):
        """after run-time registration changes (device-built Morton tree): the host's SAH build again, when a frame can afford it"""
        self._dirty, self._committed = True, False
        self._commit()

    def GeometryMoved(self, comp_index, triangles):
        """A registered geometry component (index in registration order) moved: same triangle count, new
        world-space positions.  Rewritten in place + device refit before the next trace (no rebuild)."""
        self._commit()
        first = sum(g[0].shape[0] for g in self._geom[:comp_index])
        tri = np.ascontiguousarray(triangles, dtype=np.float32).reshape(-1, 3, 3)
        if tri.shape[0] != self._geom[comp_index][0].shape[0]:
            raise ValueError("a moved component keeps its triangle count")
        self._geom[comp_index] = (tri, *self._geom[comp_index][1:])
        self.ctx.update_triangles(first, tri)

    def RegisterSource(self, InComp: FrequenSeeAudioComponent):  # ARTS.cpp:45-48
        InComp._subsys = self
        InComp._src = self.ctx.create_source(InComp._location)
        self.ActiveSources.append(InComp)

    def UnRegisterSource(self, InComp: FrequenSeeAudioComponent):  # ARTS.cpp:50-53
        if InComp in self.ActiveSources:
            self.ActiveSources.remove(InComp)
            self.ctx.destroy_source(InComp._src)
            InComp._subsys = None
            InComp._src = None

    def SetListenerLocation(self, xyz):  # PlayerPawn->GetActorLocation(), ARTS.cpp:287
        self.ctx.set_listener(xyz)

    def UpdateSource(self, Src: FrequenSeeAudioComponent, params=None):
        """ARTS.cpp:128-195: trace, evaluate, flush, deposit, reconstruct.  Returns the energy buffer."""
        self._commit()
        p = params or self.params
        e = self.ctx.compute_energy_response(Src._src, p)
        self.ctx.reconstruct_impulse_response(Src._src, p)
        return e

    def ForceUpdateSources(self):  # ARTS.cpp:883-886 — all active sources in one batched frame
        srcs = list(self.ActiveSources)
        if not srcs:
            return
        self._commit()
        self.ctx.update_sources([s._src for s in srcs], self.params)

    def SetPipelining(self, depth):
        """fs_set_pipelining: Tick then updates the sources one after the other like the reference's loop (ARTS.cpp:60-68),
        streamed — every call launches one kernel that plans this source's frame, walks the previous source's and
        connects the one before — and ends with fs_submit; the IRs appear as they are published"""
        self.ctx.set_pipelining(depth)
        self._streamed = bool(depth)

    def SetFramesPerLaunch(self, n):
        """fs_set_frames_per_launch: the streamed sources of a Tick share launches n at a time (every source keeps its own
        seed, energy buffer and IR; Tick's closing fs_submit sends a partial group off)"""
        self.ctx.set_frames_per_launch(n)

    def Tick(self, DeltaTime):  # ARTS.cpp:55-85 without the 1 s warm-up: the caller drives every frame
        if not self.ActiveSources:
            return
        if getattr(self, "_streamed", False):
            self._commit()
            for s in self.ActiveSources:
                self.ctx.compute_energy_response_async(s._src, self.params)
                self.ctx.reconstruct_impulse_response_async(s._src, self.params)
            self.ctx.submit()
        else:
            self.ForceUpdateSources()
        # the reference draws from the engine's global rand() stream: every frame sees fresh samples
        self.params.seed = (self.params.seed + 1) & 0xFFFFFFFFFFFFFFFF

    def LineTraceSingle(self, start, end):
        """closest hit on the segment [start, end] (UWorld::LineTraceSingleByObjectType contract)"""
        self._commit()
        s, e = np.asarray(start, np.float32), np.asarray(end, np.float32)
        d = e - s
        ln = float(np.linalg.norm(d))
        if ln <= 0:
            return False, 0.0, -1, np.zeros(3, np.float32)
        hit, t, tri, n = self.ctx.trace_rays(s[None], (d / ln)[None], ln)
        return bool(hit[0]), float(t[0]), int(tri[0]), n[0]


class FrequenSeeAudioReverbPlugin:
    """FFrequenSeeAudioReverbPlugin (Private/FrequenSeeAudioReverbPlugin.h:46-47, .cpp:74-213): per audio
    callback, convolve the last IR-1 + BufferLength samples of a source with its impulse response."""

    def __init__(self, subsystem: AudioRayTracingSubsystem):
        self.ctx = subsystem.ctx
        self.FrameSize = 1024      # AudioCallbackBufferFrameSize, Config/DefaultEngine.ini:13

    def Initialize(self, BufferLength=1024):
        self.FrameSize = int(BufferLength)

    def OnInitSource(self, component: FrequenSeeAudioComponent):
        self.ctx.reverb_init(component._src, self.FrameSize)

    def OnReleaseSource(self, component: FrequenSeeAudioComponent):
        self.ctx.reverb_release(component._src)

    def ProcessSourceAudio(self, component: FrequenSeeAudioComponent, AudioBuffer, literal_tail=False):
        return self.ctx.reverb_process(component._src, AudioBuffer, apply_reverb=component.bApplyReverb,
                                       literal_tail=literal_tail)


class MaterialAcousticProcessor:
    """Mirror of UMaterialAcousticProcessor (Public/MaterialAcousticProcessor.h:55-70).  Props is any object
    with Absorption / Transmission / Scattering response arrays (FMaterialAcousticFD, .h:24-37) or a 3-tuple."""

    def __init__(self, subsystem: AudioRayTracingSubsystem):
        self.subsystem = subsystem

    def ApplyMaterialFD(self, InBuffer, Props):
        if isinstance(Props, (tuple, list)):
            a, t, s = Props
        else:
            a, t, s = Props.Absorption, Props.Transmission, Props.Scattering
        spec, diff, trans = self.subsystem.ctx.apply_material_fd(InBuffer, a, t, s)
        return {"Specular": spec, "Diffuse": diff, "Transmitted": trans}   # FAcousticOutputs (.h:40-53)
