"""Latency of the legacy per-frame forward tracer (row a9: UpdateSound, FSAC.cpp:283-306 — 1500 rays x up to 10
bounces, a listener-directed transmission ray per bounce; the only tracing the reference does every frame at HEAD)
through fs_update_sound, against the CPU oracle on one core.  Run on the GPU box:
    python tests/measure_update_sound.py > gpurun_out/update_sound.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
import oracle  # noqa: E402


def main():
    pkg = graft.load_package()
    out = []
    for name, bands in (("starter_room", 4), ("old_mine", 8)):
        sc = pkg.scenes.by_name(name, bands)
        ctx = pkg.Context(num_bands=bands)
        ctx.set_scene(sc.triangles, sc.material_ids, sc.absorption)
        ctx.set_listener(sc.listener)
        src = ctx.create_source(sc.source)
        for _ in range(5):
            r = ctx.update_sound(src)
        n = 200
        t0 = time.perf_counter()
        for _ in range(n):
            r = ctx.update_sound(src)
        gpu_ms = (time.perf_counter() - t0) / n * 1e3
        ctx.close()
        osc = oracle.Scene(sc.triangles, sc.material_ids, sc.absorption)
        t0 = time.perf_counter()
        ro = osc.update_sound(sc.source, sc.listener)
        cpu_ms = (time.perf_counter() - t0) * 1e3
        out.append({"scene": name, "triangles": int(sc.num_triangles), "gpu_ms_per_update": gpu_ms,
                    "oracle_1_core_ms_per_update": cpu_ms, "traces": int(r["traces"]),
                    "occlusion_attenuation": float(r["occlusion_attenuation"]),
                    "oracle_traces": int(ro["traces"])})
    print(json.dumps({"what": "fs_update_sound latency incl. result readback (UpdateSound, FSAC.cpp:283-306)", "runs": out}))


if __name__ == "__main__":
    main()
