#!/usr/bin/env python3
"""VERDICT r3 item 4, the bounded experiment on the named bound (L1 lookups per load instruction): how coherent are the node
requests of a dense walk wave, and what is the CEILING of re-binning subpaths for coherence?
  1. one cfg3 frame through the counting instantiation: lanes per node-request instruction, distinct 64-B records per lane;
  2. the headline stream (pipelined, two frames per launch) as it is, and with FS_DEBUG_COHERENT_WAVES=1 — all 64 lanes of
     a wave walk the SAME subpath: every request of a wave is one record (the L1 broadcasts), no re-binning by hit cell and
     direction octant can be more coherent than that.  Its rate over the normal rate bounds what any re-binning can buy,
     before the cost of the binning itself.
usage (GPU box): python tools/coherence_experiment.py > gpurun_out/r04_coherence_experiment.json"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402
pkg = graft.load_package()
sc = pkg.scenes.by_name("old_mine", 8)
out = {"workload": "cfg3_old_mine: 100000 tris, 262144 rays/frame, depth 8, 8 bands"}


def ctx_for(coherent):
    os.environ["FS_DEBUG_COHERENT_WAVES"] = "1" if coherent else "0"
    c = pkg.Context(num_bands=8)
    c.set_scene(sc.triangles, sc.material_ids, sc.absorption)
    c.set_listener(sc.listener)
    return c, c.create_source(sc.source)


c, s = ctx_for(False)
p = pkg.default_params(num_rays=262144, depth=8, seed=0x5EED)
c.set_profiling(3)
c.reset_stats()
c.compute_energy_response_async(s, p)
c.synchronize()
st = c.stats()
c.set_profiling(0)
out["node_requests"] = {"instructions": st["node_request_insts"], "lanes": st["node_request_lanes"], "distinct_records": st["node_request_distinct"],
                        "lanes_per_instruction": st["node_request_lanes"] / max(st["node_request_insts"], 1),
                        "distinct_records_per_lane": st["node_request_distinct"] / max(st["node_request_lanes"], 1),
                        "distinct_records_per_instruction": st["node_request_distinct"] / max(st["node_request_insts"], 1)}
c.close()
for label, coherent in (("as_it_is", False), ("all_lanes_of_a_wave_walk_one_subpath", True), ("as_it_is_again", False)):
    c, s = ctx_for(coherent)
    c.set_pipelining(2)
    c.set_frames_per_launch(2)
    for i in range(60):
        p.seed = 100 + i
        c.compute_energy_response_async(s, p)
        c.reconstruct_impulse_response_async(s, p)
    c.synchronize()
    k = 300
    t1 = time.perf_counter()
    for i in range(k):
        p.seed = 1000 + i
        c.compute_energy_response_async(s, p)
        c.reconstruct_impulse_response_async(s, p)
    c.submit()
    c.synchronize()
    el = time.perf_counter() - t1
    out[label] = {"ms_per_frame": 1e3 * el / k, "rays_per_s": 262144 * k / el}
    c.close()
out["ceiling_of_rebinning"] = out["all_lanes_of_a_wave_walk_one_subpath"]["rays_per_s"] / out["as_it_is"]["rays_per_s"]
print(json.dumps(out))
