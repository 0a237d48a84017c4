#!/usr/bin/env python3
"""Generates tests/golden/legacy_update_sound.json: inputs and expected outputs of the legacy forward tracer
(UpdateSound / CastAudioRay / CastDirectAudioRay, FrequenSeeAudioComponent.cpp:132-306) computed by the oracle:

    python tests/golden/make_golden_legacy.py

Data only: scene name, parameter overrides, the fs_sound_result fields.  Pins the oracle (tests/test_golden.py)
and gives the GPU test an expected output that does not need the oracle at run time.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402
import oracle  # noqa: E402

CASES = [
    ("shoebox", {}),
    ("starter_room", {}),
    ("starter_room", {"seed": 99, "raycasts_per_tick": 777, "raycast_bounces": 4, "listener_radius": 60.0}),
    ("old_mine", {"raycast_distance": 1500.0}),
]


def main():
    pkg = graft.load_package()
    out = []
    for scene, kw in CASES:
        sc = pkg.scenes.by_name(scene)
        osc = oracle.Scene(sc.triangles, sc.material_ids, sc.absorption)
        osc.set_objects(sc.object_ids)
        r = osc.update_sound(sc.source, sc.listener, **kw)
        out.append({"scene": scene, "params": kw, "result": r})
        print(scene, kw, r)
    json.dump(out, open(os.path.join(HERE, "legacy_update_sound.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
