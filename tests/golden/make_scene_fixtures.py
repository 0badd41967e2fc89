#!/usr/bin/env python3
"""Regenerates tests/golden/scenes/*.json from the reference's shipped scene
files (scenes/*.blend in the reference checkout) through this repository's
own .blend loader.  The fixtures are decoded DATA (node transforms, vertex
arrays, materials, lamps, camera), in the import-level form documented in
include/acr_host.h, so that tests and the bench can run where the reference
checkout is absent.

    python tests/golden/make_scene_fixtures.py [/root/reference]
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

import cudaraytracer_amd as acr  # noqa: E402


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    out = os.path.join(HERE, "scenes")
    os.makedirs(out, exist_ok=True)
    for name in ("triangles", "plane", "cornell-box-cycles"):
        d = acr.load_blend(os.path.join(ref, "scenes", name + ".blend"))
        path = os.path.join(out, name + ".json")
        with open(path, "w") as f:
            f.write(d.to_json())
        print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
