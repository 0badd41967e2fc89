#!/usr/bin/env python3
"""Regenerates the oracle's own regression vectors:

  tests/golden/oracle_renders.npz   per scene and bounce count at 96x72: primitive ids, hit distances and
                                    RGBA (as bit patterns) of orc_render, seed 7, frame 0
  tests/golden/oracle_counters.json ray / node / slab / triangle counters of the oracle at the benchmark
                                    configurations (they give the algorithmic work per ray)

These pin the ORACLE against accidental change; they are not reference outputs
(the reference has none for this path).

    python tests/golden/make_oracle_goldens.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

import cudaraytracer_amd as acr  # noqa: E402
import oracle  # noqa: E402


def scene(name):
    d = acr.ImportData.load(os.path.join(HERE, "scenes", name + ".json"))
    view, keep = d.pack()
    return oracle.OracleScene(view), (view, keep)


def main():
    arrays = {}
    for name in ("triangles", "plane", "cornell-box-cycles"):
        o, keep = scene(name)
        for b in (1, 5):
            r = o.render(96, 72, seed=7, frames=0, max_bounces=b)
            key = f"{name}/b{b}"
            arrays[key + "/prim"] = r["prim"]
            arrays[key + "/t"] = r["t"].view(np.uint32)
            arrays[key + "/rgba"] = r["screen"].view(np.uint32)
    np.savez_compressed(os.path.join(HERE, "oracle_renders.npz"), **arrays)

    counters = {}
    configs = [("C1 triangles 640x480 primary+shadow", "triangles", 640, 480, 1, 12345),
               ("C2 cornell 1920x1080 primary+shadow", "cornell-box-cycles", 1920, 1080, 1, 12345),
               ("C3 cornell 3840x2160 4 bounces", "cornell-box-cycles", 3840, 2160, 4, 12345)]
    for label, name, w, h, b, seed in configs:
        o, keep = scene(name)
        r = o.render(w, h, seed=seed, frames=0, max_bounces=b, aov=False)
        c = r["counters"]
        rays = c["rays_primary"] + c["rays_shadow"] + c["rays_bounce"]
        c["per_ray"] = {k: round(c[k] / rays, 4) for k in ("node_visits", "slab_tests", "object_visits", "mesh_visits", "triangle_tests", "object_hits", "mesh_box_hits")}
        counters[label] = dict(scene=name, width=w, height=h, max_bounces=b, seed=seed, **c)
        print(label, c)
    with open(os.path.join(HERE, "oracle_counters.json"), "w") as f:
        json.dump(counters, f, indent=1)


if __name__ == "__main__":
    main()
