#!/usr/bin/env python3
"""Per-tile cost map of one frame: wave cycles (summed over the 16 waves of each 64x16 tile) and rays per tile.
    python tools/costmap.py [--soup N --depth D --edge E --extent X | --scene name] [--size WxH] [--bounces B]
Prints the distribution; writes gpurun_out/costmap_<tag>.npy."""
import argparse, ctypes as C, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cudaraytracer_amd as acr
from cudaraytracer_amd import _capi


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="cornell-box-cycles")
    ap.add_argument("--soup", type=int, default=0)
    ap.add_argument("--depth", type=int, default=22)
    ap.add_argument("--edge", type=float, default=0.1)
    ap.add_argument("--extent", type=float, default=10.0)
    ap.add_argument("--size", default="1920x1080")
    ap.add_argument("--bounces", type=int, default=1)
    ap.add_argument("--tag", default="map")
    a = ap.parse_args()
    w, h = map(int, a.size.split("x"))
    if a.soup:
        lib = _capi.host_lib()
        hh = lib.acr_host_import_soup(a.soup, 1, C.c_float(a.edge), C.c_float(a.extent))
        data = acr.ImportData.from_view(lib.acr_host_import_view(hh).contents)
        lib.acr_host_import_destroy(hh)
        view, keep = data.pack()
        scene = acr.Scene(import_view=view, mesh_bih_depth=a.depth)
    else:
        data = acr.ImportData.load(os.path.join(ROOT, "tests", "golden", "scenes", a.scene + ".json"))
        view, keep = data.pack()
        scene = acr.Scene(import_view=view)
    core = acr.HipCore(0)
    core.upload_scene(scene.soa)
    core.set_camera(scene.camera)
    tw, th = 64, 16
    tx, ty = (w + tw - 1) // tw, (h + th - 1) // th
    n = tx * ty
    dev = torch.device("cuda", 0)
    tiles = torch.zeros((n, tw * th, 4), dtype=torch.float32, device=dev)
    cyc = torch.zeros(n, dtype=torch.int32, device=dev)
    rays = torch.zeros(n, dtype=torch.int32, device=dev)
    ids = np.arange(n, dtype=np.uint32)
    for _ in range(2):
        core.render(w, h, seed=777, frames=0, max_bounces=a.bounces, tile_ids=ids, tile_w=tw, tile_h=th,
                    screen_dev=C.c_void_p(tiles.data_ptr()), tile_rays_dev=C.c_void_p(rays.data_ptr()),
                    tile_cycles_dev=C.c_void_p(cyc.data_ptr()))
        core.sync()
    c = cyc.cpu().numpy().astype(np.float64).reshape(ty, tx) / 16.0  # mean wave-clock ticks per wave of the tile
    r = rays.cpu().numpy().reshape(ty, tx)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.save(os.path.join(ROOT, "gpurun_out", "costmap_%s.npy" % a.tag), np.stack([c, r.astype(np.float64)]))
    q = np.percentile(c, [0, 10, 50, 90, 99, 100])
    print("mean wave ticks per tile: min %.0f p10 %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f; sum %.3g" % (*q, c.sum()))
    print("rows (mean over x) every 4th tile row:", np.round(c.mean(axis=1)[::4]).astype(int).tolist())
    print("cols (mean over y) every 2nd tile col:", np.round(c.mean(axis=0)[::2]).astype(int).tolist())
    print("rays per tile: min %d max %d" % (r.min(), r.max()))
    core.close()


if __name__ == "__main__":
    main()
