"""Ad-hoc GPU parity probe (not collected by pytest): renders the fixtures on the GPU and the oracle and prints differences."""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.join(HERE, ".."))
import numpy as np
import cudaraytracer_amd as acr
import oracle

def run(name, W, H, bounces, seed=7, frames=0):
    d = acr.ImportData.load(os.path.join(HERE, "golden", "scenes", name + ".json"))
    view, keep = d.pack()
    s = acr.Scene(import_view=view)
    o = oracle.OracleScene(view)
    core = acr.HipCore(0)
    core.upload_scene(s.soa)
    core.set_camera(s.camera)
    core.resize(W, H)
    t0 = time.time()
    core.render(W, H, seed=seed, frames=0, max_bounces=bounces, collect_stats=True)
    core.sync()
    t1 = time.time()
    st = core.read_stats()
    img = core.read_framebuffer(); pid = core.read_primid(); ht = core.read_hit_t()
    ref = o.render(W, H, seed=seed, frames=0, max_bounces=bounces)
    nd = int((pid != ref["prim"]).sum())
    td = int((ht.view(np.uint32) != ref["t"].view(np.uint32)).sum())
    idf = int((img.view(np.uint32) != ref["screen"].view(np.uint32)).sum())
    c = ref["counters"]
    print(f"{name} {W}x{H} b{bounces}: prim diff {nd}, t bit-diff {td}, rgba bit-diff words {idf}/{img.size}, "
          f"maxabs {np.abs(img-ref['screen']).max():.3g}; kernel {st.kernel_ms:.3f} ms, wall {1e3*(t1-t0):.1f} ms, lds {st.lds_bytes} mode {st.scene_in_lds}")
    print("   gpu rays", st.rays_primary, st.rays_shadow, st.rays_bounce, st.primary_hits, " oracle", c["rays_primary"], c["rays_shadow"], c["rays_bounce"], c["primary_hits"])
    if frames:
        for f in range(1, frames + 1):
            core.render(W, H, seed=seed + f, frames=f, max_bounces=bounces)
            ref = o.render(W, H, seed=seed + f, frames=f, max_bounces=bounces, screen=ref["screen"])
        core.sync()
        img = core.read_framebuffer()
        print("   after", frames, "accumulated frames: rgba bit-diff words", int((img.view(np.uint32) != ref["screen"].view(np.uint32)).sum()))
    core.close()

if __name__ == "__main__":
    run("triangles", 640, 480, 1)
    run("plane", 320, 240, 1)
    run("cornell-box-cycles", 480, 270, 1, frames=3)
    run("cornell-box-cycles", 240, 136, 5)
    run("cornell-box-cycles", 1920, 1080, 1)
