"""Camera paths built from the reference's own camera interaction, Renderer::moveCamera
(src/renderer/renderer.cu:145-166): every step is one moveCamera(pos, dir) call, evaluated by the host
library (acr_host_move_camera), so a path is exactly what a user holding the strafe key while turning
the mouse would produce."""
import ctypes as C
import math

from . import _capi


def moved(camera, pos, direction):
    """New Camera after one moveCamera(pos=(right, forward), dir=(yaw, pitch)) step."""
    cam = _capi.Camera()
    C.memmove(C.byref(cam), C.byref(camera), C.sizeof(cam))
    p = (C.c_float * 2)(*pos)
    d = (C.c_float * 2)(*direction)
    if _capi.host_lib().acr_host_move_camera(C.byref(cam), p, d) != 0:
        raise RuntimeError("acr_host_move_camera failed")
    return cam


def flythrough(camera, frames=120, sweep_degrees=30.0, radius=7.674):
    """`frames` cameras sweeping from -sweep to +sweep degrees around the point `radius` in front of the
    start camera: strafe along the chord and turn by the same angle each frame."""
    step = math.radians(2.0 * sweep_degrees / frames)
    chord = 2.0 * radius * math.sin(step / 2.0)
    cam = camera
    for _ in range(frames // 2):  # walk to the start of the sweep (not rendered)
        cam = moved(cam, (-chord, 0.0), (-step, 0.0))  # step left, turn right: stays aimed at the centre
    out = []
    for _ in range(frames):
        out.append(cam)
        cam = moved(cam, (chord, 0.0), (step, 0.0))
    return out
