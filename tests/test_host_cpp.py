"""The C++ host layer (include/tinyrenderer/*.hpp: the reference's helpers / model_renderer modules
over the C ABI).  CPU: it compiles with g++, links the library, and builds bit for bit the matrices
the Python mirror builds.  GPU: the reference's frame loop (examples/frame_loop.cpp, src/main.rs) renders
the frame the oracle renders."""
import math
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "tinyrenderer_amd")


def _col_major(m):
    return [float(v) for v in np.asarray(m, dtype=np.float64).T.reshape(16)]


def test_cpp_host_matrices_match_python_mirror(tmp_path):
    from tinyrenderer_amd import helpers as H
    from tinyrenderer_amd.model_renderer import BuiltinShader, MeshComponent

    exe = str(tmp_path / "host_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"),
                    "-o", exe, os.path.join(ROOT, "tests", "ctools", "host_check.cpp"), "-L", LIBDIR,
                    "-ltinyrenderer_hip", f"-Wl,-rpath,{LIBDIR}"], check=True)
    lines = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.splitlines()
    got = {}
    for ln in lines:
        name, *vals = ln.split()
        got[name] = vals
    persp = H.make_perspective_matrix(10.0, 1.0, 0.35 * math.pi, 853.0 / 480.0)
    view = H.look_at_rh((0.0, 0.0, 2.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0))
    view2 = H.look_at_rh((1.5, 0.3, 4.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0))
    parent, child = MeshComponent(None, None, None), MeshComponent(None, None, None)
    parent.transform.set_rotation((0.025 * 7.0, 0.0, 0.0))
    parent.transform.set_position((0.05, 0.0, -0.01))
    child.transform.set_scale((1.1, 1.2, 1.3))
    child.transform.set_position((0.0, 0.2, 0.0))
    child.transform.set_rotation((0.0, 0.4, 0.0))
    child.parent = parent
    sh = BuiltinShader(persp, H.mat4_mul(view2, child.local_to_world_matrix()), (1.5, 0.3, 4.0))
    want = {
        "rotation": H.make_rotation_matrix(0.025, 0.3, -1.2),
        "translation": H.make_translation_matrix((0.5, -2.0, 3.25)),
        "scale": H.make_scale_matrix((1.5, 0.25, -2.0)),
        "perspective": persp,
        "look_at_z2": view,
        "look_at_oblique": view2,
        "product": H.mat4_mul(H.mat4_mul(persp, view2), H.make_rotation_matrix(0.1, 0.2, 0.3)),
        "transpose": view2.T,
        "inverse": H.try_inverse4(view2.T),
        "parent_matrix": parent.transform.matrix(),
        "child_matrix": child.transform.matrix(),
        "child_local_to_world": child.local_to_world_matrix(),
        "shader_pmv": sh.pmv(),
        "shader_normal_matrix": sh.normal_matrix,
    }
    for name, m in want.items():
        vals = [float.fromhex(v) for v in got[name]]
        assert vals == _col_major(m), name
    assert got["singular"] == ["0"]
    assert H.try_inverse4(np.diag([1.0, 1.0, 0.0, 1.0])) is None
    assert got["version"][0].startswith("tinyrenderer")


@pytest.mark.gpu
def test_cpp_frame_loop_renders_the_reference_frame(oracle, tmp_path):
    """examples/frame_loop (built by __graft_entry__.build()) is src/main.rs without the window:
    its frame for time = 1 equals the oracle's, colour for colour."""
    import scene_util as SU
    from tinyrenderer_amd import helpers as H
    from tinyrenderer_amd import scenes

    exe = os.path.join(ROOT, "examples", "frame_loop")
    assert os.path.exists(exe), "examples/frame_loop is missing: run __graft_entry__.build()"
    W, Hh = 853, 480
    out = str(tmp_path / "frame.rgba")
    res = subprocess.run([exe, H.ASSET_DIR, "1", str(W), str(Hh), out], capture_output=True, text=True, timeout=110)
    assert res.returncode == 0, res.stderr
    assert "Frametime (nosync)" in res.stdout
    got = np.fromfile(out, dtype=np.uint8).reshape(Hh, W, 4)
    head, eye = SU.component("head"), SU.component("eye_inner")
    head.transform.set_rotation((0.025, 0.0, 0.0))
    eye.parent = head
    draws = [(head, scenes.shader_for(head, W, Hh)), (eye, scenes.shader_for(eye, W, Hh))]
    want = SU.oracle_render(oracle, W, Hh, draws)["present"]
    eye.parent = None
    SU.assert_image(got, want, "frame_loop")


@pytest.mark.gpu
def test_cpp_frame_loop_in_batches_of_iterations(oracle, tmp_path):
    """The same loop rendered ahead in batches of 4 iterations of BOTH meshes (render_views_multi ->
    tr_render_views_multi): 7 iterations = a batch of 4 and one of 3; the last slice is the frame for time = 7."""
    import scene_util as SU
    from tinyrenderer_amd import helpers as H
    from tinyrenderer_amd import scenes

    exe = os.path.join(ROOT, "examples", "frame_loop")
    W, Hh = 853, 480
    out = str(tmp_path / "frame7.rgba")
    res = subprocess.run([exe, H.ASSET_DIR, "7", str(W), str(Hh), out, "1", "4"], capture_output=True, text=True, timeout=110)
    assert res.returncode == 0, res.stderr
    assert "batches of 4 iterations" in res.stdout
    got = np.fromfile(out, dtype=np.uint8).reshape(Hh, W, 4)
    head, eye = SU.component("head"), SU.component("eye_inner")
    head.transform.set_rotation((0.025 * 7, 0.0, 0.0))
    eye.parent = head
    try:
        draws = [(head, scenes.shader_for(head, W, Hh)), (eye, scenes.shader_for(eye, W, Hh))]
        want = SU.oracle_render(oracle, W, Hh, draws)["present"]
    finally:
        eye.parent = None
        head.transform.set_rotation((0.025, 0.0, 0.0))
    SU.assert_image(got, want, "frame_loop batches")


@pytest.mark.gpu
def test_cpp_orbit_views_batch_renders_the_oracle_view(oracle, tmp_path):
    """examples/orbit_views: render_views (tr_render_views) from the C++ layer; view 5 of a 12-view orbit of
    diablo3_pose, taken out of the second batch of 4, equals the oracle's frame for that camera."""
    import scene_util as SU
    from tinyrenderer_amd import helpers as H
    from tinyrenderer_amd import scenes

    exe = os.path.join(ROOT, "examples", "orbit_views")
    assert os.path.exists(exe), "examples/orbit_views is missing: run __graft_entry__.build()"
    W, Hh, n, k = 480, 270, 12, 5
    out = str(tmp_path / "view.rgba")
    res = subprocess.run([exe, H.ASSET_DIR, str(n), "4", str(W), str(Hh), str(k), out], capture_output=True, text=True, timeout=110)
    assert res.returncode == 0, res.stderr
    got = np.fromfile(out, dtype=np.uint8).reshape(Hh, W, 4)
    mc = SU.component("diablo")
    mc.transform.set_rotation((0.0, 0.0, 0.0))
    want = SU.oracle_render(oracle, W, Hh, [(mc, scenes.shader_for(mc, W, Hh, eye=scenes.orbit_eye(k, n)))])["present"]
    SU.assert_image(got, want, "orbit_views")
