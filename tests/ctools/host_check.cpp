// Prints the matrices the C++ host layer builds for a fixed set of inputs, as hex floats
// (column-major), one "name v0 .. v15" line each; tests/test_host_cpp.py compares them bit for bit
// with the Python mirror (tinyrenderer_amd/helpers.py, model_renderer.py).
#include <cstdio>

#include "tinyrenderer/model_renderer.hpp"

using namespace tinyrenderer;

static void out(const char* name, const Matrix4& m) {
    std::printf("%s", name);
    for (int i = 0; i < 16; ++i) std::printf(" %a", m.m[i]);
    std::printf("\n");
}

int main() {
    const double PI = 3.14159265358979323846;
    out("rotation", make_rotation_matrix(0.025, 0.3, -1.2));
    out("translation", make_translation_matrix({0.5, -2.0, 3.25}));
    out("scale", make_scale_matrix({1.5, 0.25, -2.0}));
    const Matrix4 persp = make_perspective_matrix(10.0, 1.0, 0.35 * PI, 853.0 / 480.0);
    out("perspective", persp);
    const Matrix4 view = look_at_rh({0.0, 0.0, 2.0}, {0.0, 0.0, 0.0}, {0.0, 1.0, 0.0});
    out("look_at_z2", view);
    const Matrix4 view2 = look_at_rh({1.5, 0.3, 4.0}, {0.0, 0.0, 0.0}, {0.0, 1.0, 0.0});
    out("look_at_oblique", view2);
    out("product", persp * view2 * make_rotation_matrix(0.1, 0.2, 0.3));
    out("transpose", view2.transpose());
    out("inverse", *view2.transpose().try_inverse());
    Matrix4 singular = Matrix4::identity();
    singular.at(2, 2) = 0.0;
    std::printf("singular %d\n", singular.try_inverse() ? 1 : 0);

    MeshComponent parent_mc, child;
    parent_mc.transform.set_rotation({0.025 * 7.0, 0.0, 0.0});
    parent_mc.transform.set_position({0.05, 0.0, -0.01});
    child.transform.set_scale({1.1, 1.2, 1.3});
    child.transform.set_position({0.0, 0.2, 0.0});
    child.transform.set_rotation({0.0, 0.4, 0.0});
    child.parent = &parent_mc;
    out("parent_matrix", parent_mc.transform.matrix());
    out("child_matrix", child.transform.matrix());
    out("child_local_to_world", child.local_to_world_matrix());
    const BuiltinShader sh = BuiltinShader::make(persp, view2 * child.local_to_world_matrix(), {1.5, 0.3, 4.0});
    out("shader_pmv", sh.perspective_matrix * sh.model_view_matrix);
    out("shader_normal_matrix", sh.inverse_model_view_matrix);
    std::printf("version %s\n", tr_version());
    return 0;
}
