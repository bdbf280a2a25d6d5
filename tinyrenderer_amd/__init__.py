"""MI355X-native rasterizer behind Davidster/tinyrenderer's model_renderer / helpers surface."""
from . import helpers  # noqa: F401
from .helpers import (  # noqa: F401
    BLACK, BLUE, GREEN, RED, WHITE, flip_vertically, load_model, load_rgba_img_from_file, look_at_rh,
    make_perspective_matrix, make_rotation_matrix, make_scale_matrix, make_translation_matrix,
)
from .model_renderer import (  # noqa: F401
    BuiltinShader, MeshComponent, ModelRendererState, Transform, clear_screen, render_mesh_component,
)
