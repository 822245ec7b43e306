"""MI355X-native path-tracing core: HIP kernels behind a C ABI (libptcore.so)
plus a host-side mirror of the reference renderer interface."""
from . import renderer, scenes  # noqa: F401
from .renderer import (Camera, RenderContext, RenderResources, Scene, create_camera,  # noqa: F401
                       create_context, create_render_resources, default_camera, draw_frame,
                       reset_render, write_to_png)
