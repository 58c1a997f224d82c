"""MI355X-native render hot path of fuchsteufelswild/Ray-Tracer (package directory: ray-tracer_amd).

Import with importlib.import_module("ray-tracer_amd") (the hyphen is the name the project was given).
"""
from . import abi, scene, renderer  # noqa: F401
from .renderer import Context, Mi355Accelerator, Mi355Renderer, RtError, library  # noqa: F401
from .scene import SceneBuilder, SceneSoup, load_xml  # noqa: F401
