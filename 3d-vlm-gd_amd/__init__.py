"""MI355X-native (gfx950) engine for the geometric-distillation fine-tuning hot path of
kaist-cvml/3d-vlm-gd: hand-written HIP kernels behind a C ABI (csrc/, include/gd_hip.h) and a
host-side mirror of the reference's module / function surface.  Import as `gd_amd` via the
repo-root shim."""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
