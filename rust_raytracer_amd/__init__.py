"""MI355X-native render path of rust_raytracer: ctypes binding of librt_hip.so and the host-side mirror of the
reference's frame-level renderer interface."""


def __getattr__(name):
    # imported on first use, so that `import rust_raytracer_amd` stays cheap
    if name == "HipRenderer":
        from .renderer import HipRenderer
        return HipRenderer
    raise AttributeError(name)
