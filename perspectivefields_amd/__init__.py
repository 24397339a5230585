"""MI355X-native (gfx950) PerspectiveFields dense-field + ParamNet inference path.

    from perspectivefields_amd import PerspectiveFields      # or: from perspective2d import PerspectiveFields
    m = PerspectiveFields("Paramnet-360Cities-edina-centered").eval().cuda()
    pred = m.inference(img_bgr)

Sub-modules: config (zoo / cfg), schema (checkpoint keys), synth (seeded checkpoints and images),
engine (ctypes binding of libpf_hip.so), ops (kernel-level entry points), dist (image-level data
parallelism), build (hipcc build of csrc/).
"""
__all__ = ["PerspectiveFields", "model_zoo", "fields_from_params"]


def __getattr__(name):  # lazy: keeps `import perspectivefields_amd.synth` free of torch
    if name == "PerspectiveFields":
        from .perspectivefields import PerspectiveFields

        return PerspectiveFields
    if name == "fields_from_params":
        from .perspectivefields import fields_from_params

        return fields_from_params
    if name == "model_zoo":
        from .config import model_zoo

        return model_zoo
    raise AttributeError(name)
