"""Harness-side import of the read-only reference (this container only).

The reference pulls `cv2` (via src/data, never called on the inference path) and
`timm.models.layers` (construction/init helpers only).  Neither is installed, so two
sys.modules stubs are registered before import; no reference file is touched.
Used ONLY by tests/golden/make_golden.py to generate fixtures.  Never shipped to /
imported on the GPU box (/root/reference does not exist there).
"""
import sys, types, os
import torch

REF_ROOT = os.environ.get("FF_REFERENCE_ROOT", "/root/reference")


def install_shims():
    if "cv2" not in sys.modules:
        sys.modules["cv2"] = types.ModuleType("cv2")
    if "timm" not in sys.modules:
        timm = types.ModuleType("timm")
        models = types.ModuleType("timm.models")
        layers = types.ModuleType("timm.models.layers")

        def to_2tuple(x):
            return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

        class DropPath(torch.nn.Module):
            def __init__(self, p=0.0):
                super().__init__()

            def forward(self, x):
                return x

        layers.to_2tuple = to_2tuple
        layers.trunc_normal_ = torch.nn.init.trunc_normal_
        layers.DropPath = DropPath
        timm.models = models
        models.layers = layers
        sys.modules["timm"] = timm
        sys.modules["timm.models"] = models
        sys.modules["timm.models.layers"] = layers
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)


def build_reference_model():
    """Reference ExpertEnsemble + CompleteEnhancedFusionSR on CPU, random init, eval mode."""
    install_shims()
    import contextlib, io as _io
    with contextlib.redirect_stdout(_io.StringIO()):
        from src.models.enhanced_fusion import CompleteEnhancedFusionSR
        from src.models import expert_loader
        sys.path.insert(0, REF_ROOT)
        import importlib
        plug = importlib.import_module("models.team29_FreqFusion.io")
        cfg = plug.MODEL_CONFIG
        ens = expert_loader.ExpertEnsemble(upscale=4, device=torch.device("cpu"))
        ens.load_all_experts(checkpoint_paths={"hat": "/nonexistent", "dat": "/nonexistent", "nafnet": "/nonexistent"}, freeze=True)
        model = CompleteEnhancedFusionSR(
            expert_ensemble=ens, num_experts=cfg["num_experts"], num_bands=cfg["num_bands"],
            block_size=cfg["block_size"], upscale=cfg["scale"], fusion_dim=cfg["fusion_dim"],
            num_heads=cfg["num_heads"], refine_depth=cfg["refine_depth"], refine_channels=cfg["refine_channels"],
            enable_hierarchical=True, enable_multi_domain_freq=True, enable_lka=True, enable_edge_enhance=True,
            enable_dynamic_selection=True, enable_cross_band_attn=True, enable_adaptive_bands=True,
            enable_multi_resolution=True, enable_collaborative=True)
    model.eval()
    return model, ens
