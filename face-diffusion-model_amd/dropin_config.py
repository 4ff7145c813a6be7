"""VQ-VAE argument namespaces per dataset preset (shared by the drop-in shims and fdm_amd.pipeline)."""
import importlib.util
import os

_spec = importlib.util.spec_from_file_location("_fdm_dropin_vq_config",
                                               os.path.join(os.path.dirname(os.path.abspath(__file__)), "dropin", "models", "utils", "config.py"))
_cfg = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_cfg)


def vq_args_for(preset):
    return {"vocaset": _cfg.vocaset_vq_vae_args, "mead": _cfg.vq_vae_args, "biwi": _cfg.biwi_vq_vae_args}[preset]()
