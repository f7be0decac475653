from .sit import SiT, SiT_models  # noqa: F401
