"""lib/models/model_loader.py of the reference -> absolutetrack_amd.model.load_pretrained_model."""
from absolutetrack_amd.model import load_pretrained_model  # noqa: F401
