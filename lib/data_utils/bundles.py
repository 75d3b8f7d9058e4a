"""lib/data_utils/bundles.py of the reference -> absolutetrack_amd.bundles (the three helpers the callers use)."""
from absolutetrack_amd.bundles import asdict, collate, group, is_dictlike, map_fields, to_device  # noqa: F401
