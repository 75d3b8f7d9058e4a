"""lib/common/affine.py of the reference -> absolutetrack_amd.geometry."""
from absolutetrack_amd.geometry import (  # noqa: F401
    from_two_vectors, make_look_at_matrix, normalized, skew_matrix, transform3, transform_vec3)
