"""ModelOpts defaults of the reference (lib/models/model_opts.py:10-39); the native engine implements exactly
this configuration (absolutetrack_amd/arch.py)."""
from dataclasses import dataclass


@dataclass
class ModelOpts:
    network: str = "resnet_layers_2352-f32"
    nImageFeatureChannels: int = 72
    nSkeletonFeatureChannels: int = 4
    nTemporalMemoryChannels: int = 18
    useUnscaledAsCanonical: bool = False
    nMultiViewFusionBlocks: int = 2
    nTemporalBlocks: int = 3
    nPoseRegressionBlocks: int = 2
    spatialFTLRatio: float = 1.0
    temporalFTLRatio: float = 1.0
    nWristRigidPts: int = 7
