"""Architecture constants of the UmeTrack inference network and the
state-dict schema the reference loader expects.

Everything here is derived from the reference's constructors, not copied:
  * ModelOpts defaults                      lib/models/model_opts.py:10-39
  * stem / backbone / projection            lib/models/model_utils.py:107-138
  * ResNet "2352" layout, planes, strides   lib/models/backbone_resnet.py:75-192
  * fusion channel ramp np.linspace(144,72,3) lib/models/model_utils.py:141-163
  * temporal 90->90 x3                      lib/models/temporal.py:16-38
  * skeleton encoder Linear(132,144)+BN(4)  lib/models/skeleton_encoder.py:28-41
  * regressors (76->62, 72->63)             lib/models/model_loader.py:30-80,
                                            lib/models/regressor.py:50-73
The key names are the ones `load_state_dict` is strict about
(lib/models/model_loader.py:84-87).
"""
from typing import List, Tuple

CROP = 96                       # network input is 96x96 mono
STEM_PLANES = 32
LAYER_BLOCKS = (2, 3, 5, 2)
LAYER_PLANES = (32, 64, 128, 256)
LAYER_STRIDES = (1, 2, 2, 2)
FEAT_CH = 72                    # nImageFeatureChannels
FEAT_HW = 6                     # 96 / 2 / 8
FEAT_PIX = FEAT_HW * FEAT_HW
SKEL_CH = 4                     # nSkeletonFeatureChannels
MEM_CH = 18                     # nTemporalMemoryChannels
TEMPORAL_CH = FEAT_CH + MEM_CH  # 90
FUSION_CH = (144, 108, 72)      # np.linspace(144, 72, 3)
N_JOINTS = 22
N_FINGER_DOF = 20
N_LANDMARKS = 21
N_RIGID_PTS = 7
REG_K_IN, REG_K_OUT = FEAT_CH + SKEL_CH, 62
REG_U_IN, REG_U_OUT = FEAT_CH, 63
CANONICAL_FOCAL = 200.0         # lib/models/model_utils.py:167
BN_EPS = 1e-5

# pose record written by the native head: 22 ja + 16 xf + 1 scale + 21 sigma
POSE_REC = 60
# output slices of the regressors (lib/models/regressor.py:50-73)
REG_K_SLICES = {"joint_angles": (0, 20), "wrist_xfs": (20, 41),
                "landmark_uncertainty_sigmas": (41, 62)}
REG_U_SLICES = {"joint_angles": (0, 20), "wrist_xfs": (20, 41),
                "skel_scales": (41, 42), "landmark_uncertainty_sigmas": (42, 63)}

# FLOP accounting (2 x MAC over Conv2d/Linear only), SURVEY.md section 8(d)
FLOPS_PER_CROP_BACKBONE = 969_228_288
FLOPS_PER_HANDFRAME_KNOWN = 1_957_607_712
FLOPS_PER_HANDFRAME_UNKNOWN = 1_956_022_560

_BB = "_feature_extractor._image_backbone"


def _bn(prefix: str, c: int):
    return [(prefix + ".weight", (c,), "bn_w"), (prefix + ".bias", (c,), "bn_b"),
            (prefix + ".running_mean", (c,), "bn_mean"),
            (prefix + ".running_var", (c,), "bn_var"),
            (prefix + ".num_batches_tracked", (), "bn_nbt")]


def _basic_block(prefix: str, cin: int, cout: int, downsample: bool):
    out = [(prefix + ".conv1.weight", (cout, cin, 3, 3), "conv_w")]
    out += _bn(prefix + ".bn1", cout)
    out += [(prefix + ".conv2.weight", (cout, cout, 3, 3), "conv_w")]
    out += _bn(prefix + ".bn2", cout)
    if downsample:
        out += [(prefix + ".downsample.0.weight", (cout, cin, 1, 1), "conv_w")]
        out += _bn(prefix + ".downsample.1", cout)
    return out


def backbone_blocks() -> List[Tuple[str, int, int, int, bool]]:
    """(state-dict prefix, cin, cout, stride, has_downsample) for the 12 BasicBlocks."""
    blocks = []
    cin = STEM_PLANES
    for li, (nb, planes, stride) in enumerate(zip(LAYER_BLOCKS, LAYER_PLANES, LAYER_STRIDES)):
        for bi in range(nb):
            s = stride if bi == 0 else 1
            ds = bi == 0 and (s != 1 or cin != planes)
            blocks.append((f"{_BB}.0._layers.{li + 1}.{bi}", cin, planes, s, ds))
            cin = planes
    return blocks


def state_dict_spec() -> List[Tuple[str, tuple, str]]:
    """Ordered (key, shape, kind) list; 252 entries, 4,251,227 float parameters."""
    spec = [(f"{_BB}.0._layers.0.0.weight", (STEM_PLANES, 1, 3, 3), "conv_w"),
            (f"{_BB}.0._layers.0.0.bias", (STEM_PLANES,), "conv_b")]
    spec += _bn(f"{_BB}.0._layers.0.1", STEM_PLANES)
    for prefix, cin, cout, _s, ds in backbone_blocks():
        spec += _basic_block(prefix, cin, cout, ds)
    spec += [(f"{_BB}.1.weight", (FEAT_CH, LAYER_PLANES[-1], 1, 1), "conv_w"),
             (f"{_BB}.1.bias", (FEAT_CH,), "conv_b")]
    fu = "_feature_extractor._multi_view_fusion"
    spec += [(f"{fu}.0.weight", (FUSION_CH[1], FUSION_CH[0], 1, 1), "conv_w"),
             (f"{fu}.0.bias", (FUSION_CH[1],), "conv_b")]
    spec += _bn(f"{fu}.1", FUSION_CH[1])
    spec += [(f"{fu}.3.weight", (FUSION_CH[2], FUSION_CH[1], 1, 1), "conv_w"),
             (f"{fu}.3.bias", (FUSION_CH[2],), "conv_b")]
    spec += _bn(f"{fu}.4", FUSION_CH[2])
    spec += [(f"{fu}.6.weight", (FUSION_CH[2], FUSION_CH[2], 1, 1), "conv_w"),
             (f"{fu}.6.bias", (FUSION_CH[2],), "conv_b")]
    for i in (0, 2, 4):
        spec += [(f"_temporal._temporal_module.{i}.weight",
                  (TEMPORAL_CH, TEMPORAL_CH, 1, 1), "conv_w"),
                 (f"_temporal._temporal_module.{i}.bias", (TEMPORAL_CH,), "conv_b")]
    spec += [("_skeleton_enc._layers.0.weight", (SKEL_CH * FEAT_PIX, N_JOINTS * 6), "lin_w"),
             ("_skeleton_enc._layers.0.bias", (SKEL_CH * FEAT_PIX,), "lin_b")]
    spec += _bn("_skeleton_enc._layers.2", SKEL_CH)
    for name, cin, cout in (("_regressor_k", REG_K_IN, REG_K_OUT),
                            ("_regressor_u", REG_U_IN, REG_U_OUT)):
        p = f"{name}._pose_regression_layers"
        spec += _basic_block(f"{p}.0", cin, cin, False)
        spec += _basic_block(f"{p}.1", cin, cin, False)
        spec += [(f"{p}.2.weight", (cout, cin, 1, 1), "conv_w"),
                 (f"{p}.2.bias", (cout,), "conv_b")]
    return spec
