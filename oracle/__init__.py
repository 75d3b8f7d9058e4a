"""CPU oracle for the UmeTrack per-frame inference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under absolutetrack_amd/ or lib/ may import,
call or link anything in this package; only tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py do, and only as the checker.

The modules restate, from the text of the reference, the algorithms of
SURVEY.md section 8(a) rows a1-a13 on the CPU (numpy for geometry / FK / resampling,
torch-CPU functional ops for the fp32 CNN).  Each function cites the reference
file:line it follows.

Pinning status (see DESIGN.md "Oracle"):
  * model rows a3-a10,a13  pinned: tests/golden/model_*.npz are outputs of the
    reference's own lib.models code run in the build container
    (oracle/gen_goldens.py) on seeded synthetic weights/inputs.
  * geometry of a1/a2 (coordinate maps, intrinsics, extrinsics, crop cameras from
    points) pinned the same way from lib.common.{camera,crop,affine}.
  * FK row a12: pinned by the reference's stored outputs
    sample_data/user05/recording_{00,02,11}.npy['gt_keypoints'] (read with a
    non-executing tokenizer, oracle/stored_eval.py).  pytorch3d (`so3_exp_map`,
    unpinned "@stable") is absent from the image; its published formula is restated.
  * bilinear interpolation arithmetic of cv2.remap (a1): PARITY UNPINNED - OpenCV
    is absent and the reference stores no crop; float-bilinear and an OpenCV
    fixed-point emulation are both provided and compared self-consistently only.
  * section 8(f) rows: f1 crop-camera generation is the geometry pinned above; f2 torch_data path
    (ref_torch_data.py) pinned, bit-identical to the reference's lib.batched_dataset.data_transform on
    tests/golden/torch_data.npz (its _unpack_batched_data lives in a script importing pytorch3d-dependent
    modules and is restated from the text: index bookkeeping only); f4 metrics pinned by
    tests/golden/metrics.npz (reference load_eval._compute_metrics / lib.common.metric_utils); f3 file
    fixtures were parsed back by the reference's own TorchIdx when they were generated.
"""
