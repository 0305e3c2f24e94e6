"""CPU oracle for the VideoGPA DPO hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch (fp32 / fp64, CPU) restatement of the
reference algorithm for the path named by BASELINE.json `north_star`.  It is
the checker the HIP kernels are compared against.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it;
the product package `videogpa_amd` never does (tests/test_layout.py enforces
that) and fails loudly when its HIP library is missing.

Pinning status:
  * dpo.py, dataset.py, scorer.{project_points, batch_reproject, motion_score,
    mse (+ resize branch), psnr, pointcloud_filter, mvcs, quat_to_mat,
    pose_encoding_to_extri_intri, affine_inverse, unproject_depth}: PINNED --
    checked against outputs of the importable reference modules (train/loss.py,
    train/dataset.py, utils/projection_utils.py, utils/pointcloud_utils.py,
    metrics/consistency_score.py, metrics/mse.py, metrics/mvcs.py,
    vggt/utils/pose_enc.py, depth_anything_3/utils/geometry.py) stored in
    tests/golden/{dpo_loss,scorer,scorer2}.pt + dataset_pairs.json by
    tests/golden/make_golden.py; tests/test_oracle_golden.py is the check
    (integer / index results bit-exact, fp32 results to the tolerance stated
    per test).
  * cogvideox.py (diffusers CogVideoXTransformer3DModel + the PEFT LoRA linear),
    scheduler.py (diffusers CogVideoXDPMScheduler: add_noise / get_velocity /
    set_timesteps / step), scorer.find_fundamental /
    sampson (kornia): PARITY UNPINNED -- those third-party packages
    (diffusers>=0.31.0, peft>=0.12.0, kornia>=0.7.3, requirements.txt:20,24,33)
    are not vendored in the reference and not installed here; the restatement
    follows their published algorithms and is anchored on the reference's call
    sites plus version-independent known-answer tests (tests/test_oracle_*.py).
"""
