"""TEST STAND-IN for ``stable_baselines`` (absent from the build image): only ``common.vec_env.VecEnv``, the abstract base the
product's batched env derives from when asked to (BSKGPU_SB_VECENV=1).  tests/test_gym_boundary.py, subprocess only."""
