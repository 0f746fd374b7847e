"""Drop-in replacement for DQO-MAP's `cuda_utils` package (submodules/cuda_utils) on MI355X: `from cuda_utils._C import
accumulate_gaussian_error` (SLAM/multiprocess/mapper.py:23)."""
