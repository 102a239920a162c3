python -m pytest tests/test_gpu_kernels.py tests/test_gpu_effects_fullsize.py -x -q -m gpu 2>&1 | tail -3
python scripts/defocus_paths.py 1,2 2>&1 | grep -v amdgpu
