mkdir -p gpurun_out/r4e
python -m pytest tests -m gpu -q 2>&1 | grep -v "Warning:\|warnings.warn\|amdgpu.ids\|shared_mlp_maxpool\|^$\|^tests/" | tail -40 > gpurun_out/r4e/test.txt; grep -n "Error\|^E \|passed\|failed\|FAILED" gpurun_out/r4e/test.txt | head -30
