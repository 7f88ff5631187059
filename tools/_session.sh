mkdir -p gpurun_out/r04
O=gpurun_out/r04
bash tools/pmc_traffic.sh r04 > $O/pmc_traffic.txt 2>&1
cp gpurun_out/pmc_r04_traffic.json $O/pmc_traffic.json
rm -rf gpurun_out/pmc_r04_FETCH_SIZE gpurun_out/pmc_r04_WRITE_SIZE
tail -5 $O/pmc_traffic.txt | cut -c1-160
python -m pytest tests -m gpu -q 2>&1 | grep -v "Warning:\|warnings.warn\|amdgpu.ids\|shared_mlp_maxpool\|^$\|^tests/" | tail -6 > $O/gpu_tests.txt; tail -2 $O/gpu_tests.txt
