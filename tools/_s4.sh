mkdir -p gpurun_out/r5i
python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/r5i/gputests.txt 2>&1
tail -4 gpurun_out/r5i/gputests.txt
python tools/exp/split_precision/bench_split_conv.py 2>&1 | grep -v amdgpu > gpurun_out/r5i/split_conv.txt; tail -4 gpurun_out/r5i/split_conv.txt | cut -c1-200
python bench.py --workload istnet --steps 10 --warmup 5 --no-eager-leg --split-precision 2>/dev/null | tail -1 > gpurun_out/r5i/istnet_split.json
python -c "
import json; d=json.load(open('gpurun_out/r5i/istnet_split.json')); print(d['ms_per_step'], d['split_precision']['ms_per_step'], d['split_precision']['rgb_features_max_rel_diff_vs_fp32_mfma'])"
