mkdir -p gpurun_out/r4h
python tools/bench_fps_chain.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4h/fps_chain.txt
for v in 1024 4096; do echo -n "infer ISTNET_FPS_TRACK_MAX_N=$v: "; ISTNET_FPS_TRACK_MAX_N=$v python bench.py --workload infer --no-roofline --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))"; done
for v in 1024 4096; do echo -n "infer ISTNET_FPS_TRACK_MAX_N=$v: "; ISTNET_FPS_TRACK_MAX_N=$v python bench.py --workload infer --no-roofline --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))"; done
python -m pytest tests/test_ops_gpu.py tests/test_golden_gpu.py -m gpu -q -k "fps or chain or config5 or infer" 2>&1 | tail -2
