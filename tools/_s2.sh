python -m pytest tests/test_golden_gpu.py -q -s -k "heads_match or sa_fp_layer or encoder_b2" 2>&1 | grep -v Warning | grep "SLACK\|passed\|failed\|Error\|assert" | head -60
