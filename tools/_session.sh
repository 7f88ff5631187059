for i in 1 2 3 4; do python -m pytest tests/test_rgb_ops_gpu.py -m gpu -q -s -k "choose_equals_dense" 2>&1 | grep GATHERFIRST | awk '{printf "%s=%s ", $2, $3} END {print ""}'; done
