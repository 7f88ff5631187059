#!/bin/bash
timeout 600 python -m pytest tests/test_golden_gpu.py -q -x -k "sa_fp_layer" 2>&1 | tail -15
