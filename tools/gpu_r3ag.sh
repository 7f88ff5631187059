#!/bin/bash
timeout 600 python -m pytest tests/test_preprocess.py -q -x 2>&1 | tail -8
