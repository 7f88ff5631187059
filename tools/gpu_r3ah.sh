#!/bin/bash
timeout 600 python examples/train_synthetic.py --iters 12 --batch 8 2>&1 | tail -6
