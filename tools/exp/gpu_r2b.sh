#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r2b_tests.txt
python tools/error_budget.py golden > gpurun_out/r2b_error_budget_golden.txt 2> gpurun_out/r2b_eb.err
python tools/error_budget.py shell32 > gpurun_out/r2b_error_budget_shell32.txt 2>> gpurun_out/r2b_eb.err
tail -5 gpurun_out/r2b_tests.txt; cat gpurun_out/r2b_error_budget_golden.txt; cat gpurun_out/r2b_error_budget_shell32.txt; tail -5 gpurun_out/r2b_eb.err
