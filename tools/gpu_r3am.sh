#!/bin/bash
mkdir -p gpurun_out
python tools/aten_sources.py 2>/dev/null > gpurun_out/r3am_aten_sources.txt; head -64 gpurun_out/r3am_aten_sources.txt | cut -c1-170


