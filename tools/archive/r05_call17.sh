#!/bin/bash
# final confirmation on HEAD: whole GPU suite, smoke, the driver's bench line
mkdir -p gpurun_out
{
python -m pytest tests -m gpu -q 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py 2>&1 | tail -1
} > gpurun_out/r05_call17.log 2>&1
tail -12 gpurun_out/r05_call17.log
