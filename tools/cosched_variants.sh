#!/bin/bash
# one gpurun call: event rate of every experiment build (tools/build_variant.sh) + the operand dump
mkdir -p gpurun_out
for v in "dump 40 dump" "base 24" "waitall 24" "scalar 24" "noptrans 24"; do
  set -- $v
  timeout 300 python tools/cosched_dump.py tools/variants/$1/liblfm_hip.so $2 $3 > gpurun_out/cosched_$1.txt 2>&1
  tail -2 gpurun_out/cosched_$1.txt
done
