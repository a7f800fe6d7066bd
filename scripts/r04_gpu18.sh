#!/bin/bash
# preprocess at 8 waves per SIMD (64 registers, 13 spilled) against the product's 7 (70 registers, no spill): rebuilt ON the box, A / B / A
mkdir -p gpurun_out/r04live
run() { echo -n "$1  "; python bench.py --no-dit --no-cpu-baseline --streams 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step_serial'], d['stage_ms_per_step']['preprocess'])"
          echo -n "$1 live  "; python bench.py --live-only 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_sample'])"; }
{ run A7
cp gvfdiffusion_amd/csrc/rast.hip /tmp/rast.hip.orig
sed -i 's/^__global__ __launch_bounds__(PRE_THREADS) void preprocess_kernel(/__global__ __launch_bounds__(PRE_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void preprocess_kernel(/' gvfdiffusion_amd/csrc/rast.hip
python -m gvfdiffusion_amd._build > /dev/null 2>&1
run B8
python -m pytest tests/test_rast_gpu.py -m gpu -x -q 2>&1 | tail -1
cp /tmp/rast.hip.orig gvfdiffusion_amd/csrc/rast.hip
python -m gvfdiffusion_amd._build > /dev/null 2>&1
run A7; } 2>&1 | tee gpurun_out/r04live/pre_waves8_ab.txt
