#!/bin/bash
# round 4, GPU call 10: batch-count rule check (2 / 3 / 4 on NS, C3, C4 1M), in-process 2-stripe bench on one GPU, RCCL 1-rank combine
cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out
timeout 900 python tools/ab_env.py - "SVSDF_BATCHES=3;SVSDF_BATCHES=2;SVSDF_BATCHES=4;SVSDF_BATCHES=1" NS,C3,C4 1000000 10 > gpurun_out/r4_10_batches.txt 2>&1
cat gpurun_out/r4_10_batches.txt | cut -c1-200
timeout 600 python bench.py --gpus 2 --devices 0,0 --steps 5 --points 400000 --no-extras > gpurun_out/r4_10_bench_2stripes.json 2> gpurun_out/r4_10_bench_2stripes.err
tail -c 1500 gpurun_out/r4_10_bench_2stripes.json; tail -3 gpurun_out/r4_10_bench_2stripes.err
timeout 600 python bench.py --gpus 1 --inprocess --combine rccl --steps 5 --config C2 --no-extras --no-cpu-baseline > gpurun_out/r4_10_bench_rccl1.json 2> gpurun_out/r4_10_bench_rccl1.err
tail -c 800 gpurun_out/r4_10_bench_rccl1.json; tail -3 gpurun_out/r4_10_bench_rccl1.err
