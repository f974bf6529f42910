#!/bin/bash
# round 5, fourth GPU call: kernel trace + PMC passes of the final binary (C2, C3, C5 RK45 / M1), summarised on the box
bash tools/gpu_round5_profiles.sh r05f > gpurun_out/r05d_profiles.log 2>&1
tail -30 gpurun_out/r05d_profiles.log
ls gpurun_out/r05_profiles | head -40
