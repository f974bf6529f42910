#!/bin/bash
# round 6: SQ counters of the refilling AdvectionRK45 kernel (PK_RK45_REFILL=12) beside the one-shot kernel, config 5
# (needs tools/patches/rk45_refilling_wavefronts.patch applied and the library rebuilt)
N=r06m
OUT=$PWD/gpurun_out; mkdir -p $OUT/r06m
psteps() { python -c "
import json,sys
for l in open('$OUT/$1_run.json'):
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l)
        if d['kernels']=='$2': print(d['particle_steps'])
"; }
bash tools/gpu_profile_cfg.sh ${N}_one c5 --reps 0 --pairs-leg 0 --only rk45 > /dev/null
python tools/pmc_summary.py ${N}_one ${N}_c5_rk45_one_shot --no-latest --match rk45_kernel --evals-per-step 6 --psteps $(psteps ${N}_one AdvectionRK45) > /dev/null
export PK_RK45_REFILL=12
bash tools/gpu_profile_cfg.sh ${N}_refill c5 --reps 0 --pairs-leg 0 --only rk45 > /dev/null
python tools/pmc_summary.py ${N}_refill ${N}_c5_rk45_refill --no-latest --match rk45_refill_kernel --evals-per-step 6 --psteps $(psteps ${N}_refill AdvectionRK45) > /dev/null
cp profiles/${N}_* $OUT/r06m/
rm -rf $OUT/${N}_*_trace $OUT/${N}_*_pmc_*
ls $OUT/r06m
