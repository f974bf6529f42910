#!/bin/bash
# Runs ON THE GPU BOX (gpurun): kernel trace + PMC passes of the headline (C2), C3 and C5 of THIS binary, summarised there
# (tools/pmc_summary.py) -- only the summaries travel back (gpurun_out/r06_profiles/).
N=${1:-r06f}
OUT=$PWD/gpurun_out; mkdir -p $OUT/r06_profiles
B2="--secondary 0 --long-run 0 --reps 1 --repeat-execute 0 --user-kernels 0 --with-output 0 --check 0"
bash tools/gpu_profile_c2.sh ${N}_c2 $B2 > /dev/null
python tools/pmc_summary.py ${N}_c2 ${N}_c2 > /dev/null
psteps() { python -c "
import json,sys
for l in open('$OUT/$1_run.json'):
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l)
        if d['kernels']=='$2': print(d['particle_steps'])
"; }
bash tools/gpu_profile_cfg.sh ${N}_c3 c3 --reps 0 > /dev/null
python tools/pmc_summary.py ${N}_c3 ${N}_c3 --no-latest --psteps $(psteps ${N}_c3 AdvectionRK4_3D) --secondary AdvectionRK4_3D > /dev/null
bash tools/gpu_profile_cfg.sh ${N}_c5 c5 --reps 0 --pairs-leg 0 > /dev/null
python tools/pmc_summary.py ${N}_c5 ${N}_c5_rk45 --no-latest --match rk45_kernel --evals-per-step 6 --psteps $(psteps ${N}_c5 AdvectionRK45) --secondary AdvectionRK45 > /dev/null
python tools/pmc_summary.py ${N}_c5 ${N}_c5_m1 --no-latest --match m1_kernel --evals-per-step 7 --psteps $(psteps ${N}_c5 AdvectionDiffusionM1) --secondary AdvectionDiffusionM1 > /dev/null
cp profiles/${N}_* profiles/pmc_latest.json profiles/pmc_secondary_latest.json $OUT/r06_profiles/
rm -rf $OUT/${N}_*_trace $OUT/${N}_*_pmc_*
ls $OUT/r06_profiles | head -40
