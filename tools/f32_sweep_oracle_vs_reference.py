"""Offline sweep (build container only): random peninsula / Stommel configurations on float32 coordinates (A- and C-grid, float32 and
float64 particles, EE / RK2 / RK4, flat / spherical), the reference under oracle/ref_shim.py vs the C oracle, bit for bit.
Usage: python tools/f32_sweep_oracle_vs_reference.py N"""
import sys, time
sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo')
sys.dont_write_bytecode=True
import numpy as np, warnings, collections
warnings.simplefilter("ignore")
from oracle import cases, make_golden as mg
from case_utils import run_oracle, stop_time_of_reference
res=collections.Counter()
rng=np.random.default_rng(7)
t0=time.time()
for it in range(int(sys.argv[1])):
    kind=rng.choice(["pen","stom"])
    kern=str(rng.choice(["AdvectionRK4","AdvectionEE","AdvectionRK2"]))
    gt=str(rng.choice(["A","C"]))
    sdt=str(rng.choice(["float32","float32","float64"]))
    if kind=="pen":
        c=cases.peninsula_case("p",mesh=str(rng.choice(["flat","spherical"])),grid_type=gt,kernels=[kern],xdim=int(rng.integers(20,120)),ydim=int(rng.integers(15,60)),npart=int(rng.integers(5,40)),spatial_dtype=sdt,dt=float(rng.choice([900.,1800.,3600.])),runtime=float(rng.integers(4,30))*3600.)
    else:
        c=cases.stommel_case("s",grid_type=gt,kernels=[kern],xdim=int(rng.integers(20,80)),ydim=int(rng.integers(20,80)),npart=int(rng.integers(4,30)),spatial_dtype=sdt,dt=float(rng.choice([1800.,3600.,7200.])),runtime=float(rng.integers(2,20))*86400.)
    out,err,_=mg.ref_run_case(c)
    tstop=stop_time_of_reference(c,out,err)
    got,gerr,_=run_oracle(c,endtime=tstop)
    ok = gerr==err and len(got['x'])==len(out['x']) and all(np.array_equal(np.asarray(got[k]),np.asarray(out[k]),equal_nan=True) for k in ('x','y','z','t','ei')) and (tstop is not None or np.array_equal(got['state'],out['state']))
    res[(kind,gt,sdt,ok)]+=1
    if not ok:
        nd=sum(int((np.asarray(got[k])!=np.asarray(out[k])).sum()) for k in ('x','y','z')) if len(got['x'])==len(out['x']) else -1
        print("MISMATCH",kind,gt,sdt,kern,c['mesh'],np.asarray(c['lon']).shape,"err",err,gerr,"differing",nd,flush=True)
print(dict(res), time.time()-t0)
