import ctypes as C, sys
sys.path.insert(0,'.')
from ray_tracing_amd import _lib as L
lib=L.load()
for w in (1,2,4):
    for name in ("mul_add","cndmask","salu","valu_salu","slow_salu"):
        r=L.ValuProbe()
        assert lib.rt_probe_valu_issue(0, L.VALU_MIXES.index(name), w, 2000, C.byref(r))==0
        print(f"{name:10s} W={w}: {r.simd_cycles_per_inst:6.3f} SIMD cycles/inst  clock {r.clock_mhz:.0f} MHz", flush=True)
