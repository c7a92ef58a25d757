import sys, time, ctypes
sys.path.insert(0, '.')
import numpy as np, torch
from basilisk_env_amd._lib import GRAV_PM_J2
from basilisk_env_amd.simulators.dynamics import BatchedPropagator, default_config
from basilisk_env_amd.simulators.initial_conditions.batch import sample_ic_batch
n=65536
cfg=default_config(4,GRAV_PM_J2)
p=BatchedPropagator(cfg,n); p.reset(sample_ic_batch(n,4,seed=0))
act=torch.zeros(n,dtype=torch.int32,device='cuda'); ptr=act.data_ptr()
lib=p._lib; h=p._h
def t(f,k=2000):
    for _ in range(100): f()
    p.sync(); t0=time.perf_counter()
    for _ in range(k): f()
    p.sync(); return (time.perf_counter()-t0)/k*1e6
print('ctypes no-op (bsk_n_fields)      %.2f us'%t(lambda: lib.bsk_n_fields(h)))
print('step_device via wrapper, no prof %.2f us'%t(lambda: p.step_device(ptr,1)))
f=lib.bsk_step_device; vp=ctypes.c_void_p(ptr)
print('raw ctypes bsk_step_device       %.2f us'%t(lambda: f(h,vp,1)))
p.profile_begin(2200)
print('raw ctypes with dispatch events  %.2f us'%t(lambda: f(h,vp,1)))
print(p.profile_end())
print('K=4 sub-steps raw                %.2f us'%t(lambda: f(h,vp,4)))
