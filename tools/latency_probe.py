import time, numpy as np, sys
sys.path.insert(0, "/root/repo")
from general_motion_retargeting_amd import GeneralMotionRetargeting, synth, _lib
g = GeneralMotionRetargeting("smplx", "unitree_g1")
human, q0 = synth.make_streams(g.model, g._tables, 1, 300, seed=3)
frames = synth.streams_to_dicts(g._tables, human[0])
for f in frames[:20]: g.retarget(f)
t0=time.perf_counter()
for f in frames[20:]: g.update_targets(f)
t1=time.perf_counter()
sol=g.hip_solver; q=g.configuration.data.qpos
for i in range(280): sol.retarget_streams(q[None], human[:, 20+i:21+i])
t2=time.perf_counter()
lat=[]
for f in frames[20:]:
    t=time.perf_counter(); g.retarget(f); lat.append(time.perf_counter()-t)
print("update_targets us", (t1-t0)/280*1e6, "hip call us", (t2-t1)/280*1e6, "retarget p50 us", np.percentile(lat,50)*1e6)
