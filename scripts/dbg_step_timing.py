import sys, time, torch, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mmego_amd.train_step import StageStep
dev = torch.device("cuda:0")
t0=time.time()
imu, upper, lower, upper_frozen = bench.build_hip_models(dev)
x, imu_in, body, target = bench.synth_batch(1234, dev)
print("models+data", time.time()-t0, flush=True)
su = StageStep("upper", upper, imu, use_graph=False); su.bind(x, imu_in, body, target)
sl = StageStep("lower", lower, imu, upper_frozen=upper_frozen, use_graph=False); sl.bind(x, imu_in, body, target)
for i in range(3):
    t0=time.time(); su.step(); torch.cuda.synchronize(); t1=time.time(); sl.step(); torch.cuda.synchronize(); t2=time.time()
    print("eager step", i, "upper %.1f ms lower %.1f ms"%((t1-t0)*1e3,(t2-t1)*1e3), su.loss.item(), sl.loss.item(), flush=True)
with torch.no_grad():
    for i in range(3):
        t0=time.time(); R,t = imu(imu_in); torch.cuda.synchronize(); print("imu fwd %.2f ms"%((time.time()-t0)*1e3), flush=True)
su.use_graph=True; sl.use_graph=True
for i in range(4):
    t0=time.time(); su.step(); torch.cuda.synchronize(); t1=time.time(); sl.step(); torch.cuda.synchronize(); t2=time.time()
    print("graph step", i, "upper %.1f ms lower %.1f ms"%((t1-t0)*1e3,(t2-t1)*1e3), su.loss.item(), sl.loss.item(), flush=True)
