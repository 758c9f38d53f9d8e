import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np
import bench
from metalign_amd._hip import Hip
hip = Hip.get(0)
def rep(tag):
    hip.sync()
    f, t, p = hip.mem_info()
    print("%-34s in use %7.1f GB, of which cached by the allocator %7.1f GB" % (tag, (t - f) / 2**30, p / 2**30), flush=True)
rep("start")
cfg = dict(bench.PRESETS[3], config=3)
w = bench.build_workload(cfg, 1000, 0, hip)
rep("workload built (genome sketches)")
job = bench.make_job(hip, None, 0, 1, cfg, w)
rep("job loaded (primed, trimmed)")
job.step(); rep("after 1 step")
job.run(4); rep("after run(4)")
job.run(20); rep("after run(20)")
hip.mem_trim(); rep("after trim")
