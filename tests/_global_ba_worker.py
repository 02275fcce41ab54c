"""One rank of the landmark-sharded global BA rehearsal (launched by test_global_ba_gpu.py; RANK/WORLD_SIZE/MASTER_* in env)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch.distributed as dist
from boslam_amd import Context, ba, synth

out = sys.argv[1]
n_kf, n_pts, window, iters = (int(v) for v in sys.argv[2:6])
dist.init_process_group(backend=os.environ.get("BH_TEST_BACKEND", "gloo"))
rank = dist.get_rank()
pr = synth.ba_problem_sequence(n_kf=n_kf, n_pts=n_pts, window=window, seed=int(os.environ.get("BH_TEST_SEED", "3")))
ctx = Context(0)
s = ba.GlobalBA(ctx, pr["K"], dist)
s.set_problem(pr["R"], pr["t"], pr["fixed"], pr["pts"], pr["ecam"], pr["ept"], pr["meas"])
done = s.optimize(iters)
R, t = s.poses()
pts = s.points_all()
ms, nbytes = s.comm()
det = s.comm_detail()
np.savez(os.path.join(out, f"rank{rank}.npz"), R=R, t=t, pts=pts, log=s.log()[:done], comm_ms=ms, comm_bytes=nbytes, solver=s.solver(),
         **{"det_" + k: v for k, v in det.items()})
s.close(); ctx.close()
dist.barrier()
dist.destroy_process_group()
