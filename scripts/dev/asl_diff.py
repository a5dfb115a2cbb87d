"""dev aid: run_sequence.py hip vs cpu on a synthetic ASL folder; where do the trajectories start to differ?"""
import json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from flvis_amd import traj_io
import test_dataset_runner as T
root, yaml, frames, imu_sensor, _, _ = T.make_asl_folder(17, int(sys.argv[1]) if len(sys.argv) > 1 else 1403636579000000000)
res = {}
for backend in ("cpu", "hip"):
    out, imu_out = os.path.join(root, "traj_%s.txt" % backend), os.path.join(root, "imu_%s.txt" % backend)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "run_sequence.py"), root, yaml, out, "--backend", backend, "--imu-out", imu_out],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    print(backend, r.stdout.decode().strip().splitlines()[-1] if r.returncode == 0 else r.stderr.decode()[-1500:])
    res[backend] = (traj_io.read_stamped(out), traj_io.read_stamped(imu_out))
(ta, pa, qa), (tb, pb, qb) = res["hip"][0], res["cpu"][0]
print("pose traj: n", len(ta), len(tb), "max dpos", np.abs(np.asarray(pa) - np.asarray(pb)).max())
(ta, pa, qa), (tb, pb, qb) = res["hip"][1], res["cpu"][1]
d = np.abs(np.asarray(pa) - np.asarray(pb)).max(axis=1)
print("imu traj: n", len(ta), len(tb), "max dpos", d.max(), "first differing row", int(np.argmax(d > 1e-7)), "of", len(d))
k = int(np.argmax(d > 1e-7))
print("t at first diff", ta[k] - ta[0], "frame times", [round(f[0] * 1e-9 - ta[0], 3) for f in frames][:12])
print(np.asarray(pa)[k - 2:k + 3], np.asarray(pb)[k - 2:k + 3])
