"""Extract a small data fixture from the KITTI ground-truth poses the reference ships (bag/KITTI/dataset/poses.zip):
the first 120 poses of sequence 04 -> tests/golden/kitti_poses_04_head.txt.  Run in the build container only."""
import os
import zipfile

src = "/root/reference/bag/KITTI/dataset/poses.zip"
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "kitti_poses_04_head.txt")
with zipfile.ZipFile(src) as z:
    lines = z.read("poses/04.txt").decode().splitlines()[:120]
open(dst, "w").write("\n".join(lines) + "\n")
print(len(lines), "poses ->", dst)
