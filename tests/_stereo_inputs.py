"""Inputs of CameraFrame::recover3DPts_c_FromStereo for the tests: the oracle front-end is run to a tracked frame of a synthetic stream
and hands over what getAll2dPlaneUndistort3d_cvPf / hasDepthInf() would -- the landmarks' pixel, undistorted pixel, world point and
depth flag -- with the frame's two images and pose."""
import numpy as np

import _oracle as O


def tracked_frame(ocfg, rig, stream, nframes, device="cpu", drop_depth_every=3):
    from flvis_amd import synth
    tr = synth.Trajectory(stream)
    rnd = synth.Renderer(device, rig=rig)
    trk = O.Tracker(ocfg, 0xF1715 + stream)
    imu = ocfg.type_of_vi != 4          # (the KITTI rig has no IMU)
    t_prev = -0.05
    standin = None
    out = None
    for f in range(nframes):
        t = f / synth.FRAME_HZ
        if imu:
            for s in synth.imu_samples(tr, stream, t_prev, t):
                trk.imu(s[0], s[1:4], s[4:7])
        t_prev = t
        if f >= ocfg.skip_first_n_imgs or standin is None:
            i0, i1 = rnd.stereo_frame([tr], t, f)
            standin = (i0[0].cpu().numpy(), i1[0].cpu().numpy())
        out = trk.image(t, standin[0], standin[1])
    assert out["state"] == 1
    lm = trk.landmarks()
    has = (lm["flags"] & 1).astype(np.uint8)
    if drop_depth_every:
        has[::drop_depth_every] = 0          # landmarks without depth yet: their seed is the pixel itself
    return dict(img0=standin[0], img1=standin[1], pose7=out["pose7"], p2d=lm["p2d"].astype(np.float32), p2u=lm["p2u"].astype(np.float32),
                p3w=lm["p3w"].astype(np.float32), has=has, p3w_exact=lm["p3w"])
