#!/usr/bin/env python3
"""Throughput of rt_trace_closest_dev / rt_trace_any_dev on three ray distributions over the cornellBox
fixture: camera rays (coherent), rays leaving surfaces in random directions (the bounce rays of the path
tracer) and rays with uniform origins and directions. usage: tools/trace_bench.py [n_rays] [repeats]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from raytracing_amd.device import Context  # noqa: E402
from raytracing_amd.scene import FlatScene, camera_ray  # noqa: E402
import raygen  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 23
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    scene = FlatScene.load_npz(os.path.join(ROOT, "tests", "golden", "cornellBox_path.npz"))
    cfg = scene.config.with_resolution(1920, 1080)
    ctx = Context(0)
    ctx.upload_scene(scene)
    ctx.set_camera(cfg)
    dev = torch.device("cuda", 0)
    prim = torch.empty(n, dtype=torch.int32, device=dev)
    t = torch.empty(n, dtype=torch.float32, device=dev)
    bary = torch.empty((n, 3), dtype=torch.float32, device=dev)
    occ = torch.empty(n, dtype=torch.uint8, device=dev)

    def timed(fn):
        torch.cuda.synchronize()   # the batch was made by torch kernels on torch's stream; the library launches on its own
        fn()
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        ctx.synchronize()
        return (time.perf_counter() - t0) / reps

    all_schedules = tuple(os.environ.get("RT_TB_SCHEDULES", "lockstep,stream").split(","))
    quick = os.environ.get("RT_TB_QUICK") == "1"   # only the ungrouped batches
    only = [w for w in os.environ.get("RT_TB_ONLY", "").split(",") if w]   # e.g. "cosine,cosine2": only these batches are timed

    def run(name, o, d, schedules=None):
        if quick and ("/" in name):
            return
        if only and name not in only:
            return
        o = o.contiguous()
        d = d.contiguous()
        torch.cuda.synchronize()
        for sch in (schedules or all_schedules):
            ctx.set_trace_schedule(sch)
            run1(name + " " + sch, o, d)

    def run1(name, o, d):
        dt = timed(lambda: ctx.trace_closest_dev(n, o.data_ptr(), d.data_ptr(), 0, prim.data_ptr(), t.data_ptr(), bary.data_ptr()))
        da = timed(lambda: ctx.trace_any_dev(n, o.data_ptr(), d.data_ptr(), 0, occ.data_ptr()))
        ctx.synchronize()
        print(f"{name:>22}: closest {n / dt / 1e9:7.2f} Grays/s ({dt * 1e3:7.3f} ms)   any {n / da / 1e9:7.2f} Grays/s   "
              f"hit {float((prim >= 0).float().mean()):.3f} checksum {int(prim.long().sum())} {float(t[prim >= 0].double().sum()):.6e}")

    # camera rays in pixel order (pinhole at the camera position; directions through a regular grid)
    eye, dir00 = [np.asarray(v, np.float32) for v in camera_ray(cfg, 0.5, 0.5)]
    _, dir10 = [np.asarray(v, np.float32) for v in camera_ray(cfg, cfg.width - 0.5, 0.5)]
    _, dir01 = [np.asarray(v, np.float32) for v in camera_ray(cfg, 0.5, cfg.height - 0.5)]
    side = int(np.sqrt(n))
    u = torch.linspace(0, 1, side, device=dev)
    gy, gx = torch.meshgrid(u, u, indexing="ij")
    d00, d10, d01 = [torch.tensor(v, device=dev) for v in (dir00, dir10, dir01)]
    dcam = d00 + gx.reshape(-1, 1) * (d10 - d00) + gy.reshape(-1, 1) * (d01 - d00)
    dcam = torch.nn.functional.normalize(dcam, dim=1)
    if dcam.shape[0] < n:
        dcam = torch.cat([dcam, dcam[: n - dcam.shape[0]]])
    ocam = torch.tensor(eye, device=dev).expand(n, 3)
    run("camera", ocam, dcam)

    # bounce rays: start at the camera rays' hit points, uniform random directions
    p = ocam + t.unsqueeze(1) * dcam
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    drand = torch.nn.functional.normalize(torch.randn((n, 3), device=dev, generator=g), dim=1)
    keep = prim >= 0
    p = torch.where(keep.unsqueeze(1), p, torch.tensor(eye, device=dev).expand(n, 3))
    run("bounce", p, drand)
    # the same rays grouped by direction octant (stable: pixel order kept inside a group)
    octant = (drand[:, 0] > 0).int() + 2 * (drand[:, 1] > 0).int() + 4 * (drand[:, 2] > 0).int()
    order = torch.sort(octant, stable=True).indices
    run("bounce/oct", p[order], drand[order])
    # cosine-weighted directions about +y from floor-like points would be the path tracer's; here: hemisphere facing the box centre
    centre = torch.tensor([278.0, 274.4, 279.6], device=dev)
    flip = ((centre - p) * drand).sum(1) < 0
    dhemi = torch.where(flip.unsqueeze(1), -drand, drand)
    run("inward", p, dhemi)
    octant = (dhemi[:, 0] > 0).int() + 2 * (dhemi[:, 1] > 0).int() + 4 * (dhemi[:, 2] > 0).int()
    order = torch.sort(octant, stable=True).indices
    run("inward/oct", p[order], dhemi[order])

    # grouping inside windows of w consecutive rays only (what a workgroup could do on the fly), and a finer key
    idx = torch.arange(n, device=dev)
    for w in (256, 1024, 4096):
        order = torch.sort((idx // w) * 8 + octant, stable=True).indices
        run(f"inward/oct{w}", p[order], dhemi[order], schedules=("lockstep",))
    major = torch.argmax(dhemi.abs(), dim=1).int()
    order = torch.sort(octant * 3 + major, stable=True).indices
    run("inward/oct+axis", p[order], dhemi[order], schedules=("lockstep",))

    # what the path tracer's second bounce traces: hits only (compacted, pixel order), cosine-distributed about the
    # surface normal that faces the incoming ray
    tri = torch.tensor(scene.tri_p.reshape(-1, 3, 3), device=dev)
    ng_all = torch.nn.functional.normalize(torch.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0], dim=1), dim=1)
    run1("camera (for hits)", ocam.contiguous(), dcam.contiguous())
    hitmask = prim >= 0
    ph = (ocam + t.unsqueeze(1) * dcam)[hitmask]
    nh = ng_all[prim[hitmask].long()]
    nh = torch.where(((nh * dcam[hitmask]).sum(1) > 0).unsqueeze(1), -nh, nh)
    dcos = torch.nn.functional.normalize(nh + torch.nn.functional.normalize(torch.randn(ph.shape, device=dev, generator=g), dim=1) * 0.999, dim=1)
    m = ph.shape[0]
    reps_idx = torch.arange(n, device=dev) % m   # pad to n rays by repeating the sequence
    run("cosine", ph[reps_idx], dcos[reps_idx])
    # the same rays grouped by the primitive they start on, and by primitive and direction octant (global stable sorts:
    # an upper bound on what binning the queue appends could give)
    ctx.set_trace_schedule("lockstep")
    torch.cuda.synchronize()
    ctx.trace_closest_dev(n, ocam.contiguous().data_ptr(), dcam.contiguous().data_ptr(), 0, prim.data_ptr(), t.data_ptr(), bary.data_ptr())
    ctx.synchronize()
    start_prim = prim[hitmask][reps_idx].clone()
    oc, dc = ph[reps_idx], dcos[reps_idx]
    order = torch.sort(start_prim, stable=True).indices
    run("cosine/prim", oc[order], dc[order], schedules=("lockstep",))
    octc = (dc[:, 0] > 0).int() + 2 * (dc[:, 1] > 0).int() + 4 * (dc[:, 2] > 0).int()
    order = torch.sort(start_prim * 8 + octc, stable=True).indices
    run("cosine/prim+oct", oc[order], dc[order], schedules=("lockstep",))
    order = torch.sort(octc, stable=True).indices
    run("cosine/oct", oc[order], dc[order], schedules=("lockstep",))
    # third bounce: from the hit points of those, again cosine-distributed
    ctx.set_trace_schedule("lockstep")
    o2, d2 = ph[reps_idx].contiguous(), dcos[reps_idx].contiguous()
    torch.cuda.synchronize()
    ctx.trace_closest_dev(n, o2.data_ptr(), d2.data_ptr(), 0, prim.data_ptr(), t.data_ptr(), bary.data_ptr())
    ctx.synchronize()
    h2 = prim >= 0
    p2 = (o2 + t.unsqueeze(1) * d2)[h2]
    n2 = ng_all[prim[h2].long()]
    n2 = torch.where(((n2 * d2[h2]).sum(1) > 0).unsqueeze(1), -n2, n2)
    dc2 = torch.nn.functional.normalize(n2 + torch.nn.functional.normalize(torch.randn(p2.shape, device=dev, generator=g), dim=1) * 0.999, dim=1)
    idx2 = torch.arange(n, device=dev) % p2.shape[0]
    run("cosine2", p2[idx2], dc2[idx2])

    o, d = raygen.incoherent_rays(scene.bounds, n, 3)
    run("uniform", torch.tensor(o, device=dev), torch.tensor(d, device=dev))
    ctx.close()


if __name__ == "__main__":
    main()
