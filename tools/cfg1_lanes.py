import sys, os
sys.path.insert(0, os.getcwd())
import bench, path_tracer_amd as pt
rt = bench.build_raytracer(pt, "rtow", 400, 225, 10)
r = pt.ParallelRenderer.new(0)
r.bind(rt)
for lanes in (0, 4, 8, 16, 32, 64):
    r.set_lanes_per_pixel(lanes)
    best = 1e9
    for _ in range(6):
        r.render(rt, seed=1)
        best = min(best, r.last_path_kernel_ms())
    print("lanes", lanes, "->", r.lanes_per_pixel(10, 0, 100), "kernel ms %.3f" % best)
