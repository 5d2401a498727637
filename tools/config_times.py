"""All five BASELINE configurations on one GPU: frame times (pipelined and per kernel)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import piet_metal_amd as pm

def main():
    r = pm.Renderer(0)
    W = pm.workloads
    cases = [("config1 rect 512x512", W.config1_rect()), ("config2 tiger 1920x1080 fills", W.tiger(1920, 1080, fills_only=True)),
             ("config3 tiger 3840x2160", W.tiger(3840, 2160)), ("config4 10k blobs 4096x4096", W.config4_blobs()),
             ("config5 5x5 tigers 8192x8192", W.config5_tiger_grid())]
    for name, wl in cases:
        t = time.time()
        r.resize(wl.width, wl.height)
        nb, ni = r.flatten_and_encode(wl.paths, wl.affine, wl.width_scale)
        r.render(); r.sync()
        t_first = time.time() - t
        st = r.stats()
        iters = 200 if wl.width <= 4096 and ni < 5000 else 30
        tm = r.time_frames(iters)
        print(json.dumps({"case": name, "items": ni, "scene_bytes": nb, "setup_plus_first_frame_s": round(t_first, 3),
                          "pipelined_ms": round(tm["total_ms"] / iters, 4), "bin_ms": round(tm["bin_ms"], 4), "coarse_ms": round(tm["coarse_ms"], 4),
                          "fine_ms": round(tm["fine_ms"], 4), "mpix_s": round(wl.width * wl.height / (tm["total_ms"] / iters) / 1e3, 1),
                          "queued_tiles": st["queued_tiles"], "heavy_tiles": st["heavy_tiles"], "arena_cap_MB": round(st["arena_cap_dwords"] * 4 / 1e6, 1),
                          "arena_used_MB": round(st["arena_used_dwords"] * 4 / 1e6, 1), "ptcl_cmds": st["ptcl_used_cmds"]}), flush=True)

main()
