import json, os, sys, time
ROOT = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, ROOT)
import tweeker_raytracer_amd as twk
def run(name, scene_text):
    system = open(os.path.join(ROOT, "scenes", "system_rtigo3_cornell_box.txt")).read()
    app = twk.Application(system_text=system, scene_text=scene_text)
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    app.initDevice(dev)
    for it in range(64): dev.render(it)
    dev.synchronizeStream()
    dev.statsEnable(True); dev.statsGet(reset=True)
    for it in range(64, 128): dev.render(it)
    dev.synchronizeStream()
    st = dev.statsGet(reset=True); dev.statsEnable(False)
    dev.profileReset(); dev.profileEnable(True)
    t0 = time.perf_counter()
    for it in range(128, 192): dev.render(it)
    dev.synchronizeStream()
    dt = time.perf_counter() - t0
    prof = dev.profileGet()
    print(json.dumps({"scene": name, "ms_per_step": dt * 1e3 / 64, "shaded_hits_per_step": st["shadedHits"] / 64, "missed": st["missed"] / 64, "prof": prof}), flush=True)
    dev.close()
c2 = open(os.path.join(ROOT, "scenes", "scene_rtigo3_cornell_box.txt")).read()
run("C2", c2)
alld = c2.replace("brdf_ggx_smith", "brdf_diffuse").replace("bsdf_specular", "brdf_diffuse").replace("brdf_specular", "brdf_diffuse")
run("C2 all diffuse", alld)
