import os, sys, json
ROOT='/root/repo'
sys.path.insert(0, ROOT)
import tweeker_raytracer_amd as twk
app = twk.Application(os.path.join(ROOT, "scenes", "system_rtigo3_cornell_box.txt"), os.path.join(ROOT, "scenes", "scene_rtigo3_cornell_box.txt"))
for maxd in (1, 2, 10):
    st = app.state
    st.pathLengths[1] = maxd
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    app.initDevice(dev); dev.setState(st)
    dev.setLaunchBatch(1)
    dev.statsEnable(True); dev.statsGet(True)
    for it in range(4): dev.render(it)
    dev.synchronizeStream()
    s = dev.statsGet(True)
    print(json.dumps({"maxDepth": maxd, "maxNodesPerRay": s["maxNodesPerRay"], "rays": s["radianceRays"]+s["shadowRays"], "nodes_per_ray": s["nodesVisited"]/max(1,s["radianceRays"]+s["shadowRays"]), "overflow": s["overflowRays"], "waveCycles": s["waveCycles"]}))
    dev.close()
