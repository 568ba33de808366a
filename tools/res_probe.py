"""Probe: average trace/shade launch time versus frame size (how long does a nearly empty launch take?)."""
import os, sys
sys.path.insert(0, os.getcwd())
import tweeker_raytracer_amd as twk
for res in ((16, 9), (64, 36), (256, 144), (640, 360), (1920, 1080)):
    app = twk.Application('scenes/system_rtigo3_cornell_box.txt', 'scenes/scene_rtigo3_cornell_box.txt')
    app.setResolution(*res)
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    app.initDevice(dev)
    for it in range(3):
        dev.render(it)
    dev.synchronizeStream()
    dev.profileEnable(True); dev.profileReset()
    for it in range(3, 11):
        dev.render(it)
    p = dev.profileGet()
    dev.profileEnable(False)
    dev.statsEnable(True); dev.statsGet(True); dev.render(3); s = dev.statsGet(True)
    print(res, 'trace avg us', 1e3 * p['trace']['ms'] / p['trace']['launches'], 'shade avg us', 1e3 * p['shade']['ms'] / p['shade']['launches'],
          'rays/step', s['radianceRays'] + s['shadowRays'], 'max nodes', s['maxNodesPerRay'])
    dev.close()
