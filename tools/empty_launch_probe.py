"""Probe: cost of trace launches with (almost) no rays — camera looks past the box, every primary ray misses."""
import os, sys
sys.path.insert(0, os.getcwd())
import tweeker_raytracer_amd as twk
scene = open('scenes/scene_rtigo3_cornell_box.txt').read()
system = open('scenes/system_rtigo3_cornell_box.txt').read().replace('center 0 1 0', 'center 0 100 0')
app = twk.Application(system_text=system, scene_text=scene)
dev = twk.Device(ordinal=0, miss=app.info.miss)
app.initDevice(dev)
for it in range(3):
    dev.render(it)
dev.synchronizeStream()
dev.profileEnable(True); dev.profileReset()
for it in range(3, 13):
    dev.render(it)
print(dev.profileGet())
dev.profileEnable(False)
dev.statsEnable(True); dev.statsGet(True); dev.render(0); print(dev.statsGet(True))
