import sys, os
sys.path.insert(0, os.getcwd())
import tweeker_raytracer_amd as twk
app = twk.Application('scenes/system_rtigo3_cornell_box.txt','scenes/scene_rtigo3_cornell_box.txt')
for maxd in (1,2,3,5,10):
    st = app.state
    st.pathLengths[1] = maxd
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    app.initDevice(dev)
    dev.setState(st)
    dev.statsEnable(True); dev.statsGet(True)
    dev.render(5); dev.synchronizeStream()
    print(maxd, dev.statsGet(True))
    dev.close()
