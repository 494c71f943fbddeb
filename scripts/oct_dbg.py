import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccm_slam_amd import orb, synth
from ccm_slam_amd._lib import Context
ctx = Context(0)
ex = orb.ORBextractor(ctx, 1000)
img = synth.gen_image(1000, 0)
ex(img)
ex.close()
