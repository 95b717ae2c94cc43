"""Training-step time with a given build of the library (A/B of two builds inside one gpurun call).  usage: ab_lib.py <lib.so> [mode] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neural_marionette_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
from neural_marionette_amd import NeuralMarionette, HotPathOptions, synth
from neural_marionette_amd.train import DetectorTrainer
mode = sys.argv[2] if len(sys.argv) > 2 else "bf16"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 10
o = HotPathOptions(grid_size=64)
net = NeuralMarionette(o); net.load_state_dict(synth.make_state_dict(o, seed=42, variant="peaky"))
net = net.cuda().train(); net.anneal(1); net.set_conv_mode(mode)
vox = synth.figure_clip(4, 16, 64, seed=77).cuda()
tr = DetectorTrainer(net, lr=4e-4)
for _ in range(3): tr.step(vox, sync=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n): tr.step(vox, sync=False)
torch.cuda.synchronize()
print("%s %s: %.2f ms per step" % (os.path.basename(sys.argv[1]), mode, (time.perf_counter() - t0) * 1e3 / n))
